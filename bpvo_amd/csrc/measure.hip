// libbpvo_hip, host side: measurement and diagnostics — profiling events, kernel statistics, counters of the work done.
#include "host_ctx.h"

using namespace bpvo_hip;
using namespace bpvo_hip_host;

extern "C" {

// ---- measurement ----------------------------------------------------------------------------------------------------
int bpvo_hip_profiling(bpvo_hip_ctx* c, int enable)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  resolve_events(c);
  c->profiling = enable != 0;
  c->profile_all = enable == 2;
  c->profile_k6_all = enable == 3;
  for(int k = 0; k < KC_COUNT; ++k) { c->kc_ms[k] = 0; c->kc_units[k] = 0; c->kc_launches[k] = 0; }
  HIP_CK(c, hipMemset(c->d_counters, 0, kWsCounters * sizeof(unsigned long long) * c->n_pairs));
  c->total_lin = 0;
  for(auto& ln : c->lanes) ln.k6_seq = 0;
  return BPVO_OK;
}
int bpvo_hip_get_kernel_stats(bpvo_hip_ctx* c, bpvo_hip_kernel_stat* out, int max_out, int* n_out)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  resolve_events(c);
  int rc = refresh_counters(c);
  if(rc) return rc;
  // algorithmic bytes per unit: unit = template point for the GN kernels, pixel otherwise
  const double C = c->C;
  const double bpu[KC_COUNT] = {2.0, 1.0 + 4.0 * C, 4.0 * C + 4.0 + 4.0, 32.0, 16.0 + 4.0 + 5.0 * 4.0 * C + 28.0 * C, 18.0 + 24.0 * C,
                                4.0 * C, 2.0 + 28.0 * C, 0.0};
  int n = 0;
  for(int k = 0; k < KC_COUNT && n < max_out; ++k, ++n) {
    std::memset(&out[n], 0, sizeof(out[n]));
    std::snprintf(out[n].name, sizeof(out[n].name), "%s", kKernelNames[k]);
    out[n].launches = c->kc_launches[k];
    out[n].total_ms = c->kc_ms[k];
    out[n].units = c->kc_units[k];
    out[n].bytes_per_unit = bpu[k];
    if(k == KC_IRLS_REDUCE && c->kc_units[k] > 0)   // fused points carry warp_residual's bytes as well
      out[n].bytes_per_unit = bpu[k] + bpu[KC_WARP_RESIDUAL] * c->points_fused / c->kc_units[k];
  }
  *n_out = n;
  return BPVO_OK;
}
int bpvo_hip_fused_point_counts(bpvo_hip_ctx* c, uint64_t* fused, uint64_t* total)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  int rc = refresh_counters(c);
  if(rc) return rc;
  *fused = (uint64_t) c->points_fused;
  *total = (uint64_t) c->kc_units[KC_IRLS_REDUCE];
  return BPVO_OK;
}
int bpvo_hip_persistent_counts(bpvo_hip_ctx* c, uint64_t* levels, int* gave_up)
{
  if(!c) return BPVO_ERR_INVALID_ARG;
  if(levels) *levels = c->persistent_levels.load();
  if(gave_up) *gave_up = c->persistent_failed.load() ? 1 : 0;
  return BPVO_OK;
}

int bpvo_hip_upload_stats(bpvo_hip_ctx* c, double* seconds, uint64_t* bytes)
{
  if(!c) return BPVO_ERR_INVALID_ARG;
  if(seconds) *seconds = c->up_last_seconds;
  if(bytes) *bytes = (uint64_t) c->up_last_bytes;
  return BPVO_OK;
}
int bpvo_hip_team_counts(bpvo_hip_ctx* c, uint64_t* launches)
{
  if(!c || !launches) return BPVO_ERR_INVALID_ARG;
  *launches = c->team_launches.load();
  return BPVO_OK;
}

int bpvo_hip_median_path_counts(bpvo_hip_ctx* c, uint64_t* bracketed, uint64_t* full)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  int rc = refresh_counters(c);
  if(rc) return rc;
  *bracketed = c->median_bracketed;
  *full = c->median_full;
  return BPVO_OK;
}
int bpvo_hip_tap_cache_counts(bpvo_hip_ctx* c, uint64_t out[4])
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  int rc = refresh_counters(c);
  if(rc) return rc;
  for(int k = 0; k < 4; ++k) out[k] = c->tap_counts[k];
  return BPVO_OK;
}
// diagnostics: the Gauss-Newton state of a workspace after its last call — out[0..15] T, [16..51] H, [52..57] G, [58..63] dp,
// [64] f_norm, [65] scale, [66] delta_scale, [67] g_norm, [68..83] T_lin (pose of the last linearisation)
int bpvo_hip_debug_gn_state(bpvo_hip_ctx* c, int ws, float out[84])
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  (void) hipSetDevice(c->device);
  GNState st;
  HIP_CK(c, hipStreamSynchronize(c->stream));
  HIP_CK(c, hipMemcpy(&st, c->d_states + ws, sizeof(st), hipMemcpyDeviceToHost));
  std::memcpy(out, st.T, 64); std::memcpy(out + 16, st.H, 144); std::memcpy(out + 52, st.G, 24); std::memcpy(out + 58, st.dp, 24);
  out[64] = st.f_norm; out[65] = st.scale; out[66] = st.delta_scale; out[67] = st.g_norm;
  std::memcpy(out + 68, st.T_lin, 64);
  return BPVO_OK;
}

int bpvo_hip_total_linearizations(bpvo_hip_ctx* c, uint64_t* n)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  int rc = refresh_counters(c);
  if(rc) return rc;
  *n = c->total_lin;
  return BPVO_OK;
}

}  // extern "C"
