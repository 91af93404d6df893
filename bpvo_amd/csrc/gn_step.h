// K9 Gauss-Newton step, device side: f64 sum of the tile partials, 6x6 solve, SE(3) update, PoseEstimatorBase::run's bookkeeping.
#pragma once
#include "gn_common.h"

namespace bpvo_hip {

// ------------------------------------------------------------------------------------------------------------------
// K9 gn_step: PoseEstimatorBase::run as a device-side state machine (reference: bpvo/pose_estimator_base.h:324-407 with
// testConvergence :258-282, PoseEstimatorData_::solve :90-148, RigidBodyWarp::paramsToPose bpvo/rigid_body_warp.h:130-138).
// One wave per workspace: lanes 0..28 sum the per-block partials in block order in f64 (deterministic), lane 0 runs the
// 6x6 solve, pose update and bookkeeping — Q1 (pose updated again after convergence) and Q2 (iteration count) included.
__device__ __forceinline__ float inf_norm6(const float* g)
{
  float m = 0.0f;
  for(int i = 0; i < 6; ++i) m = fmaxf(m, fabsf(g[i]));
  return m;
}

__device__ void gn_update_pose(GNState* st, const float* nrm)
{
  float mdp[6];
  for(int i = 0; i < 6; ++i) mdp[i] = -st->dp[i];
  M44 T;
  for(int i = 0; i < 16; ++i) T.m[i] = st->T[i];
  // nrm[4] != 0: DisparitySpaceWarp::paramsToPose = TwistToMatrix(p), scalePose is the identity (disparity_space_warp.h:79-91)
  const M44 Tn = m44_mul(T, nrm[4] != 0.0f ? twist_to_matrix(mdp) : params_to_pose(nrm, mdp));
  for(int i = 0; i < 16; ++i) st->T[i] = Tn.m[i];
}

__device__ void gn_finalize(GNState* st)
{
  if(st->status != BPVO_STATUS_SOLVER_ERROR)
    for(int i = 0; i < 16; ++i) st->T_out[i] = st->T[i];
  st->num_iterations -= 1;
  bpvo_hip_stats& s = st->stats[st->level];
  s.numIterations = st->num_iterations;
  s.finalError = st->f_norm;
  s.firstOrderOptimality = st->g_norm;
  s.status = st->status;
  st->phase = PHASE_DONE;
  st->active = 0;
}

// PoseEstimatorBase::reset + the head of run() (bpvo/pose_estimator_base.h:287-293,327-335): the state a pyramid level starts from —
// level_begin_kernel on the state in HBM, the persistent single-pair kernel on its workgroups' copies (kernels_gn_team.hip).
__device__ __forceinline__ void gn_level_reset(GNState* st, int level, int scale_is_moot, int n_points)
{
  st->scale = 1.0f;
  st->delta_scale = scale_is_moot ? 0.0f : 1e10f;
  st->f_norm_prev = 0.0f;
  st->g_tol = 0.0f;
  st->g_norm = 0.0f;
  st->num_fun_evals = 0;
  st->num_iterations = 0;
  st->status = BPVO_STATUS_MAX_ITERATIONS;
  st->phase = PHASE_FIRST;
  st->has_converged = 0;
  st->level = level;
  st->median_valid = 0;
  st->last_median = 0.0f;
  for(int i = 0; i < 16; ++i) st->T[i] = st->T_out[i];
  for(int i = 0; i < 6; ++i) st->dp[i] = 0.0f;
  st->active = (n_points > 0) ? 1 : 0;
}

// the serial part of gn_step, executed by lane 0 on the LDS copy of the state; returns true if another linearisation
// is requested (the workspace stays active)
#ifdef BPVO_PK_TIMING
__shared__ unsigned pk_sub[8];      // timing build: 10-ns ticks of the serial step's parts (unpack, solve, pose update, tests), summed
#define GN_SUBTICK(k) do { const long long t_ = wall_clock64(); pk_sub[k] += (unsigned) (t_ - sub_t); sub_t = t_; } while(0)
#else
#define GN_SUBTICK(k) do { } while(0)
#endif
__device__ bool gn_logic(GNState* st, const float* nrm, const float* s_sum, SolveScratch* scratch, int mode, int max_iterations,
                         int max_fun_evals, float p_tol, float f_tol, float g_tol_param)
{
#ifdef BPVO_PK_TIMING
  long long sub_t = wall_clock64();
#endif
  // unpack: upper triangle -> symmetric H (toEigen + selfadjointView<Upper>, linear_system_builder.cc:207-221)
  {
    int idx = 0;
    for(int a = 0; a < 6; ++a)
      for(int b = a; b < 6; ++b) {
        st->H[a * 6 + b] = s_sum[idx];
        st->H[b * 6 + a] = s_sum[idx];
        ++idx;
      }
    for(int a = 0; a < 6; ++a) st->G[a] = s_sum[21 + a];
  }
  const float f_norm = sqrtf(s_sum[27]);               // LinearSystemBuilder::Run returns sqrt (:349)
  st->f_norm = f_norm;
  st->n_valid = (uint32_t) s_sum[28];
  st->num_fun_evals += 1;
  if(mode == 1) return true;
  GN_SUBTICK(0);

  const float sqrt_eps = sqrtf(FLT_EPSILON);

  const bool first = st->phase == PHASE_FIRST;
  if(first) {
    const float g_norm = inf_norm6(st->G);
    st->g_norm = g_norm;
    st->g_tol = g_tol_param * fmaxf(g_norm, sqrt_eps);
    if(g_norm < st->g_tol) {                            // :343-354 initial value is optimal
      bpvo_hip_stats& s = st->stats[st->level];
      s.status = BPVO_STATUS_GRADIENT_TOL; s.finalError = f_norm; s.numIterations = 1; s.firstOrderOptimality = g_norm;
      st->status = BPVO_STATUS_GRADIENT_TOL;
      st->phase = PHASE_DONE; st->active = 0;
      return false;
    }
  }
  // ONE call site for both phases: the solve is then inlined where it is used (a shared out-of-line copy saves its callee-saved
  // registers to scratch, and every kernel that holds the step — irls_reduce with step_in_reduce — would carry a private segment)
  const bool solved = solve_system(st->H, st->G, st->dp, scratch);
  if(first) {
    if(!solved) {                                                // :356-362
      bpvo_hip_stats& s = st->stats[st->level];
      s.status = BPVO_STATUS_SOLVER_ERROR; s.finalError = f_norm; s.numIterations = 0; s.firstOrderOptimality = 0.0f;
      st->status = BPVO_STATUS_SOLVER_ERROR;
      st->phase = PHASE_DONE; st->active = 0;
      return false;
    }
    st->f_norm_prev = 0.0f;
    st->dp_norm_prev = 0.0f;
    st->has_converged = 0;
    gn_update_pose(st, nrm);                            // :371
  } else {
    // runIteration's solve (pose_estimator_gn.h:89-97)
    if(!solved) {
      st->status = BPVO_STATUS_SOLVER_ERROR;
      gn_finalize(st);                                  // `break`: no ++ on the way out
      return false;
    }
    GN_SUBTICK(1);
    gn_update_pose(st, nrm);                            // :390
    GN_SUBTICK(2);
    const bool cont = (st->num_iterations++ < max_iterations) && !st->has_converged && (st->num_fun_evals < max_fun_evals);
    if(!cont) { gn_finalize(st); return false; }
  }

  // top of the do-loop body (:374-383)
  float dp_norm = 0.0f;
  for(int i = 0; i < 6; ++i) dp_norm += st->dp[i] * st->dp[i];
  dp_norm = sqrtf(dp_norm);
  const float g_norm = inf_norm6(st->G);
  st->g_norm = g_norm;
  bool conv = false;
  if(dp_norm < p_tol || dp_norm < p_tol * (sqrt_eps + st->dp_norm_prev)) {
    st->status = BPVO_STATUS_PARAMETER_TOL; conv = true;
  } else if(f_norm < f_tol || f_norm < f_tol * (sqrt_eps + st->f_norm_prev) || fabsf(f_norm - st->f_norm_prev) < f_tol) {
    st->status = BPVO_STATUS_FUNCTION_TOL; conv = true;
  } else if(g_norm < st->g_tol) {
    st->status = BPVO_STATUS_GRADIENT_TOL; conv = true;
  }
  st->has_converged = conv ? 1 : 0;
  st->dp_norm_prev = dp_norm;
  st->f_norm_prev = f_norm;
  GN_SUBTICK(3);
  if(!conv) {
    st->phase = PHASE_LOOP;                             // next launch: linearize at the updated pose
    return true;
  }
  gn_update_pose(st, nrm);                              // Q1: applied again with the stale dp
  st->num_iterations++;                                 // the `numIterations++ <` of the failing while test
  gn_finalize(st);
  return false;
}

// lanes 0 .. kNumAcc-1 of one wave: deterministic sum (tile order, f64) of the tile partials of workspace j.
// The loads of 32 tiles are issued back to back, UNCONDITIONALLY (the tile index is clamped, the add is what the bound selects: a
// conditional load makes the compiler wait per branch), so a level costs one global-memory round trip per 32 tiles instead of one per
// 8: 2.2 -> 1.3 us of the serial step at the finest level of a 1241x376 pair (profiles/r02_persistent_phases.txt).  Same order of additions.
template <bool COHERENT = false>
__device__ __forceinline__ void gn_sum_partials(const PairJob& j, int pts_per_block, int lane, float* s_sum /*[kPartialStride]*/,
                                                const float* __restrict__ partials)
{
  // (wide descriptors: the tiles of every channel group, one run after the other — except in reference order, pts_per_block = 2^30, where one
  // partial holds the sums over all channels)
  const int groups = (pts_per_block >= (1 << 30)) ? 1 : j.n_groups;
  const int nblk = ((j.n + pts_per_block - 1) / pts_per_block) * groups;
  if(lane < kNumAcc) {
    double s = 0.0;
    const float* __restrict__ pp = partials + lane;
    auto chunked = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      for(int b0 = 0; b0 < nblk; b0 += U) {
        float v[U];
#pragma unroll
        for(int u = 0; u < U; ++u) {
          const float* q = pp + (size_t) min(b0 + u, nblk - 1) * kPartialStride;
          if constexpr(COHERENT) v[u] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else v[u] = *q;
        }
#pragma unroll
        for(int u = 0; u < U; ++u) {
          const double t = s + (double) v[u];
          s = (b0 + u < nblk) ? t : s;
        }
      }
    };
    if(nblk <= 8) chunked(std::integral_constant<int, 8>());      // (coarse levels: no point in 32 loads for 6 tiles)
    else chunked(std::integral_constant<int, 32>());
    if(lane == 28 && groups > 1) s = s / (double) groups;      // every group counted the valid points
    s_sum[lane] = (float) s;
  }
}

// one thread, on an LDS copy `st` of the state: the step that consumes the linearisation summed in s_sum.  `stats`: this copy
// is the one that keeps the workspace's measurement counters (the persistent kernel runs the step redundantly in every workgroup)
__device__ __forceinline__ void gn_serial_step(const PairJob& j, GNState* st, const float* s_nrm, const float* s_sum, SolveScratch* scratch,
                                               int mode, int max_iterations, int max_fun_evals, float p_tol, float f_tol, float g_tol_param,
                                               int fuse_frozen, bool stats)
{
  // the linearisation consumed here was taken at st->T; with the fused path its residuals were never written
  for(int i = 0; i < 16; ++i) st->T_lin[i] = st->T[i];
  const bool fused_lin = fuse_frozen && !(st->delta_scale > 1e-6f);
  st->r_stale = fused_lin ? 1 : 0;
  const bool again = gn_logic(st, s_nrm, s_sum, scratch, mode, max_iterations, max_fun_evals, p_tol, f_tol, g_tol_param);
  (void) again;   // who is still active is read from st->active (compact_active_kernel once per host round / the persistent loop)
  if(stats && mode == 0 && j.trace) {
    // bpvo_hip_estimate_pose_trace: the linearisation just consumed (pose, system, function value, scale, valid count) and the step
    // solved from it; record layout: BPVO_HIP_TRACE_FLOATS in c_api.h
    if(st->trace_n < j.trace_cap) {
      float* o = j.trace + (size_t) st->trace_n * kTraceFloats;
      for(int i = 0; i < 16; ++i) o[i] = st->T_lin[i];
      for(int i = 0; i < 36; ++i) o[16 + i] = st->H[i];
      for(int i = 0; i < 6; ++i) { o[52 + i] = st->G[i]; o[61 + i] = st->dp[i]; }
      o[58] = st->f_norm; o[59] = st->scale; o[60] = (float) st->n_valid; o[67] = (float) st->level;
    }
    st->trace_n += 1;
  }
  if(stats) {
    j.cnt[0] += (unsigned long long) j.n;     // measurement: points and linearisations processed (bench.py roofline)
    j.cnt[1] += 1ull;
    if(fused_lin) {                           // the fused path keeps its own tap-cache statistics (the others: median_finish)
      j.cnt[5] += (unsigned long long) s_sum[29]; j.cnt[6] += j.tapcache_on ? (unsigned long long) s_sum[28] : 0ull;
      j.cnt[10] += (unsigned long long) j.n;
    }
  }
}

// The step of one workspace by ONE wavefront (lanes 0..63 of the calling workgroup's first wave; nobody else may touch `s`): the state
// lives in HBM between launches, the serial bookkeeping runs on an LDS copy (global-memory round trips would otherwise dominate:
// every field access is a dependent ~1 us load).  Lanes 0 .. kNumAcc-1 sum the partials, lane 0 runs the serial step.  COHERENT: the
// partials were stored by other workgroups of the launch that is still running (gn_last_tile) and are read past the caches.
struct GNStepLds {
  uint32_t state[sizeof(GNState) / sizeof(uint32_t)];
  float sum[kPartialStride];
  float nrm[5];
  SolveScratch scratch;
};
static_assert(sizeof(GNState) % sizeof(uint32_t) == 0, "GNState must be word sized");
__device__ __forceinline__ void wave_lds_sync()      // LDS written by some lanes of the wave, read by others: program order is enough for the
{                                                    // hardware (one wave's LDS operations complete in order), the compiler must keep it
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool COHERENT>
__device__ __forceinline__ void gn_step_wave(const PairJob& j, GNStepLds& s, int pts_per_block, int mode, const GNParams& prm, int fuse_frozen)
{
  constexpr int kWords = (int) (sizeof(GNState) / sizeof(uint32_t));
  const int lane = threadIdx.x;
  GNState* gst = j.st;
  for(int i = lane; i < kWords; i += 64) s.state[i] = reinterpret_cast<const uint32_t*>(gst)[i];
  if(lane < 4) s.nrm[lane] = j.nrm[lane];
  if(lane == 4) s.nrm[4] = j.dspace ? 1.0f : 0.0f;
  gn_sum_partials<COHERENT>(j, pts_per_block, lane, s.sum, j.partials);
  wave_lds_sync();
  if(lane == 0)
    gn_serial_step(j, reinterpret_cast<GNState*>(s.state), s.nrm, s.sum, &s.scratch, mode, prm.max_iterations, prm.max_fun_evals, prm.p_tol, prm.f_tol,
                   prm.g_tol, fuse_frozen, true);
  wave_lds_sync();
  for(int i = lane; i < kWords; i += 64) reinterpret_cast<uint32_t*>(gst)[i] = s.state[i];
}

// K8 + K9 in one launch.  The first wave of every tile of a workspace, once its partial is stored THROUGH the caches (irls_tile,
// agent_store) and acknowledged, draws a ticket; the wave that draws the LAST one has all partials of the workspace in reach — read
// past the caches, gn_sum_partials<true> — and takes the step: summed in tile order as ever, so the result does not depend on which
// tile that is.  No fences: an agent-scope release writes back the whole L2 of the XCD and an acquire invalidates it, once per tile —
// measured, the launch took twice as long (profiles/r04_step_in_reduce.txt).  Called by the first wave only (the others are done);
// true for the wave that takes the step.  `tiles`: tickets of the workspace in this launch.
__device__ __forceinline__ bool gn_last_tile(const PairJob& j, int tiles)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the partial has arrived where the other XCDs read it
  unsigned ticket = 0;
  if(threadIdx.x == 0) ticket = __hip_atomic_fetch_add(j.ticket.p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket = (unsigned) __builtin_amdgcn_readfirstlane((int) ticket);
  if(ticket != (unsigned) (tiles - 1)) return false;
  if(threadIdx.x == 0) __hip_atomic_store(j.ticket.p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
  return true;
}

}  // namespace bpvo_hip
