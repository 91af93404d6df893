// libbpvo_hip, host side: create / destroy, frame and workspace storage, job tables, events, estimation lanes, per-context options.
#include "host_ctx.h"

using namespace bpvo_hip;
using namespace bpvo_hip_host;

namespace bpvo_hip_host {

std::atomic<int> g_live_ctx[64];
thread_local std::string g_create_error;

int fail(bpvo_hip_ctx* c, int code, const char* msg)
{
  c->err = msg;
  return code;
}


// cv::getGaussianKernel(5, sigma, CV_32F) (OpenCV 2.4 smooth.cpp; reference call site bpvo/bitplanes_descriptor.cc:56):
// exp in double, stored as float, normalised by the double sum of the floats.
void gaussian_kernel5(double sigma, float k[3])
{
  float kk[5];
  const double sigmaX = sigma > 0 ? sigma : ((5 - 1) * 0.5 - 1) * 0.3 + 0.8;
  const double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for(int i = 0; i < 5; ++i) {
    const double x = i - 2.0;
    kk[i] = (float) std::exp(scale2X * x * x);
    sum += kk[i];
  }
  sum = 1. / sum;
  for(int i = 0; i < 5; ++i) kk[i] = (float) (kk[i] * sum);
  k[0] = kk[2]; k[1] = kk[3]; k[2] = kk[4];
}

// cv::getGaussianKernel(n, sigma, CV_32F) for sigma > 0 and the 8-bit fixed-point taps cvRound(k * 256) of the u8 filters
void gaussian_taps(int n, double sigma, GaussTaps* g)
{
  *g = GaussTaps();
  if(!(sigma > 0) || n <= 0 || n > kMaxGaussTaps) return;
  g->n = n;
  const double scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for(int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    g->k[i] = (float) std::exp(scale2X * x * x);
    sum += g->k[i];
  }
  sum = 1. / sum;
  for(int i = 0; i < n; ++i) {
    g->k[i] = (float) (g->k[i] * sum);
    g->ki[i] = (int) std::nearbyint((double) g->k[i] * 256.0);
  }
}
// imsmooth (bpvo/imgproc.cc:166-171): max(5, 2 * round(sigma) + 1) taps
int imsmooth_taps(float sigma) { return std::max(5, 2 * (int) std::round((double) sigma) + 1); }
// cv::GaussianBlur(Size(), sigma) on a CV_32F image (OpenCV 2.4 createGaussianFilter): cvRound(sigma * 4 * 2 + 1) | 1
int auto_gauss_taps_f32(float sigma) { return ((int) std::nearbyint((double) sigma * 8.0 + 1.0)) | 1; }

void carve_frame_data(bpvo_hip_ctx* c, FrameSlot& f, unsigned char* base, size_t* total)
{
  Carver cv{base};
  for(int l = 0; l < c->L; ++l) f.img[l] = cv.take<uint8_t>(c->geom[l].npix);
  f.disp = cv.take<float>(c->geom[0].npix);
  for(int l = 0; l < c->L; ++l) f.desc[l] = cv.take<float>(c->geom[l].npix * c->C);
  for(int l = 0; l < c->L; ++l) f.cen[l] = (c->C == 8) ? cv.take<uint8_t>(c->geom[l].npix) : nullptr;
  for(int l = 0; l < c->L; ++l) f.ch0[l] = (c->C == 8) ? cv.take<float>(c->geom[l].npix) : nullptr;
  f.scratch = c->plane_scratch ? cv.take<float>((size_t) c->scratch_planes * c->geom[0].npix) : nullptr;
  if(total) *total = cv.off;
}

void carve_frame_tmpl(bpvo_hip_ctx* c, FrameSlot& f, unsigned char* base, size_t* total)
{
  Carver cv{base};
  for(int l = 0; l < c->L; ++l) {
    const LevelGeom& g = c->geom[l];
    f.sal[l] = cv.take<float>(g.npix);
    const size_t nwords = (size_t) g.rows * ((g.cols + 63) / 64);      // candidate bits of the tiled selection (kernels_frame.hip)
    f.flag[l] = (uint8_t*) cv.take<unsigned long long>((std::max(g.npix, nwords * 8) + 7) / 8);
    f.blk_count[l] = cv.take<int>(std::max((size_t) g.nblk, nwords));
    f.pts[l] = cv.take<float4>(g.cap);
    f.inds[l] = cv.take<int>(g.cap);
    f.pix[l] = cv.take<float>((size_t) g.cap * c->C);
    f.grad[l] = cv.take<float>((size_t) g.cap * c->C * 2);
  }
  f.nrm = cv.take<float>(4 * kMaxLevels);
  f.n_dev = cv.take<int>(kMaxLevels);
  if(total) *total = cv.off;
}

int ensure_template_storage(bpvo_hip_ctx* c, FrameSlot& f)
{
  if(f.tmpl_slab) return BPVO_OK;
  size_t total = 0;
  FrameSlot tmp;
  carve_frame_tmpl(c, tmp, nullptr, &total);
  HIP_CK(c, hipMalloc(&f.tmpl_slab, total));
  HIP_CK(c, hipMemsetAsync(f.tmpl_slab, 0, total, c->stream));
  carve_frame_tmpl(c, f, (unsigned char*) f.tmpl_slab, nullptr);
  return BPVO_OK;
}

FrameJob make_frame_job(bpvo_hip_ctx* c, FrameSlot& f, int l)
{
  const LevelGeom& g = c->geom[l];
  FrameJob j;
  std::memset(&j, 0, sizeof(j));
  j.img = f.img[l];
  j.cen = f.cen[l];
  j.ch0 = f.ch0_valid ? f.ch0[l] : nullptr;
  j.scratch = f.scratch;
  j.desc = f.desc[l];
  j.sal = f.sal[l];
  j.flag = f.flag[l];
  j.words = reinterpret_cast<unsigned long long*>(f.flag[l]);
  j.blk_count = f.blk_count[l];
  j.n_out = f.n_dev ? f.n_dev + l : nullptr;
  j.disp = f.disp;
  j.pts = f.pts[l];
  j.inds = f.inds[l];
  j.pix = f.pix[l];
  j.grad = f.grad[l];
  j.nrm = f.nrm ? f.nrm + 4 * l : nullptr;
  j.rows = g.rows; j.cols = g.cols; j.level = l; j.disp_cols = c->cols;
  j.cap = g.cap;
  j.nms_radius = g.nms_radius;
  std::memcpy(j.K, g.K, sizeof(j.K));
  j.b = g.b;
  j.dspace = c->dspace;
  j.lazy = f.lazy[l] ? 1 : 0;
  return j;
}

PairJob make_pair_job(bpvo_hip_ctx* c, int ws, int ref, int cur, int l)
{
  FrameSlot& fr = c->frames[ref];
  FrameSlot& fc = c->frames[cur];
  const LevelGeom& g = c->geom[l];
  PairJob j;
  std::memset(&j, 0, sizeof(j));
  j.pts = fr.pts[l];
  j.pix = fr.pix[l];
  j.grad = fr.grad[l];
  j.nrm = fr.nrm + 4 * l;
  j.n = fr.n_host[l];
  j.desc = fc.desc[l];
  j.rows = g.rows; j.cols = g.cols;
  std::memcpy(j.K, g.K, sizeof(j.K));
  j.b = g.b;
  j.dspace = c->dspace;
  j.r = c->ws[ws].r;
  j.valid = c->ws[ws].valid;
  j.cand = c->ws[ws].cand;
  j.tapkey = c->ws[ws].tapkey;
  j.tapcache_on = c->ws[ws].tapkey != nullptr;
  j.tapcache = c->ws[ws].tapcache;
  j.med_blk = c->ws[ws].med_blk;
  j.partials = c->ws[ws].partials;
  j.st = c->d_states + ws;
  j.cnt = c->d_counters + kWsCounters * (size_t) ws;
  j.ticket = c->d_tickets + ws;
  if(ws == c->trace_ws) { j.trace = c->d_trace; j.trace_cap = c->trace_cap; }
  j.pitch = c->C;
  j.n_groups = c->G;
  j.med_tot = (int) med_totals_at(c);
  return j;
}

// channel group k (0 .. G-1) of a whole job: the same point set, the descriptor / template / residual records entered at the group's first
// channel, its own run of bracket segments and tile partials behind those of the groups before it
PairJob group_pair_job(const bpvo_hip_ctx* c, const PairJob& whole, int k)
{
  PairJob j = whole;
  const size_t off = (size_t) k * c->Cg;
  const size_t nblk = (size_t) ((whole.n + kChunkPoints - 1) / kChunkPoints);
  const size_t ntiles = (size_t) std::max(1, (whole.n + gn_pts_per_block(c->Cg) - 1) / gn_pts_per_block(c->Cg));
  j.desc = whole.desc.get() + off;
  j.pix = whole.pix.get() + off;
  j.grad = whole.grad.get() + off;
  j.r = whole.r.get() + off;
  j.cand = whole.cand.get() + (size_t) k * nblk * kChunkPoints * c->Cg;
  j.med_blk = whole.med_blk.get() + (size_t) k * nblk * 4;
  j.partials = whole.partials.get() + (size_t) k * ntiles * kPartialStride;
  j.n_groups = 1;
  return j;
}

// ---- measurement: HIP events on the ctx stream around kernel classes ------------------------------------------------
hipEvent_t take_event(Lane* ln)
{
  if(!ln->ev_pool.empty()) {
    hipEvent_t e = ln->ev_pool.back();
    ln->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void) hipEventCreate(&e);
  return e;
}
void resolve_events(bpvo_hip_ctx* c)   // call from the API thread after the lanes' streams are synchronised
{
  for(auto& ln : c->lanes) {
    for(auto& ep : ln.ev_pending) {
      float ms = 0.0f;
      if(hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
        c->kc_ms[ep.kc] += ms;
        c->kc_units[ep.kc] += ep.units;
        c->kc_launches[ep.kc] += 1;
      }
      ln.ev_pool.push_back(ep.a);
      ln.ev_pool.push_back(ep.b);
    }
    ln.ev_pending.clear();
  }
}

// ---- estimation lanes: allocated on demand (create: up to two; option "lanes" and the first batch that wants more: the rest) ---------------
// lanes a batch of n pairs fans out over (at most `cap`), allocating those the context does not hold yet; a (negative) error code on failure
int lanes_for(bpvo_hip_ctx* c, int n, int cap)
{
  if(g_live_ctx[c->device & 63].load() > 1) return 1;      // several contexts on the device: one lane each
  const int want = std::max(1, std::min(std::min(c->max_lanes_now, cap), n / kMinPairsPerLane));
  if(want > (int) c->lanes.size()) {
    const int rc = ensure_lanes(c, want);
    if(rc) return rc;
  }
  return want;
}
int ensure_lanes(bpvo_hip_ctx* c, int n)
{
  const int n_pairs = c->n_pairs;
  while((int) c->lanes.size() < n) {
    c->lanes.emplace_back();
    Lane& ln = c->lanes.back();
    const bool first = c->lanes.size() == 1;
    if(first) { ln.stream = c->stream; ln.owns_stream = false; }
    else { HIP_CK(c, hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking)); ln.owns_stream = true; }
    // (wide descriptors: the table of the whole jobs, then one table per channel group)
    HIP_CK(c, hipMalloc((void**) &ln.d_pjobs, sizeof(PairJob) * (size_t) c->L * n_pairs * job_tables(c)));
    HIP_CK(c, hipMalloc((void**) &ln.d_Tinit, sizeof(float) * 16 * n_pairs));
    HIP_CK(c, hipMalloc((void**) &ln.d_active, 8 * sizeof(int)));
    HIP_CK(c, hipMalloc((void**) &ln.d_list, 3 * sizeof(int) * (size_t) n_pairs));
    for(auto& e : ln.round_ev) HIP_CK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_CK(c, hipEventCreateWithFlags(&ln.selected_ev, hipEventDisableTiming));
    for(auto& e : ln.staging_ev) HIP_CK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_CK(c, hipHostMalloc((void**) &ln.h_pjobs, sizeof(PairJob) * std::max((size_t) c->L * n_pairs * job_tables(c), (size_t) (1 + kMaxGroups))));
    HIP_CK(c, hipHostMalloc((void**) &ln.h_T, sizeof(float) * 16 * n_pairs));
    HIP_CK(c, hipHostMalloc((void**) &ln.h_active, 8 * sizeof(int)));
    HIP_CK(c, hipMalloc((void**) &ln.d_pk_ctl, sizeof(unsigned) * kPkCtlWords * kMaxLevels));
    HIP_CK(c, hipHostMalloc((void**) &ln.h_pk_ctl, sizeof(unsigned) * kPkCtlWords * kMaxLevels));
    if(first) {
      HIP_CK(c, hipMalloc((void**) &ln.d_team_ctl, sizeof(unsigned) * (size_t) gn_team_ctl_words(kMaxTeams)));
      HIP_CK(c, hipHostMalloc((void**) &ln.h_team_ctl, sizeof(unsigned) * 64));      // (two launches' first lines: estimate.hip)
      std::memset(ln.h_team_ctl, 0, sizeof(unsigned) * 64);
    }
    HIP_CK(c, hipHostMalloc((void**) &ln.h_states, sizeof(GNState) * n_pairs));
  }
  return BPVO_OK;
}

// ---- per-context options (bpvo_hip_set_option / bpvo_hip_get_option; the table in include/bpvo_hip/c_api.h) -----------------------------
const std::vector<OptionDef>& option_table()
{
#define OPT_INT(key_, field_, lo_, hi_) OptionDef{key_, lo_, hi_, [](bpvo_hip_ctx* c) { return (double) c->field_; }, \
                                                  [](bpvo_hip_ctx* c, double v) { c->field_ = (decltype(c->field_)) v; return BPVO_OK; }}
  static const std::vector<OptionDef> t = {
    // estimation lanes a batch may fan out over (streams driven by host threads); more than the context holds are allocated here
    OptionDef{"lanes", 1, 8, [](bpvo_hip_ctx* c) { return (double) std::max(1, std::min(c->max_lanes_now, std::max(1, c->n_pairs / kMinPairsPerLane))); },
              [](bpvo_hip_ctx* c, double v) {
                const int n = std::max(1, std::min((int) v, std::max(1, c->n_pairs / kMinPairsPerLane)));
                const int rc = ensure_lanes(c, n);
                if(rc == BPVO_OK) c->max_lanes_now = (int) v;
                return rc;
              }},
    // "persistent" = 1 also re-arms a context whose persistent / team launch once gave up at a barrier (persistent_failed is sticky otherwise)
    OptionDef{"persistent", 0, 1, [](bpvo_hip_ctx* c) { return (double) c->persistent; },
              [](bpvo_hip_ctx* c, double v) { c->persistent = (int) v; if(c->persistent) c->persistent_failed.store(false); return BPVO_OK; }},
    OPT_INT("persist_max_ws", persist_max_ws, 1, kPersistMaxWs),
    OPT_INT("persist_grid", persist_grid, 1, 128),
    OPT_INT("persist_max_points", persist_max_points, 0, 1 << 30),
    OPT_INT("dense_candidates_from", dense_candidates_from, 0, 1 << 30),
    OPT_INT("persist_timeout_ticks", persist_timeout, 1, 1e15),
    OPT_INT("team", team_mode, 0, 1),
    OPT_INT("team_max_pairs", team_max_pairs, 0, 1 << 20),
    OPT_INT("team_full_pairs", team_full_pairs, 0, 1 << 20),
    OPT_INT("team_size", team_size_env, 0, 256),
    // CUs the team grid may claim: never more than the device has (a larger grid cannot be co-resident and would only time out)
    OptionDef{"team_cus", 1, 1 << 16, [](bpvo_hip_ctx* c) { return (double) c->num_cus; },
              [](bpvo_hip_ctx* c, double v) {
                if(c->device_cus > 0 && (int) v > c->device_cus) { c->err = "option team_cus: more CUs than the device has"; return BPVO_ERR_INVALID_ARG; }
                c->num_cus = (int) v;
                return BPVO_OK;
              }},
    OPT_INT("team_local_barriers", team_local_barriers, 0, 1),
    OPT_INT("team_join", team_join, 0, 2),
    OptionDef{"team_joins_seen", 0, 0, [](bpvo_hip_ctx* c) { return (double) c->team_joins.load(); },      // read-only counter (bpvo_hip_get_option)
              [](bpvo_hip_ctx* c, double) { c->team_joins.store(0); return BPVO_OK; }},
    OPT_INT("team_join_from_pairs", team_join_from_pairs, 0, 1 << 20),
    OPT_INT("team_spares", team_spares, 0, 1),
    OPT_INT("vo_disparity_late", vo_disparity_late, 0, 1),
    OPT_INT("normalization_side_stream", nrm_side_stream, 0, 1),
    OPT_INT("normalization_form", nrm_dpp_asm, 0, 4),
    OPT_INT("small_batch_fused", small_batch_fused, 0, 1),
    OPT_INT("levels_in_one_launch_max_frames", merge_levels_max_frames, 0, 1 << 20),
    OPT_INT("normalization_deferred", nrm_defer, 0, 1),
    OPT_INT("team_split_max_pairs", team_split_max_pairs, 0, 1 << 20),
    OPT_INT("fuse_frozen", fuse_frozen, 0, 1),
    OPT_INT("reference_reduction", reference_reduction, 0, 1),
    OPT_INT("step_in_reduce_max_pairs", step_in_reduce_max, 0, 1 << 20),
    OPT_INT("stagger", stagger, 0, 1),
    OPT_INT("stagger_min_pairs", stagger_min_pairs, 0, 1 << 20),
    OPT_INT("upload_workers", up_workers, 0, 32),
    OPT_INT("keep_current_disparity", keep_current_disparity, 0, 1),
    OPT_INT("lazy_template_descriptor", lazy_template, 0, 1),
    OptionDef{"tapcache_max_density", 0.0, 1e9, [](bpvo_hip_ctx* c) { return c->tapcache_max_density; },
              [](bpvo_hip_ctx* c, double v) { c->tapcache_max_density = v; return BPVO_OK; }},
    OptionDef{"upload_plan_first", 0.0, 0.9, [](bpvo_hip_ctx* c) { return c->up_plan[0]; },
              [](bpvo_hip_ctx* c, double v) { if(v + c->up_plan[1] >= 1.0) return BPVO_ERR_INVALID_ARG; c->up_plan[0] = v; return BPVO_OK; }},
    OptionDef{"upload_plan_second", 0.01, 0.99, [](bpvo_hip_ctx* c) { return c->up_plan[1]; },
              [](bpvo_hip_ctx* c, double v) { if(v + c->up_plan[0] >= 1.0) return BPVO_ERR_INVALID_ARG; c->up_plan[1] = v; return BPVO_OK; }},
  };
#undef OPT_INT
  return t;
}
int set_option(bpvo_hip_ctx* c, const std::string& key, double v)
{
  for(const OptionDef& o : option_table())
    if(key == o.key) {
      if(!(v >= o.lo && v <= o.hi)) { c->err = "option " + key + ": value out of range"; return BPVO_ERR_INVALID_ARG; }
      const int rc = o.set(c, v);
      if(rc == BPVO_ERR_INVALID_ARG) c->err = "option " + key + ": invalid value";
      return rc;
    }
  c->err = "unknown option: " + key;
  return BPVO_ERR_INVALID_ARG;
}
int apply_options_string(bpvo_hip_ctx* c, const char* str)
{
  std::string s(str);
  size_t i = 0;
  while(i < s.size()) {
    size_t j = s.find(',', i);
    if(j == std::string::npos) j = s.size();
    const std::string kv = s.substr(i, j - i);
    i = j + 1;
    if(kv.empty()) continue;
    const size_t eq = kv.find('=');
    if(eq == std::string::npos) { c->err = "expected key=value: " + kv; return BPVO_ERR_INVALID_ARG; }
    char* end = nullptr;
    const double v = std::strtod(kv.c_str() + eq + 1, &end);
    if(end == kv.c_str() + eq + 1) { c->err = "not a number: " + kv; return BPVO_ERR_INVALID_ARG; }
    const int rc = set_option(c, kv.substr(0, eq), v);
    if(rc) return rc;
  }
  return BPVO_OK;
}

}  // namespace bpvo_hip_host

extern "C" {

void bpvo_hip_default_params(bpvo_hip_params* p)   // AlgorithmParameters() (reference: bpvo/types.cc:31-66)
{
  p->numPyramidLevels = -1;
  p->minImageDimensionForPyramid = 40;
  p->sigmaPriorToCensusTransform = -1.0f;
  p->sigmaBitPlanes = 0.5f;
  p->dfSigma1 = 0.75f;
  p->dfSigma2 = 1.75f;
  p->latchNumBytes = 1;
  p->latchRotationInvariance = 0;
  p->latchHalfSsdSize = 1;
  p->centralDifferenceRadius = 3;
  p->centralDifferenceSigmaBefore = 0.75f;
  p->centralDifferenceSigmaAfter = 1.75f;
  p->laplacianKernelSize = 1;
  p->maxIterations = 50;
  p->parameterTolerance = 1e-7f;
  p->functionTolerance = 1e-6f;
  p->gradientTolerance = 1e-8f;
  p->relaxTolerancesForCoarseLevels = 1;
  p->gradientEstimation = BPVO_GRAD_CD3;
  p->interp = BPVO_INTERP_LINEAR;
  p->lossFunction = BPVO_LOSS_TUKEY;
  p->descriptor = BPVO_DESC_INTENSITY;
  p->verbosity = BPVO_VERB_ITERATION;
  p->minTranslationMagToKeyFrame = 0.15f;
  p->minRotationMagToKeyFrame = 5.0f;
  p->maxFractionOfGoodPointsToKeyFrame = 0.6f;
  p->goodPointThreshold = 0.85f;
  p->minNumPixelsForNonMaximaSuppression = 320 * 240;
  p->nonMaxSuppRadius = 1;
  p->minNumPixelsToWork = 256;
  p->minSaliency = 0.1f;
  p->minValidDisparity = 0.001f;
  p->maxValidDisparity = 512.0f;
  p->maxTestLevel = 0;
  p->withNormalization = 1;
}

int bpvo_hip_create(bpvo_hip_ctx** out, const float K[9], float baseline, int rows, int cols, const bpvo_hip_params* p,
                    int device, int n_frames, int n_pairs)
{
  if(!out || !K || !p || rows < 8 || cols < 8 || n_frames < 1 || n_pairs < 1) {
    g_create_error = "invalid argument";
    return BPVO_ERR_INVALID_ARG;
  }
  // pixel coordinates travel as 16-bit values (selection lists are uint16_t in the reference as well, Q9; the tap-cache key
  // packs (yi << 16 | xi)), and linear pixel indices as int
  if(rows > 65535 || cols > 65535 || (long long) rows * cols * 8 > 0x7fffffffLL) {
    g_create_error = "image too large (at most 65535 x 65535 and 2^28 pixels)";
    return BPVO_ERR_INVALID_ARG;
  }
  int ndev = 0;
  if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device: libbpvo_hip has no CPU fallback";
    return BPVO_ERR_NO_DEVICE;
  }
  if(device < 0 || device >= ndev) {
    g_create_error = "bad device ordinal";
    return BPVO_ERR_INVALID_ARG;
  }
  std::unique_ptr<bpvo_hip_ctx> c(new bpvo_hip_ctx);
  c->params = *p;
  std::memcpy(c->K, K, sizeof(c->K));
  c->baseline = baseline;
  c->rows = rows; c->cols = cols; c->device = device;
  c->n_frames = n_frames; c->n_pairs = n_pairs;
  if(c->params.numPyramidLevels <= 0)   // bpvo/vo.cc:101-105
    c->params.numPyramidLevels = 1 + (int) std::round(std::log2(std::min(rows, cols) / (double) p->minImageDimensionForPyramid));
  c->L = c->params.numPyramidLevels;
  auto unsupported = [&](const char* m) { g_create_error = m; return BPVO_ERR_UNSUPPORTED; };
  if(c->L < 1 || c->L > kMaxLevels) return unsupported("numPyramidLevels out of range (1..8)");
  if(c->params.maxTestLevel < 0 || c->params.maxTestLevel >= c->L) { g_create_error = "invalid maxTestLevel"; return BPVO_ERR_INVALID_ARG; }
  const bool desc_fields = c->params.descriptor == BPVO_DESC_FIELDS_FIRST_ORDER || c->params.descriptor == BPVO_DESC_FIELDS_SECOND_ORDER;
  if(c->params.descriptor != BPVO_DESC_INTENSITY && c->params.descriptor != BPVO_DESC_BITPLANES && c->params.descriptor != BPVO_DESC_LAPLACIAN &&
     c->params.descriptor != BPVO_DESC_INTENSITY_AND_GRADIENT && c->params.descriptor != BPVO_DESC_CENTRAL_DIFFERENCE && c->params.descriptor != BPVO_DESC_LATCH &&
     !desc_fields) {
    g_create_error = "unknown DescriptorType";      // DenseDescriptor::Create's default branch (bpvo/dense_descriptor.cc:86-87)
    return BPVO_ERR_INVALID_ARG;
  }
  if(c->params.descriptor == BPVO_DESC_LATCH) {
    const int nb = c->params.latchNumBytes;
    if(nb != 1 && nb != 2 && nb != 4 && nb != 8 && nb != 16 && nb != 32 && nb != 64) {       // bpvo/latch_descriptor.cc:104
      g_create_error = "descriptorSize must be 1, 2, 4, 8, 16, 32, or 64";
      return BPVO_ERR_INVALID_ARG;
    }
    // (8 - 64 bytes: 64 - 512 channels, in channel groups of 32 through the per-point kernels — types.h PairJob::pitch)
    if(c->params.latchHalfSsdSize < 0 || c->params.latchHalfSsdSize > 8) return unsupported("latchHalfSsdSize: 0 .. 8 are on the device path");
  }
  if(c->params.descriptor == BPVO_DESC_CENTRAL_DIFFERENCE) {
    if(c->params.centralDifferenceRadius <= 0) { g_create_error = "invalid radius"; return BPVO_ERR_INVALID_ARG; }   // central_difference_descriptor.cc:19
    // radius > 3: (2r + 1)^2 - 1 > 48 channels, in channel groups (types.h PairJob::pitch) where the count splits into at most kMaxGroups groups of a
    // channel count the per-point kernels are built for: r = 4 (80 = 5 x 16), 5 (120 = 5 x 24), 6 (168 = 7 x 24), 7 (224 = 7 x 32), 8 (288 = 6 x 48), 9 (360 = 15 x 24)
    if(c->params.centralDifferenceRadius > 9) return unsupported("centralDifferenceRadius: 1 .. 9 (8 .. 360 channels) are on the device path");
    if((c->params.centralDifferenceSigmaBefore > 0.0f && imsmooth_taps(c->params.centralDifferenceSigmaBefore) > kMaxGaussTaps) ||
       (c->params.centralDifferenceSigmaAfter > 0.0f && imsmooth_taps(c->params.centralDifferenceSigmaAfter) > kMaxGaussTaps))
      return unsupported("centralDifferenceSigmaBefore / After: imsmooth kernels of up to 31 taps (sigma < 15.5) are on the device path");
  }
  if(desc_fields) {   // imsmooth (bpvo/imgproc.cc:166-171): max(5, 2*round(sigma)+1) taps
    if((c->params.dfSigma1 > 0.0f && imsmooth_taps(c->params.dfSigma1) > kMaxGaussTaps) ||
       (c->params.dfSigma2 > 0.0f && imsmooth_taps(c->params.dfSigma2) > kMaxGaussTaps))
      return unsupported("dfSigma1 / dfSigma2: imsmooth kernels of up to 31 taps (sigma < 15.5) are on the device path");
  }
  if(c->params.descriptor == BPVO_DESC_INTENSITY_AND_GRADIENT && c->params.sigmaPriorToCensusTransform > 0.0f) {
    const int k = auto_gauss_taps_f32(c->params.sigmaPriorToCensusTransform);   // cv::GaussianBlur(Size(), sigma): automatic kernel size
    if(k < 5 || k > kMaxGaussTaps)
      return unsupported("IntensityAndGradient: pre-smoothing kernels of 5 to 31 taps (0.44 <= sigmaPriorToCensusTransform <= 3.8) are on the device path");
  }
  if(c->params.descriptor == BPVO_DESC_LAPLACIAN && c->params.laplacianKernelSize != 1 && c->params.laplacianKernelSize != 3 &&
     c->params.laplacianKernelSize != 5 && c->params.laplacianKernelSize != 7)
    return unsupported("laplacianKernelSize: 1, 3, 5 and 7 are on the device path (from 11 on OpenCV's f32 sums are no longer exact integers)");
  if(c->params.interp < BPVO_INTERP_LINEAR || c->params.interp > BPVO_INTERP_CUBIC_HERMITE) return unsupported("unknown interp");
  if(c->params.lossFunction != BPVO_LOSS_HUBER && c->params.lossFunction != BPVO_LOSS_TUKEY && c->params.lossFunction != BPVO_LOSS_L2)
    return unsupported("unknown lossFunction");
  if(c->params.gradientEstimation != BPVO_GRAD_CD3 && c->params.gradientEstimation != BPVO_GRAD_CD5) return unsupported("unknown gradientEstimation");
  switch(c->params.descriptor) {
    case BPVO_DESC_BITPLANES: c->C = 8; break;
    case BPVO_DESC_INTENSITY_AND_GRADIENT: c->C = 3; break;
    case BPVO_DESC_FIELDS_FIRST_ORDER: c->C = 5; break;
    case BPVO_DESC_FIELDS_SECOND_ORDER: c->C = 10; break;
    case BPVO_DESC_CENTRAL_DIFFERENCE: c->C = (2 * c->params.centralDifferenceRadius + 1) * (2 * c->params.centralDifferenceRadius + 1) - 1; break;
    case BPVO_DESC_LATCH: c->C = 8 * c->params.latchNumBytes; break;
    default: c->C = 1; break;
  }
  c->G = 1; c->Cg = 0;
  if(c->C > 48) {
    // a wide descriptor: the largest group size the per-point kernels are instantiated for that divides C into at most kMaxGroups groups
    for(int cg : {48, 32, 24, 16}) {
      if(c->C % cg == 0 && c->C / cg <= kMaxGroups) { c->Cg = cg; c->G = c->C / cg; break; }
    }
    if(c->G == 1) return unsupported("descriptor channel count does not split into channel groups of 16 / 24 / 32 / 48");
  }
  const bool grad_smoothed = c->params.descriptor == BPVO_DESC_INTENSITY_AND_GRADIENT && c->params.sigmaPriorToCensusTransform > 0.0f;
  // (LATCH keeps its [key points][bytes] buffer behind three work planes: kernels_planes.hip)
  // (... and both wide descriptors eight more: the row passes of a group of eight channels, kernels_planes.hip df_col8_kernel)
  c->scratch_planes = c->params.descriptor == BPVO_DESC_LATCH ? std::max(kDfPlanes, 3 + (c->params.latchNumBytes + 3) / 4 + 8)
                      : c->params.descriptor == BPVO_DESC_CENTRAL_DIFFERENCE ? std::max(kDfPlanes, 3 + 8) : (c->C == 5 || c->C == 10) ? kDfPlanes + 8 : kDfPlanes;
  c->plane_scratch = c->C == 5 || c->C == 10 || c->params.descriptor == BPVO_DESC_CENTRAL_DIFFERENCE || c->params.descriptor == BPVO_DESC_LATCH || grad_smoothed;
  if(c->params.descriptor == BPVO_DESC_CENTRAL_DIFFERENCE) {
    gaussian_taps(imsmooth_taps(c->params.centralDifferenceSigmaBefore), c->params.centralDifferenceSigmaBefore, &c->cd_before);
    gaussian_taps(imsmooth_taps(c->params.centralDifferenceSigmaAfter), c->params.centralDifferenceSigmaAfter, &c->cd_after);
  }
  if(desc_fields) {
    gaussian_taps(imsmooth_taps(c->params.dfSigma1), c->params.dfSigma1, &c->df_g1);
    gaussian_taps(imsmooth_taps(c->params.dfSigma2), c->params.dfSigma2, &c->df_g2);
  }
  if(grad_smoothed) gaussian_taps(auto_gauss_taps_f32(c->params.sigmaPriorToCensusTransform), c->params.sigmaPriorToCensusTransform, &c->grad_pre);
  gaussian_kernel5(c->params.sigmaBitPlanes, c->gauss_k);
  auto gauss3_fixed = [](double sg, int taps[2]) {      // cv::getGaussianKernel(3, sigma) in f32, then cvRound(k * 256)
    const double scale2X = -0.5 / (sg * sg);
    float kk[3];
    double sum = 0;
    for(int i = 0; i < 3; ++i) { const double x = i - 1.0; kk[i] = (float) std::exp(scale2X * x * x); sum += kk[i]; }
    sum = 1. / sum;
    for(int i = 0; i < 3; ++i) kk[i] = (float) (kk[i] * sum);
    taps[0] = (int) std::nearbyint((double) kk[1] * 256.0);
    taps[1] = (int) std::nearbyint((double) kk[2] * 256.0);
  };
  if(c->params.sigmaPriorToCensusTransform > 0.0f) gauss3_fixed(c->params.sigmaPriorToCensusTransform, c->census_taps);
  if(c->params.descriptor == BPVO_DESC_LATCH) {
    gauss3_fixed(2.0, c->latch_taps);
    gaussian_taps(imsmooth_taps(1.75f), 1.75f, &c->latch_after);
  }

  // level geometry (bpvo/vo_frame.cc:21-28: K *= 0.5, K(2,2) = 1, b *= 2; pyrDown sizes)
  {
    int r = rows, w = cols;
    float Kp[9];
    std::memcpy(Kp, K, sizeof(Kp));
    float bp = baseline;
    for(int l = 0; l < c->L; ++l) {
      if(l > 0) {
        r = (r + 1) / 2; w = (w + 1) / 2;
        for(int k = 0; k < 9; ++k) Kp[k] *= 0.5f;
        Kp[8] = 1.0f;
        bp *= 2.0f;
      }
      LevelGeom& g = c->geom[l];
      g.rows = r; g.cols = w; g.npix = (size_t) r * w;
      g.nblk = (int) ((g.npix + 255) / 256);
      const bool nms = (r * w >= c->params.minNumPixelsForNonMaximaSuppression) && c->params.nonMaxSuppRadius > 0;   // template_data.cc:43-49
      g.nms_radius = nms ? c->params.nonMaxSuppRadius : -1;
      // strict local maxima: at most one per 2x2 block (two adjacent pixels cannot both be strict maxima)
      const size_t cap = nms ? (size_t) ((r + 1) / 2) * ((w + 1) / 2) : g.npix;
      g.cap = (int) ((cap + kTile - 1) / kTile * kTile);   // whole 64-point tiles (tiled per-point layout, types.h)
      std::memcpy(g.K, Kp, sizeof(Kp));
      g.b = bp;
      c->cap_max = std::max(c->cap_max, g.cap);
      if(r < 8 || w < 8) return unsupported("pyramid level smaller than 8 pixels");
    }
  }

  bpvo_hip_ctx* cp = c.get();
  auto dev_fail = [&](hipError_t e, const char* what) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(e);
    bpvo_hip_destroy(c.release());   // frees whatever was allocated so far (a failed create must not leak device memory)
    return BPVO_ERR_DEVICE;
  };
#define CREATE_CK(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) return dev_fail(e_, #expr); } while(0)
  CREATE_CK(hipSetDevice(device));
  CREATE_CK(hipStreamCreateWithFlags(&cp->stream, hipStreamNonBlocking));
  cp->frames.resize(n_frames);
  size_t data_total = 0;
  { FrameSlot tmp; carve_frame_data(cp, tmp, nullptr, &data_total); }
  for(auto& f : cp->frames) {
    CREATE_CK(hipMalloc(&f.data_slab, data_total));
    carve_frame_data(cp, f, (unsigned char*) f.data_slab, nullptr);
  }
  // sequential-VO contexts (up to 3 slots) get their template storage now: allocated on first use it is a ~10 ms hiccup on
  // the frame that switches keyframes; batch contexts keep it lazy (only every other slot of a pair batch is a template)
  if(n_frames <= 3)
    for(auto& f : cp->frames)
      if(ensure_template_storage(cp, f) != BPVO_OK) return dev_fail(hipErrorOutOfMemory, "template storage");
  cp->ws.resize(n_pairs);
  const size_t nblk_max = (size_t) gn_num_blocks(cp->cap_max);
  // the 4 x 4 tap cache of kCubic / kCubicHermite (512 B per point for eight channels) is an optimisation, not a requirement: a context whose
  // workspaces would spend more than a third of the free memory on it runs without (the kernel gathers every footprint from the descriptor)
  bool tap_cache = cp->C == 8 || cp->C == 1;
  {
    const bool wide_taps = cp->params.interp == BPVO_INTERP_CUBIC || cp->params.interp == BPVO_INTERP_CUBIC_HERMITE;
    size_t free_b = 0, total_b = 0;
    if(tap_cache && wide_taps && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const size_t need = (size_t) n_pairs * sizeof(float) * 16 * cp->C * (((size_t) cp->cap_max + kTile - 1) / kTile * kTile);
      if(need > free_b / 3) tap_cache = false;
    }
  }
  for(auto& w : cp->ws) {
    CREATE_CK(hipMalloc((void**) &w.r, sizeof(float) * (size_t) cp->cap_max * cp->C));
    CREATE_CK(hipMalloc((void**) &w.valid, (size_t) cp->cap_max));
    // (channel groups: a group's candidate segments and bracket counters follow those of the group before it, whole 256-point chunks each)
    // (+ kDenseRuns chunks: the dense form's runs are whole multiples of a chunk; + kDenseRuns lines of totals behind the chunks' counters)
    CREATE_CK(hipMalloc((void**) &w.cand, sizeof(uint32_t) * (nblk_max + bpvo_hip::kDenseRuns) * kChunkPoints * (size_t) cp->C));
    CREATE_CK(hipMalloc((void**) &w.med_blk, sizeof(uint32_t) * 4 * (med_totals_at(cp) + 8 * (size_t) bpvo_hip::kDenseRuns)));
    CREATE_CK(hipMemset(w.med_blk + 4 * med_totals_at(cp), 0, sizeof(uint32_t) * 4 * 8 * bpvo_hip::kDenseRuns));      // (every finish leaves the totals at zero again)
    if(tap_cache) {      // tap cache of warp_residual: the footprint's taps x C floats per point (2 x 2; kCubic / kCubicHermite: 4 x 4), whole tiles
      const bool wide_taps = cp->params.interp == BPVO_INTERP_CUBIC || cp->params.interp == BPVO_INTERP_CUBIC_HERMITE;
      const size_t cap_tiles = ((size_t) cp->cap_max + kTile - 1) / kTile * kTile;
      CREATE_CK(hipMalloc((void**) &w.tapkey, sizeof(uint32_t) * (size_t) cp->cap_max));
      CREATE_CK(hipMalloc((void**) &w.tapcache, sizeof(float) * (wide_taps ? 16 : 4) * cp->C * cap_tiles));
    }
    // two buffers of tile partials (the persistent kernels double-buffer them by iteration parity, kernels_gn.hip pk_partials)
    CREATE_CK(hipMalloc((void**) &w.partials, sizeof(float) * (size_t) gn_partials_entries(cp->cap_max, cp->G > 1 ? cp->Cg : cp->C) * std::max(1, (cp->G + 1) / 2) * kPartialStride));
  }
  CREATE_CK(hipMalloc((void**) &cp->d_states, sizeof(GNState) * n_pairs));
  CREATE_CK(hipMemset(cp->d_states, 0, sizeof(GNState) * n_pairs));
  CREATE_CK(hipMalloc((void**) &cp->d_fjobs, 2 * sizeof(FrameJob) * (size_t) cp->L * n_frames));
  CREATE_CK(hipMalloc((void**) &cp->d_job1, sizeof(PairJob) * (1 + kMaxGroups)));      // the whole job + its channel groups (wide descriptors)
  if(cp->params.descriptor == BPVO_DESC_LATCH) {
    // The triplet coordinates as CalcuateSums uses them (bpvo/latch_descriptor.cc:170-236): the table's, or — latchRotationInvariance —
    // rotated by the key point's angle and clamped to the patch.  The dense evaluation builds its key points with cv::KeyPoint() (:135-141),
    // angle -1, so the rotation is one and the same for every pixel: angle = -1 * (float)(CV_PI / 180.f), cos / sin of that float
    // (:259-262), (int)((float) ax * cos - (float) ay * sin) (:193-200).
    const int n_ints = 48 * cp->params.latchNumBytes;
    std::vector<signed char> off(n_ints);
    const float angle = -1.0f * (float) (3.1415926535897932384626433832795 / 180.f);
    const float cos_theta = std::cos(angle), sin_theta = std::sin(angle);
    for(int t = 0; t < n_ints; t += 2) {
      int x = kLatchTable[t], y = kLatchTable[t + 1];
      if(cp->params.latchRotationInvariance) {
        const int xr = (int) (((float) x) * cos_theta - ((float) y) * sin_theta), yr = (int) (((float) x) * sin_theta + ((float) y) * cos_theta);
        x = std::max(-24, std::min(24, xr));
        y = std::max(-24, std::min(24, yr));
      }
      off[t] = (signed char) x; off[t + 1] = (signed char) y;
    }
    CREATE_CK(hipMalloc((void**) &cp->d_latch_off, (size_t) n_ints));
    CREATE_CK(hipMemcpy(cp->d_latch_off, off.data(), (size_t) n_ints, hipMemcpyHostToDevice));
  }
  {
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, device) == hipSuccess) cp->num_cus = cp->device_cus = prop.multiProcessorCount;
  }
  cp->max_lanes_now = cp->C == 8 ? kDefaultLanes : kDefaultLanesNarrow;
  cp->persist_max_points = cp->C == 8 ? 32768 : 65536;
  // BPVO_HIP_OPTIONS="key=value,key=value": bpvo_hip_set_option applied to every context the process creates (measurement scripts and
  // tests; a drop-in caller uses the function) — the library's one environment variable
  if(const char* e = std::getenv("BPVO_HIP_OPTIONS")) {
    const int rc = apply_options_string(cp, e);
    if(rc) { g_create_error = "BPVO_HIP_OPTIONS: " + cp->err; return rc; }
  }
  {
    // (a third lane — the default for eight channels — gets its stream when a batch first wants it: a context that only ever runs
    // host-buffer batches, which stay on their two-lane upload plan, then holds three streams + the copy stream, not four + one: the
    // streams of a process share four hardware queues by default, and a fifth busy one made the host-buffer step 1.33 x instead of 1.15 x)
    const int rc = ensure_lanes(cp, std::max(1, std::min(std::min(cp->max_lanes_now, 2), n_pairs / kMinPairsPerLane)));
    if(rc) { g_create_error = cp->err; return rc; }
  }
  CREATE_CK(hipMalloc((void**) &cp->d_records, sizeof(float) * kRecordFloats * n_pairs));
  CREATE_CK(hipMalloc((void**) &cp->d_wtmp, sizeof(float) * (size_t) cp->cap_max * cp->C));
  CREATE_CK(hipMalloc((void**) &cp->d_count, sizeof(unsigned int)));
  CREATE_CK(hipMalloc((void**) &cp->d_tickets, sizeof(unsigned) * n_pairs));
  CREATE_CK(hipMemset(cp->d_tickets, 0, sizeof(unsigned) * n_pairs));
  CREATE_CK(hipMalloc((void**) &cp->d_counters, kWsCounters * sizeof(unsigned long long) * n_pairs));
  CREATE_CK(hipMemset(cp->d_counters, 0, kWsCounters * sizeof(unsigned long long) * n_pairs));
  CREATE_CK(hipHostMalloc((void**) &cp->h_fjobs, 2 * sizeof(FrameJob) * (size_t) cp->L * n_frames));
  CREATE_CK(hipHostMalloc((void**) &cp->h_ints, sizeof(int) * std::max((size_t) n_frames * kMaxLevels, (size_t) 16)));
  CREATE_CK(hipMalloc((void**) &cp->d_ints, sizeof(int) * std::max((size_t) n_frames * kMaxLevels, (size_t) 16)));
#undef CREATE_CK
  cp->T_kf = m44_identity();
  cp->cloud_pose = m44_identity();
  g_live_ctx[device & 63].fetch_add(1);
  cp->counted_live = true;
  *out = c.release();
  return BPVO_OK;
}

void bpvo_hip_destroy(bpvo_hip_ctx* c)
{
  if(!c) return;
  if(c->counted_live) g_live_ctx[c->device & 63].fetch_sub(1);
  (void) hipSetDevice(c->device);
  if(c->stream) (void) hipStreamSynchronize(c->stream);
  for(auto& f : c->frames) { (void) hipFree(f.data_slab); (void) hipFree(f.tmpl_slab); }
  for(auto& w : c->ws) { (void) hipFree(w.r); (void) hipFree(w.valid); (void) hipFree(w.cand); (void) hipFree(w.med_blk); (void) hipFree(w.tapkey); (void) hipFree(w.tapcache); (void) hipFree(w.partials); }
  (void) hipFree(c->d_states); (void) hipFree(c->d_fjobs); (void) hipFree(c->d_job1); (void) hipFree(c->d_latch_off); (void) hipFree(c->d_cloud);
  (void) hipFree(c->d_records); (void) hipFree(c->d_wtmp);
  (void) hipFree(c->d_count); (void) hipFree(c->d_counters); (void) hipFree(c->d_tickets); (void) hipFree(c->d_trace);
  (void) hipFree(c->st_left); (void) hipFree(c->st_right); (void) hipFree(c->st_left_pre); (void) hipFree(c->st_right_pre); (void) hipFree(c->st_disp);
  (void) hipFree(c->st_sgm);
  for(auto st : c->up_streams) if(st) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); }
  for(auto p : c->up_pinned) (void) hipHostFree(p);
  for(auto e : c->up_slot_free) if(e) (void) hipEventDestroy(e);
  for(auto e : c->up_chunk_done) if(e) (void) hipEventDestroy(e);
  (void) hipFree(c->up_d_img); (void) hipFree(c->up_d_disp);
  (void) hipHostFree(c->h_fjobs); (void) hipHostFree(c->h_ints); (void) hipFree(c->d_ints);
  for(auto& ln : c->lanes) {
    if(ln.stream) (void) hipStreamSynchronize(ln.stream);
    (void) hipFree(ln.d_pjobs); (void) hipFree(ln.d_Tinit); (void) hipFree(ln.d_active); (void) hipFree(ln.d_list);
    (void) hipHostFree(ln.h_pjobs); (void) hipHostFree(ln.h_T); (void) hipHostFree(ln.h_active); (void) hipHostFree(ln.h_states);
    (void) hipFree(ln.d_pk_ctl); (void) hipHostFree(ln.h_pk_ctl);
    (void) hipFree(ln.d_team_ctl); (void) hipHostFree(ln.h_team_ctl);
    for(auto& ep : ln.ev_pending) { (void) hipEventDestroy(ep.a); (void) hipEventDestroy(ep.b); }
    for(auto e : ln.ev_pool) (void) hipEventDestroy(e);
    for(auto e : ln.round_ev) if(e) (void) hipEventDestroy(e);
    if(ln.selected_ev) (void) hipEventDestroy(ln.selected_ev);
    for(auto e : ln.staging_ev) if(e) (void) hipEventDestroy(e);
    if(ln.owns_stream && ln.stream) (void) hipStreamDestroy(ln.stream);
  }
  if(c->side_stream) { (void) hipStreamSynchronize(c->side_stream); (void) hipStreamDestroy(c->side_stream); }
  if(c->side_stream2) { (void) hipStreamSynchronize(c->side_stream2); (void) hipStreamDestroy(c->side_stream2); }
  if(c->copy_stream) { (void) hipStreamSynchronize(c->copy_stream); (void) hipStreamDestroy(c->copy_stream); }
  if(c->copy_ev) (void) hipEventDestroy(c->copy_ev);
  for(auto e : c->side_ev) if(e) (void) hipEventDestroy(e);
  if(c->stream) (void) hipStreamDestroy(c->stream);
  delete c;
}

const char* bpvo_hip_last_error(const bpvo_hip_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
int bpvo_hip_num_levels(const bpvo_hip_ctx* c) { return c ? c->L : 0; }
int bpvo_hip_num_channels(const bpvo_hip_ctx* c) { return c ? c->C : 0; }
int bpvo_hip_level_size(const bpvo_hip_ctx* c, int level, int* rows, int* cols)
{
  if(!c || level < 0 || level >= c->L) return BPVO_ERR_INVALID_ARG;
  *rows = c->geom[level].rows; *cols = c->geom[level].cols;
  return BPVO_OK;
}

int bpvo_hip_set_max_lanes(bpvo_hip_ctx* c, int n)      // = bpvo_hip_set_option(c, "lanes", n), kept for callers of round 2
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  return set_option(c, "lanes", (double) (n <= 0 ? (c->C == 8 ? kDefaultLanes : kDefaultLanesNarrow) : std::min(8, n)));
}
int bpvo_hip_set_option(bpvo_hip_ctx* c, const char* key, double value)
{
  CHECK_CTX(c);
  if(!key) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr key");
  (void) hipSetDevice(c->device);
  return set_option(c, key, value);
}
int bpvo_hip_get_option(bpvo_hip_ctx* c, const char* key, double* value)
{
  CHECK_CTX(c);
  if(!key || !value) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr argument");
  for(const OptionDef& o : option_table())
    if(std::string(key) == o.key) { *value = o.get(c); return BPVO_OK; }
  return fail(c, BPVO_ERR_INVALID_ARG, "unknown option");
}

}  // extern "C"
