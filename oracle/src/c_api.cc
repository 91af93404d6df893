// ORACLE — test infrastructure only (see orc.h).  C ABI (oracle/bpvo_oracle.h) over the C++ restatement.
#include "../bpvo_oracle.h"
#include "orc.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <exception>
#include <string>

using namespace orc;

struct bpvo_orc_ctx {
  Params params;
  float K[9];
  float baseline;
  int rows, cols;
  int nthreads = 1;
  std::vector<std::unique_ptr<Frame>> frames;
  std::vector<VoPoseEstimator> ws;
  std::vector<int> ws_ref_slot, ws_level;   // what the last linearize of a workspace ran on
  VisualOdometry vo;
  Result last_result;
  uint64_t total_lin = 0;
  std::string err;
};

static std::string g_create_err;

#define ORC_TRY(ctx_) try {
#define ORC_CATCH(ctx_)                                     \
  }                                                          \
  catch(const std::exception& e) {                           \
    (ctx_)->err = e.what();                                  \
    return -1;                                               \
  }                                                          \
  return 0;

static int fail(bpvo_orc_ctx* c, const char* msg, int code = -1)
{
  c->err = msg;
  return code;
}

static M44 toM44(const float* T) { M44 m; std::memcpy(m.m, T, sizeof(m.m)); return m; }

extern "C" {

void bpvo_orc_default_params(void* p) { defaultParams(*reinterpret_cast<Params*>(p)); }

int bpvo_orc_create(bpvo_orc_ctx** out, const float K[9], float baseline, int rows, int cols, const void* params,
                    int, int n_frames, int n_pairs)
{
  try {
    auto* c = new bpvo_orc_ctx;
    c->params = *reinterpret_cast<const Params*>(params);
    std::memcpy(c->K, K, sizeof(c->K));
    c->baseline = baseline;
    c->rows = rows;
    c->cols = cols;
    if(c->params.numPyramidLevels <= 0)   // bpvo/vo.cc:101-105
      c->params.numPyramidLevels =
          1 + (int) std::round(std::log2(std::min(rows, cols) / (double) c->params.minImageDimensionForPyramid));
    if(c->params.maxTestLevel < 0 || c->params.maxTestLevel >= c->params.numPyramidLevels) {
      g_create_err = "invalid maxTestLevel";
      delete c;
      return -1;
    }
    for(int i = 0; i < n_frames; ++i) {
      c->frames.emplace_back(new Frame);
      c->frames.back()->init(K, baseline, rows, cols, c->params);
    }
    c->ws.resize(n_pairs);
    c->ws_ref_slot.assign(n_pairs, -1);
    c->ws_level.assign(n_pairs, -1);
    for(auto& w : c->ws) w.init(c->params, 1);
    c->vo.init(K, baseline, rows, cols, c->params, 1);
    *out = c;
    return 0;
  } catch(const std::exception& e) {
    g_create_err = e.what();
    return -1;
  }
}

void bpvo_orc_destroy(bpvo_orc_ctx* ctx) { delete ctx; }
const char* bpvo_orc_last_error(const bpvo_orc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int bpvo_orc_set_num_threads(bpvo_orc_ctx* c, int n)
{
  c->nthreads = std::max(1, n);
  for(auto& f : c->frames) f->nthreads = c->nthreads;
  for(auto& w : c->ws) w.est.nthreads = c->nthreads;
  c->vo.vo_pose.est.nthreads = c->nthreads;
  for(Frame* f : {c->vo.ref.get(), c->vo.cur.get(), c->vo.prev.get()}) f->nthreads = c->nthreads;
  return 0;
}

// 0: the reference's f32 accumulation of the normal equations; 1: the same terms accumulated in f64 (test instrument, orc.h)
int bpvo_orc_set_reduction(bpvo_orc_ctx* c, int mode)
{
  for(auto& w : c->ws) w.est.reduction = mode;
  c->vo.vo_pose.est.reduction = mode;
  return 0;
}
// test instrument (orc.h PoseEstimator::perturb_rel): relative noise on every linearisation's (H, G); rel = 0 switches it off
int bpvo_orc_set_perturbation(bpvo_orc_ctx* c, int seed, double rel)
{
  for(auto& w : c->ws) { w.est.perturb_rel = (float) rel; w.est.perturb_seed = (uint32_t) seed; w.est.perturb_count = 0; }
  c->vo.vo_pose.est.perturb_rel = (float) rel; c->vo.vo_pose.est.perturb_seed = (uint32_t) seed; c->vo.vo_pose.est.perturb_count = 0;
  return 0;
}
int bpvo_orc_set_warp_formulation(bpvo_orc_ctx* c, int mode)
{
  if(mode != 0 && mode != 1 && mode != 2) return fail(c, "unknown warp formulation");
  // mode 2: DisparitySpaceWarp as the warp of TemplateData (points, Jacobians, pose update) + the all-f32 formulation;
  // templates built before the switch hold points of the other warp and must be rebuilt by the caller
  for(auto& f : c->frames)
    for(auto& td : f->tdata) { td.fast_warp = mode; td.warp.dspace = (mode == 2); }
  for(Frame* f : {c->vo.ref.get(), c->vo.cur.get(), c->vo.prev.get()})
    for(auto& td : f->tdata) { td.fast_warp = mode; td.warp.dspace = (mode == 2); }
  return 0;
}

int bpvo_orc_num_levels(const bpvo_orc_ctx* c) { return c->params.numPyramidLevels; }
int bpvo_orc_num_channels(const bpvo_orc_ctx* c)
{
  switch(c->params.descriptor) {
    case kBitPlanes: return 8;
    case kIntensityAndGradient: return 3;
    case kDescriptorFieldsFirstOrder: return 5;
    case kDescriptorFieldsSecondOrder: return 10;
    case kCentralDifference: return (2 * c->params.centralDifferenceRadius + 1) * (2 * c->params.centralDifferenceRadius + 1) - 1;
    case kLatch: return 8 * c->params.latchNumBytes;
    default: return 1;
  }
}
int bpvo_orc_level_size(const bpvo_orc_ctx* c, int level, int* rows, int* cols)
{
  int r = c->rows, w = c->cols;
  for(int i = 0; i < level; ++i) { r = (r + 1) / 2; w = (w + 1) / 2; }
  *rows = r; *cols = w;
  return 0;
}

#define CHECK_SLOT(c, s) if((s) < 0 || (s) >= (int) (c)->frames.size()) return fail(c, "bad frame slot")
#define CHECK_LEVEL(c, l) if((l) < (c)->params.maxTestLevel || (l) >= (c)->params.numPyramidLevels) return fail(c, "bad level")
#define CHECK_WS(c, w) if((w) < 0 || (w) >= (int) (c)->ws.size()) return fail(c, "bad workspace")

int bpvo_orc_frame_set_data(bpvo_orc_ctx* c, int slot, const uint8_t* image, const float* disparity)
{
  CHECK_SLOT(c, slot);
  if(!image || !disparity) return fail(c, "nullptr image/disparity");
  ORC_TRY(c) c->frames[slot]->setData(image, disparity); ORC_CATCH(c)
}
int bpvo_orc_frame_set_template(bpvo_orc_ctx* c, int slot)
{
  CHECK_SLOT(c, slot);
  if(!c->frames[slot]->has_data) return fail(c, "no data in frame", -3);
  ORC_TRY(c) c->frames[slot]->setTemplate(); ORC_CATCH(c)
}
int bpvo_orc_frame_clear(bpvo_orc_ctx* c, int slot) { CHECK_SLOT(c, slot); c->frames[slot]->clear(); return 0; }
int bpvo_orc_frame_state(const bpvo_orc_ctx* c, int slot, int* has_data, int* has_template)
{
  if(slot < 0 || slot >= (int) c->frames.size()) return -1;
  *has_data = c->frames[slot]->has_data;
  *has_template = c->frames[slot]->has_template;
  return 0;
}
int bpvo_orc_frames_set_data(bpvo_orc_ctx* c, int first, int stride, int count, const uint8_t* images,
                             const float* disps, int)
{
  const size_t n = (size_t) c->rows * c->cols;
  for(int i = 0; i < count; ++i) {
    int rc = bpvo_orc_frame_set_data(c, first + i * stride, images + i * n, disps + i * n);
    if(rc) return rc;
  }
  return 0;
}
int bpvo_orc_frames_set_template(bpvo_orc_ctx* c, int first, int stride, int count)
{
  for(int i = 0; i < count; ++i) {
    int rc = bpvo_orc_frame_set_template(c, first + i * stride);
    if(rc) return rc;
  }
  return 0;
}

int bpvo_orc_get_image(bpvo_orc_ctx* c, int slot, int level, uint8_t* out)
{
  CHECK_SLOT(c, slot);
  if(level < 0 || level >= c->params.numPyramidLevels) return fail(c, "bad level");
  Frame& f = *c->frames[slot];
  if(!f.has_data) return fail(c, "no data", -3);
  std::memcpy(out, f.pyr[level].data(), f.pyr[level].size());
  return 0;
}
int bpvo_orc_get_descriptor_channel(bpvo_orc_ctx* c, int slot, int level, int channel, float* out)
{
  CHECK_SLOT(c, slot); CHECK_LEVEL(c, level);
  Frame& f = *c->frames[slot];
  if(!f.has_data) return fail(c, "no data", -3);
  if(channel < 0 || channel >= f.desc[level].numChannels()) return fail(c, "bad channel");
  std::memcpy(out, f.desc[level].ch[channel].data(), f.desc[level].ch[channel].size() * sizeof(float));
  return 0;
}
int bpvo_orc_get_saliency(bpvo_orc_ctx* c, int slot, int level, float* out)
{
  CHECK_SLOT(c, slot); CHECK_LEVEL(c, level);
  Frame& f = *c->frames[slot];
  if(!f.has_template) return fail(c, "no template", -4);
  std::memcpy(out, f.tdata[level].saliency.data(), f.tdata[level].saliency.size() * sizeof(float));
  return 0;
}
#define TD(c, slot, level)                                            \
  CHECK_SLOT(c, slot); CHECK_LEVEL(c, level);                         \
  if(!(c)->frames[slot]->has_template) return fail(c, "no template", -4); \
  TemplateData& td = (c)->frames[slot]->tdata[level]

int bpvo_orc_num_points(bpvo_orc_ctx* c, int slot, int level, int* n) { TD(c, slot, level); *n = td.numPoints(); return 0; }
int bpvo_orc_get_points(bpvo_orc_ctx* c, int slot, int level, float* xyzw)
{ TD(c, slot, level); std::memcpy(xyzw, td.points.data(), td.points.size() * sizeof(float)); return 0; }
int bpvo_orc_get_point_indices(bpvo_orc_ctx* c, int slot, int level, int* inds)
{ TD(c, slot, level); std::memcpy(inds, td.inds.data(), td.inds.size() * sizeof(int)); return 0; }
int bpvo_orc_get_pixels(bpvo_orc_ctx* c, int slot, int level, float* pixels)
{ TD(c, slot, level); std::memcpy(pixels, td.pixels.data(), td.pixels.size() * sizeof(float)); return 0; }
int bpvo_orc_get_jacobians(bpvo_orc_ctx* c, int slot, int level, float* J)
{ TD(c, slot, level); std::memcpy(J, td.jacobians.data(), td.jacobians.size() * sizeof(float)); return 0; }
int bpvo_orc_get_normalization(bpvo_orc_ctx* c, int slot, int level, float T[16], float T_inv[16])
{ TD(c, slot, level); std::memcpy(T, td.warp.T.m, 64); std::memcpy(T_inv, td.warp.T_inv.m, 64); return 0; }

int bpvo_orc_linearize(bpvo_orc_ctx* c, int ws, int ref_slot, int cur_slot, int level, const float T[16],
                       int reset_scale, float H[36], float G[6], float* f_norm, float* sigma, int* num_valid)
{
  CHECK_WS(c, ws); CHECK_SLOT(c, ref_slot); CHECK_SLOT(c, cur_slot); CHECK_LEVEL(c, level);
  Frame& ref = *c->frames[ref_slot];
  Frame& cur = *c->frames[cur_slot];
  if(!ref.has_template) return fail(c, "no template", -4);
  if(!cur.has_data) return fail(c, "no data", -3);
  ORC_TRY(c)
  PoseEstimator& est = c->ws[ws].est;
  if(reset_scale) est.scale_estimator.reset();
  *f_norm = est.linearize(&ref.tdata[level], cur.desc[level], toM44(T), H, G);
  *sigma = est.last_sigma;
  *num_valid = est.last_num_valid;
  c->ws_ref_slot[ws] = ref_slot;
  c->ws_level[ws] = level;
  c->total_lin += 1;
  ORC_CATCH(c)
}

// residuals / weights are [C*N] channel-major; valid is handed back as the un-replicated [N] vector
int bpvo_orc_get_residuals(bpvo_orc_ctx* c, int ws, float* r, size_t* n)
{
  CHECK_WS(c, ws);
  const auto& v = c->ws[ws].est.residuals;
  if(n) *n = v.size();
  if(r) std::memcpy(r, v.data(), v.size() * sizeof(float));
  return 0;
}
int bpvo_orc_get_valid(bpvo_orc_ctx* c, int ws, uint16_t* v, size_t* n)
{
  CHECK_WS(c, ws);
  const auto& est = c->ws[ws].est;
  const int C = bpvo_orc_num_channels(c);
  const size_t N = est.valid.size() == est.residuals.size() ? est.valid.size() / C : est.valid.size();
  if(n) *n = N;
  if(v) std::memcpy(v, est.valid.data(), N * sizeof(uint16_t));
  return 0;
}
int bpvo_orc_get_weights(bpvo_orc_ctx* c, int ws, float* w, size_t* n)
{
  CHECK_WS(c, ws);
  const auto& v = c->ws[ws].est.weights;
  if(n) *n = v.size();
  if(w) std::memcpy(w, v.data(), v.size() * sizeof(float));
  return 0;
}
int bpvo_orc_fraction_good(bpvo_orc_ctx* c, int ws, float threshold, float* frac)
{
  CHECK_WS(c, ws);
  *frac = c->ws[ws].getFractionOfGoodPoints(threshold);
  return 0;
}

int bpvo_orc_estimate_pose_trace(bpvo_orc_ctx* c, int ws, int ref_slot, int cur_slot, const float T_init[16],
                                 float T_est[16], bpvo_orc_stats* stats, float* records, int max_records, int* n_records)
{
  CHECK_WS(c, ws); CHECK_SLOT(c, ref_slot); CHECK_SLOT(c, cur_slot);
  Frame& ref = *c->frames[ref_slot];
  Frame& cur = *c->frames[cur_slot];
  if(!ref.has_template) return fail(c, "no template", -4);
  if(!cur.has_data) return fail(c, "no data", -3);
  ORC_TRY(c)
  VoPoseEstimator& vp = c->ws[ws];
  std::vector<IterationRecord> trace;
  std::vector<int> levels;
  vp.est.trace = records ? &trace : nullptr;
  // VisualOdometryPoseEstimator::estimatePose (bpvo/vo_pose_estimator.cc:63-93), unrolled to tag records by level
  const int L = c->params.numPyramidLevels;
  Stats def = {0, -1.0f, -1.0f, kSolverError};
  std::vector<Stats> st(L, def);
  M44 T = toM44(T_init);
  for(int i = L - 1; i >= c->params.maxTestLevel; --i) {
    const size_t before = trace.size();
    st[i] = vp.est.run(&ref.tdata[i], cur.desc[i], T);
    c->total_lin += vp.est.num_fun_evals;
    for(size_t k = before; k < trace.size(); ++k) levels.push_back(i);
  }
  vp.est.trace = nullptr;
  c->ws_ref_slot[ws] = ref_slot;
  c->ws_level[ws] = c->params.maxTestLevel;
  std::memcpy(T_est, T.m, 64);
  for(int i = 0; i < L; ++i) {
    stats[i].numIterations = st[i].numIterations;
    stats[i].finalError = st[i].finalError;
    stats[i].firstOrderOptimality = st[i].firstOrderOptimality;
    stats[i].status = st[i].status;
  }
  if(records) {
    const int n = (int) std::min<size_t>(trace.size(), (size_t) max_records);
    for(int k = 0; k < n; ++k) {
      float* o = records + (size_t) k * BPVO_ORC_TRACE_FLOATS;
      std::memcpy(o, trace[k].T.m, 64);
      std::memcpy(o + 16, trace[k].H, 144);
      std::memcpy(o + 52, trace[k].G, 24);
      o[58] = trace[k].f_norm;
      o[59] = trace[k].sigma;
      o[60] = (float) trace[k].num_valid;
      std::memcpy(o + 61, trace[k].dp, 24);
      o[67] = (float) levels[k];
    }
    if(n_records) *n_records = (int) trace.size();
  }
  ORC_CATCH(c)
}

int bpvo_orc_estimate_pose(bpvo_orc_ctx* c, int ws, int ref_slot, int cur_slot, const float T_init[16], float T_est[16],
                           bpvo_orc_stats* stats)
{
  return bpvo_orc_estimate_pose_trace(c, ws, ref_slot, cur_slot, T_init, T_est, stats, nullptr, 0, nullptr);
}

int bpvo_orc_add_frame(bpvo_orc_ctx* c, const uint8_t* image, const float* disparity, bpvo_orc_result* result)
{
  if(!image || !disparity) return fail(c, "nullptr image/disparity");   // bpvo/vo.cc:68-69
  ORC_TRY(c)
  Result& r = c->last_result;
  c->vo.addFrame(image, disparity, r);
  std::memcpy(result->pose, r.pose.m, 64);
  std::memcpy(result->covariance, r.covariance, sizeof(r.covariance));
  result->numLevels = (int) r.stats.size();
  for(int i = 0; i < result->numLevels && i < 8; ++i) {
    result->optimizerStatistics[i].numIterations = r.stats[i].numIterations;
    result->optimizerStatistics[i].finalError = r.stats[i].finalError;
    result->optimizerStatistics[i].firstOrderOptimality = r.stats[i].firstOrderOptimality;
    result->optimizerStatistics[i].status = r.stats[i].status;
  }
  result->isKeyFrame = r.isKeyFrame;
  result->keyFramingReason = r.keyFramingReason;
  result->hasPointCloud = r.hasPointCloud;
  ORC_CATCH(c)
}
int bpvo_orc_vo_num_points_at_level(bpvo_orc_ctx* c, int level, int* n)
{
  if(level < 0) level = c->vo.params.maxTestLevel;
  *n = c->vo.ref->tdata[level].numPoints();
  return 0;
}
int bpvo_orc_vo_points_at_level(bpvo_orc_ctx* c, int level, float* xyzw)
{
  if(level < 0) level = c->vo.params.maxTestLevel;
  const auto& p = c->vo.ref->tdata[level].points;
  std::memcpy(xyzw, p.data(), p.size() * sizeof(float));
  return 0;
}
int bpvo_orc_get_point_cloud(bpvo_orc_ctx* c, bpvo_orc_point_with_info* pts, size_t* n, float pose[16])
{
  const Result& r = c->last_result;
  if(n) *n = r.cloud.size();
  if(pts) std::memcpy(pts, r.cloud.data(), r.cloud.size() * sizeof(PointWithInfo));
  if(pose) std::memcpy(pose, r.cloudPose.m, 64);
  return 0;
}
int bpvo_orc_trajectory_size(bpvo_orc_ctx* c, int* n) { *n = (int) c->vo.trajectory.size(); return 0; }
int bpvo_orc_get_trajectory(bpvo_orc_ctx* c, float* poses)
{
  for(size_t i = 0; i < c->vo.trajectory.size(); ++i) std::memcpy(poses + 16 * i, c->vo.trajectory[i].m, 64);
  return 0;
}

int bpvo_orc_batch_estimate(bpvo_orc_ctx* c, int n_pairs, const float* T_init, float* poses, bpvo_orc_stats* stats)
{
  const int L = c->params.numPyramidLevels;
  const M44 I = identity44();
  for(int p = 0; p < n_pairs; ++p) {
    int rc = bpvo_orc_estimate_pose(c, p, 2 * p, 2 * p + 1, T_init ? T_init + 16 * p : I.m, poses + 16 * p, stats + (size_t) p * L);
    if(rc) return rc;
  }
  return 0;
}
int bpvo_orc_batch_run(bpvo_orc_ctx* c, int n_pairs, const uint8_t* images, const float* disps, int, float* poses,
                       bpvo_orc_stats* stats)
{
  if(2 * n_pairs > (int) c->frames.size() || n_pairs > (int) c->ws.size()) return fail(c, "batch exceeds ctx capacity");
  int rc = bpvo_orc_frames_set_data(c, 0, 1, 2 * n_pairs, images, disps, 0);
  if(rc) return rc;
  rc = bpvo_orc_frames_set_template(c, 0, 2, n_pairs);
  if(rc) return rc;
  return bpvo_orc_batch_estimate(c, n_pairs, nullptr, poses, stats);
}
int bpvo_orc_total_linearizations(bpvo_orc_ctx* c, uint64_t* n) { *n = c->total_lin; return 0; }

int bpvo_orc_pyrdown_u8(const uint8_t* src, int rows, int cols, uint8_t* dst)
{
  std::vector<uint8_t> d; int dr, dc;
  pyrDownU8(src, rows, cols, d, dr, dc);
  std::memcpy(dst, d.data(), d.size());
  return 0;
}
int bpvo_orc_census(const uint8_t* src, int rows, int cols, float sigma_ct, uint8_t* dst)
{ census(src, rows, cols, sigma_ct, dst); return 0; }
int bpvo_orc_gaussian_f32(const float* src, int rows, int cols, int ksize, float sigma, float* dst)
{
  try { gaussianBlurF32(src, rows, cols, ksize, sigma, dst); } catch(const std::exception&) { return -1; }
  return 0;
}
int bpvo_orc_gaussian_u8(const uint8_t* src, int rows, int cols, int ksize, float sigma, uint8_t* dst)
{
  try { gaussianBlurU8(src, rows, cols, ksize, sigma, dst); } catch(const std::exception&) { return -1; }
  return 0;
}
int bpvo_orc_imsmooth_taps(float sigma) { return imsmoothTaps(sigma); }
int bpvo_orc_auto_gauss_taps_f32(float sigma) { return autoGaussTapsF32(sigma); }
int bpvo_orc_gaussian5x5_f32(const float* src, int rows, int cols, float sigma, float* dst)
{ gaussianBlurF32_5x5(src, rows, cols, sigma, dst); return 0; }
float bpvo_orc_median(const float* data, size_t n)
{ std::vector<float> v(data, data + n); return medianOf(v); }
int bpvo_orc_solve(const float H[36], const float G[6], float dp[6]) { return solveSystem(H, G, dp) ? 1 : 0; }
void bpvo_orc_twist_to_matrix(const float p[6], float T[16]) { M44 m = twistToMatrix(p); std::memcpy(T, m.m, 64); }
// StereoAlgorithm::run (BlockMatching): params = {preFilterCap, SADWindowSize, minDisparity, numberOfDisparities, textureThreshold, uniquenessRatio}
int bpvo_orc_stereo_bm(const uint8_t* left, const uint8_t* right, int rows, int cols, const int params[6], float* dmap)
{
  StereoParams sp;
  sp.preFilterCap = params[0]; sp.SADWindowSize = params[1]; sp.minDisparity = params[2]; sp.numberOfDisparities = params[3];
  sp.textureThreshold = params[4]; sp.uniquenessRatio = params[5];
  if(sp.SADWindowSize < 5 || sp.SADWindowSize > 255 || sp.SADWindowSize % 2 == 0 || sp.numberOfDisparities <= 0 || sp.numberOfDisparities % 16 != 0 ||
     sp.preFilterCap < 1 || sp.preFilterCap > 63) return 1;     // the argument checks of cvFindStereoCorrespondenceBM
  stereoBM(left, right, rows, cols, sp, dmap);
  return 0;
}
int bpvo_orc_stereo_prefilter(const uint8_t* src, int rows, int cols, int cap, uint8_t* dst) { stereoPrefilterXSobel(src, rows, cols, cap, dst); return 0; }
// StereoAlgorithm::run (SemiGlobalMatching): iparams = {numberOfDisparities, sobelCapValue, censusRadius, windowRadius, smoothnessPenaltySmall,
// smoothnessPenaltyLarge, consistencyThreshold}, dparams = {disparityFactor, censusWeightFactor}; 1 = arguments the original throws on
int bpvo_orc_stereo_sgm(const uint8_t* left, const uint8_t* right, int rows, int cols, const int iparams[7], const double dparams[2], float* dmap)
{
  SgmParams sp;
  sp.numberOfDisparities = iparams[0]; sp.sobelCapValue = iparams[1]; sp.censusRadius = iparams[2]; sp.windowRadius = iparams[3];
  sp.smoothnessPenaltySmall = iparams[4]; sp.smoothnessPenaltyLarge = iparams[5]; sp.consistencyThreshold = iparams[6];
  sp.disparityFactor = dparams[0]; sp.censusWeightFactor = dparams[1];
  return stereoSGM(left, right, rows, cols, sp, dmap) ? 0 : 1;
}
// StereoAlgorithm::run (SemiGlobalBlockMatching): params = the cv::StereoSGBM fields {minDisparity, numberOfDisparities, SADWindowSize, P1, P2,
// disp12MaxDiff, preFilterCap, uniquenessRatio, speckleWindowSize, speckleRange, fullDP}; 1 = arguments outside what is restated
int bpvo_orc_stereo_sgbm(const uint8_t* left, const uint8_t* right, int rows, int cols, const int params[11], float* dmap)
{
  SgbmParams sp;
  sp.minDisparity = params[0]; sp.numberOfDisparities = params[1]; sp.SADWindowSize = params[2]; sp.P1 = params[3]; sp.P2 = params[4];
  sp.disp12MaxDiff = params[5]; sp.preFilterCap = params[6]; sp.uniquenessRatio = params[7]; sp.speckleWindowSize = params[8];
  sp.speckleRange = params[9]; sp.fullDP = params[10];
  return stereoSGBM(left, right, rows, cols, sp, dmap) ? 0 : 1;
}
// MEstimator::ComputeWeights on raw arrays (r [n], valid [n] u16 -> w [n]); n a multiple of 16 runs the SIMD body only
int bpvo_orc_compute_weights(int loss, const float* r, const uint16_t* valid, size_t n, float sigma, float* w)
{
  std::vector<float> rv(r, r + n), wv;
  std::vector<uint16_t> vv(valid, valid + n);
  computeWeights(loss, rv, vv, sigma, wv);
  std::memcpy(w, wv.data(), n * sizeof(float));
  return 0;
}

}  // extern "C"
