// libbpvo_hip, host side: VisualOdometry::addFrame (reference: bpvo/vo.cc:125-224) on device-resident frame slots, point cloud, trajectory, and the stereo
// front-end (block matching, SGM) that feeds it.
#include "host_ctx.h"

using namespace bpvo_hip;
using namespace bpvo_hip_host;

extern "C" {

// ---- VisualOdometry -------------------------------------------------------------------------------------------------
static int should_keyframe(bpvo_hip_ctx* c, const M44& pose, int* reason)   // reference: bpvo/vo.cc:199-224
{
  const bpvo_hip_params& p = c->params;
  const float t_norm = pose.m[3] * pose.m[3] + pose.m[7] * pose.m[7] + pose.m[11] * pose.m[11];
  if(t_norm > p.minTranslationMagToKeyFrame * p.minTranslationMagToKeyFrame) { *reason = BPVO_KF_LARGE_TRANSLATION; return BPVO_OK; }
  // math::RotationMatrixToEulerAngles (bpvo/math_utils.h:203-216); compared in radians (Q17)
  const float R00 = pose.m[0], R10 = pose.m[4], R20 = pose.m[8], R21 = pose.m[9];
  const float eta = (float) (1.0 / (std::sqrt(R00 * R00 + R10 * R10)));
  const float rz = std::asin(eta * R10), ry = std::asin(-R20), rx = std::asin(eta * R21);
  const float r_norm = rx * rx + ry * ry + rz * rz;
  if(r_norm > p.minRotationMagToKeyFrame * p.minRotationMagToKeyFrame) { *reason = BPVO_KF_LARGE_ROTATION; return BPVO_OK; }
  float frac = 0.0f;
  int rc = fraction_good(c, 0, p.goodPointThreshold, &frac);
  if(rc) return rc;
  *reason = (frac < p.maxFractionOfGoodPointsToKeyFrame) ? BPVO_KF_SMALL_FRAC_GOOD : BPVO_KF_NO_KEYFRAMING;
  return BPVO_OK;
}

// getPointCloudFromRefFrame + GetColor (reference: bpvo/vo.cc:250-281)
static int build_point_cloud(bpvo_hip_ctx* c)
{
  // One kernel over the template points of the level the estimate ended on; nothing crosses the bus here: the 32-byte records wait in HBM for
  // bpvo_hip_get_point_cloud (the reference's Result owns its cloud; here the caller fetches it — vo.hpp does, into the Result's vector).
  // (Before: every channel's weights + the points + the image copied out and a host loop over the points — 5 - 9 ms per key frame of a dense
  // 640 x 480 template, more than the estimate itself.)
  const int lvl = c->params.maxTestLevel;
  FrameSlot& ref = c->frames[c->vo_ref];
  const int n = ref.n_host[lvl];
  Workspace& w = c->ws[0];
  if(w.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  if(n > c->frames[w.last_ref].n_host[w.last_level]) return fail(c, BPVO_ERR_INVALID_ARG, "size mismatch");
  int rc = ensure_residuals(c, 0);      // (fused path: the residual buffer may lag behind the last linearisation)
  if(rc) return rc;
  rc = upload_single_job(c, 0, w.last_ref, w.last_cur, w.last_level);
  if(rc) return rc;
  if((size_t) n > c->d_cloud_cap) {
    HIP_CK(c, hipStreamSynchronize(c->stream));
    (void) hipFree(c->d_cloud);
    c->d_cloud = nullptr; c->d_cloud_cap = 0;
    const size_t cap = std::max<size_t>((size_t) c->geom[lvl].cap, (size_t) n);
    HIP_CK(c, hipMalloc((void**) &c->d_cloud, cap * sizeof(bpvo_hip_point_with_info)));
    c->d_cloud_cap = cap;
  }
  launch_point_cloud(c->stream, c->d_job1, n, c->C, c->params.lossFunction, ref.img[0], c->rows, c->cols, c->geom[lvl].K, c->dspace, c->d_cloud);
  HIP_CK(c, hipGetLastError());
  c->cloud_n = (size_t) n;
  return BPVO_OK;
}

// ---- stereo front-end (reference: utils/stereo_algorithm.cc:63-82,98-111 -> OpenCV 2.4 cvFindStereoCorrespondenceBM) ----
static int stereo_check(bpvo_hip_ctx* c, const bpvo_hip_stereo_params* sp)
{
  // the argument checks of cvFindStereoCorrespondenceBM (stereobm.cpp) + what the kernel serves
  if(!sp) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr stereo parameters");
  if(sp->algorithm == BPVO_STEREO_SGM) {
    // the checks of SgmStereo::compute / the SGMStereo setters (utils/sgm.cc:168-171,208-254)
    if(sp->numberOfDisparities <= 0 || sp->numberOfDisparities % 16) return fail(c, BPVO_ERR_INVALID_ARG, "numberOfDisparities must be a multiple of 16");
    if(sp->censusRadius < 1 || sp->censusRadius > 2) return fail(c, BPVO_ERR_INVALID_ARG, "window radius of Census transform must be 1 or 2");
    if(sp->censusWeightFactor < 0) return fail(c, BPVO_ERR_INVALID_ARG, "weight of Census transform must be positive");
    if(sp->smoothnessPenaltySmall < 0 || sp->smoothnessPenaltyLarge < 0) return fail(c, BPVO_ERR_INVALID_ARG, "smoothness penalty value is less than zero");
    if(sp->smoothnessPenaltySmall >= sp->smoothnessPenaltyLarge) return fail(c, BPVO_ERR_INVALID_ARG, "small value of smoothness penalty must be smaller than large penalty value");
    if(sp->consistencyThreshold < 0) return fail(c, BPVO_ERR_INVALID_ARG, "threshold for LR consistency must be positive");
    if(!(sp->disparityFactor > 0)) return fail(c, BPVO_ERR_INVALID_ARG, "disparity factor is less than zero");
    if(sp->numberOfDisparities > 256) return fail(c, BPVO_ERR_UNSUPPORTED, "SGM: numberOfDisparities <= 256 are on the device path");
    // (2r+1)^2 * 255 must stay below 2^15: the original's sliding sums are int16 saturating additions (_mm_adds_epi16) and the cost is read
    // back as int16; up to radius 5 (121 * 255 = 30855) nothing saturates and the plain integer sums of the device path are the same numbers
    if(sp->windowRadius < 0 || sp->windowRadius > 5 || c->rows <= sp->windowRadius) return fail(c, BPVO_ERR_UNSUPPORTED, "SGM: windowRadius 0..5 (and fewer than image rows) are on the device path");
    // int16 path costs: the sums of four paths stay clear of saturation for penalties below this (the original saturates silently)
    if(sp->smoothnessPenaltyLarge > 4000) return fail(c, BPVO_ERR_UNSUPPORTED, "SGM: smoothnessPenaltyLarge <= 4000 on the device path");
    return BPVO_OK;
  }
  if(sp->algorithm == BPVO_STEREO_SGBM) {
    if(sp->numberOfDisparities <= 0 || sp->numberOfDisparities % 16) return fail(c, BPVO_ERR_INVALID_ARG, "numberOfDisparities must be a positive multiple of 16");   // CV_Assert(D % 16 == 0)
    SgbmLaunch g = {};
    g.rows = c->rows; g.cols = c->cols;
    g.min_disp = sp->minDisparity; g.ndisp = sp->numberOfDisparities; g.sad_window = sp->SADWindowSize; g.P1 = sp->P1; g.P2 = sp->P2;
    g.disp12_max_diff = sp->disp12MaxDiff; g.pre_filter_cap = sp->preFilterCap; g.uniqueness_ratio = sp->uniquenessRatio;
    g.speckle_window = sp->speckleWindowSize; g.speckle_range = sp->speckleRange; g.full_dp = sp->fullDP;
    const char* why = nullptr;
    if(!sgbm_serves(g, &why)) return fail(c, BPVO_ERR_UNSUPPORTED, why);
    return BPVO_OK;
  }
  if(sp->algorithm != BPVO_STEREO_BLOCK_MATCHING) return fail(c, BPVO_ERR_UNSUPPORTED, "StereoAlgorithm: BlockMatching, SGM and SGBM are on the device path (RSGM is GPL-gated in the reference and not built)");
  if(sp->preFilterCap < 1 || sp->preFilterCap > 63) return fail(c, BPVO_ERR_INVALID_ARG, "preFilterCap must be within 1..63");
  if(sp->SADWindowSize < 5 || sp->SADWindowSize > 255 || sp->SADWindowSize % 2 == 0 || sp->SADWindowSize >= std::min(c->cols, c->rows))
    return fail(c, BPVO_ERR_INVALID_ARG, "SADWindowSize must be odd, be within 5..255 and be not larger than image width or height");
  if(sp->numberOfDisparities <= 0 || sp->numberOfDisparities % 16 != 0) return fail(c, BPVO_ERR_INVALID_ARG, "numberOfDisparities must be positive and divisble by 16");
  if(sp->textureThreshold < 0) return fail(c, BPVO_ERR_INVALID_ARG, "texture threshold must be non-negative");
  if(sp->uniquenessRatio < 0) return fail(c, BPVO_ERR_INVALID_ARG, "uniqueness ratio must be non-negative");
  if(sp->SADWindowSize > 21) return fail(c, BPVO_ERR_UNSUPPORTED, "SADWindowSize: 5..21 are on the device path");
  if(sp->minDisparity < 0 || sp->numberOfDisparities > 256) return fail(c, BPVO_ERR_UNSUPPORTED, "minDisparity >= 0 and numberOfDisparities <= 256 are on the device path");
  return BPVO_OK;
}
static int stereo_reserve(bpvo_hip_ctx* c, int count)
{
  if(count <= c->st_frames) return BPVO_OK;
  HIP_CK(c, hipStreamSynchronize(c->stream));
  (void) hipFree(c->st_left); (void) hipFree(c->st_right); (void) hipFree(c->st_left_pre); (void) hipFree(c->st_right_pre); (void) hipFree(c->st_disp);
  c->st_left = c->st_right = c->st_left_pre = c->st_right_pre = nullptr; c->st_disp = nullptr; c->st_frames = 0;
  const size_t npix = c->geom[0].npix * (size_t) count;
  HIP_CK(c, hipMalloc((void**) &c->st_left, npix)); HIP_CK(c, hipMalloc((void**) &c->st_right, npix));
  HIP_CK(c, hipMalloc((void**) &c->st_left_pre, npix)); HIP_CK(c, hipMalloc((void**) &c->st_right_pre, npix));
  HIP_CK(c, hipMalloc((void**) &c->st_disp, npix * sizeof(float)));
  c->st_frames = count;
  return BPVO_OK;
}
// disparities of `count` rectified pairs into c->st_disp (device); d_left: where the left images are on the device afterwards
static int stereo_run(bpvo_hip_ctx* c, int count, const uint8_t* left, const uint8_t* right, bool on_device, const bpvo_hip_stereo_params* sp,
                      const uint8_t** d_left)
{
  int rc = stereo_check(c, sp);
  if(rc) return rc;
  if(count <= 0 || !left || !right) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr image");
  rc = stereo_reserve(c, count);
  if(rc) return rc;
  const size_t npix = c->geom[0].npix * (size_t) count;
  const uint8_t* dl = left;
  const uint8_t* dr = right;
  if(!on_device) {
    HIP_CK(c, hipMemcpyAsync(c->st_left, left, npix, hipMemcpyHostToDevice, c->stream));
    HIP_CK(c, hipMemcpyAsync(c->st_right, right, npix, hipMemcpyHostToDevice, c->stream));
    dl = c->st_left; dr = c->st_right;
  }
  if(sp->algorithm == BPVO_STEREO_SGM) {
    const size_t need = sgm_scratch_bytes(c->rows, c->cols, sp->numberOfDisparities);
    if(need > c->st_sgm_bytes) {
      HIP_CK(c, hipStreamSynchronize(c->stream));
      (void) hipFree(c->st_sgm);
      c->st_sgm = nullptr; c->st_sgm_bytes = 0;
      HIP_CK(c, hipMalloc(&c->st_sgm, need));
      c->st_sgm_bytes = need;
    }
    SgmLaunch g;
    g.left = dl; g.right = dr; g.disp = c->st_disp; g.scratch = c->st_sgm;
    g.rows = c->rows; g.cols = c->cols; g.nframes = count;
    g.ndisp = sp->numberOfDisparities; g.sobel_cap = sp->sobelCapValue; g.census_radius = sp->censusRadius; g.window_radius = sp->windowRadius;
    g.P1 = sp->smoothnessPenaltySmall; g.P2 = sp->smoothnessPenaltyLarge; g.consistency_threshold = sp->consistencyThreshold;
    g.disparity_factor = sp->disparityFactor; g.census_weight = sp->censusWeightFactor;
    if(!launch_stereo_sgm(c->stream, g)) return fail(c, BPVO_ERR_UNSUPPORTED, "semi-global matching: disparity range not served by the kernels");
    HIP_CK(c, hipGetLastError());
    if(d_left) *d_left = dl;
    return BPVO_OK;
  }
  if(sp->algorithm == BPVO_STEREO_SGBM) {
    const size_t need = sgbm_scratch_bytes(c->rows, c->cols, sp->minDisparity, sp->numberOfDisparities);
    if(need > c->st_sgm_bytes) {
      HIP_CK(c, hipStreamSynchronize(c->stream));
      (void) hipFree(c->st_sgm);
      c->st_sgm = nullptr; c->st_sgm_bytes = 0;
      HIP_CK(c, hipMalloc(&c->st_sgm, need));
      c->st_sgm_bytes = need;
    }
    SgbmLaunch g = {};
    g.left = dl; g.right = dr; g.disp = c->st_disp; g.scratch = c->st_sgm;
    g.rows = c->rows; g.cols = c->cols; g.nframes = count;
    g.min_disp = sp->minDisparity; g.ndisp = sp->numberOfDisparities; g.sad_window = sp->SADWindowSize; g.P1 = sp->P1; g.P2 = sp->P2;
    g.disp12_max_diff = sp->disp12MaxDiff; g.pre_filter_cap = sp->preFilterCap; g.uniqueness_ratio = sp->uniquenessRatio;
    g.speckle_window = sp->speckleWindowSize; g.speckle_range = sp->speckleRange; g.full_dp = sp->fullDP;
    if(!launch_stereo_sgbm(c->stream, g)) return fail(c, BPVO_ERR_UNSUPPORTED, "semi-global block matching: parameters not served by the kernels");
    HIP_CK(c, hipGetLastError());
    if(d_left) *d_left = dl;
    return BPVO_OK;
  }
  launch_stereo_prefilter(c->stream, dl, c->st_left_pre, c->rows, c->cols, sp->preFilterCap, count);
  launch_stereo_prefilter(c->stream, dr, c->st_right_pre, c->rows, c->cols, sp->preFilterCap, count);
  StereoLaunch g;
  g.left_pre = c->st_left_pre; g.right_pre = c->st_right_pre; g.disp = c->st_disp;
  g.rows = c->rows; g.cols = c->cols; g.nframes = count;
  g.wsz = sp->SADWindowSize; g.ndisp = sp->numberOfDisparities; g.mindisp = sp->minDisparity; g.cap = sp->preFilterCap;
  g.texture_threshold = sp->textureThreshold; g.uniqueness_ratio = sp->uniquenessRatio;
  if(!launch_stereo_bm(c->stream, g)) return fail(c, BPVO_ERR_UNSUPPORTED, "stereo block matching: window / disparity range not served by the kernel");
  HIP_CK(c, hipGetLastError());
  if(d_left) *d_left = dl;
  return BPVO_OK;
}

void bpvo_hip_default_stereo_params(bpvo_hip_stereo_params* p)   // utils/stereo_algorithm.cc:63-82 (numberOfDisparities has no default there)
{
  std::memset(p, 0, sizeof(*p));
  p->preFilterCap = 31; p->SADWindowSize = 15; p->minDisparity = 0; p->numberOfDisparities = 64; p->textureThreshold = 10; p->uniquenessRatio = 15;
  // SgmStereo::Config() (utils/sgm.cc:47-56); algorithm: "BlockMatching" is the config file's default (utils/stereo_algorithm.cc:25)
  p->algorithm = BPVO_STEREO_BLOCK_MATCHING;
  p->sobelCapValue = 15; p->censusRadius = 2; p->windowRadius = 2; p->smoothnessPenaltySmall = 100; p->smoothnessPenaltyLarge = 1600;
  p->consistencyThreshold = 1; p->disparityFactor = 256.0; p->censusWeightFactor = 1.0 / 6.0;
}
void bpvo_hip_stereo_params_sgbm_from_config(bpvo_hip_stereo_params* p, int minDisparity, int numberOfDisparities, int SADWindowSize, int P1, int P2,
                                             int uniquenessRatio, int speckleWindowSize, int speckleRange, int fullDP)
{
  // make_unique<cv::StereoSGBM>(minDisparity, numberOfDisparities, SADWindowSize, P1, P2, uniquenessRatio, speckleWindowSize, speckleRange,
  // (bool) fullDP) against StereoSGBM(minDisparity, numDisparities, SADWindowSize, P1, P2, disp12MaxDiff, preFilterCap, uniquenessRatio,
  // speckleWindowSize, speckleRange = 0, fullDP = false)  (utils/stereo_algorithm.cc:30-39)
  bpvo_hip_default_stereo_params(p);
  p->algorithm = BPVO_STEREO_SGBM;
  p->minDisparity = minDisparity; p->numberOfDisparities = numberOfDisparities; p->SADWindowSize = SADWindowSize; p->P1 = P1; p->P2 = P2;
  p->disp12MaxDiff = uniquenessRatio;
  p->preFilterCap = speckleWindowSize;
  p->uniquenessRatio = speckleRange;
  p->speckleWindowSize = fullDP ? 1 : 0;
  p->speckleRange = 0;
  p->fullDP = 0;
  p->textureThreshold = 0;
}
int bpvo_hip_stereo_bm(bpvo_hip_ctx* c, int count, const uint8_t* left, const uint8_t* right, int on_device, const bpvo_hip_stereo_params* sp,
                       float* disparity, int disparity_on_device)
{
  CHECK_CTX(c);
  if(!disparity) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr disparity");
  (void) hipSetDevice(c->device);
  int rc = stereo_run(c, count, left, right, on_device != 0, sp, nullptr);
  if(rc) return rc;
  HIP_CK(c, hipMemcpyAsync(disparity, c->st_disp, c->geom[0].npix * (size_t) count * sizeof(float),
                           disparity_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}

static int add_frame_impl(bpvo_hip_ctx* c, const uint8_t* image, const float* disparity, bool on_device, bpvo_hip_result* ret);
int bpvo_hip_add_frame(bpvo_hip_ctx* c, const uint8_t* image, const float* disparity, bpvo_hip_result* ret)
{
  CHECK_CTX(c);
  if(!image || !disparity) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr image/disparity");   // bpvo/vo.cc:68-69
  return add_frame_impl(c, image, disparity, false, ret);
}
// addFrame fed by the stereo front-end: the reference's apps run StereoAlgorithm::run on the rectified pair and hand the f32
// disparity to VisualOdometry::addFrame (apps/vo_app.cc, utils/dataset.h); here the disparity never leaves the device
int bpvo_hip_add_frame_stereo(bpvo_hip_ctx* c, const uint8_t* left, const uint8_t* right, const bpvo_hip_stereo_params* sp, bpvo_hip_result* ret)
{
  CHECK_CTX(c);
  if(!left || !right) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr image");
  (void) hipSetDevice(c->device);
  const uint8_t* d_left = nullptr;
  int rc = stereo_run(c, 1, left, right, false, sp, &d_left);
  if(rc) return rc;
  return add_frame_impl(c, d_left, c->st_disp, true, ret);
}
static int add_frame_impl(bpvo_hip_ctx* c, const uint8_t* image, const float* disparity, bool on_device, bpvo_hip_result* ret)
{
  if(!ret) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr result");
  if(c->n_frames < 3) return fail(c, BPVO_ERR_INVALID_ARG, "add_frame needs a ctx with n_frames >= 3");
  (void) hipSetDevice(c->device);
  const M44 I = m44_identity();
  std::memset(ret, 0, sizeof(*ret));
  std::memcpy(ret->pose, I.m, 64);
  for(int i = 0; i < 36; ++i) ret->covariance[i] = (i % 7 == 0) ? 1.0f : 0.0f;   // Q16
  ret->numLevels = c->L;
  for(int l = 0; l < kMaxLevels; ++l) ret->optimizerStatistics[l] = bpvo_hip_stats{0, -1.0f, -1.0f, BPVO_STATUS_SOLVER_ERROR};
  ret->isKeyFrame = 0;
  ret->keyFramingReason = BPVO_KF_NO_KEYFRAMING;
  ret->hasPointCloud = 0;
  c->cloud_n = 0;                   // the point cloud belongs to one Result (bpvo/types.h:549-563)
  c->cloud_pose = I;

  // _cur_frame->setData (vo.cc:131).  No host synchronisation behind it: the estimate queues behind the data stage on the same stream and ends
  // with one (every copy from the caller's buffers is complete when this function returns — its early returns synchronise themselves).  The
  // disparity of a host frame, which only a later template stage reads, is uploaded once the estimate is queued (upload_disparity): the copy
  // from pageable memory holds the host for 90 us, which then lie under the Gauss-Newton kernels instead of in front of them.
  if(c->vo_cur < 0 || c->vo_cur >= c->n_frames) return fail(c, BPVO_ERR_INVALID_ARG, "bad frame slot range");
  FrameRun data_run = ctx_run(c);
  data_run.skip_disparity_upload = !on_device && c->vo_disparity_late && single_pair_is_queued_at_once(c);
  int rc = frames_set_data(c, c->vo_cur, 1, 1, image, disparity, on_device, data_run, 0);
  if(rc) { (void) hipStreamSynchronize(c->stream); return rc; }
  const int data_slot = c->vo_cur;
  bool disparity_pending = data_run.skip_disparity_upload;
  auto upload_now = [&]() -> int {
    c->before_final_sync = nullptr;
    if(!disparity_pending) return BPVO_OK;
    disparity_pending = false;
    return upload_disparity(c, data_slot, disparity);
  };

  if(!c->frames[c->vo_ref].has_template) {            // first frame (vo.cc:133-139)
    std::swap(c->vo_ref, c->vo_cur);
    rc = upload_now();                                 // (its template stage reads the disparity)
    if(rc == BPVO_OK) rc = frames_set_template(c, c->vo_ref, 1, 1);
    if(rc) { (void) hipStreamSynchronize(c->stream); return rc; }
    trajectory_push(c, c->T_kf);
    ret->isKeyFrame = 1;
    ret->keyFramingReason = BPVO_KF_FIRST_FRAME;
    return BPVO_OK;
  }

  M44 T_est;
  const int ws0 = 0;
  rc = check_template_not_empty(c, c->vo_ref);
  if(rc) { (void) upload_now(); (void) hipStreamSynchronize(c->stream); return rc; }
  c->prefetch_frac_thr = c->params.goodPointThreshold;      // should_keyframe's fraction of good points rides behind the estimate
  if(disparity_pending) c->before_final_sync = [&]() { return upload_now(); };
  rc = estimate_batch(c, 1, &ws0, &c->vo_ref, &c->vo_cur, c->T_kf.m, T_est.m, ret->optimizerStatistics);
  c->prefetch_frac_thr = -1.0f;
  {      // (an estimate that did not come by its final synchronisation — an error on the way, another path — : now)
    const int rcu = upload_now();
    if(rc == BPVO_OK) rc = rcu;
  }
  if(rc) { (void) hipStreamSynchronize(c->stream); return rc; }
  int reason = BPVO_KF_NO_KEYFRAMING;
  rc = should_keyframe(c, T_est, &reason);
  if(rc) return rc;
  ret->keyFramingReason = reason;
  ret->isKeyFrame = reason != BPVO_KF_NO_KEYFRAMING;

  M44 pose;
  if(!ret->isKeyFrame) {
    std::swap(c->vo_prev, c->vo_cur);
    pose = m44_mul(T_est, m44_inverse(c->T_kf));
    c->T_kf = T_est;
  } else {
    rc = build_point_cloud(c);
    if(rc) return rc;
    ret->hasPointCloud = 1;
    if(!c->frames[c->vo_prev].has_data) {               // vo.cc:161-173
      std::swap(c->vo_cur, c->vo_ref);
      rc = frames_set_template(c, c->vo_ref, 1, 1);
      if(rc) return rc;
      pose = m44_mul(T_est, m44_inverse(c->T_kf));
      c->T_kf = m44_identity();
    } else {                                            // vo.cc:174-188
      std::swap(c->vo_prev, c->vo_ref);
      c->frames[c->vo_prev].has_data = false;
      c->frames[c->vo_prev].has_template = false;
      // the estimate against the new key frame follows on the same stream: the template stage ends without a host round trip of its own and
      // leaves the normalisation sums of the levels below the coarsest on the side streams, under the Gauss-Newton iterations of the levels
      // above them (a dense 640x480 template, conf/tsukuba.cfg: 2.3 ms of dependent adds for the finest level)
      FrameRun fr = ctx_run(c);
      fr.no_final_sync = !c->profiling;
      fr.defer_finest_nrm = true;
      rc = frames_set_template(c, c->vo_ref, 1, 1, fr);
      if(rc == BPVO_OK) rc = estimate_batch(c, 1, &ws0, &c->vo_ref, &c->vo_cur, I.m, T_est.m, ret->optimizerStatistics);
      // (an error on the way: nothing of this call stays in flight)
      if(c->nrm_pending) { (void) hipEventSynchronize(c->nrm_pending); c->nrm_pending = nullptr; }
      if(c->nrm_pending_finest) { (void) hipEventSynchronize(c->nrm_pending_finest); c->nrm_pending_finest = nullptr; }
      if(rc) return rc;
      pose = T_est;
      c->T_kf = T_est;
    }
  }
  std::memcpy(ret->pose, pose.m, 64);
  trajectory_push(c, pose);
  if(ret->hasPointCloud) c->cloud_pose = c->trajectory.back();
  return BPVO_OK;
}

int bpvo_hip_vo_num_points_at_level(bpvo_hip_ctx* c, int level, int* n)
{
  CHECK_CTX(c);
  if(level < 0) level = c->params.maxTestLevel;
  if(level >= c->L) return fail(c, BPVO_ERR_INVALID_ARG, "bad level");
  *n = c->frames[c->vo_ref].has_template ? c->frames[c->vo_ref].n_host[level] : 0;
  return BPVO_OK;
}
int bpvo_hip_vo_points_at_level(bpvo_hip_ctx* c, int level, float* xyzw)
{
  CHECK_CTX(c);
  if(level < 0) level = c->params.maxTestLevel;
  return bpvo_hip_get_points(c, c->vo_ref, level, xyzw);
}
int bpvo_hip_get_point_cloud(bpvo_hip_ctx* c, bpvo_hip_point_with_info* pts, size_t* n, float pose[16])
{
  CHECK_CTX(c);
  if(n) *n = c->cloud_n;
  if(pts && c->cloud_n) {
    (void) hipSetDevice(c->device);
    HIP_CK(c, hipMemcpyAsync(pts, c->d_cloud, c->cloud_n * sizeof(bpvo_hip_point_with_info), hipMemcpyDeviceToHost, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
  }
  if(pose) std::memcpy(pose, c->cloud_pose.m, 64);
  return BPVO_OK;
}
int bpvo_hip_trajectory_size(bpvo_hip_ctx* c, int* n) { CHECK_CTX(c); *n = (int) c->trajectory.size(); return BPVO_OK; }
int bpvo_hip_get_trajectory(bpvo_hip_ctx* c, float* poses)
{
  CHECK_CTX(c);
  for(size_t i = 0; i < c->trajectory.size(); ++i) std::memcpy(poses + 16 * i, c->trajectory[i].m, 64);
  return BPVO_OK;
}

}  // extern "C"
