// ORACLE — test infrastructure only (see orc.h).  Gauss-Newton / IRLS driver, frames, estimatePose, VisualOdometry.
#include "orc.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace orc {

// PoseEstimatorParameters(const AlgorithmParameters&) (bpvo/pose_estimator_params.cc:27-33): maxFuncEvals is not
// copied and stays 6*200 (Q4).
void PoseEstimator::setParameters(const Params& p)
{
  maxIterations = p.maxIterations;
  functionTolerance = p.functionTolerance;
  parameterTolerance = p.parameterTolerance;
  gradientTolerance = p.gradientTolerance;
  lossFunction = p.lossFunction;
  maxFuncEvals = 6 * 200;
}

// test instrument (orc.h, PoseEstimator::perturb_rel): (H, G) <- (H, G) .* (1 + rel * n), H kept symmetric
static void perturbSystem(float H[36], float G[6], float rel, uint32_t seed, uint32_t count)
{
  uint64_t s = 0x9e3779b97f4a7c15ull * (uint64_t) (seed + 1) + 0xbf58476d1ce4e5b9ull * (uint64_t) (count + 1);
  auto next = [&]() {      // splitmix64
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  };
  auto gauss = [&]() {     // sum of 12 uniforms - 6
    double a = 0.0;
    for(int i = 0; i < 12; ++i) a += (double) (next() >> 11) * (1.0 / 9007199254740992.0);
    return a - 6.0;
  };
  for(int i = 0; i < 6; ++i)
    for(int j = i; j < 6; ++j) {
      const float v = (float) ((double) H[i * 6 + j] * (1.0 + (double) rel * gauss()));
      H[i * 6 + j] = v;
      H[j * 6 + i] = v;
    }
  for(int i = 0; i < 6; ++i) G[i] = (float) ((double) G[i] * (1.0 + (double) rel * gauss()));
}

// PoseEstimatorGN::linearize (bpvo/pose_estimator_gn.h:70-81) incl. replicateValidFlags (pose_estimator_base.h:307-320).
float PoseEstimator::linearize(TemplateData* tdata, const Descriptor& desc, const M44& T, float H[36], float G[6])
{
  tdata->computeResiduals(desc, T, residuals, valid, nthreads);
  int nv = 0;
  for(uint16_t v : valid) nv += v;
  last_num_valid = nv;
  if(residuals.size() != valid.size()) {
    const size_t m = residuals.size() / valid.size(), n = valid.size();
    std::vector<uint16_t> tmp(residuals.size());
    for(size_t i = 0; i < m; ++i) std::memcpy(tmp.data() + i * n, valid.data(), n * sizeof(uint16_t));
    valid.swap(tmp);
  }
  const float sigma = scale_estimator.estimateScale(residuals, valid);
  last_sigma = sigma;
  computeWeights(lossFunction, residuals, valid, sigma, weights);
  num_fun_evals += 1;
  const float f = linearSystemRun(tdata->jacobians, residuals, weights, valid, H, G, reduction == 1 ? -1 : nthreads);
  if(perturb_rel > 0.0f) perturbSystem(H, G, perturb_rel, perturb_seed, perturb_count++);
  return f;
}

static inline float infNorm6(const float* g)
{
  float m = 0.0f;
  for(int i = 0; i < 6; ++i) m = std::max(m, std::fabs(g[i]));
  return m;
}

// PoseEstimatorBase::run (bpvo/pose_estimator_base.h:324-407) with testConvergence (:258-282) and reset (:287-293);
// Q1 (pose update repeated after convergence) and Q2 (iteration bookkeeping) reproduced.
Stats PoseEstimator::run(TemplateData* tdata, const Descriptor& desc, M44& T)
{
  scale_estimator.reset();
  f_norm_prev = 0.0f;
  g_tol = 0.0f;
  num_fun_evals = 0;

  Stats ret;
  ret.numIterations = 0;
  ret.finalError = -1.0f;
  ret.firstOrderOptimality = 0;
  ret.status = kMaxIterations;

  float H[36], G[6], dp[6] = {0, 0, 0, 0, 0, 0};
  M44 dT = T;
  const float sqrt_eps = std::sqrt(std::numeric_limits<float>::epsilon());

  auto record = [&](const M44& Tlin, float f) {
    if(!trace) return;
    IterationRecord rec;
    rec.T = Tlin;
    std::memcpy(rec.H, H, sizeof(H));
    std::memcpy(rec.G, G, sizeof(G));
    rec.f_norm = f;
    rec.sigma = last_sigma;
    rec.num_valid = last_num_valid;
    std::memset(rec.dp, 0, sizeof(rec.dp));
    trace->push_back(rec);
  };
  auto solve = [&]() {
    const bool ok = solveSystem(H, G, dp);
    if(trace && !trace->empty()) std::memcpy(trace->back().dp, dp, sizeof(dp));
    return ok;
  };
  auto update = [&]() {
    const float mdp[6] = {-dp[0], -dp[1], -dp[2], -dp[3], -dp[4], -dp[5]};
    dT = mul44(dT, tdata->warp.paramsToPose(mdp));
  };

  float f_norm = linearize(tdata, desc, dT, H, G);
  record(dT, f_norm);
  float g_norm = infNorm6(G);
  g_tol = gradientTolerance * std::max(g_norm, sqrt_eps);

  if(g_norm < g_tol) {
    ret.status = kGradientTolReached;
    ret.finalError = f_norm;
    ret.numIterations = 1;
    ret.firstOrderOptimality = g_norm;
    return ret;
  }

  if(!solve()) {
    ret.status = kSolverError;
    ret.finalError = f_norm;
    return ret;
  }

  f_norm_prev = 0.0f;
  float dp_norm_prev = 0.0f;
  bool has_converged = false;

  update();

  do {
    float dp_norm = 0.0f;
    for(int i = 0; i < 6; ++i) dp_norm += dp[i] * dp[i];
    dp_norm = std::sqrt(dp_norm);
    g_norm = infNorm6(G);

    // testConvergence
    has_converged = false;
    if(dp_norm < parameterTolerance || dp_norm < parameterTolerance * (sqrt_eps + dp_norm_prev)) {
      ret.status = kParameterTolReached;
      has_converged = true;
    } else if(f_norm < functionTolerance || f_norm < functionTolerance * (sqrt_eps + f_norm_prev) ||
              std::fabs(f_norm - f_norm_prev) < functionTolerance) {
      ret.status = kFunctionTolReached;
      has_converged = true;
    } else if(g_norm < g_tol) {
      ret.status = kGradientTolReached;
      has_converged = true;
    }

    dp_norm_prev = dp_norm;
    f_norm_prev = f_norm;

    if(!has_converged) {
      // PoseEstimatorGN::runIteration (bpvo/pose_estimator_gn.h:83-100)
      f_norm = linearize(tdata, desc, dT, H, G);
      record(dT, f_norm);
      if(!solve()) {
        ret.status = kSolverError;
        break;
      }
    }

    update();
  } while(ret.numIterations++ < maxIterations && !has_converged && num_fun_evals < maxFuncEvals);

  if(ret.status != kSolverError) T = dT;

  ret.numIterations -= 1;
  ret.finalError = f_norm;
  ret.firstOrderOptimality = g_norm;
  return ret;
}

// VisualOdometryFrame ctor (bpvo/vo_frame.cc:13-29): per level K *= 0.5 with K(2,2) = 1, b *= 2 (Q20).
void Frame::init(const float K[9], float b, int rows_, int cols_, const Params& p)
{
  params = p;
  rows = rows_;
  cols = cols_;
  const int L = p.numPyramidLevels;
  pyr.resize(L); prow.resize(L); pcol.resize(L);
  desc.resize(L);
  tdata.resize(L);
  float Kp[9];
  std::memcpy(Kp, K, sizeof(Kp));
  float bp = b;
  for(int i = 0; i < L; ++i) {
    if(i > 0) {
      for(int k = 0; k < 9; ++k) Kp[k] *= 0.5f;
      Kp[8] = 1.0f;
      bp *= 2.0f;
    }
    tdata[i].level = i;
    tdata[i].params = p;
    tdata[i].warp.init(Kp, bp);
  }
}

// VisualOdometryFrame::setData (bpvo/vo_frame.cc:48-55) -> DenseDescriptorPyramid::init
// (bpvo/dense_descriptor_pyramid.cc:67-78) -> ImagePyramid::compute (bpvo/image_pyramid.cc:43-50)
void Frame::setData(const uint8_t* img, const float* disp)
{
  const size_t n = (size_t) rows * cols;
  image.assign(img, img + n);
  disparity.assign(disp, disp + n);
  const int L = params.numPyramidLevels;
  pyr[0] = image; prow[0] = rows; pcol[0] = cols;
  for(int i = 1; i < L; ++i) pyrDownU8(pyr[i - 1].data(), prow[i - 1], pcol[i - 1], pyr[i], prow[i], pcol[i]);
  for(int i = L - 1; i >= params.maxTestLevel; --i)
    computeDescriptor(params, pyr[i].data(), prow[i], pcol[i], desc[i], nthreads);
  has_data = true;
}

// VisualOdometryFrame::setTemplate (bpvo/vo_frame.cc:61-93)
void Frame::setTemplate()
{
  if(!has_data) throw std::logic_error("no data in frame");
  for(int i = (int) tdata.size() - 1; i >= params.maxTestLevel; --i)
    tdata[i].setData(desc[i], disparity.data(), cols);
  has_template = true;
}

// VisualOdometryPoseEstimator (bpvo/vo_pose_estimator.cc:55-107); Q3: both parameter sets are identical.
void VoPoseEstimator::init(const Params& p, int nthreads)
{
  params = p;
  est.setParameters(p);
  est.nthreads = nthreads;
}

void VoPoseEstimator::estimatePose(Frame* ref, Frame* cur, const M44& T_init, M44& T_est, std::vector<Stats>& stats)
{
  const int L = (int) ref->desc.size();
  Stats def = {0, -1.0f, -1.0f, kSolverError};                     // OptimizerStatistics() (bpvo/types.cc:306-310)
  stats.assign(L, def);
  T_est = T_init;
  for(int i = L - 1; i >= params.maxTestLevel; --i)
    stats[i] = est.run(&ref->tdata[i], cur->desc[i], T_est);
}

float VoPoseEstimator::getFractionOfGoodPoints(float thresh) const
{
  const auto& w = est.weights;
  const auto n = std::count_if(w.begin(), w.end(), [=](float wi) { return wi > thresh; });
  return n / static_cast<float>(w.size());
}

// VisualOdometry::Impl ctor (bpvo/vo.cc:97-115)
void VisualOdometry::init(const float K_[9], float b, int rows_, int cols_, const Params& p, int nthreads)
{
  params = p;
  rows = rows_;
  cols = cols_;
  std::memcpy(K, K_, sizeof(K));
  vo_pose.init(p, nthreads);
  T_kf = identity44();
  if(params.numPyramidLevels <= 0)
    params.numPyramidLevels = 1 + (int) std::round(std::log2(std::min(rows, cols) / (double) p.minImageDimensionForPyramid));
  ref.reset(new Frame); cur.reset(new Frame); prev.reset(new Frame);
  for(Frame* f : {ref.get(), cur.get(), prev.get()}) {
    f->init(K, b, rows, cols, params);
    f->nthreads = nthreads;
  }
  vo_pose.params.numPyramidLevels = params.numPyramidLevels;
  trajectory.clear();
}

// Trajectory::push_back + InvertPose (bpvo/trajectory.cc:30-50)
void VisualOdometry::trajectoryPush(const M44& T)
{
  M44 Ti = identity44();
  for(int i = 0; i < 3; ++i)
    for(int j = 0; j < 3; ++j) Ti.m[i * 4 + j] = T.m[j * 4 + i];
  for(int i = 0; i < 3; ++i) {
    // -(R^T)^T * t = -R * t with R^T already stored in Ti: ret.block(0,3) = -Ti_R.transpose() * t
    float s = Ti.m[0 * 4 + i] * T.m[3];
    s += Ti.m[1 * 4 + i] * T.m[7];
    s += Ti.m[2 * 4 + i] * T.m[11];
    Ti.m[i * 4 + 3] = -s;
  }
  if(!trajectory.empty()) trajectory.push_back(mul44(trajectory.back(), Ti));
  else trajectory.push_back(Ti);
}

// VisualOdometry::Impl::shouldKeyFrame (bpvo/vo.cc:199-224) with math::RotationMatrixToEulerAngles
// (bpvo/math_utils.h:203-216); Q17: the rotation threshold is compared in radians.
int VisualOdometry::shouldKeyFrame(const M44& pose) const
{
  const float t_norm = pose.m[3] * pose.m[3] + pose.m[7] * pose.m[7] + pose.m[11] * pose.m[11];
  if(t_norm > params.minTranslationMagToKeyFrame * params.minTranslationMagToKeyFrame) return kLargeTranslation;

  const float R00 = pose.m[0], R10 = pose.m[4], R20 = pose.m[8], R21 = pose.m[9];
  const float eta = (float) (1.0 / (std::sqrt(R00 * R00 + R10 * R10)));
  const float rz = std::asin(eta * R10);
  const float ry = std::asin(-R20);
  const float rx = std::asin(eta * R21);
  const float r_norm = rx * rx + ry * ry + rz * rz;
  if(r_norm > params.minRotationMagToKeyFrame * params.minRotationMagToKeyFrame) return kLargeRotation;

  const float frac_good = vo_pose.getFractionOfGoodPoints(params.goodPointThreshold);
  if(frac_good < params.maxFractionOfGoodPointsToKeyFrame) return kSmallFracOfGoodPoints;
  return kNoKeyFraming;
}

// VisualOdometry::Impl::addFrame (bpvo/vo.cc:125-197) + getPointCloudFromRefFrame (:260-281) + GetColor (:250-258)
void VisualOdometry::addFrame(const uint8_t* I, const float* D, Result& ret)
{
  ret.pose = identity44();
  for(int i = 0; i < 36; ++i) ret.covariance[i] = (i % 7 == 0) ? 1.0f : 0.0f;   // Q16
  ret.isKeyFrame = false;
  ret.keyFramingReason = kNoKeyFraming;
  ret.hasPointCloud = false;
  ret.cloud.clear();
  ret.cloudPose = identity44();

  cur->setData(I, D);

  if(!ref->has_template) {
    std::swap(ref, cur);
    ref->setTemplate();
    trajectoryPush(T_kf);
    // FirstFrameResult (vo.cc:111-123)
    Stats def = {0, -1.0f, -1.0f, kSolverError};
    ret.stats.assign(ref->desc.size(), def);
    ret.isKeyFrame = true;
    ret.keyFramingReason = kFirstFrame;
    return;
  }

  M44 T_est;
  vo_pose.estimatePose(ref.get(), cur.get(), T_kf, T_est, ret.stats);
  ret.keyFramingReason = shouldKeyFrame(T_est);
  ret.isKeyFrame = kNoKeyFraming != ret.keyFramingReason;

  if(!ret.isKeyFrame) {
    std::swap(prev, cur);
    ret.pose = mul44(T_est, inverse44(T_kf));
    T_kf = T_est;
  } else {
    // getPointCloudFromRefFrame
    {
      const TemplateData& td = ref->tdata[params.maxTestLevel];
      const auto& weights = vo_pose.est.weights;
      const size_t n = td.numPoints();
      if(n > weights.size()) throw std::logic_error("size mismatch");
      ret.cloud.resize(n);
      for(size_t i = 0; i < n; ++i) {
        const float* X = td.points.data() + 4 * i;
        // getImagePoint (bpvo/rigid_body_warp.h:123-128): x = K * X.head<3>(), z_i = 1/x.z
        const float* Kl = td.warp.K;
        float x[3];
        for(int r = 0; r < 3; ++r) {
          float s = Kl[r * 3 + 0] * X[0];
          s += Kl[r * 3 + 1] * X[1];
          s += Kl[r * 3 + 2] * X[2];
          x[r] = s;
        }
        const float z_i = 1.0f / x[2];
        float u = z_i * x[0], v = z_i * x[1];
        if(td.warp.dspace) { u = X[0] + Kl[2]; v = X[1] + Kl[5]; }   // DisparitySpaceWarp::getImagePoint (disparity_space_warp.h:73-76)
        uint8_t c = 0;
        if(v >= 0 && v < rows && u >= 0 && u < cols) c = ref->image[(size_t) ((int) v) * cols + (int) u];
        PointWithInfo& p = ret.cloud[i];
        std::memset(&p, 0, sizeof(p));
        std::memcpy(p.xyzw, X, 4 * sizeof(float));
        p.rgba[0] = c; p.rgba[1] = c; p.rgba[2] = c; p.rgba[3] = 255;
        p.weight = weights[i];
      }
      ret.hasPointCloud = true;
    }

    if(!prev->has_data) {
      std::swap(cur, ref);
      ref->setTemplate();
      ret.pose = mul44(T_est, inverse44(T_kf));
      T_kf = identity44();
    } else {
      std::swap(prev, ref);
      prev->clear();
      ref->setTemplate();
      M44 T_init = identity44();
      vo_pose.estimatePose(ref.get(), cur.get(), T_init, T_est, ret.stats);
      ret.pose = T_est;
      T_kf = T_est;
    }
  }

  trajectoryPush(ret.pose);
  if(ret.hasPointCloud) ret.cloudPose = trajectory.back();
}

}  // namespace orc
