/*
 * bpvo_hip/vo.hpp — header-only C++11 facade over the C ABI (c_api.h) with the reference's class names, argument
 * meaning and error behaviour, so that code written against bpvo/vo.h keeps compiling after swapping the include:
 *
 *   bpvo::VisualOdometry                (reference: bpvo/vo.h:31-105, bpvo/vo.cc:66-94)
 *   bpvo::VisualOdometryFrame           (reference: bpvo/vo_frame.h:21-90)
 *   bpvo::VisualOdometryPoseEstimator   (reference: bpvo/vo_pose_estimator.h:34-64)
 *   bpvo::AlgorithmParameters, Result, OptimizerStatistics, PointWithInfo, PointCloud, Trajectory, ImageSize, enums
 *                                       (reference: bpvo/types.h:127-589, bpvo/point_cloud.h, bpvo/trajectory.h)
 *
 * Differences a caller sees:
 *   - no Eigen / OpenCV in the interface: Matrix33 / Matrix44 are row-major std::array<float,9|16>
 *     (an Eigen user maps them with Eigen::Map<const Eigen::Matrix<float,4,4,Eigen::RowMajor>>);
 *   - frames live on the GPU: VisualOdometryFrame is a handle (device context + slot), not a host container;
 *   - a non-zero C status becomes bpvo::Error (std::logic_error) exactly where the reference uses THROW_ERROR
 *     (bpvo/utils.h:211-220), e.g. null image/disparity pointers (bpvo/vo.cc:68-69).
 */
#ifndef BPVO_HIP_VO_HPP
#define BPVO_HIP_VO_HPP

#include <array>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "c_api.h"

namespace bpvo {

struct Error : public std::logic_error {                       // bpvo/utils.h:205-209
  explicit Error(const std::string& what) : std::logic_error(what) {}
};

typedef std::array<float, 9> Matrix33;                          // row-major
typedef std::array<float, 16> Matrix44;                         // row-major
typedef Matrix44 Pose;
typedef std::array<float, 36> PoseCovariance;
typedef std::array<float, 4> Point;                             // (X, Y, Z, 1)  bpvo/types.h:84
typedef std::vector<Point> PointVector;
typedef std::vector<float> WeightsVector;

enum LossFunctionType { kHuber = BPVO_LOSS_HUBER, kTukey = BPVO_LOSS_TUKEY, kL2 = BPVO_LOSS_L2 };
enum VerbosityType { kIteration = BPVO_VERB_ITERATION, kFinal, kSilent, kDebug };
/* all eight values of the reference (bpvo/types.h:142-152), all on the device path */
enum DescriptorType { kIntensity = BPVO_DESC_INTENSITY, kIntensityAndGradient, kDescriptorFieldsFirstOrder, kDescriptorFieldsSecondOrder,
                      kLatch, kCentralDifference, kLaplacian, kBitPlanes = BPVO_DESC_BITPLANES };
static_assert(kLaplacian + 1 == kBitPlanes, "DescriptorType numbering follows the reference");
enum GradientEstimationType { kCentralDifference_3 = BPVO_GRAD_CD3, kCentralDifference_5 = BPVO_GRAD_CD5 };
enum InterpolationType { kLinear = BPVO_INTERP_LINEAR, kCosine, kCubic, kCubicHermite };
enum PoseEstimationStatus { kParameterTolReached = BPVO_STATUS_PARAMETER_TOL, kFunctionTolReached, kGradientTolReached,
                            kMaxIterations, kSolverError };
enum KeyFramingReason { kLargeTranslation = BPVO_KF_LARGE_TRANSLATION, kLargeRotation, kSmallFracOfGoodPoints, kNoKeyFraming,
                        kFirstFrame };

struct ImageSize {                                              // bpvo/types.h:572-582
  int rows, cols;
  ImageSize(int r = 0, int c = 0) : rows(r), cols(c) {}
  int numel() const { return rows * cols; }
};

/* bpvo::AlgorithmParameters (bpvo/types.h:171-413): the same 35 fields; defaults = bpvo/types.cc:31-66 */
struct AlgorithmParameters : public bpvo_hip_params {
  AlgorithmParameters() { bpvo_hip_default_params(this); }
};

struct OptimizerStatistics {                                    // bpvo/types.h:444-482
  int numIterations;
  float finalError;
  float firstOrderOptimality;
  PoseEstimationStatus status;
  OptimizerStatistics() : numIterations(0), finalError(-1.0f), firstOrderOptimality(-1.0f), status(kSolverError) {}
  explicit OptimizerStatistics(const bpvo_hip_stats& s)
      : numIterations(s.numIterations), finalError(s.finalError), firstOrderOptimality(s.firstOrderOptimality),
        status(static_cast<PoseEstimationStatus>(s.status)) {}
};

typedef bpvo_hip_point_with_info PointWithInfo;                 // bpvo/point_cloud.h:30-62 (32-byte record)

class PointCloud {                                              // bpvo/point_cloud.h:67-120
 public:
  PointCloud() { setIdentity(); }
  const std::vector<PointWithInfo>& points() const { return _points; }
  std::vector<PointWithInfo>& points() { return _points; }
  size_t size() const { return _points.size(); }
  bool empty() const { return _points.empty(); }
  const Matrix44& pose() const { return _pose; }
  Matrix44& pose() { return _pose; }
 private:
  void setIdentity() { _pose.fill(0.0f); _pose[0] = _pose[5] = _pose[10] = _pose[15] = 1.0f; }
  std::vector<PointWithInfo> _points;
  Matrix44 _pose;
};

struct Result {                                                 // bpvo/types.h:489-563 (move-only)
  Pose pose;
  PoseCovariance covariance;
  std::vector<OptimizerStatistics> optimizerStatistics;
  bool isKeyFrame;
  KeyFramingReason keyFramingReason;
  std::unique_ptr<PointCloud> pointCloud;
  Result() : isKeyFrame(false), keyFramingReason(kNoKeyFraming) {}
  Result(Result&&) = default;
  Result& operator=(Result&&) = default;
  Result(const Result&) = delete;
  Result& operator=(const Result&) = delete;
};

class Trajectory {                                              // bpvo/trajectory.h
 public:
  size_t size() const { return _poses.size(); }
  const Matrix44& operator[](size_t i) const { return _poses[i]; }
  const Matrix44& back() const { return _poses.back(); }
  const std::vector<Matrix44>& poses() const { return _poses; }
  /* Trajectory::push_back (bpvo/trajectory.cc:29-50): appends back() * InvertPose(T) with InvertPose as written there,
   * R' = R^T, t' = -(R'^T) t = -R t.  VisualOdometry fills its trajectory on the device side with the same rule; this
   * member is for callers that assemble a trajectory themselves (apps/eval_kitti.cc:120-130). */
  void push_back(const Matrix44& T)
  {
    Matrix44 Ti;
    for(int i = 0; i < 3; ++i)
      for(int j = 0; j < 3; ++j) Ti[i * 4 + j] = T[j * 4 + i];
    for(int i = 0; i < 3; ++i) Ti[i * 4 + 3] = -((T[i * 4 + 0] * T[3] + T[i * 4 + 1] * T[7]) + T[i * 4 + 2] * T[11]);
    Ti[12] = Ti[13] = Ti[14] = 0.0f;
    Ti[15] = 1.0f;
    if(_poses.empty()) { _poses.push_back(Ti); return; }
    const Matrix44& A = _poses.back();
    Matrix44 C;
    for(int r = 0; r < 4; ++r)
      for(int c = 0; c < 4; ++c) {
        float acc = A[r * 4 + 0] * Ti[0 * 4 + c];
        for(int k = 1; k < 4; ++k) acc += A[r * 4 + k] * Ti[k * 4 + c];
        C[r * 4 + c] = acc;
      }
    _poses.push_back(C);
  }
 private:
  friend class VisualOdometry;
  std::vector<Matrix44> _poses;
};

namespace detail {
/* owns one bpvo_hip_ctx; maps a non-zero status to bpvo::Error (THROW_ERROR, bpvo/utils.h:211-220) */
class Device {
 public:
  Device(const Matrix33& K, float baseline, ImageSize size, const AlgorithmParameters& p, int n_frames, int n_pairs, int device = 0)
      : _ctx(nullptr), _size(size)
  {
    const int rc = bpvo_hip_create(&_ctx, K.data(), baseline, size.rows, size.cols, &p, device, n_frames, n_pairs);
    if(rc != BPVO_OK) throw Error(std::string("bpvo_hip_create: ") + bpvo_hip_last_error(nullptr));
  }
  ~Device() { bpvo_hip_destroy(_ctx); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  bpvo_hip_ctx* ctx() const { return _ctx; }
  ImageSize imageSize() const { return _size; }
  void check(int rc) const { if(rc != BPVO_OK) throw Error(bpvo_hip_last_error(_ctx)); }
  /* how the device schedules its work (c_api.h "Options"); no counterpart in the reference */
  void setOption(const std::string& name, double value) { check(bpvo_hip_set_option(_ctx, name.c_str(), value)); }
  double getOption(const std::string& name) const { double v = 0.0; check(bpvo_hip_get_option(_ctx, name.c_str(), &v)); return v; }
 private:
  bpvo_hip_ctx* _ctx;
  ImageSize _size;
};
}  // namespace detail

/* bpvo::VisualOdometryFrame (bpvo/vo_frame.h:21-90): a device-resident frame = slot of a Device */
class VisualOdometryFrame {
 public:
  VisualOdometryFrame(std::shared_ptr<detail::Device> dev, int slot) : _dev(dev), _slot(slot) {}
  void setData(const uint8_t* image, const float* disparity) { _dev->check(bpvo_hip_frame_set_data(_dev->ctx(), _slot, image, disparity)); }
  void setTemplate() { _dev->check(bpvo_hip_frame_set_template(_dev->ctx(), _slot)); }
  void clear() { _dev->check(bpvo_hip_frame_clear(_dev->ctx(), _slot)); }
  bool empty() const { int d = 0, t = 0; _dev->check(bpvo_hip_frame_state(_dev->ctx(), _slot, &d, &t)); return !d; }
  bool hasTemplate() const { int d = 0, t = 0; _dev->check(bpvo_hip_frame_state(_dev->ctx(), _slot, &d, &t)); return t != 0; }
  int numLevels() const { return bpvo_hip_num_levels(_dev->ctx()); }
  int numPointsAtLevel(int level) const { int n = 0; _dev->check(bpvo_hip_num_points(_dev->ctx(), _slot, level, &n)); return n; }
  /* The frame lives in device memory; what the reference hands out as references to host containers
   * (imagePointer, getDenseDescriptorAtLevel()->getChannel, getTemplateDataAtLevel()->{points, pixels, jacobians},
   * bpvo/vo_frame.h:62-80, bpvo/template_data.h:68-75) is copied out on request. */
  ImageSize levelSize(int level) const
  {
    int r = 0, c = 0;
    _dev->check(bpvo_hip_level_size(_dev->ctx(), level, &r, &c));
    return ImageSize(r, c);
  }
  int numChannels() const { return bpvo_hip_num_channels(_dev->ctx()); }
  std::vector<uint8_t> image(int level = 0) const
  {
    std::vector<uint8_t> v((size_t) levelSize(level).numel());
    _dev->check(bpvo_hip_get_image(_dev->ctx(), _slot, level, v.data()));
    return v;
  }
  std::vector<float> descriptorChannel(int level, int channel) const
  {
    std::vector<float> v((size_t) levelSize(level).numel());
    _dev->check(bpvo_hip_get_descriptor_channel(_dev->ctx(), _slot, level, channel, v.data()));
    return v;
  }
  PointVector points(int level) const
  {
    PointVector v((size_t) numPointsAtLevel(level));
    if(!v.empty()) _dev->check(bpvo_hip_get_points(_dev->ctx(), _slot, level, v[0].data()));
    return v;
  }
  std::vector<float> pixels(int level) const           // [channel][point], TemplateData::pixels()
  {
    std::vector<float> v((size_t) numPointsAtLevel(level) * numChannels());
    if(!v.empty()) _dev->check(bpvo_hip_get_pixels(_dev->ctx(), _slot, level, v.data()));
    return v;
  }
  std::vector<float> jacobians(int level) const        // [channel][point][6], TemplateData::jacobians()
  {
    std::vector<float> v((size_t) numPointsAtLevel(level) * numChannels() * 6);
    if(!v.empty()) _dev->check(bpvo_hip_get_jacobians(_dev->ctx(), _slot, level, v.data()));
    return v;
  }
  int slot() const { return _slot; }
  const std::shared_ptr<detail::Device>& device() const { return _dev; }
 private:
  std::shared_ptr<detail::Device> _dev;
  int _slot;
};

/* bpvo::VisualOdometryPoseEstimator (bpvo/vo_pose_estimator.h:34-64) */
class VisualOdometryPoseEstimator {
 public:
  explicit VisualOdometryPoseEstimator(std::shared_ptr<detail::Device> dev, int workspace = 0) : _dev(dev), _ws(workspace) {}
  std::vector<OptimizerStatistics> estimatePose(const VisualOdometryFrame* ref_frame, const VisualOdometryFrame* cur_frame,
                                                const Matrix44& T_init, Matrix44& T_est)
  {
    std::vector<bpvo_hip_stats> st(bpvo_hip_num_levels(_dev->ctx()));
    _dev->check(bpvo_hip_estimate_pose(_dev->ctx(), _ws, ref_frame->slot(), cur_frame->slot(), T_init.data(), T_est.data(), st.data()));
    std::vector<OptimizerStatistics> ret;
    for(const auto& s : st) ret.push_back(OptimizerStatistics(s));
    return ret;
  }
  float getFractionOfGoodPoints(float thresh) const
  {
    float f = 0.0f;
    _dev->check(bpvo_hip_fraction_good(_dev->ctx(), _ws, thresh, &f));
    return f;
  }
  const WeightsVector& getWeights() const
  {
    size_t n = 0;
    _dev->check(bpvo_hip_get_weights(_dev->ctx(), _ws, nullptr, &n));
    _weights.resize(n);
    if(n) _dev->check(bpvo_hip_get_weights(_dev->ctx(), _ws, _weights.data(), &n));
    return _weights;
  }
 private:
  std::shared_ptr<detail::Device> _dev;
  int _ws;
  mutable WeightsVector _weights;
};

/* Block-matching parameters of the reference's StereoAlgorithm (utils/stereo_algorithm.cc:63-82: the CvStereoBMState fields it sets,
 * with its defaults; numberOfDisparities "must be provided") */
struct StereoParameters : bpvo_hip_stereo_params {
  explicit StereoParameters(int number_of_disparities)
  {
    bpvo_hip_default_stereo_params(this);
    numberOfDisparities = number_of_disparities;
  }
  /* `StereoAlgorithm = SGM` (utils/stereo_algorithm.cc:42-59): the in-tree semi-global matcher with SgmStereo::Config() defaults
   * (utils/sgm.cc:47-56; the KITTI evaluation of the reference runs it: conf/kitti_eval.cfg:27) */
  static StereoParameters SemiGlobalMatching(int number_of_disparities = 128)
  {
    StereoParameters p(number_of_disparities);
    p.algorithm = BPVO_STEREO_SGM;
    return p;
  }
  /* `StereoAlgorithm = SGBM` (utils/stereo_algorithm.cc:27-40; conf/kitti_seq_0.cfg:6): cv::StereoSGBM as the reference's constructor call
   * builds it from the config KEYS (defaults of the cf.get calls) — nine positional arguments into a constructor of eleven, the keys land one
   * slot off (include/bpvo_hip/c_api.h).  Set the cv::StereoSGBM fields of the struct directly for the matcher as OpenCV documents it. */
  static StereoParameters SemiGlobalBlockMatching(int minDisparity, int numberOfDisparities, int SADWindowSize = 3, int P1 = 0, int P2 = 0,
                                                  int uniquenessRatio = 0, int speckleWindowSize = 0, int speckleRange = 0, int fullDP = 0)
  {
    StereoParameters p(numberOfDisparities);
    bpvo_hip_stereo_params_sgbm_from_config(&p, minDisparity, numberOfDisparities, SADWindowSize, P1, P2, uniquenessRatio, speckleWindowSize,
                                            speckleRange, fullDP);
    return p;
  }
};

/* bpvo::VisualOdometry (bpvo/vo.h:31-105).  The keyframe state machine of bpvo/vo.cc:125-224 runs inside the library
 * on three device-resident frames. */
class VisualOdometry {
 public:
  VisualOdometry(const Matrix33& K, float baseline, ImageSize image_size, const AlgorithmParameters& params = AlgorithmParameters(),
                 int device = 0)
      : _dev(std::make_shared<detail::Device>(K, baseline, image_size, params, 3, 1, device)), _max_test_level(params.maxTestLevel) {}

  /* reference: template <class CalibrationT> VisualOdometry(const CalibrationT&, ImageSize, const AlgorithmParameters&)
   * (bpvo/vo.h:49-52): any type with members K (Matrix33) and baseline */
  template <class CalibrationT>
  VisualOdometry(const CalibrationT& calib, ImageSize image_size, const AlgorithmParameters& params = AlgorithmParameters())
      : VisualOdometry(calib.K, calib.baseline, image_size, params) {}

  /* reference: template <class DataLoaderT> VisualOdometry(const DataLoaderT*, const AlgorithmParameters&) (bpvo/vo.h:57-60):
   * any type with calibration() and imageSize() — what apps/vo_perf.cc:56 calls with its Dataset */
  template <class DataLoaderT>
  VisualOdometry(const DataLoaderT* data_loader, const AlgorithmParameters& params = AlgorithmParameters())
      : VisualOdometry(data_loader->calibration(), data_loader->imageSize(), params) {}

  /* reference: template <class FramePointer> Result addFrame(const FramePointer&) (bpvo/vo.h:76-80): a (smart) pointer to a
   * frame whose image() / disparity() return matrices with a cv::Mat-style ptr<T>() (utils/dataset.h DatasetFrame) */
  template <class FramePointer>
  Result addFrame(const FramePointer& frame)
  {
    return this->addFrame(frame->image().template ptr<const uint8_t>(), frame->disparity().template ptr<const float>());
  }

  /* reference: Result addFrame(const uint8_t* image, const float* disparity) (bpvo/vo.h:71, bpvo/vo.cc:66-72) */
  Result addFrame(const uint8_t* image, const float* disparity)
  {
    if(image == nullptr || disparity == nullptr) throw Error("nullptr image/disparity");
    bpvo_hip_result r;
    _dev->check(bpvo_hip_add_frame(_dev->ctx(), image, disparity, &r));
    return makeResult(r);
  }

  /* The reference's apps compute the disparity with StereoAlgorithm::run(left, right, dmap) (utils/stereo_algorithm.h:24-30; block
   * matching by default) and pass it to addFrame.  Here both steps run on the device and the f32 map never crosses the bus. */
  Result addFrame(const uint8_t* left, const uint8_t* right, const StereoParameters& stereo)
  {
    if(left == nullptr || right == nullptr) throw Error("nullptr image");
    bpvo_hip_result r;
    _dev->check(bpvo_hip_add_frame_stereo(_dev->ctx(), left, right, &stereo, &r));
    return makeResult(r);
  }

  int numPointsAtLevel(int level = -1) const                     // bpvo/vo.h:86
  {
    int n = 0;
    _dev->check(bpvo_hip_vo_num_points_at_level(_dev->ctx(), level, &n));
    return n;
  }
  const PointVector& pointsAtLevel(int level = -1) const         // bpvo/vo.h:92
  {
    _points.resize(numPointsAtLevel(level));
    if(!_points.empty()) _dev->check(bpvo_hip_vo_points_at_level(_dev->ctx(), level, _points[0].data()));
    return _points;
  }
  const Trajectory& trajectory() const { return _trajectory; }   // bpvo/vo.h:98
  /* scheduling options of the device context (c_api.h "Options"; the reference has none) */
  void setOption(const std::string& name, double value) { _dev->setOption(name, value); }
  double getOption(const std::string& name) const { return _dev->getOption(name); }

 private:
  Result makeResult(const bpvo_hip_result& r)
  {
    Result ret;
    std::memcpy(ret.pose.data(), r.pose, sizeof(r.pose));
    std::memcpy(ret.covariance.data(), r.covariance, sizeof(r.covariance));
    for(int i = 0; i < r.numLevels; ++i) ret.optimizerStatistics.push_back(OptimizerStatistics(r.optimizerStatistics[i]));
    ret.isKeyFrame = r.isKeyFrame != 0;
    ret.keyFramingReason = static_cast<KeyFramingReason>(r.keyFramingReason);
    if(r.hasPointCloud) {
      size_t n = 0;
      _dev->check(bpvo_hip_get_point_cloud(_dev->ctx(), nullptr, &n, nullptr));
      ret.pointCloud.reset(new PointCloud);
      ret.pointCloud->points().resize(n);
      _dev->check(bpvo_hip_get_point_cloud(_dev->ctx(), ret.pointCloud->points().data(), &n, ret.pointCloud->pose().data()));
    }
    int nt = 0;
    _dev->check(bpvo_hip_trajectory_size(_dev->ctx(), &nt));
    _trajectory._poses.resize(nt);
    if(nt) _dev->check(bpvo_hip_get_trajectory(_dev->ctx(), _trajectory._poses[0].data()));
    return ret;
  }
  std::shared_ptr<detail::Device> _dev;
  int _max_test_level;
  Trajectory _trajectory;
  mutable PointVector _points;
};

}  // namespace bpvo

#endif  // BPVO_HIP_VO_HPP
