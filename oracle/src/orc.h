/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Plain C++ CPU restatement of bpvo's dense photometric-alignment hot path.  It exists to
 * check the HIP path (tests/, __graft_entry__.smoke()) and to be timed as the CPU baseline
 * (bench.py `cpu_baseline`, kind "port").  Nothing in the product path (bpvo_amd/, include/)
 * may include, link or call it.
 *
 * PARITY UNPINNED (except three small pieces, below): the reference (/root/reference) cannot be
 * built here (Eigen, OpenCV, TBB, Boost absent; every hot-path source includes them) and its
 * test/ directory holds no assertion, golden value or fixture.
 * Pinned against the reference's own code, compiled from its tree into oracle/_ref (make ref;
 * tests/test_reference_pins_cpu.py): bpvo::median (the selection rule of the robust scale), the
 * v128 byte operators the census is composed of, simd::dot / simd::abs, and the ConfigFile reader
 * — the only parts on or next to the path that compile without Eigen / OpenCV.  The oracle therefore restates the
 * reference's sources line by line (citations on every function, paths relative to the
 * reference checkout) and restates the published semantics of the third-party calls on the
 * path (OpenCV 2.4.x pyrDown / GaussianBlur / convertTo, Eigen 3.2.x LDLT / isApprox /
 * fixed-size products), which nothing in the reference pins.
 *
 * Deviations from the reference, all deliberate:
 *   Q13  Jacobians keep the SSE code's a * rcp(b) structure with the exact reciprocal 1.0f/b instead of the
 *        approximate, vendor-specific _mm_rcp_ps.
 *   Q15  the serial (non-TBB) index-order reduction is what is restated.
 */
#ifndef BPVO_ORACLE_ORC_H
#define BPVO_ORACLE_ORC_H

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace orc {

// ---- enums (bpvo/types.h:127-169, 418-441): same numeric values
enum { kHuber = 0x10, kTukey = 0x11, kL2 = 0x12 };
enum { kIntensity = 0x30, kIntensityAndGradient = 0x31, kDescriptorFieldsFirstOrder = 0x32, kDescriptorFieldsSecondOrder = 0x33,
       kLatch = 0x34, kCentralDifference = 0x35, kLaplacian = 0x36, kBitPlanes = 0x37 };
enum { kCD3 = 0, kCD5 = 1 };
enum { kLinear = 0, kCosine = 1, kCubic = 2, kCubicHermite = 3 };
enum { kParameterTolReached = 0x30, kFunctionTolReached, kGradientTolReached, kMaxIterations, kSolverError };
enum { kLargeTranslation = 0x40, kLargeRotation, kSmallFracOfGoodPoints, kNoKeyFraming, kFirstFrame };

// POD mirror of AlgorithmParameters (bpvo/types.h:171-413), layout shared with the C ABI.
struct Params {
  int numPyramidLevels, minImageDimensionForPyramid;
  float sigmaPriorToCensusTransform, sigmaBitPlanes, dfSigma1, dfSigma2;
  int latchNumBytes, latchRotationInvariance, latchHalfSsdSize, centralDifferenceRadius;
  float centralDifferenceSigmaBefore, centralDifferenceSigmaAfter;
  int laplacianKernelSize, maxIterations;
  float parameterTolerance, functionTolerance, gradientTolerance;
  int relaxTolerancesForCoarseLevels, gradientEstimation, interp, lossFunction, descriptor, verbosity;
  float minTranslationMagToKeyFrame, minRotationMagToKeyFrame, maxFractionOfGoodPointsToKeyFrame, goodPointThreshold;
  int minNumPixelsForNonMaximaSuppression, nonMaxSuppRadius, minNumPixelsToWork;
  float minSaliency, minValidDisparity, maxValidDisparity;
  int maxTestLevel, withNormalization;
};
void defaultParams(Params& p);   // bpvo/types.cc:31-66

struct Stats { int numIterations; float finalError; float firstOrderOptimality; int status; };  // types.cc:306-310

// Row-major small matrices. M44 m[r*4+c].
struct M44 { float m[16]; };
struct M33 { float m[9]; };
M44 identity44();
M44 mul44(const M44& a, const M44& b);          // Eigen fixed 4x4 f32 product, index-order sums
M44 inverse44(const M44& a);                    // general cofactor inverse (Eigen Matrix4f::inverse, unpinned)

// ---- imgproc.cc
void pyrDownU8(const uint8_t* src, int rows, int cols, std::vector<uint8_t>& dst, int& drows, int& dcols); // cv::pyrDown
void gaussianBlurF32_5x5(const float* src, int rows, int cols, float sigma, float* dst);   // cv::GaussianBlur f32 5x5
constexpr int kMaxGaussTaps = 31;                                                          // widest kernel restated
int imsmoothTaps(float sigma);                                                             // bpvo/imgproc.cc:168: max(5, 2*round(sigma)+1)
int autoGaussTapsF32(float sigma);                                                         // cv::GaussianBlur(Size(), sigma) on CV_32F: cvRound(sigma*8+1)|1
void gaussianBlurF32(const float* src, int rows, int cols, int ksize, float sigma, float* dst);    // k = 5: the small-kernel form, k >= 7: generic
void gaussianBlurU8(const uint8_t* src, int rows, int cols, int ksize, float sigma, uint8_t* dst); // 8-bit fixed point, any odd k >= 5
void gaussianBlurU8_3x3(const uint8_t* src, int rows, int cols, float sigma, uint8_t* dst); // cv::GaussianBlur u8 3x3 (2.4)
void gaussianBlurU8_5x5(const uint8_t* src, int rows, int cols, float sigma, uint8_t* dst); // cv::GaussianBlur u8 5x5 (2.4)
void census(const uint8_t* src, int rows, int cols, float sigma_ct, uint8_t* dst);          // bpvo/census.cc:59-91
void gradientAbsoluteMagnitude(const float* src, int rows, int cols, float* dst);           // bpvo/imgproc.cc:45-74
void gradientAbsoluteMagnitudeAcc(const float* src, int rows, int cols, float* dst);        // bpvo/imgproc.cc:104-127
bool isLocalMax(const float* ptr, int stride, int radius, int row, int col);                // bpvo/imgproc.h:117-160

// stereo.cc — the stereo front-end: OpenCV 2.4 block matching with the reference's parameters
// (utils/stereo_algorithm.cc:63-82); defaults = cvCreateStereoBMState(CV_STEREO_BM_BASIC) as the reference overrides them
struct StereoParams {
  int preFilterCap = 31, SADWindowSize = 15, minDisparity = 0, numberOfDisparities = 64, textureThreshold = 10, uniquenessRatio = 15;
};
void stereoPrefilterXSobel(const uint8_t* src, int rows, int cols, int ftzero, uint8_t* dst);
void stereoBlockMatching(const uint8_t* left, const uint8_t* right, int rows, int cols, const StereoParams& sp, int16_t* disp);
void stereoBM(const uint8_t* left, const uint8_t* right, int rows, int cols, const StereoParams& sp, float* dmap);
// SgmStereo::Config (utils/sgm.h:33-46; defaults utils/sgm.cc:47-56 = what utils/stereo_algorithm.cc:46-56 reads from the config)
struct SgmParams {
  int numberOfDisparities = 128, sobelCapValue = 15, censusRadius = 2, windowRadius = 2;
  int smoothnessPenaltySmall = 100, smoothnessPenaltyLarge = 1600, consistencyThreshold = 1;
  double disparityFactor = 256.0, censusWeightFactor = 1.0 / 6.0;
};
bool stereoSGM(const uint8_t* left, const uint8_t* right, int rows, int cols, const SgmParams& sp, float* dmap);
// sgbm.cc — cv::StereoSGBM of OpenCV 2.4 (the fields of the class; the reference's positional constructor call that fills them:
// utils/stereo_algorithm.cc:27-40, Q22 in sgbm.cc)
struct SgbmParams {
  int minDisparity = 0, numberOfDisparities = 64, SADWindowSize = 3, P1 = 0, P2 = 0, disp12MaxDiff = 0, preFilterCap = 0, uniquenessRatio = 0,
      speckleWindowSize = 0, speckleRange = 0, fullDP = 0;
};
bool stereoSGBM(const uint8_t* left, const uint8_t* right, int rows, int cols, const SgbmParams& sp, float* dmap);

// ---- descriptor (dense_descriptor.*, intensity_descriptor.cc, bitplanes_descriptor.cc)
struct Descriptor {
  int rows = 0, cols = 0;
  std::vector<std::vector<float>> ch;     // planar channels, like std::array<cv::Mat,8>
  int numChannels() const { return (int) ch.size(); }
};
void computeDescriptor(const Params& p, const uint8_t* img, int rows, int cols, Descriptor& d, int nthreads);
void computeSaliencyMap(const Descriptor& d, std::vector<float>& S);   // dense_descriptor.cc:92-100

// ---- warp (rigid_body_warp.h, warps.cc)
struct Warp {
  float K[9]; float b;           // level intrinsics / baseline
  float P[12];                   // 3x4 K*T[0:3]
  M44 T, T_inv;                  // normalisation
  int dspace = 0;                // 1: DisparitySpaceWarp (bpvo/disparity_space_warp.{h,cc}) instead of RigidBodyWarp; P then
                                 // holds rows 0, 1, 3 of H = G * T * G_inv (operator(): x = pw0/pw3 + cx, y = pw1/pw3 + cy)
  void init(const float K_[9], float b_);
  void makePoint(float x, float y, float d, float out[4]) const;          // rigid_body_warp.h:47-60
  void setNormalization(const std::vector<float>& pts);                   // warps.cc:27-48, rigid_body_warp.h:62-71
  void setPose(const M44& pose);                                          // rigid_body_warp.h:111-114
  void computeJacobian(const float* pts, int N, const float* IxIy, float* J) const;  // rigid_body_warp.cc:60-315 (Q13)
  M44 paramsToPose(const float p[6]) const;                               // rigid_body_warp.h:130-138
};
M44 twistToMatrix(const float p[6]);                                      // bpvo/math_utils.h:140-168

// ---- template (template_data.cc)
struct TemplateData {
  int level = 0; Params params; Warp warp;
  std::vector<float> points;      // [N][4]
  std::vector<int>   inds;        // [N] y*cols+x (valid_inds)
  std::vector<float> pixels;      // [C*N] channel-major
  std::vector<float> jacobians;   // [C*N][6] (+ the reference's zero pad is not materialised, Q10)
  std::vector<float> saliency;    // kept for parity inspection
  int numChannels = 0;
  int fast_warp = 0;              // 1: projectPoints / BilinearInterp all-f32 formulation (PHOTO_ERROR_OPT branch)
                                  // 2: the same with DisparitySpaceWarp as the warp (warp.dspace = 1)
  int numPoints() const { return (int)(points.size() / 4); }
  void setData(const Descriptor& desc, const float* D, int Dcols);        // template_data.cc:37-142
  // template_data.cc:174-189 + photo_error.cc:344-451 (standard branch)
  void computeResiduals(const Descriptor& desc, const M44& pose, std::vector<float>& residuals,
                        std::vector<uint16_t>& valid, int nthreads);
};

// ---- mestimator.cc
struct AutoScaleEstimator {
  float scale = 1.0f, delta_scale = 1e10f, tol = 1e-6f;
  std::vector<float> buffer;
  void reset() { delta_scale = 1e10f; scale = 1.0f; }
  float estimateScale(const std::vector<float>& r, const std::vector<uint16_t>& valid);   // mestimator.cc:467-490
};
float medianOf(std::vector<float>& data);                                                   // bpvo/utils.h:224-252
void computeWeights(int loss, const std::vector<float>& r, const std::vector<uint16_t>& valid, float sigma,
                    std::vector<float>& w);                                                 // mestimator.cc:390-415

// ---- linear_system_builder.cc:140-266,334-350 ; returns sqrt(sum w v r^2)
float linearSystemRun(const std::vector<float>& J, const std::vector<float>& r, const std::vector<float>& w,
                      const std::vector<uint16_t>& valid, float H[36], float G[6], int nthreads);

// ---- solver (pose_estimator_base.h:67-151 with Eigen 3.2 LDLT semantics)
bool solveSystem(const float H[36], const float G[6], float dp[6]);

// ---- GN driver (pose_estimator_base.h:324-407, pose_estimator_gn.h:70-100)
struct IterationRecord { M44 T; float H[36]; float G[6]; float f_norm; float sigma; int num_valid; float dp[6]; };
struct PoseEstimator {
  // PoseEstimatorParameters (pose_estimator_params.h:30-56, .cc:27-33; Q4: maxFuncEvals stays 1200)
  int maxIterations = 50, maxFuncEvals = 1200;
  float functionTolerance = 1e-6f, parameterTolerance = 1e-6f, gradientTolerance = 1e-6f;
  int lossFunction = kHuber;
  int nthreads = 1;
  // 0: the reference's f32 accumulation (serial, or range-split with nthreads > 1).  1: the same sums accumulated in f64 — NOT a
  // mode of the reference: a test instrument that shows what a reduction without f32 accumulation noise (the GPU's tree +
  // f64 combine is within 4e-6 of it) does to the termination tests (tests/test_oracle_cpu.py, tests/tools/fuzz_regressions.txt)
  int reduction = 0;
  // Test instrument, NOT a mode of the reference: every linearisation's (H, G) multiplied entry by entry (H symmetrically) by
  // 1 + perturb_rel * n, n ~ N(0, 1) from a generator seeded with (perturb_seed, linearisation count) — the size of the difference
  // between two f32 summation orders.  The randomised parity tool asks the oracle with it whether a problem's final pose is
  // decided by such differences (tests/tools/fuzz_parity.py, rule "solver-fallback-edge").
  float perturb_rel = 0.0f; uint32_t perturb_seed = 0; uint32_t perturb_count = 0;
  AutoScaleEstimator scale_estimator;
  std::vector<float> residuals, weights;
  std::vector<uint16_t> valid;        // replicated to C*N after linearize (Q12 / base.h:307-320)
  float f_norm_prev = 0.0f, g_tol = 0.0f; int num_fun_evals = 0;
  std::vector<IterationRecord>* trace = nullptr;   // optional per-linearisation trace
  float last_sigma = 1.0f; int last_num_valid = 0;
  void setParameters(const Params& p);
  float linearize(TemplateData* tdata, const Descriptor& desc, const M44& T, float H[36], float G[6]);
  Stats run(TemplateData* tdata, const Descriptor& desc, M44& T);
};

// ---- frame (vo_frame.cc)
struct Frame {
  Params params; int rows = 0, cols = 0;
  bool has_data = false, has_template = false;
  std::vector<uint8_t> image; std::vector<float> disparity;
  std::vector<std::vector<uint8_t>> pyr; std::vector<int> prow, pcol;
  std::vector<Descriptor> desc;
  std::vector<TemplateData> tdata;
  int nthreads = 1;
  void init(const float K[9], float b, int rows_, int cols_, const Params& p);   // vo_frame.cc:13-29
  void setData(const uint8_t* img, const float* disp);                           // vo_frame.cc:48-55
  void setTemplate();                                                            // vo_frame.cc:61-93
  void clear() { has_data = false; has_template = false; }
};

// ---- estimatePose (vo_pose_estimator.cc:63-107)
struct VoPoseEstimator {
  Params params; PoseEstimator est;
  void init(const Params& p, int nthreads);
  void estimatePose(Frame* ref, Frame* cur, const M44& T_init, M44& T_est, std::vector<Stats>& stats);
  float getFractionOfGoodPoints(float thresh) const;
};

struct PointWithInfo { float xyzw[4]; uint8_t rgba[4]; float weight; char pad[8]; };
struct Result {
  M44 pose; float covariance[36]; std::vector<Stats> stats; bool isKeyFrame; int keyFramingReason;
  bool hasPointCloud; std::vector<PointWithInfo> cloud; M44 cloudPose;
};

// ---- VisualOdometry (vo.cc:97-281) + Trajectory (trajectory.cc:30-50)
struct VisualOdometry {
  Params params; int rows, cols; float K[9];
  VoPoseEstimator vo_pose;
  std::unique_ptr<Frame> ref, cur, prev;
  M44 T_kf;
  std::vector<M44> trajectory;
  void init(const float K[9], float b, int rows_, int cols_, const Params& p, int nthreads);
  void addFrame(const uint8_t* I, const float* D, Result& r);
  int shouldKeyFrame(const M44& pose) const;
  void trajectoryPush(const M44& T);
};

}  // namespace orc
#endif
