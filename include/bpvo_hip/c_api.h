/*
 * bpvo_hip — C ABI of the MI355X-native dense photometric alignment path.
 *
 * This is the drop-in boundary.  bpvo has no FFI
 * layer of its own; the seams this ABI replaces are cited per entry point as
 * `reference: <file>:<line>` (paths relative to the reference checkout).
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, <0 = error (never throws).
 *     bpvo_hip_last_error() gives the message; the C++ facade (vo.hpp) maps a
 *     non-zero status to bpvo::Error like THROW_ERROR does (bpvo/utils.h:211-220).
 *   - matrices are ROW-MAJOR float arrays (K[9], T[16], H[36]).
 *   - host pointers unless a function name ends in `_device`.
 *   - images are contiguous row-major rows x cols, u8 image + f32 disparity, as
 *     VisualOdometry::addFrame takes them (bpvo/vo.h:71).  The caller keeps
 *     ownership; data is copied before the call returns (bpvo/vo_frame.cc:50-51).
 *   - per-point arrays handed back to the host use the REFERENCE layouts:
 *     channel-major `a[c*N + i]` (bpvo/template_data.cc:96-97,108,133),
 *     Jacobians `[C*N][6]`, valid flags uint16_t (bpvo/types.h:69-73).
 *   - one ctx per (host thread, device); calls on one ctx are serialised by the
 *     caller (the reference objects are not re-entrant either,
 *     bpvo/template_data.h:80,86).
 */
#ifndef BPVO_HIP_C_API_H
#define BPVO_HIP_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums: numeric values equal to the reference's (bpvo/types.h:127-169,418-441) */
enum { BPVO_LOSS_HUBER = 0x10, BPVO_LOSS_TUKEY = 0x11, BPVO_LOSS_L2 = 0x12 };
enum { BPVO_VERB_ITERATION = 0x20, BPVO_VERB_FINAL = 0x21, BPVO_VERB_SILENT = 0x22, BPVO_VERB_DEBUG = 0x23 };
enum { BPVO_DESC_INTENSITY = 0x30, BPVO_DESC_INTENSITY_AND_GRADIENT = 0x31, BPVO_DESC_FIELDS_FIRST_ORDER = 0x32,
       BPVO_DESC_FIELDS_SECOND_ORDER = 0x33, BPVO_DESC_LATCH = 0x34, BPVO_DESC_CENTRAL_DIFFERENCE = 0x35, BPVO_DESC_LAPLACIAN = 0x36, BPVO_DESC_BITPLANES = 0x37 };
enum { BPVO_GRAD_CD3 = 0, BPVO_GRAD_CD5 = 1 };
enum { BPVO_INTERP_LINEAR = 0, BPVO_INTERP_COSINE = 1, BPVO_INTERP_CUBIC = 2, BPVO_INTERP_CUBIC_HERMITE = 3 };
enum { BPVO_STATUS_PARAMETER_TOL = 0x30, BPVO_STATUS_FUNCTION_TOL = 0x31, BPVO_STATUS_GRADIENT_TOL = 0x32,
       BPVO_STATUS_MAX_ITERATIONS = 0x33, BPVO_STATUS_SOLVER_ERROR = 0x34 };
enum { BPVO_KF_LARGE_TRANSLATION = 0x40, BPVO_KF_LARGE_ROTATION = 0x41, BPVO_KF_SMALL_FRAC_GOOD = 0x42,
       BPVO_KF_NO_KEYFRAMING = 0x43, BPVO_KF_FIRST_FRAME = 0x44 };

/* ---- error codes */
enum { BPVO_OK = 0, BPVO_ERR_INVALID_ARG = -1, BPVO_ERR_UNSUPPORTED = -2, BPVO_ERR_NO_DATA = -3,
       BPVO_ERR_NO_TEMPLATE = -4, BPVO_ERR_DEVICE = -5, BPVO_ERR_NO_DEVICE = -6 };

/* POD mirror of bpvo::AlgorithmParameters, same 35 fields in the same order
 * (reference: bpvo/types.h:171-413; defaults bpvo/types.cc:31-66). bool -> int. */
typedef struct bpvo_hip_params {
  int   numPyramidLevels;
  int   minImageDimensionForPyramid;
  float sigmaPriorToCensusTransform;
  float sigmaBitPlanes;
  float dfSigma1;
  float dfSigma2;
  int   latchNumBytes;
  int   latchRotationInvariance;
  int   latchHalfSsdSize;
  int   centralDifferenceRadius;
  float centralDifferenceSigmaBefore;
  float centralDifferenceSigmaAfter;
  int   laplacianKernelSize;
  int   maxIterations;
  float parameterTolerance;
  float functionTolerance;
  float gradientTolerance;
  int   relaxTolerancesForCoarseLevels;
  int   gradientEstimation;
  int   interp;
  int   lossFunction;
  int   descriptor;
  int   verbosity;
  float minTranslationMagToKeyFrame;
  float minRotationMagToKeyFrame;
  float maxFractionOfGoodPointsToKeyFrame;
  float goodPointThreshold;
  int   minNumPixelsForNonMaximaSuppression;
  int   nonMaxSuppRadius;
  int   minNumPixelsToWork;
  float minSaliency;
  float minValidDisparity;
  float maxValidDisparity;
  int   maxTestLevel;
  int   withNormalization;
} bpvo_hip_params;

/* reference: bpvo::OptimizerStatistics (bpvo/types.h:444-482; ctor bpvo/types.cc:306-310) */
typedef struct bpvo_hip_stats {
  int   numIterations;
  float finalError;
  float firstOrderOptimality;
  int   status;
} bpvo_hip_stats;

#define BPVO_HIP_MAX_LEVELS 8

/* reference: bpvo::Result (bpvo/types.h:489-563) minus the point cloud, which is
 * fetched separately with bpvo_hip_get_point_cloud when isKeyFrame && hasPointCloud */
typedef struct bpvo_hip_result {
  float pose[16];
  float covariance[36];                   /* never written by the reference: Identity (Q16) */
  bpvo_hip_stats optimizerStatistics[BPVO_HIP_MAX_LEVELS];
  int   numLevels;
  int   isKeyFrame;
  int   keyFramingReason;
  int   hasPointCloud;
} bpvo_hip_result;

/* reference: bpvo::PointWithInfo, 32-byte record (bpvo/point_cloud.h:30-62) */
typedef struct bpvo_hip_point_with_info {
  float   xyzw[4];
  uint8_t rgba[4];
  float   weight;
  char    pad[8];
} bpvo_hip_point_with_info;

typedef struct bpvo_hip_ctx bpvo_hip_ctx;

/* Fill `p` with AlgorithmParameters() defaults (reference: bpvo/types.cc:31-66). */
void bpvo_hip_default_params(bpvo_hip_params* p);

/*
 * Create a context: the device-resident equivalent of
 * VisualOdometry::Impl's three VisualOdometryFrame objects + pose estimator
 * (reference: bpvo/vo.cc:97-115, bpvo/vo_frame.cc:13-29).
 *   n_frames   frame slots (each = image pyramid + disparity + descriptor pyramid + template pyramid)
 *   n_pairs    estimation workspaces (residuals, valid, weights, GN state); 1 for sequential VO,
 *              B for batches of independent frame pairs
 *   device     HIP device ordinal
 * numPyramidLevels <= 0 is resolved like bpvo/vo.cc:103-107.
 */
int bpvo_hip_create(bpvo_hip_ctx** out, const float K[9], float baseline, int rows, int cols,
                    const bpvo_hip_params* p, int device, int n_frames, int n_pairs);
void bpvo_hip_destroy(bpvo_hip_ctx* ctx);
const char* bpvo_hip_last_error(const bpvo_hip_ctx* ctx);   /* ctx may be NULL: last create error */
int bpvo_hip_num_levels(const bpvo_hip_ctx* ctx);
int bpvo_hip_num_channels(const bpvo_hip_ctx* ctx);
int bpvo_hip_level_size(const bpvo_hip_ctx* ctx, int level, int* rows, int* cols);

/* ---- VisualOdometryFrame (reference: bpvo/vo_frame.h:21-90) ------------------------------- */
/* setData: copy image+disparity, build image pyramid + descriptor pyramid
 * (reference: bpvo/vo_frame.cc:48-55 -> dense_descriptor_pyramid.cc:67-78, image_pyramid.cc:43-50). */
int bpvo_hip_frame_set_data(bpvo_hip_ctx* ctx, int slot, const uint8_t* image, const float* disparity);
int bpvo_hip_frame_set_data_device(bpvo_hip_ctx* ctx, int slot, const uint8_t* d_image, const float* d_disparity);
/* setTemplate: pixel selection, 3-D points, normalisation, pixels+gradients+Jacobians for levels
 * L-1..maxTestLevel (reference: bpvo/vo_frame.cc:61-93 -> bpvo/template_data.cc:37-142). */
int bpvo_hip_frame_set_template(bpvo_hip_ctx* ctx, int slot);
int bpvo_hip_frame_clear(bpvo_hip_ctx* ctx, int slot);                /* vo_frame.h:47 */
int bpvo_hip_frame_state(const bpvo_hip_ctx* ctx, int slot, int* has_data, int* has_template);

/* batched forms: `count` slots first_slot, first_slot+stride, ... ; inputs are `count` images /
 * disparities back to back.  One launch per stage covers all of them. */
int bpvo_hip_frames_set_data(bpvo_hip_ctx* ctx, int first_slot, int slot_stride, int count,
                             const uint8_t* images, const float* disparities, int on_device);
int bpvo_hip_frames_set_template(bpvo_hip_ctx* ctx, int first_slot, int slot_stride, int count);

/* accessors (parity surface) */
int bpvo_hip_get_image(bpvo_hip_ctx* ctx, int slot, int level, uint8_t* out);               /* image_pyramid.cc:43-50 */
int bpvo_hip_get_descriptor_channel(bpvo_hip_ctx* ctx, int slot, int level, int channel,
                                    float* out);                                         /* dense_descriptor.h getChannel */
int bpvo_hip_get_saliency(bpvo_hip_ctx* ctx, int slot, int level, float* out);               /* dense_descriptor.cc:92-100 */
int bpvo_hip_num_points(bpvo_hip_ctx* ctx, int slot, int level, int* n);                      /* template_data.h numPoints */
int bpvo_hip_get_points(bpvo_hip_ctx* ctx, int slot, int level, float* xyzw /*[N][4]*/);      /* template_data.h points */
int bpvo_hip_get_point_indices(bpvo_hip_ctx* ctx, int slot, int level, int* inds /*[N], y*W+x*/);
int bpvo_hip_get_pixels(bpvo_hip_ctx* ctx, int slot, int level, float* pixels /*[C*N]*/);     /* template_data.h pixels */
int bpvo_hip_get_jacobians(bpvo_hip_ctx* ctx, int slot, int level, float* J /*[C*N][6]*/);    /* template_data.h jacobians */
int bpvo_hip_get_normalization(bpvo_hip_ctx* ctx, int slot, int level, float T[16], float T_inv[16]); /* rigid_body_warp.h:62-71 */

/* ---- operator-level seam used by PoseEstimatorGN::linearize (reference: bpvo/pose_estimator_gn.h:70-81):
 * computeResiduals + estimateScale + ComputeWeights + LinearSystemBuilder::Run at pose T.
 * reset_scale != 0 resets the AutoScaleEstimator first (pose_estimator_base.h:287-293).
 * Outputs: H (6x6 row-major, symmetric), G, f_norm = sqrt(sum w v r^2), sigma, number of valid points. */
int bpvo_hip_linearize(bpvo_hip_ctx* ctx, int ws, int ref_slot, int cur_slot, int level, const float T[16],
                       int reset_scale, float H[36], float G[6], float* f_norm, float* sigma, int* num_valid);
/* The same seam with the robust scale GIVEN instead of estimated: computeResiduals at T, then MEstimator::ComputeWeights(r, sigma) and
 * LinearSystemBuilder::Run (bpvo/pose_estimator_gn.h:72,76-79 without :74).  The scale estimator of the workspace is left frozen at
 * sigma.  For comparing H, G and f_norm with a reference run at a pose where that run's sigma is known (its freeze history, Q6, is
 * not reproducible from the pose alone). */
int bpvo_hip_linearize_at_scale(bpvo_hip_ctx* ctx, int ws, int ref_slot, int cur_slot, int level, const float T[16],
                                float sigma, float H[36], float G[6], float* f_norm, int* num_valid);
int bpvo_hip_get_residuals(bpvo_hip_ctx* ctx, int ws, float* r /*[C*N]*/, size_t* n);
int bpvo_hip_get_valid(bpvo_hip_ctx* ctx, int ws, uint16_t* v /*[N]*/, size_t* n);
int bpvo_hip_get_weights(bpvo_hip_ctx* ctx, int ws, float* w /*[C*N]*/, size_t* n);          /* vo_pose_estimator.cc:95-99 */
int bpvo_hip_fraction_good(bpvo_hip_ctx* ctx, int ws, float threshold, float* frac);         /* vo_pose_estimator.cc:101-107 */

/* Selects how template points are warped and interpolated (both formulations exist in the reference):
 *   BPVO_WARP_PHOTO_ERROR_F64 (default) — the ACTIVE reference path: PhotoError "standard" branch, projection and bilinear
 *       interpolation in double, Floor(), invalid -> r = 0 (bpvo/photo_error.cc:336-459);
 *   BPVO_WARP_PROJECT_POINTS_F32 — the inactive all-float path (PHOTO_ERROR_OPT): projectPoints + interpolation
 *       coefficients + dot product (bpvo/project_points.cc:180-214, bpvo/photo_error.cc:82-214, bpvo/interp_util.h:49-71):
 *       (int) truncation, C = [(1-xf)(1-yf), xf(1-yf), (1-xf)yf, xf*yf] in its expanded form, invalid -> r = -I0;
 *   BPVO_WARP_DISPARITY_SPACE_F32 — DisparitySpaceWarp in the place of RigidBodyWarp (bpvo/disparity_space_warp.{h,cc}; the
 *       reference declares the class and its warp_traits but typedefs TemplateData::WarpType to RigidBodyWarp,
 *       bpvo/template_data.h:42, so no build of it runs this): template points (x - cx, y - cy, d, 1) (makePoint :31-34),
 *       H = G * T * G_inv (setPose :36), x = (H p)_0 / (H p)_3 + cx (operator() :66-71, all f32) followed by the
 *       interpolation of the projectPoints formulation, Jacobian rows of jacobian() (:40-64), no normalisation (:87-91),
 *       paramsToPose = TwistToMatrix (:79-84).  bpvo_hip_get_points then returns disparity-space points.
 * Switching to or from the disparity-space warp drops the templates (has_template = 0 on every slot; the frame data
 * stay): call it before bpvo_hip_frame_set_template / the first bpvo_hip_add_frame. */
enum { BPVO_WARP_PHOTO_ERROR_F64 = 0, BPVO_WARP_PROJECT_POINTS_F32 = 1, BPVO_WARP_DISPARITY_SPACE_F32 = 2 };
int bpvo_hip_set_warp_formulation(bpvo_hip_ctx* ctx, int mode);

/* ---- VisualOdometryPoseEstimator::estimatePose (reference: bpvo/vo_pose_estimator.cc:63-93):
 * coarse-to-fine PoseEstimatorGN::run (pose_estimator_base.h:324-407). stats[numLevels]. */
int bpvo_hip_estimate_pose(bpvo_hip_ctx* ctx, int ws, int ref_slot, int cur_slot, const float T_init[16],
                           float T_est[16], bpvo_hip_stats* stats);

/* The same estimate with one record per linearisation of the Gauss-Newton runs, in the order they happen (coarse to fine) — what the
 * reference prints per iteration at Verbosity kIteration (PoseEstimatorBase::run, bpvo/pose_estimator_base.h:231-247,373-393: iteration,
 * function value, first-order optimality, step size), with the pose of the linearisation and the system that was solved added:
 *   [0..15] T the linearisation was taken at (row-major)   [16..51] H   [52..57] G   [58] f_norm = sqrt(sum w v r^2)
 *   [59] robust scale sigma the weights used   [60] valid points   [61..66] dp solved from (H, G)   [67] pyramid level
 * Written on the device by the thread that runs the solve / pose update (gn_step, or the persistent single-pair kernel); the
 * estimate itself is the one bpvo_hip_estimate_pose gives, bit for bit.  *n_records = records written (may exceed max_records:
 * only the first max_records are copied). */
#define BPVO_HIP_TRACE_FLOATS 68
int bpvo_hip_estimate_pose_trace(bpvo_hip_ctx* ctx, int ws, int ref_slot, int cur_slot, const float T_init[16], float T_est[16],
                                 bpvo_hip_stats* stats, float* records /*[max_records][68]*/, int max_records, int* n_records);

/* ---- VisualOdometry (reference: bpvo/vo.h:42-100, bpvo/vo.cc:125-224). Uses frame slots 0..2 and
 * workspace 0 of the ctx (needs n_frames >= 3). */
int bpvo_hip_add_frame(bpvo_hip_ctx* ctx, const uint8_t* image, const float* disparity, bpvo_hip_result* result);
int bpvo_hip_vo_num_points_at_level(bpvo_hip_ctx* ctx, int level, int* n);                    /* vo.cc:226-238 */
int bpvo_hip_vo_points_at_level(bpvo_hip_ctx* ctx, int level, float* xyzw);                   /* vo.cc:240-248 */
int bpvo_hip_get_point_cloud(bpvo_hip_ctx* ctx, bpvo_hip_point_with_info* pts, size_t* n, float pose[16]); /* vo.cc:260-281 */
int bpvo_hip_trajectory_size(bpvo_hip_ctx* ctx, int* n);                                      /* trajectory.cc:42-50 */
int bpvo_hip_get_trajectory(bpvo_hip_ctx* ctx, float* poses /*[n][16]*/);

/* ---- batches of independent frame pairs (BASELINE.json config 5).
 * Pair p uses frame slots 2p (reference/template frame A) and 2p+1 (current frame B) and workspace p.
 * For each pair: A.setData, A.setTemplate, B.setData, estimatePose(A, B, Identity) -> poses[p].
 * images = [A0,B0,A1,B1,...] (2*n_pairs images), disparities likewise — B's disparity is never read: estimatePose uses the current
 * frame's descriptor only (bpvo/vo_pose_estimator.cc:63-93), so it is neither uploaded nor stored (40 % of the input bytes); the B slots
 * of a batch therefore cannot be turned into templates afterwards (bpvo_hip_frame_set_template: BPVO_ERR_NO_DATA).
 * stats = [n_pairs][numLevels].
 * A pyramid level of a pair whose template keeps no point does not fail the batch (the reference's computeResiduals throws there,
 * bpvo/template_data.cc:177, and so do the single-pair entry points: BPVO_ERR_NO_TEMPLATE): that level of that pair is skipped, its
 * statistics stay at the OptimizerStatistics() defaults {0, -1, -1, BPVO_STATUS_SOLVER_ERROR} and its pose passes through.
 * With two or more estimation lanes each lane runs its share of the pairs end to end on its own stream (frame stages queued one behind
 * the other): same results, bit for bit, as the three stages called one after the other. */
int bpvo_hip_batch_run(bpvo_hip_ctx* ctx, int n_pairs, const uint8_t* images, const float* disparities,
                       int on_device, float* poses /*[n_pairs][16]*/, bpvo_hip_stats* stats);
/* same, but only the estimatePose stage on already prepared slots */
int bpvo_hip_batch_estimate(bpvo_hip_ctx* ctx, int n_pairs, const float* T_init /*[n_pairs][16] or NULL=Identity*/,
                            float* poses, bpvo_hip_stats* stats);
/* device address of the packed result records of the last batch (32 floats per pair:
 * pose 3x4 row-major (12), twist-free pad, per-level numIterations (4..), status ...) for an RCCL gather. */
int bpvo_hip_batch_result_records_device(bpvo_hip_ctx* ctx, const float** d_records, int* floats_per_pair);
/* device-to-device copy of the first n_pairs records into caller-owned device memory (e.g. the tensor handed to
 * the RCCL gather), complete on return */
int bpvo_hip_batch_copy_records_device(bpvo_hip_ctx* ctx, float* d_dst, int n_pairs);

/* ---- stereo front-end.  reference: StereoAlgorithm (utils/stereo_algorithm.{h,cc}), BlockMatching branch:
 * cvFindStereoCorrespondenceBM with the state of utils/stereo_algorithm.cc:63-82, then disp16.convertTo(CV_32F, 1/16) (:98-111).
 * The matcher is OpenCV 2.4's (third party): restated, parity unpinned.  Invalid pixels carry minDisparity - 1 (getInvalidValue).
 * Two documented deviations where the original depends on memory layout: right-image window columns past the row end are read by
 * linear addressing, clamped at the end of the image; and for minDisparity > 0 the original's column loop runs past the end of the
 * row (its results there depend on the next row's bytes) — here and in the oracle that overrun is CUT at the last column, not
 * reproduced. */
enum { BPVO_STEREO_BLOCK_MATCHING = 0, BPVO_STEREO_SGM = 1, BPVO_STEREO_SGBM = 2 };
typedef struct bpvo_hip_stereo_params {
  int preFilterCap;          /* 31 */
  int SADWindowSize;         /* 15; odd, 5..21 on the device path */
  int minDisparity;          /* 0 */
  int numberOfDisparities;   /* no default in the reference ("must be provided"); multiple of 16, <= 256 */
  int textureThreshold;      /* 10 */
  int uniquenessRatio;       /* 15 */
  /* `StereoAlgorithm` of the reference's config file (utils/stereo_algorithm.cc:25-27,42,62): BPVO_STEREO_BLOCK_MATCHING (the default,
   * the fields above) or BPVO_STEREO_SGM — the in-tree semi-global matcher SgmStereo (utils/sgm.{h,cc}; conf/kitti_eval.cfg:27,
   * conf/kitti_stereo.cfg:5) with SgmStereo::Config below (defaults utils/sgm.cc:47-56 = utils/stereo_algorithm.cc:46-56); it reads
   * numberOfDisparities from the field above.  Invalid pixels carry 0.  Integer arithmetic restated line by line from the source,
   * which includes OpenCV and cannot be built in this image: parity unpinned, HIP = oracle bit for bit. */
  int    algorithm;
  int    sobelCapValue;            /* 15 (clamped to 15..127, made odd: utils/sgm.cc:225-226) */
  int    censusRadius;             /* 2; 1 or 2 */
  int    windowRadius;             /* 2 */
  int    smoothnessPenaltySmall;   /* 100 */
  int    smoothnessPenaltyLarge;   /* 1600 */
  int    consistencyThreshold;     /* 1 */
  int    reserved_;
  double disparityFactor;          /* 256.0 */
  double censusWeightFactor;       /* 1.0 / 6.0 */
  /* BPVO_STEREO_SGBM — `StereoAlgorithm = SGBM | SemiGlobalBlockMatching` (utils/stereo_algorithm.cc:25-40, run :113-121; selected by
   * conf/kitti_seq_0.cfg:6): cv::StereoSGBM of OpenCV 2.4 + its medianBlur(3) + filterSpeckles + convertTo(CV_32F, 1/16).  These are the
   * FIELDS OF cv::StereoSGBM; it also reads minDisparity, numberOfDisparities, SADWindowSize, preFilterCap and uniquenessRatio above.
   * The reference's constructor call passes nine positional arguments to a constructor of eleven, so its config keys land one slot off
   * (uniquenessRatio -> disp12MaxDiff, speckleWindowSize -> preFilterCap, speckleRange -> uniquenessRatio, (bool) fullDP -> speckleWindowSize):
   * bpvo_hip_stereo_params_sgbm_from_config fills the struct from the KEYS exactly like that call.  `<= 0 means` rules of
   * computeDisparitySGBM apply (SADWindowSize <= 0: 5, P1 <= 0: 2, P2 <= 0: 5 then max(P2, P1 + 1), disp12MaxDiff <= 0: 1, uniquenessRatio < 0:
   * 10, preFilterCap: max(cap, 15) | 1).  Invalid pixels carry minDisparity - 1.  OpenCV's source is absent from the reference tree:
   * restated, parity unpinned; the GPU tests hold the kernels to that restatement bit for bit.  Device path: single-pass mode (fullDP = 0: the only mode the
   * reference's call can reach), minDisparity >= 0, numberOfDisparities <= 256, window and penalties that keep the int16 buffers of the
   * original from wrapping (BPVO_ERR_UNSUPPORTED names the limit otherwise). */
  int    P1;                       /* 0 */
  int    P2;                       /* 0 */
  int    disp12MaxDiff;            /* 0 */
  int    speckleWindowSize;        /* 0 */
  int    speckleRange;             /* 0 */
  int    fullDP;                   /* 0 */
} bpvo_hip_stereo_params;
void bpvo_hip_default_stereo_params(bpvo_hip_stereo_params* p);
/* StereoAlgorithm::Impl's SGBM branch (utils/stereo_algorithm.cc:27-40): the struct the reference's constructor call produces from the config
 * KEYS minDisparity, numberOfDisparities, SADWindowSize (3), P1 (0), P2 (0), uniquenessRatio (0), speckleWindowSize (0), speckleRange (0),
 * fullDP (0) — defaults in brackets; algorithm = BPVO_STEREO_SGBM */
void bpvo_hip_stereo_params_sgbm_from_config(bpvo_hip_stereo_params* p, int minDisparity, int numberOfDisparities, int SADWindowSize, int P1, int P2,
                                             int uniquenessRatio, int speckleWindowSize, int speckleRange, int fullDP);
/* StereoAlgorithm::run for `count` rectified pairs of the ctx's image size ([count][rows*cols] u8 each, host or device) ->
 * f32 disparities [count][rows*cols] (host or device); sp->algorithm selects the matcher (the entry point keeps its name) */
int bpvo_hip_stereo_bm(bpvo_hip_ctx* ctx, int count, const uint8_t* left, const uint8_t* right, int on_device,
                       const bpvo_hip_stereo_params* sp, float* disparity, int disparity_on_device);
/* VisualOdometry::addFrame(left, StereoAlgorithm::run(left, right)): the disparity map stays on the device */
int bpvo_hip_add_frame_stereo(bpvo_hip_ctx* ctx, const uint8_t* left, const uint8_t* right, const bpvo_hip_stereo_params* sp,
                              bpvo_hip_result* result);

/* Pyramid levels that were run by the persistent single-launch Gauss-Newton kernel (groups of BPVO_HIP_PERSIST_MAX_WS or fewer
 * pairs) since the context was created, and whether such a launch ever gave up at a grid barrier (the
 * context then stays on the four-kernel chain; results are the same either way). */
int bpvo_hip_persistent_counts(bpvo_hip_ctx* ctx, uint64_t* levels, int* gave_up);

/* Upload pipeline of the last bpvo_hip_batch_run that was handed HOST buffers (batches of at least 32 pairs): wall time from the start of
 * the call until the last chunk had landed on the device, and the bytes that crossed the bus (both images, the template frame's
 * disparity).  The uploads run under the compute of the chunks before them. */
int bpvo_hip_upload_stats(bpvo_hip_ctx* ctx, double* seconds, uint64_t* bytes);

/* Batch estimates that ran their whole Gauss-Newton stage in one launch of the team-persistent kernel (batches of 2 ..
 * BPVO_HIP_TEAM_MAX_PAIRS pairs) since the context was created. */
int bpvo_hip_team_counts(bpvo_hip_ctx* ctx, uint64_t* launches);

/* ---- measurement hooks (bench.py): per-kernel HIP-event timing on the ctx's own stream */
typedef struct bpvo_hip_kernel_stat {
  char     name[48];
  uint64_t launches;
  double   total_ms;          /* sum of HIP-event durations */
  double   units;             /* units processed (points or pixels), summed over launches */
  double   bytes_per_unit;    /* algorithmic bytes per unit */
} bpvo_hip_kernel_stat;
/* 0 off; 1: HIP events around the frame stages and around every 5th warp_residual launch of a batch estimate (the
 * reported units are scaled to the sampled launches); 2: around every launch of every kernel; 3: as 1, but around EVERY
 * warp_residual and EVERY irls_reduce launch (the averages are then over the same launches as a rocprofv3 kernel trace's).
 * Resets the counters. */
int bpvo_hip_profiling(bpvo_hip_ctx* ctx, int enable);
int bpvo_hip_get_kernel_stats(bpvo_hip_ctx* ctx, bpvo_hip_kernel_stat* out, int max_out, int* n_out);
int bpvo_hip_total_linearizations(bpvo_hip_ctx* ctx, uint64_t* n);  /* GN iterations done since create/reset */
/* exact median selections served by the bracketed path / by the full 3-pass path since the last counter reset */
int bpvo_hip_median_path_counts(bpvo_hip_ctx* ctx, uint64_t* bracketed, uint64_t* full);
/* points linearised since the last counter reset: `fused` of `total` went through the fused residual + reduction path that
 * irls_reduce takes once a workspace's robust scale is frozen for the level (warp_residual skips those workspaces) */
int bpvo_hip_fused_point_counts(bpvo_hip_ctx* ctx, uint64_t* fused, uint64_t* total);
/* tap cache of warp_residual since the last counter reset: out[0] hits, out[1] lookups (= valid points), out[2] / out[3] the same over
 * the first 8 linearisations of every level (the moving-pose regime) */
int bpvo_hip_tap_cache_counts(bpvo_hip_ctx* ctx, uint64_t out[4]);
/* ---- per-context options: how the library schedules its work.  None of them changes a result (every setting is covered by a
 * bit-identity test) — with ONE exception, the validation mode "reference_reduction" at the end of the table, which changes the summation
 * order of the normal equations to the reference's own; they exist so that a caller — not the process environment — decides, per context.  Call between API calls, from the
 * thread that drives the context.  Unknown key or value out of range: BPVO_ERR_INVALID_ARG (bpvo_hip_last_error names the key).
 *
 *   key                      default   meaning
 *   "lanes"                  3 / 2     (8-channel / single-channel descriptors) estimation lanes (HIP streams driven by host threads) a pair batch fans out over, 1 .. 8 (at least 8
 *                                      pairs per lane).  The narrow per-pair kernels of one lane overlap the chip-filling kernels of another;
 *                                      per-launch timings are only clean with 1.  A context is created with up to two; more are allocated by this option
 *                                      or by the first batch that fans out over them (host-buffer batches stay on their two-lane upload plan).
 *   "persistent"             1         single pairs (estimatePose, addFrame) run every pyramid level in ONE persistent launch; 0: the
 *                                      four-kernel chain.  Also the master switch of "team".
 *   "persist_max_ws"         1         groups of up to this many pairs take the persistent kernel (1 .. 8; more than 1 measured slower)
 *   "persist_grid"           64        workgroups of the persistent kernel (one per CU)
 *   "persist_max_points"     32768 / 65536  (8 channels / 1) a pyramid level with more template points than this, and the finer levels behind it, take the
 *                                      four-kernel chain even for a single pair: dense templates (no non-maximum suppression: conf/tsukuba.cfg) are
 *                                      bandwidth work for the whole chip, not latency work for 64 workgroups (same bits either way)
 *   "dense_candidates_from"  32768     chain launches over a pyramid level with at least this many template points keep the candidates of
 *                                      the exact median in one contiguous run per workspace instead of one segment per 256-point chunk: the
 *                                      selection reads four totals and one array (a 300 k-point level: 10 us instead of 120); same median
 *   "persist_timeout_ticks"  5e7       100 MHz ticks a device-side barrier waits before the launch gives up and the call falls back to
 *                                      the chain (0.5 s; the tests of that path set 1)
 *   "team"                   1         batches of 2 .. team_max_pairs pairs run their whole Gauss-Newton stage in one team-persistent launch
 *   "team_max_pairs"         128       ... up to this many pairs; above "team_full_pairs" (80) only when CUs / pairs workgroups per pair fill at least
 *                                      95 % of the CUs (128 pairs on 256 CUs do, 96 do not: the chain is faster there)
 *   "team_full_pairs"        80        batches of up to this many pairs take the team kernel whatever the fill of the chip (see "team_max_pairs", "team_spares")
 *   "team_size"              0         workgroups per team; 0 = CUs / pairs, at most an eighth of the CUs (32 of 256: larger teams were slower at 2 - 7 pairs)
 *   "team_cus"               (device)  CUs the team kernel may claim (tests: fewer teams than pairs)
 *   "team_local_barriers"    1         teams whose workgroups all run on one XCD (checked on the device) skip the L2 write-back of their barriers
 *   "team_join"              2         workgroups of a team that has run out of pairs join the teams still at work (a batch ends with its slowest
 *                                      pair: + 7 % at 128 pairs): 0 never, 1 teams on the workgroup's own XCD only, 2 any team; same bits either way
 *   "team_join_from_pairs"   48        smaller batches (few, large teams: little to balance) run the team kernel with teams of fixed size
 *   "team_spares"            1         the growing form's grid fills the chip: workgroups beyond pairs x (CUs / pairs) start without a team and join one
 *                                      (96 pairs: 2 x 96 + 64 spares, + 4 %; 112: + 6 %) — with it every batch of up to team_max_pairs pairs takes the team kernel
 *   "team_joins_seen"        (counter) workgroups that joined another team so far (set: resets it)
 *   "normalization_side_stream" 1      a frame stage that runs alone on the context's stream (single frames, batches on one lane) queues the
 *                                      Hartley normalisation sums on a stream of its own, next to template_build, and joins them before it returns
 *   "vo_disparity_late"      1         bpvo_hip_add_frame uploads the disparity of a host frame — which only a later template stage reads — once the
 *                                      estimate is queued: the copy from pageable memory holds the host for 90 us (640x480), which then lie under the
 *                                      Gauss-Newton kernels — where the estimate is the persistent kernel's launch per level, queued in one go (0, and on
 *                                      the chain, whose rounds the host paces: in the data stage)
 *   "normalization_form"     4         the sequential (reference-order) Hartley sums: 1 = hand-scheduled DPP add chains (170 us for a 1241x376 template,
 *                                      2.28 ms for a dense 640x480 one); 0 = the compiler's DPP form (281 us / 3.6 ms); 2 = every lane of a row reads the same
 *                                      four consecutive elements from LDS and adds them with plain adds — no cross-lane traffic, no asm (311 us / 4.0 ms);
 *                                      3 = those reads with the adds as asm blocks of back-to-back plain adds on a wave that does nothing else (170 us /
 *                                      2.04 ms, 118 registers); 4 = 3 for launches of at most 1024 workgroups, 1 for larger ones.  Same sums bit for bit:
 *                                      tests/test_gpu_parity.py runs the four against each other
 *   "normalization_deferred" 1         ... and inside bpvo_hip_batch_run on one lane (not the team kernel), and where addFrame re-estimates against a new key
 *                                      frame, only the coarsest level's sums are joined: the others run on under the Gauss-Newton iterations of the levels
 *                                      above them (the finest level's on a stream of its own), the estimation waits for them level by level
 *   "team_split_max_pairs"   4         ... and team batches of up to this many pairs run the coarsest level of every pair in a launch of its own, the
 *                                      deferred sums under it, the other levels in a second launch behind them (2 / 4 pairs + 1.2 / + 1.5 %; from 8 pairs on the
 *                                      launch boundary — every pair waits for the slowest — costs 4 - 7 %: DESIGN.md 7; 0: one launch, sums joined first)
 *   "levels_in_one_launch_max_frames" 8  frame stages of at most this many frames run ALL levels of the pyramid (three pyrDown steps per launch), of
 *                                      the bit-planes, of the tiled selection and of the template build in one launch each (a single pair: 33 -> 14 launches)
 *   "small_batch_fused"      1         contexts of a few pairs: job table + initial poses in one launch, states copied out by the record-packing launch
 *   "fuse_frozen"            1         residuals recomputed inside the reduction once a workspace's robust scale is frozen (C = 8)
 *   "step_in_reduce_max_pairs" 128     groups (the pairs of one lane) of up to this many pairs: the Gauss-Newton step is taken by the last tile of a pair
 *                                      inside the reduction launch — three kernels per iteration instead of four, same bits (0: never)
 *   "stagger"                1         lanes run their pairs end to end (frame stage of one lane under the estimation of another) ...
 *   "stagger_min_pairs"      192       ... for batches of at least this many pairs (smaller ones: the frame stage of all pairs first, then the lanes)
 *   "tapcache_max_density"   0.5       pyramid levels with more template points per pixel than this gather straight from the descriptor
 *   "upload_workers"         6         host threads that stage a HOST-buffer batch in pinned chunks (0: plain copies)
 *   "upload_plan_first"      0.19      fractions of a host batch in the first (lane 0) and second (lane 1) group of the upload plan;
 *   "upload_plan_second"     0.50      first = 0: two equal groups
 *   "lazy_template_descriptor" 1       bpvo_hip_batch_run keeps, for the TEMPLATE frame (A) of every pair, census bytes + channel 0 instead of the
 *                                      32-byte descriptor records at the pyramid levels with non-maximum suppression (bit-planes, CD3 gradients):
 *                                      one pixel in ~18 is a template point there, template_build forms the records of its stencils from the census
 *                                      bytes (same operations, same bits).  The accessors and a later estimate with such a slot as the CURRENT frame
 *                                      rebuild the records on demand.
 *   "reference_reduction"    0         VALIDATION MODE (the one option that changes numbers).  1: H, G and the squared residual norm of every
 *                                      linearisation are accumulated exactly as the reference's default (serial, WITH_TBB OFF) build does —
 *                                      f32, in index order over the channel-major arrays, w' = w * float(valid), (w' J_a) J_b, (w' r) J, (w' r) r,
 *                                      one multiply and one add per slot (bpvo/linear_system_builder.cc:140-205,239-266) — by one wavefront per
 *                                      pair (kernels_gn_ref.hip): an estimate takes 30 - 50 x as long (scripts/reference_mode_cost.py).  Every iterate, the final pose,
 *                                      numIterations and status of every level then equal the reference path's BIT FOR BIT
 *                                      (tests/test_gpu_reference_order.py, the conf/ sequences, the goldens, a 330-case randomised run).  The
 *                                      default (0) regroups the same sums (rank-2 form, FMA, wave tree, f64 block combine): H, G, f within
 *                                      4e-6 of an f64 evaluation, poses within 1e-4 rad / 1e-3 m, iteration counts inside the reference's own
 *                                      spread between its serial and its TBB build.  In this mode every estimate takes the four-kernel chain
 *                                      (no persistent / team kernel, no fused path, no step inside the reduction).
 *   "keep_current_disparity" 0         bpvo_hip_batch_run stores the disparity of the CURRENT frame (B) of every pair too.  By default it is
 *                                      neither uploaded nor stored — nothing on the path reads it — and bpvo_hip_frame(s)_set_template on
 *                                      such a slot returns BPVO_ERR_NO_DATA; set 1 before batches whose B frames become templates later.
 *
 * The environment variable BPVO_HIP_OPTIONS="key=value,key=value" applies the same settings to every context a process creates
 * (measurement scripts and tests; the library reads no other variable). */
int bpvo_hip_set_option(bpvo_hip_ctx* ctx, const char* key, double value);
int bpvo_hip_get_option(bpvo_hip_ctx* ctx, const char* key, double* value);
/* = bpvo_hip_set_option(ctx, "lanes", n); n <= 0 restores the default */
int bpvo_hip_set_max_lanes(bpvo_hip_ctx* ctx, int n);
/* diagnostics: Gauss-Newton state of a workspace after its last call: T (16), H (36), G (6), dp (6), f_norm, scale, delta_scale,
 * g_norm, pose of the last linearisation (16) */
int bpvo_hip_debug_gn_state(bpvo_hip_ctx* ctx, int ws, float out[84]);

#ifdef __cplusplus
}
#endif
#endif /* BPVO_HIP_C_API_H */
