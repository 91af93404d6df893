/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/src/orc.h for the full notice; PARITY UNPINNED but for median / census operators / ConfigFile).
 *
 * C ABI of the CPU restatement, deliberately the same shape as include/bpvo_hip/c_api.h so that one Python
 * wrapper can drive both and compare them call by call.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 */
#ifndef BPVO_ORACLE_H
#define BPVO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bpvo_orc_ctx bpvo_orc_ctx;
struct bpvo_orc_params;   /* same layout as bpvo_hip_params / orc::Params (35 x 4 bytes) */
typedef struct bpvo_orc_stats { int numIterations; float finalError; float firstOrderOptimality; int status; } bpvo_orc_stats;
typedef struct bpvo_orc_result {
  float pose[16]; float covariance[36]; bpvo_orc_stats optimizerStatistics[8];
  int numLevels; int isKeyFrame; int keyFramingReason; int hasPointCloud;
} bpvo_orc_result;
typedef struct bpvo_orc_point_with_info { float xyzw[4]; uint8_t rgba[4]; float weight; char pad[8]; } bpvo_orc_point_with_info;

/* one record per linearisation of estimate_pose_trace: 68 floats
 * [0..15] T (row-major)  [16..51] H  [52..57] G  [58] f_norm  [59] sigma  [60] num_valid  [61..66] dp  [67] level */
#define BPVO_ORC_TRACE_FLOATS 68

void bpvo_orc_default_params(void* p);
int bpvo_orc_create(bpvo_orc_ctx** out, const float K[9], float baseline, int rows, int cols, const void* params,
                    int device_ignored, int n_frames, int n_pairs);
void bpvo_orc_destroy(bpvo_orc_ctx* ctx);
const char* bpvo_orc_last_error(const bpvo_orc_ctx* ctx);
int bpvo_orc_set_num_threads(bpvo_orc_ctx* ctx, int n);
int bpvo_orc_set_reduction(bpvo_orc_ctx* ctx, int mode);   /* 0: reference (f32 accumulation); 1: f64 accumulation — test instrument only */
int bpvo_orc_set_perturbation(bpvo_orc_ctx* ctx, int seed, double rel);   /* test instrument: (H, G) .* (1 + rel N(0,1)) per linearisation; 0 = off */
int bpvo_orc_set_warp_formulation(bpvo_orc_ctx* ctx, int mode);   /* 0: PhotoError f64 (active), 1: projectPoints f32 (inactive branch) */
int bpvo_orc_num_levels(const bpvo_orc_ctx* ctx);
int bpvo_orc_num_channels(const bpvo_orc_ctx* ctx);
int bpvo_orc_level_size(const bpvo_orc_ctx* ctx, int level, int* rows, int* cols);

int bpvo_orc_frame_set_data(bpvo_orc_ctx* ctx, int slot, const uint8_t* image, const float* disparity);
int bpvo_orc_frame_set_template(bpvo_orc_ctx* ctx, int slot);
int bpvo_orc_frame_clear(bpvo_orc_ctx* ctx, int slot);
int bpvo_orc_frame_state(const bpvo_orc_ctx* ctx, int slot, int* has_data, int* has_template);
int bpvo_orc_frames_set_data(bpvo_orc_ctx* ctx, int first_slot, int slot_stride, int count, const uint8_t* images,
                             const float* disparities, int on_device_ignored);
int bpvo_orc_frames_set_template(bpvo_orc_ctx* ctx, int first_slot, int slot_stride, int count);

int bpvo_orc_get_image(bpvo_orc_ctx* ctx, int slot, int level, uint8_t* out);
int bpvo_orc_get_descriptor_channel(bpvo_orc_ctx* ctx, int slot, int level, int channel, float* out);
int bpvo_orc_get_saliency(bpvo_orc_ctx* ctx, int slot, int level, float* out);
int bpvo_orc_num_points(bpvo_orc_ctx* ctx, int slot, int level, int* n);
int bpvo_orc_get_points(bpvo_orc_ctx* ctx, int slot, int level, float* xyzw);
int bpvo_orc_get_point_indices(bpvo_orc_ctx* ctx, int slot, int level, int* inds);
int bpvo_orc_get_pixels(bpvo_orc_ctx* ctx, int slot, int level, float* pixels);
int bpvo_orc_get_jacobians(bpvo_orc_ctx* ctx, int slot, int level, float* J);
int bpvo_orc_get_normalization(bpvo_orc_ctx* ctx, int slot, int level, float T[16], float T_inv[16]);

int bpvo_orc_linearize(bpvo_orc_ctx* ctx, int ws, int ref_slot, int cur_slot, int level, const float T[16],
                       int reset_scale, float H[36], float G[6], float* f_norm, float* sigma, int* num_valid);
int bpvo_orc_get_residuals(bpvo_orc_ctx* ctx, int ws, float* r, size_t* n);
int bpvo_orc_get_valid(bpvo_orc_ctx* ctx, int ws, uint16_t* v, size_t* n);
int bpvo_orc_get_weights(bpvo_orc_ctx* ctx, int ws, float* w, size_t* n);
int bpvo_orc_fraction_good(bpvo_orc_ctx* ctx, int ws, float threshold, float* frac);

int bpvo_orc_estimate_pose(bpvo_orc_ctx* ctx, int ws, int ref_slot, int cur_slot, const float T_init[16],
                           float T_est[16], bpvo_orc_stats* stats);
int bpvo_orc_estimate_pose_trace(bpvo_orc_ctx* ctx, int ws, int ref_slot, int cur_slot, const float T_init[16],
                                 float T_est[16], bpvo_orc_stats* stats, float* records, int max_records,
                                 int* n_records);

int bpvo_orc_add_frame(bpvo_orc_ctx* ctx, const uint8_t* image, const float* disparity, bpvo_orc_result* result);
int bpvo_orc_vo_num_points_at_level(bpvo_orc_ctx* ctx, int level, int* n);
int bpvo_orc_vo_points_at_level(bpvo_orc_ctx* ctx, int level, float* xyzw);
int bpvo_orc_get_point_cloud(bpvo_orc_ctx* ctx, bpvo_orc_point_with_info* pts, size_t* n, float pose[16]);
int bpvo_orc_trajectory_size(bpvo_orc_ctx* ctx, int* n);
int bpvo_orc_get_trajectory(bpvo_orc_ctx* ctx, float* poses);

int bpvo_orc_batch_run(bpvo_orc_ctx* ctx, int n_pairs, const uint8_t* images, const float* disparities,
                       int on_device_ignored, float* poses, bpvo_orc_stats* stats);
int bpvo_orc_batch_estimate(bpvo_orc_ctx* ctx, int n_pairs, const float* T_init, float* poses, bpvo_orc_stats* stats);
int bpvo_orc_total_linearizations(bpvo_orc_ctx* ctx, uint64_t* n);

/* stand-alone operators (parity of single stages on arbitrary inputs) */
int bpvo_orc_pyrdown_u8(const uint8_t* src, int rows, int cols, uint8_t* dst /* ((rows+1)/2)*((cols+1)/2) */);
int bpvo_orc_census(const uint8_t* src, int rows, int cols, float sigma_ct, uint8_t* dst);
int bpvo_orc_gaussian5x5_f32(const float* src, int rows, int cols, float sigma, float* dst);
int bpvo_orc_gaussian_f32(const float* src, int rows, int cols, int ksize, float sigma, float* dst);     /* ksize 5 or 7..31 */
int bpvo_orc_gaussian_u8(const uint8_t* src, int rows, int cols, int ksize, float sigma, uint8_t* dst);   /* 8-bit fixed point */
int bpvo_orc_imsmooth_taps(float sigma);            /* bpvo/imgproc.cc:168 */
int bpvo_orc_auto_gauss_taps_f32(float sigma);      /* cv::GaussianBlur(Size(), sigma) on CV_32F */
float bpvo_orc_median(const float* data, size_t n);            /* bpvo/utils.h:224-252 on a copy */
int bpvo_orc_solve(const float H[36], const float G[6], float dp[6]);   /* returns 1 if solved */
void bpvo_orc_twist_to_matrix(const float p[6], float T[16]);
int bpvo_orc_stereo_bm(const uint8_t* left, const uint8_t* right, int rows, int cols, const int params[6], float* dmap);
int bpvo_orc_stereo_prefilter(const uint8_t* src, int rows, int cols, int cap, uint8_t* dst);
int bpvo_orc_stereo_sgm(const uint8_t* left, const uint8_t* right, int rows, int cols, const int iparams[7], const double dparams[2], float* dmap);
int bpvo_orc_stereo_sgbm(const uint8_t* left, const uint8_t* right, int rows, int cols, const int params[11], float* dmap);
int bpvo_orc_compute_weights(int loss, const float* r, const uint16_t* valid, size_t n, float sigma, float* w);

#ifdef __cplusplus
}
#endif
#endif
