// Launchers of the HIP kernels (kernels_frame.hip, kernels_gn.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "types.h"

#include <type_traits>

// The channel counts the kernels are instantiated for: Intensity / Laplacian 1, IntensityAndGradient 3, DescriptorFields 5,
// BitPlanes 8, DescriptorFields2ndOrder 10, CentralDifference 8 / 24 / 48, Latch 8 / 16 / 32.  f receives std::integral_constant<int, C>.
template <class F>
static inline void dispatch_channels(int C, F&& f)
{
  switch(C) {
    case 1: f(std::integral_constant<int, 1>()); break;
    case 3: f(std::integral_constant<int, 3>()); break;
    case 5: f(std::integral_constant<int, 5>()); break;
    case 10: f(std::integral_constant<int, 10>()); break;
    case 16: f(std::integral_constant<int, 16>()); break;
    case 32: f(std::integral_constant<int, 32>()); break;
    case 24: f(std::integral_constant<int, 24>()); break;
    case 48: f(std::integral_constant<int, 48>()); break;
    default: f(std::integral_constant<int, 8>()); break;
  }
}

// ... and the WIDE descriptors (more than 48 channels): LATCH with 8 / 16 / 32 / 64 bytes, CentralDifference radii 4 .. 9.  Only kernels whose
// per-thread state does not grow with C are instantiated for these (the saliency forms: C is a stride and a loop bound there); the per-point
// kernels run the channels in groups (types.h PairJob::pitch) or loop over them at run time.
template <class F>
static inline bool dispatch_wide_channels(int C, F&& f)
{
  switch(C) {
    case 64: f(std::integral_constant<int, 64>()); return true;
    case 80: f(std::integral_constant<int, 80>()); return true;
    case 120: f(std::integral_constant<int, 120>()); return true;
    case 128: f(std::integral_constant<int, 128>()); return true;
    case 168: f(std::integral_constant<int, 168>()); return true;
    case 224: f(std::integral_constant<int, 224>()); return true;
    case 256: f(std::integral_constant<int, 256>()); return true;
    case 288: f(std::integral_constant<int, 288>()); return true;
    case 360: f(std::integral_constant<int, 360>()); return true;
    case 512: f(std::integral_constant<int, 512>()); return true;
    default: return false;
  }
}

namespace bpvo_hip {

// A Gaussian smoothing kernel of cv::GaussianBlur as the descriptors use it: n taps (odd; 0 = no smoothing), k the f32 taps
// of cv::getGaussianKernel, ki their 8-bit fixed-point form cvRound(k * 256) for u8 images.  n = 5 runs the small-kernel
// forms of OpenCV's filter engine, n >= 7 the generic ones (kernels_frame.hip, df_plane_kernel).
constexpr int kMaxGaussTaps = 31;
struct GaussTaps { int n = 0; float k[kMaxGaussTaps] = {}; int ki[kMaxGaussTaps] = {}; };

// per-frame stage (batched over frames)
void launch_ingest(hipStream_t s, const FrameJob* jobs_level0, const uint8_t* d_images, const float* d_disps, size_t npix, int nframes, int skip_odd_disp = 0);
void launch_pyrdown(hipStream_t s, const FrameJob* src, const FrameJob* dst, int dW, int dR, int nframes);
// few frames: `steps` (1..3) pyrDown steps in one launch — src_row is the source level's row of the table [level][job_pitch], dW x dR the COARSEST level of the group
void launch_pyramid_levels(hipStream_t s, const FrameJob* src_row, int job_pitch, int steps, int dW, int dR, int nframes);
void launch_intensity(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int nlevels = 1, int job_pitch = 0);
void launch_gradient_descriptor(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, const GaussTaps& pre);   // (I, Ix, Iy), C = 3
void launch_descriptor_fields(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int second_order, const GaussTaps& g1,
                              const GaussTaps& g2);   // C = 5 / 10
void launch_central_difference(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int radius, const GaussTaps& before,
                               const GaussTaps& after);   // C = 8 / 24 / 48
void launch_latch(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int bytes, int half_ssd, const signed char* d_offsets, int kc, int ks,
                  const GaussTaps& after);   // C = 8 * bytes, bytes = 1 / 2 / 4
void launch_laplacian(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int ksize /*1 or 3*/);
void launch_census(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, const int* blur_taps /*null: no smoothing*/, int nlevels = 1, int job_pitch = 0);
// from_image: the census transform is computed inside the bit-planes kernel (no launch_census, sigma_bp > 0 and sigma_ct <= 0)
// nlevels > 1 (bit-planes, tiled selection, template build): `jobs` is the FINEST level's row of the table [level][job_pitch], W / R / max_points that level's — the
// levels in one launch (kernels_frame.hip level_job)
void launch_bitplanes(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, float sigma, const float k[3], int from_image, int nlevels = 1,
                      int job_pitch = 0);
void launch_saliency_select(hipStream_t s, const FrameJob* jobs, int C, int W, int R, int nframes, int nms_radius, float min_saliency,
                            float min_disp, float max_disp, int border, int nlevels = 1, int job_pitch = 0);
void launch_copy_rows(hipStream_t s, void* dst, const void* src_host_pinned, size_t pitch_bytes, size_t width_bytes, int rows);   // multiples of 8 bytes
void launch_gather_counts(hipStream_t s, const FrameJob* jobs /*[L][job_pitch]*/, int job_pitch, int nframes, int first_level, int num_levels,
                          int* out /*[nframes][kMaxLevels]*/);
void launch_normalization(hipStream_t s, const FrameJob* jobs /*[L][job_pitch]*/, int job_pitch, int nframes, int first_level,
                          int num_levels, int with_normalization, int form = 1);   // form of the sequential sums (same sums): 1 hand-scheduled DPP add chains, 0 the compiler's DPP form, 2 broadcast LDS reads + plain adds
void launch_export_jacobians(hipStream_t s, const FrameJob* job /*device, one job*/, int C, int n, float* out /*[C*n][6]*/);
void launch_template_build(hipStream_t s, const FrameJob* jobs, int C, int max_points, int nframes, int grad_cd5, const float gauss_k[3], int nlevels = 1,
                           int job_pitch = 0);   // gauss_k: the bit-planes blur taps (lazy levels)

// stereo front-end (kernels_stereo.hip): OpenCV 2.4 block matching with the reference's parameters, batched over frames
struct StereoLaunch {
  const uint8_t* left_pre;    // [nframes][rows*cols] pre-filtered images
  const uint8_t* right_pre;
  float* disp;                // [nframes][rows*cols]
  int rows, cols, nframes;
  int wsz, ndisp, mindisp, cap, texture_threshold, uniqueness_ratio;
};
void launch_stereo_prefilter(hipStream_t s, const uint8_t* src, uint8_t* dst, int rows, int cols, int cap, int nframes);
bool launch_stereo_bm(hipStream_t s, const StereoLaunch& g);   // false: window size / disparity range outside what the kernel serves

// semi-global matching (kernels_sgm.hip): the reference's in-tree SgmStereo (utils/sgm.cc)
struct SgmLaunch {
  const uint8_t* left;      // [nframes][rows*cols]
  const uint8_t* right;
  float* disp;              // [nframes][rows*cols]
  void* scratch;            // sgm_scratch_bytes(rows, cols, ndisp) bytes, shared by the frames (processed one after the other)
  int rows, cols, nframes;
  int ndisp, sobel_cap, census_radius, window_radius, P1, P2, consistency_threshold;
  double disparity_factor, census_weight;
};
size_t sgm_scratch_bytes(int rows, int cols, int ndisp);
bool launch_stereo_sgm(hipStream_t s, const SgmLaunch& g);   // false: disparity range outside what the kernels serve (multiple of 16, <= 256)
// filterSpeckles on a u16 map whose invalid value is 0 (kernels_sgm.hip's union-find kernels): 4-connected regions (neighbours both non-zero,
// |difference| <= max_diff) of at most max_size pixels are zeroed; lab / size: rows * cols ints each
void launch_speckle_filter_u16(hipStream_t s, uint16_t* img, int* lab, int* size, int rows, int cols, int max_diff, int max_size);

// semi-global block matching (kernels_sgbm.hip): cv::StereoSGBM of OpenCV 2.4, single-pass mode, + medianBlur(3) + filterSpeckles + / 16
struct SgbmLaunch {
  const uint8_t* left; const uint8_t* right;   // [nframes][rows*cols] u8 (device)
  float* disp;                                 // [nframes][rows*cols] f32 (device)
  void* scratch;                               // sgbm_scratch_bytes(rows, cols, min_disp, ndisp) bytes, shared by the frames
  int rows, cols, nframes;
  // the cv::StereoSGBM fields (their "<= 0 means" defaults are applied by the launcher like computeDisparitySGBM does)
  int min_disp, ndisp, sad_window, P1, P2, disp12_max_diff, pre_filter_cap, uniqueness_ratio, speckle_window, speckle_range, full_dp;
};
size_t sgbm_scratch_bytes(int rows, int cols, int min_disp, int ndisp);
bool sgbm_serves(const SgbmLaunch& g, const char** why);      // false + the reason: what the device path does not take
bool launch_stereo_sgbm(hipStream_t s, const SgbmLaunch& g);

// Gauss-Newton stage (batched over workspaces / pairs)
// Compacted list of the workspaces of a launch that are still iterating (estimate loops).  The host rebuilds it (on the
// device) once per round of kItersPerSync iterations, together with the read-back of the count, and sizes the workspace
// dimension of the next round's grids with that count; entry k of the list names the k-th active workspace.  Without it
// every launch dispatches the workgroups of the finished workspaces as well: ~0.9 ns each, ~150 µs per launch in the tail
// of a level at 1024 pairs.  list == nullptr: every workspace of the launch, in order.
constexpr int kDenseRuns = 64;         // runs of candidates per workspace in the dense form of the exact median's bracket step (gn_common.h, bracket_chunk)
constexpr int kChunkPoints = 256;      // points per chunk of the warp + residual kernels = per bracket segment of the exact median (gn_common.h K6_BLOCK)
struct ActiveSet {
  const int* list = nullptr;   // device [npairs]
};

struct GNLaunch {
  const PairJob* jobs;   // device, [npairs] for the level
  int npairs;            // grid size in workspaces (= number of list entries when an ActiveSet is given)
  ActiveSet active;
  int max_points;        // max n over the pairs (grid sizing)
  int C;
  int loss;
  int interp = 0;        // BPVO_INTERP_* (kLinear uses the tap-cached kernel, the others warp_residual_interp_kernel)
  int fuse_frozen = 0;   // estimate loops, C = 8, kLinear, f64 formulation: once a workspace's scale is frozen, irls_reduce
                         // recomputes the residuals itself and warp_residual skips the workspace
  // 1: the bracket step of warp_residual and median_finish keep a workspace's candidates in one contiguous run with four totals
  // (bracket_chunk<C, true>, gn_common.h) instead of per-chunk segments and counters; both launches of an iteration must agree
  int dense_candidates = 0;
  int fast_warp = 0;     // 1: projectPoints / BilinearInterp all-f32 formulation (bpvo_hip_set_warp_formulation)
  // 1: the tile of a workspace that stores its partial LAST in an irls_reduce launch also takes the Gauss-Newton step (what
  // gn_step_kernel does, with step_prm) — the chain is then three kernels per iteration and launch_gn_step is not called
  int step_in_reduce = 0;
  GNParams step_prm = {0, 0, 0.0f, 0.0f, 0.0f};
  // gn_persistent_kernel only: begin_level >= 0 — the kernel takes the level's start itself (level_begin_kernel's state reset; the tap-cache
  // keys were invalidated by the kernel of the level before, which was handed this level's jobs as next_jobs)
  int begin_level = -1, begin_moot = 0;
  const PairJob* next_jobs = nullptr;
  // 1: validation mode "reference_reduction" (kernels_gn_ref.hip) — launch_irls_reduce sums H, G and the squared norm in the reference's
  // index order (one partial per workspace), launch_gn_step reads that one partial; the fused path and step_in_reduce are ignored
  int reference_reduction = 0;
};
int  gn_num_blocks(int max_points);
void launch_set_pose(hipStream_t s, const PairJob* jobs, const float* T_init, int n, unsigned* clear = nullptr, int clear_words = 0);   // clear: words zeroed by the same launch
// PoseEstimatorBase::reset of every workspace + invalidation of its tap-cache keys (max_points: the largest template of the launch)
void launch_level_begin(hipStream_t s, const PairJob* jobs, int npairs, int max_points, int level, int scale_is_moot = 0);
void launch_reset_tapkeys(hipStream_t s, const GNLaunch& g);
// out_list / out_count <- the still-active workspaces among the n_in entries of `in` (or of 0..n_in-1), in order
void launch_compact_active(hipStream_t s, const PairJob* jobs, ActiveSet in, int n_in, int* out_list, int* out_count);
void launch_warp_residual(hipStream_t s, const GNLaunch& g);
void launch_refresh_residuals(hipStream_t s, const GNLaunch& g);   // fused path: rebuild r / valid of stale workspaces from T_lin
void launch_median(hipStream_t s, const GNLaunch& g);
void launch_irls_reduce(hipStream_t s, const GNLaunch& g);
void launch_reference_reduce(hipStream_t s, const GNLaunch& g);   // what launch_irls_reduce does with g.reference_reduction (kernels_gn_ref.hip)
// mode 0: full PoseEstimatorBase::run step (solve, update, convergence); mode 1: linearize only (H, G, f_norm)
void launch_gn_step(hipStream_t s, const GNLaunch& g, int mode, int max_iterations, int max_fun_evals, float p_tol,
                    float f_tol, float g_tol);
// Persistent kernel for small groups: one launch runs a whole level of up to kPersistMaxWs workspaces (kernels_gn.hip).  ctl: two
// zeroed words {arrivals, abort}; after the launch ctl[1] != 0 says the kernel gave up (states untouched: rerun the chain).
constexpr int kPersistMaxWs = 8;
bool gn_persistent_serves(const GNLaunch& g);
int  gn_persistent_grid(const GNLaunch& g, int max_grid);
hipError_t launch_gn_persistent(hipStream_t s, const GNLaunch& g, int max_iterations, int max_fun_evals, float p_tol, float f_tol, float g_tol,
                                unsigned* ctl, int grid, long long timeout_ticks);
// Team-persistent kernel for small batches (kernels_gn.hip, gn_team_kernel): ONE launch runs every pair of the group through all its
// pyramid levels; grid = team_size x n_teams workgroups, one per CU, all co-resident.  ctl: gn_team_ctl_words(n_teams) zeroed words;
// after the launch ctl[1] != 0 says a team barrier gave up (rerun the group on the chain).
struct GNTeamLaunch {
  const PairJob* jobs_all;   // device [levels][job_pitch]
  int job_pitch, n_pairs, level_hi, level_lo;
  int C, loss, fuse_frozen, scale_is_moot;
  int team_size, n_teams;
  unsigned* ctl;
  long long timeout_ticks;
  int local_barriers = 1;    // teams whose workgroups share one XCD (checked on the device) keep their barriers inside that XCD's L2
  int spare_workgroups = 0;  // workgroups beyond n_teams * team_size that start without a team and join one (join_mode != 0 only): the grid fills the chip
  int join_mode = 2;         // workgroups whose team has run out of pairs join the teams still at work: 0 never, 1 teams on their own XCD, 2 any team
};
int  gn_team_ctl_words(int n_teams);
void launch_team_ctl_reset_keep_abort(hipStream_t s, unsigned* ctl, int n_teams);   // between the launches of a split run: every control word to 0 but the abort word
int  gn_team_max_size();       // teams stop admitting newcomers at this size
hipError_t launch_gn_team(hipStream_t s, const GNTeamLaunch& t, int max_iterations, int max_fun_evals, float p_tol, float f_tol, float g_tol);
void launch_prepare_linearize(hipStream_t s, const PairJob* job, const float* T /*device [16]*/, int reset_scale, int level, float given_scale = 0.0f);
int  gn_pts_per_block(int C);
int  gn_partials_entries(int cap, int C);   // kPartialStride-float entries of a workspace's (double-buffered) tile partials
// (C > 48: run-time channel loops over the point-major records)
void launch_weights(hipStream_t s, const PairJob* job, int n, int C, int loss, float* w_out /*[n][C]*/);
void launch_count_good(hipStream_t s, const PairJob* job, int n, int C, int loss, float thr, unsigned int* count);
// the key frame's point cloud (bpvo/vo.cc:250-281) as 32-byte records on the device; K: the level's intrinsics, img: the key frame's level-0 image
void launch_point_cloud(hipStream_t s, const PairJob* job, int n, int C, int loss, const uint8_t* img, int rows, int cols, const float K[9], int dspace,
                        bpvo_hip_point_with_info* out);
void launch_pack_records(hipStream_t s, const PairJob* jobs, int n, int L, float* records, const GNState* d_states = nullptr, GNState* h_states = nullptr,
                         const unsigned* d_ctl = nullptr, unsigned* h_ctl = nullptr, int ctl_words = 0, unsigned* zero = nullptr);   // h_states / h_ctl (pinned host): copied out by the same launch; zero: a word cleared by it
// a few pairs: job table upload (from the pinned host rows) + initial poses + cleared control words in one launch
void launch_set_pose_upload(hipStream_t s, PairJob* d_table, const PairJob* h_table, size_t table_jobs, const PairJob* h_jobs_coarsest, const float* T_init,
                            int n, unsigned* clear, int clear_words);

}  // namespace bpvo_hip
