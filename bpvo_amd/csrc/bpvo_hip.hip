// libbpvo_hip: host driver + C ABI (include/bpvo_hip/c_api.h) of the MI355X-native dense alignment path.
//
// The host keeps bpvo's object model (frames with a descriptor pyramid and a template pyramid, a pose estimator with
// per-level Gauss-Newton runs, the VisualOdometry keyframe state machine) but every O(pixels) / O(points) array lives
// in HBM; per GN iteration the host sees one 4-byte "pairs still active" counter.  See DESIGN.md.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "latch_table.h"

using namespace bpvo_hip;

namespace {

// live contexts per device: the estimation lanes of one context are streams = hardware queues, of which a process has a handful;
// a second context's lanes end up multiplexed on the same queues and LOSE (measured: 505 k instead of 666 k GN it/s for a
// 128-pair batch next to a second context), so batches only fan out over lanes while theirs is the only context on the device
std::atomic<int> g_live_ctx[64];
thread_local std::string g_create_error;   // bpvo_hip_last_error(nullptr): the failed create of THIS thread (contexts are created
                                           // concurrently by the per-GPU host threads of multi_gpu.hip)

struct LevelGeom {
  int rows, cols;
  size_t npix;
  int nblk;        // 256-pixel chunks of the row-major scan
  int cap;         // capacity of the template arrays
  int nms_radius;  // <= 0: off
  float K[9];
  float b;
};

struct FrameSlot {
  bool has_data = false, has_template = false;
  bool has_disp = false;     // the slot holds the disparity of its image (false for the current frames B of a pair batch: never uploaded)
  void* data_slab = nullptr;
  void* tmpl_slab = nullptr;
  uint8_t* img[kMaxLevels] = {};
  uint8_t* cen[kMaxLevels] = {};
  float* ch0[kMaxLevels] = {};
  bool ch0_valid = true;          // false: the descriptor of the slot's current data was computed without the compact channel-0 plane
  float* desc[kMaxLevels] = {};
  float* disp = nullptr;
  float* scratch = nullptr;   // descriptor fields: kDfPlanes work planes
  float* sal[kMaxLevels] = {};
  uint8_t* flag[kMaxLevels] = {};
  int* blk_count[kMaxLevels] = {};
  float4* pts[kMaxLevels] = {};
  int* inds[kMaxLevels] = {};
  float* pix[kMaxLevels] = {};
  float* grad[kMaxLevels] = {};
  float* nrm = nullptr;    // [L][4]
  int* n_dev = nullptr;    // [L]
  int n_host[kMaxLevels] = {};
};

struct Workspace {
  float* r = nullptr;
  uint8_t* valid = nullptr;
  uint32_t* cand = nullptr;
  uint32_t* med_blk = nullptr;
  uint32_t* tapkey = nullptr;
  float* tapcache = nullptr;
  float* partials = nullptr;
  int last_ref = -1, last_cur = -1, last_level = -1;
};

enum KernelClass { KC_PYRAMID = 0, KC_DESCRIPTOR, KC_SALIENCY_SELECT, KC_NORMALIZATION, KC_TEMPLATE, KC_WARP_RESIDUAL, KC_MEDIAN,
                   KC_IRLS_REDUCE, KC_GN_STEP, KC_COUNT };
const char* kKernelNames[KC_COUNT] = {"pyramid", "descriptor", "saliency_select", "normalization", "template_build", "warp_residual",
                                      "median", "irls_reduce", "gn_step"};

struct EventPair { hipEvent_t a, b; int kc; double units; };

// An estimation lane: one HIP stream plus the host staging it needs.  Batches of independent pairs are split over
// several lanes driven by their own host threads, so that the narrow per-pair kernels of one group (median select,
// gn_step: one workgroup per pair) overlap with the chip-filling kernels (warp_residual, irls_reduce) of another.
struct Lane {
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  PairJob* h_pjobs = nullptr;      // pinned [L][n_pairs]
  PairJob* d_pjobs = nullptr;      // [L][n_pairs]
  float* h_T = nullptr;            // pinned [n_pairs][16]
  float* d_Tinit = nullptr;
  int* d_active = nullptr;         // [3][2] per round in flight: entries of the active list, and how many of them still estimate their scale
  int* d_list = nullptr;           // [3][n_pairs] active-workspace lists of the host rounds in flight (ActiveSet)
  int* h_active = nullptr;         // pinned [3][2]
  hipEvent_t round_ev[3] = {};     // "compaction of round r and its count have landed"
  hipEvent_t staging_ev[2] = {};   // "the upload of this lane's rows of FrameJob table 0 / 1 has left the pinned staging"
  hipEvent_t selected_ev = nullptr; // staggered batches: "the selection of this lane's templates has been queued" (FrameRun)
  unsigned* d_pk_ctl = nullptr;    // [kMaxLevels][kPkCtlWords] {arrivals, abort} of the persistent kernel, one slot per level
  unsigned* h_pk_ctl = nullptr;    // pinned copy
  unsigned* d_team_ctl = nullptr;  // gn_team_ctl_words(kMaxTeams) words of the team-persistent kernel (lane 0 only)
  unsigned* h_team_ctl = nullptr;  // pinned copy of its first line (abort word)
  GNState* h_states = nullptr;     // pinned [n_pairs]
  std::vector<EventPair> ev_pending;
  std::vector<hipEvent_t> ev_pool;
  unsigned k6_seq = 0;             // warp_residual launches of this lane since bpvo_hip_profiling (event sampling)
  std::string err;
};
// Estimation lanes (streams driven by host threads) of a batch: the narrow per-pair kernels (median_finish, gn_step: one
// workgroup / wave per pair) of one lane overlap the chip-filling kernels of the other.  Two lanes: +2 % at 1024 pairs of
// 1241x376 bit-planes, +3.7 % at 128, +7 % for 640x480 intensity; four lanes lose at every size.  Per-launch durations of
// overlapping lanes include the time shared with the other lane: measurements that need clean per-kernel times run with
// bpvo_hip_set_max_lanes(ctx, 1) / BPVO_HIP_LANES=1.  Results do not depend on the number of lanes (test_gpu_parity.py).
constexpr int kDefaultLanes = 2;
constexpr int kDefaultLanesNarrow = 2;
constexpr int kMinPairsPerLane = 8;
constexpr int kPkCtlWords = 32;    // one 128-byte line per level
constexpr int kMaxTeams = 1024;

}  // namespace

struct bpvo_hip_ctx {
  bpvo_hip_params params;
  float K[9];
  float baseline;
  int rows, cols, L, C, device;
  int n_frames, n_pairs;
  LevelGeom geom[kMaxLevels];
  float gauss_k[3];
  GaussTaps df_g1, df_g2;       // imsmooth kernels of dfSigma1 / dfSigma2 (descriptor fields); n = 0: sigma <= 0
  GaussTaps cd_before, cd_after; // imsmooth kernels of centralDifferenceSigmaBefore (u8 fixed point) / After (f32)
  GaussTaps grad_pre;           // cv::GaussianBlur(Size(), sigma) of GradientDescriptor (sigmaPriorToCensusTransform > 0)
  GaussTaps latch_after;        // imsmooth(1.75) of every LATCH channel (bpvo/latch_descriptor.cc:1082)
  int latch_taps[2] = {0, 0};   // fixed-point {centre, side} taps of LATCH's cv::GaussianBlur(Size(3,3), 2, 2) (:147)
  signed char* d_latch_off = nullptr;   // [48 * latchNumBytes] triplet coordinates as CalcuateSums uses them (:170-236)
  bool plane_scratch = false; // descriptor built from plane operations (descriptor fields, central difference, smoothed gradient)
  hipStream_t stream = nullptr;
  std::vector<FrameSlot> frames;
  std::vector<Workspace> ws;
  GNState* d_states = nullptr;
  FrameJob* d_fjobs = nullptr;     // [2][L][n_frames]: the table of the setData stage, then the one of the setTemplate stage
  std::vector<Lane> lanes;         // lanes[0] shares the ctx stream
  PairJob* d_job1 = nullptr;       // scratch single job (linearize / weights)
  float* d_records = nullptr;      // [n_pairs][kRecordFloats]
  float* d_wtmp = nullptr;         // [cap_max * C] weights scratch
  unsigned int* d_count = nullptr;
  unsigned long long* d_counters = nullptr;   // [4] points, linearisations, bracketed / full median selections
  // pinned staging
  FrameJob* h_fjobs = nullptr;
  int* h_ints = nullptr;           // [max(n_frames*kMaxLevels, 16)] pinned
  int* d_ints = nullptr;           // same size, device
  int cap_max = 0;
  // VisualOdometry state (bpvo/vo.cc:45-52)
  int vo_ref = 0, vo_cur = 1, vo_prev = 2;
  M44 T_kf;
  std::vector<M44> trajectory;
  std::vector<bpvo_hip_point_with_info> cloud;
  M44 cloud_pose;
  // measurement
  double points_fused = 0;     // points linearised through the fused path since the last counter reset
  int fast_warp = 0;           // bpvo_hip_set_warp_formulation
  int dspace = 0;              // BPVO_WARP_DISPARITY_SPACE_F32: DisparitySpaceWarp as the warp (implies fast_warp)
  int fuse_frozen = 1;         // estimate loops: fused residual + reduction once a workspace's scale is frozen (bit-identical,
                               // +3 % GN iterations/s).  BPVO_HIP_FUSE_FROZEN=0 turns it off.
  int irls_merge_below = 1 << 30;   // fuse_frozen: fewer active workspaces than this -> ONE irls_reduce launch with a per-workspace branch
                               // instead of two instantiations sharing the slot.  Measured faster at every batch size (1 pair +6 %, 8 / 32
                               // pairs +13 %, 128 +7 %, 1024 +3.7 %: the second, half-empty launch costs more than the fourth wave per SIMD
                               // buys the plain form), so it is always on; BPVO_HIP_IRLS_MERGE_BELOW=0 restores the two launches
  // Groups of at most persist_max_ws workspaces (a single pair: sequential addFrame) run every pyramid level in ONE persistent
  // launch (kernels_gn.hip, gn_persistent_kernel) instead of rounds of four kernels per iteration; bit-identical results.
  // BPVO_HIP_PERSISTENT=0 turns it off, BPVO_HIP_PERSIST_MAX_WS / _GRID size it.  persistent_failed: a launch gave up at a barrier
  // (workgroups not co-resident) — the context stays on the four-kernel chain from then on.
  int persistent = 1, persist_max_ws = 1, persist_grid = 64;
  long long persist_timeout = 50000000ll;   // ticks of the 100 MHz wall clock a grid barrier waits before it gives up (0.5 s)
  // Batches of 2 .. team_max_pairs pairs run their whole Gauss-Newton stage in ONE launch of the team-persistent kernel
  // (kernels_gn.hip, gn_team_kernel): teams of team_size workgroups, one workgroup per CU, a pair per team at a time.
  // BPVO_HIP_TEAM=0 turns it off, BPVO_HIP_TEAM_MAX_PAIRS / BPVO_HIP_TEAM_SIZE (0 = CUs / pairs) size it.
  int team_mode = 1, team_max_pairs = 64, team_size_env = 0, num_cus = 0;
  int team_single = 0;         // BPVO_HIP_TEAM_SINGLE=1: single pairs through the team kernel too (all levels in one launch) instead of gn_persistent_kernel per level
  std::atomic<uint64_t> team_launches{0};
  std::atomic<bool> persistent_failed{false};      // (atomics: estimate_group runs on the lane threads)
  std::atomic<uint64_t> persistent_levels{0};      // levels run by the persistent kernel (measurement)
  // bpvo_hip_estimate_pose_trace: while trace_ws >= 0 the jobs of that workspace carry the device trace buffer
  float* d_trace = nullptr;
  int trace_cap = 0, trace_ws = -1;
  int max_lanes_now = 1 << 30; // bpvo_hip_set_max_lanes: measurement runs that need per-launch timings without overlap
  // stereo front-end scratch (lazily sized for the largest frame count seen): raw and pre-filtered u8 pairs, f32 disparities
  uint8_t* st_left = nullptr; uint8_t* st_right = nullptr; uint8_t* st_left_pre = nullptr; uint8_t* st_right_pre = nullptr;
  float* st_disp = nullptr;
  int st_frames = 0;
  void* st_sgm = nullptr;      // scratch of the semi-global matcher (cost volumes: sized for the largest disparity range seen)
  size_t st_sgm_bytes = 0;
  // Upload pipeline of pair batches handed over in HOST buffers (bpvo_hip_batch_run, on_device = 0): worker threads stage chunks of
  // kUploadChunkPairs pairs in pinned memory and copy them on streams of their own into a device staging area, chunk after chunk in lane
  // order, while the lanes already work on the chunks that have landed (upload_pipeline below).  BPVO_HIP_UPLOAD_WORKERS (0 = off).
  int up_workers = 6;
  int up_group = 4;                      // chunks a lane's frame stage takes at once (BPVO_HIP_UPLOAD_GROUP): 16-pair launches are too small to fill the chip
  int up_subbatches = 1;                 // groups per lane of a host batch (BPVO_HIP_UPLOAD_SUBBATCHES; host_groups): 2 and more measured slower
  int ctl_kernel_mode = 1;               // BPVO_HIP_CTL_KERNEL_COPY: control tables by copy_rows_kernel never (0) / while an upload pipeline runs (1) / always (2)
  std::atomic<bool> ctl_by_kernel{false};
  int up_streams_n = 1;                  // copy streams the workers share (BPVO_HIP_UPLOAD_STREAMS).  A process has a handful of hardware queues
                                         // and HIP streams are multiplexed onto them: with a stream per worker the lanes' kernels queued behind
                                         // other workers' copies and nothing ran until the last chunk had landed (profiles/r03_host_timeline.txt)
  std::vector<hipStream_t> up_streams;
  std::vector<uint8_t*> up_pinned;       // [worker]: 2 slots of up_slot_bytes
  std::vector<hipEvent_t> up_slot_free;  // [worker * 2 + slot]
  std::vector<hipEvent_t> up_chunk_done; // pool, one per chunk of a call
  size_t up_slot_bytes = 0;
  uint8_t* up_d_img = nullptr;           // device staging: images [2 n][npix]
  float* up_d_disp = nullptr;            //                 disparities of the A frames [n][npix]
  int up_cap_pairs = 0;
  // host batches on two lanes: the pairs are cut into a SMALL first group (lane 0 starts its Gauss-Newton stage while most of the batch
  // is still crossing the bus), a large second one for lane 1, and the rest for lane 0 again (host_groups_plan); fractions of the batch
  double up_plan[2] = {0.19, 0.50};      // BPVO_HIP_UPLOAD_PLAN="f0:f1"; f0 = 0: two equal groups.  Measured: profiles/r03_host_buffers_plan.txt
  double up_last_seconds = 0.0;          // wall time the workers of the last call needed for all chunks (measurement)
  size_t up_last_bytes = 0;
  bool counted_live = false;   // this context is in g_live_ctx
  // addFrame: the fraction of good points (should_keyframe's last criterion) is queued right behind the estimation, before the host
  // waits for the pose, instead of in a second round trip; frac_* hold it for fraction_good (same kernels, same count)
  float prefetch_frac_thr = -1.0f;    // >= 0 while bpvo_hip_add_frame runs its estimate
  bool frac_valid = false; int frac_ws = -1; float frac_thr = 0.0f; unsigned frac_cnt = 0; int frac_n = 0;
  double tapcache_max_density = 0.5;   // BPVO_HIP_TAPCACHE_MAX_DENSITY: levels with more template points per pixel than this run without the tap cache (batches)
  bool skip_frozen_launches = true;   // BPVO_HIP_SKIP_FROZEN=0: keep launching warp_residual / median when every active scale is frozen (A/B)
  bool stagger = true;         // BPVO_HIP_STAGGER=0: batches run stage by stage over all pairs (batch_run_staggered)
  bool sync_rounds = false;    // BPVO_HIP_SYNC_ROUNDS=1: no pipelining of the host rounds (A/B measurements)
  bool split_census = false;   // BPVO_HIP_SPLIT_CENSUS=1: census as its own kernel even where it can be fused (A/B measurements)
  int census_taps[2] = {0, 0}; // fixed-point {centre, side} taps of the 3x3 u8 blur before the census (sigma_ct > 0)
  bool profiling = false;      // HIP events around warp_residual (the roofline kernel) and the frame stages
  bool profile_all = false;    // ... and around every GN kernel (diagnostics; costs ~10 % throughput)
  bool profile_k6_all = false; // level 3: events around EVERY warp_residual launch (and nothing else in the loop): bench.py's roofline pass
  double kc_ms[KC_COUNT] = {};
  double kc_units[KC_COUNT] = {};
  uint64_t kc_launches[KC_COUNT] = {};
  uint64_t total_lin = 0, median_bracketed = 0, median_full = 0;
  uint64_t tap_counts[4] = {};
  std::mutex units_mu;         // kc_units updates of concurrent frame stages
  std::string err;
};

namespace {

#define HIP_CK(ctx_, expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if(e_ != hipSuccess) {                                                                  \
      (ctx_)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
      return BPVO_ERR_DEVICE;                                                               \
    }                                                                                       \
  } while(0)

int fail(bpvo_hip_ctx* c, int code, const char* msg)
{
  c->err = msg;
  return code;
}

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {
  unsigned char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t count)
  {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += align_up(count * sizeof(T));
    return p;
  }
};

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
  f.scratch = c->plane_scratch ? cv.take<float>((size_t) kDfPlanes * c->geom[0].npix) : nullptr;
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
  if(ws == c->trace_ws) { j.trace = c->d_trace; j.trace_cap = c->trace_cap; }
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
struct ScopedTimer {
  Lane* ln;
  EventPair ep;
  bool on;
  ScopedTimer(bpvo_hip_ctx* c_, int kc, double units, Lane* lane = nullptr, bool sampled = true)
      : ln(lane ? lane : &c_->lanes[0]), on(c_->profiling && sampled)
  {
    if(!on) return;
    ep.kc = kc; ep.units = units;
    ep.a = take_event(ln); ep.b = take_event(ln);
    (void) hipEventRecord(ep.a, ln->stream);
  }
  ~ScopedTimer()
  {
    if(!on) return;
    (void) hipEventRecord(ep.b, ln->stream);
    ln->ev_pending.push_back(ep);
  }
};
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

// ---- frame stages ---------------------------------------------------------------------------------------------------
// slots: first, first+stride, ...; uploads the FrameJob table [L][count] and returns its device base
// A frame stage runs either on the ctx stream (single frames, batches on one lane) or, for staggered batches, on a lane's own stream
// with its own rows [tab, tab + count) of the job tables and count staging (FrameRun); errors of a lane go to the lane's string.
struct FrameRun {
  hipStream_t stream;
  Lane* ln;          // timing events are taken from / queued on this lane
  int tab;           // first row of the FrameJob table [L][n_frames] and of h_ints / d_ints [n_frames][kMaxLevels] used by this run
  bool own_thread;   // run by a lane thread next to others: no resolve_events, errors into ln->err
  hipEvent_t selected_ev;   // recorded once the selection of all levels has been queued (the next lane's frame stage starts behind it), or null
  std::function<void()> on_selected;   // ... and called right after that record (releases the next lane's host thread)
};
#define FR_CK(c_, fr_, expr)                                                                \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if(e_ != hipSuccess) {                                                                  \
      ((fr_).own_thread ? (fr_).ln->err : (c_)->err) = std::string(#expr) + ": " + hipGetErrorString(e_); \
      return BPVO_ERR_DEVICE;                                                               \
    }                                                                                       \
  } while(0)
FrameRun ctx_run(bpvo_hip_ctx* c) { return FrameRun{c->stream, &c->lanes[0], 0, false, nullptr, nullptr}; }

// which: 0 = table of the setData stage, 1 = table of the setTemplate stage (two tables, so that queueing the template stage does not
// have to wait for the descriptor kernels that still read the first).  Returns the device table through *tab (row fr.tab of level 0).
int upload_frame_jobs(bpvo_hip_ctx* c, int first, int stride, int count, const FrameRun& fr, int which, const FrameJob** tab)
{
  const size_t table = (size_t) which * c->L * c->n_frames;
  FR_CK(c, fr, hipEventSynchronize(fr.ln->staging_ev[which]));   // the pinned rows may still feed the copy of an earlier call
  for(int l = 0; l < c->L; ++l) {
    FrameJob* row = c->h_fjobs + table + (size_t) l * c->n_frames + fr.tab;
    for(int i = 0; i < count; ++i) row[i] = make_frame_job(c, c->frames[first + i * stride], l);
  }
  // rows [tab, tab + count) of every level in one copy
  const size_t pitch = sizeof(FrameJob) * (size_t) c->n_frames;
  static_assert(sizeof(FrameJob) % 8 == 0 && sizeof(PairJob) % 8 == 0, "copy_rows_kernel moves 8-byte words");
  if(c->ctl_by_kernel.load())
    launch_copy_rows(fr.stream, c->d_fjobs + table + fr.tab, c->h_fjobs + table + fr.tab, pitch, sizeof(FrameJob) * (size_t) count, c->L);
  else
    FR_CK(c, fr, hipMemcpy2DAsync(c->d_fjobs + table + fr.tab, pitch, c->h_fjobs + table + fr.tab, pitch, sizeof(FrameJob) * (size_t) count, (size_t) c->L,
                                  hipMemcpyHostToDevice, fr.stream));
  FR_CK(c, fr, hipEventRecord(fr.ln->staging_ev[which], fr.stream));
  *tab = c->d_fjobs + table + fr.tab;
  return BPVO_OK;
}

// VisualOdometryFrame::setData (reference: bpvo/vo_frame.cc:48-55) for `count` frames at once
// skip_odd_disp: the frames are the (A, B) frames of pairs, in that order: B (odd i) only ever serves as the CURRENT frame of its pair,
// whose disparity nothing reads (the reference copies what it is handed, bpvo/vo_frame.cc:50-51; estimatePose never looks at it) — it is
// neither uploaded nor copied: 40 % of a pair's input bytes
// (2: as 1, with the device-resident disparities packed for the even frames only — the staging area of the upload pipeline)
int frames_set_data(bpvo_hip_ctx* c, int first, int stride, int count, const uint8_t* images, const float* disps, bool on_device,
                    const FrameRun& fr, int skip_odd_disp = 0)
{
  if(count <= 0) return BPVO_OK;
  const size_t npix = c->geom[0].npix;
  hipStream_t s = fr.stream;
  if(!on_device) {
    for(int i = 0; i < count; ++i) {
      FrameSlot& f = c->frames[first + i * stride];
      FR_CK(c, fr, hipMemcpyAsync(f.img[0], images + (size_t) i * npix, npix, hipMemcpyHostToDevice, s));
      if(!(skip_odd_disp && (i & 1)))
        FR_CK(c, fr, hipMemcpyAsync(f.disp, disps + (size_t) i * npix, npix * sizeof(float), hipMemcpyHostToDevice, s));
    }
  }
  // pair batches: the compact channel-0 plane serves the saliency map of TEMPLATE frames only; the current frames' descriptor kernel
  // skips its store (the selection reads channel 0 from the records should such a frame be made a template later)
  for(int i = 0; i < count; ++i) c->frames[first + i * stride].ch0_valid = !(skip_odd_disp && (i & 1));
  const FrameJob* tab = nullptr;
  int rc = upload_frame_jobs(c, first, stride, count, fr, 0, &tab);
  if(rc) return rc;
  const int NF = c->n_frames;
  if(on_device) launch_ingest(s, tab, images, disps, npix, count, skip_odd_disp);   // one launch instead of 2 copies per frame
  {
    double px = 0;
    for(int l = 1; l < c->L; ++l) px += (double) c->geom[l].npix * count;
    ScopedTimer t(c, KC_PYRAMID, px, fr.ln);
    for(int l = 1; l < c->L; ++l)   // ImagePyramid::compute (bpvo/image_pyramid.cc:43-50)
      launch_pyrdown(s, tab + (size_t) (l - 1) * NF, tab + (size_t) l * NF, c->geom[l].cols, c->geom[l].rows, count);
  }
  {
    double px = 0;
    for(int l = c->L - 1; l >= c->params.maxTestLevel; --l) px += (double) c->geom[l].npix * count;
    ScopedTimer t(c, KC_DESCRIPTOR, px, fr.ln);
    for(int l = c->L - 1; l >= c->params.maxTestLevel; --l) {   // DenseDescriptorPyramid::init (dense_descriptor_pyramid.cc:67-71)
      const FrameJob* jobs = tab + (size_t) l * NF;
      const LevelGeom& g = c->geom[l];
      if(c->params.descriptor == BPVO_DESC_CENTRAL_DIFFERENCE) {
        launch_central_difference(s, jobs, g.cols, g.rows, count, c->params.centralDifferenceRadius, c->cd_before, c->cd_after);
      } else if(c->params.descriptor == BPVO_DESC_LATCH) {
        launch_latch(s, jobs, g.cols, g.rows, count, c->params.latchNumBytes, c->params.latchHalfSsdSize, c->d_latch_off, c->latch_taps[0], c->latch_taps[1],
                     c->latch_after);
      } else if(c->C == 5 || c->C == 10) {
        launch_descriptor_fields(s, jobs, g.cols, g.rows, count, c->C == 10, c->df_g1, c->df_g2);
      } else if(c->C == 3) {
        launch_gradient_descriptor(s, jobs, g.cols, g.rows, count, c->grad_pre);
      } else if(c->C == 1) {
        if(c->params.descriptor == BPVO_DESC_LAPLACIAN) launch_laplacian(s, jobs, g.cols, g.rows, count, c->params.laplacianKernelSize);
        else launch_intensity(s, jobs, g.cols, g.rows, count);
      } else {
        // census fused into the bit-planes kernel unless the census is taken of the smoothed image or the planes stay unsmoothed
        const bool fused_census = !(c->params.sigmaPriorToCensusTransform > 0.0f) && c->params.sigmaBitPlanes > 0.0f && !c->split_census;
        if(!fused_census)
          launch_census(s, jobs, g.cols, g.rows, count, c->params.sigmaPriorToCensusTransform > 0.0f ? c->census_taps : nullptr);
        launch_bitplanes(s, jobs, g.cols, g.rows, count, c->params.sigmaBitPlanes, c->gauss_k, fused_census ? 1 : 0);
      }
    }
  }
  FR_CK(c, fr, hipGetLastError());
  for(int i = 0; i < count; ++i) {
    FrameSlot& f = c->frames[first + i * stride];
    f.has_data = true;
    f.has_disp = !(skip_odd_disp && (i & 1));
  }
  return BPVO_OK;
}
int frames_set_data(bpvo_hip_ctx* c, int first, int stride, int count, const uint8_t* images, const float* disps, bool on_device, int skip_odd_disp = 0)
{
  if(count <= 0) return BPVO_OK;
  if(first < 0 || stride < 1 || first + (count - 1) * stride >= c->n_frames) return fail(c, BPVO_ERR_INVALID_ARG, "bad frame slot range");
  if(!images || !disps) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr image/disparity");
  return frames_set_data(c, first, stride, count, images, disps, on_device, ctx_run(c), skip_odd_disp);
}

// VisualOdometryFrame::setTemplate (reference: bpvo/vo_frame.cc:61-93 -> bpvo/template_data.cc:37-142) for `count` frames
int frames_set_template(bpvo_hip_ctx* c, int first, int stride, int count, const FrameRun& fr)
{
  if(count <= 0) return BPVO_OK;
  hipStream_t s = fr.stream;
  for(int i = 0; i < count; ++i) {
    FrameSlot& f = c->frames[first + i * stride];
    if(f.tmpl_slab) continue;
    size_t total = 0;
    FrameSlot tmp;
    carve_frame_tmpl(c, tmp, nullptr, &total);
    FR_CK(c, fr, hipMalloc(&f.tmpl_slab, total));
    FR_CK(c, fr, hipMemsetAsync(f.tmpl_slab, 0, total, s));
    carve_frame_tmpl(c, f, (unsigned char*) f.tmpl_slab, nullptr);
  }
  const FrameJob* tab = nullptr;
  int rc = upload_frame_jobs(c, first, stride, count, fr, 1, &tab);
  if(rc) return rc;
  const int NF = c->n_frames;
  int* const h_ints = c->h_ints + (size_t) fr.tab * kMaxLevels;
  int* const d_ints = c->d_ints + (size_t) fr.tab * kMaxLevels;
  const bpvo_hip_params& p = c->params;
  const int border = std::max(p.nonMaxSuppRadius, 3);   // template_data.cc:51
  for(int l = c->L - 1; l >= p.maxTestLevel; --l) {
    const FrameJob* jobs = tab + (size_t) l * NF;
    const LevelGeom& g = c->geom[l];
    ScopedTimer t(c, KC_SALIENCY_SELECT, (double) g.npix * count, fr.ln);
    launch_saliency_select(s, jobs, c->C, g.cols, g.rows, count, g.nms_radius, p.minSaliency, p.minValidDisparity, p.maxValidDisparity, border);
  }
  {
    // the sequential (reference-order) normalisation sums of all levels and frames run side by side in one launch
    ScopedTimer t(c, KC_NORMALIZATION, 0.0, fr.ln);
    // DisparitySpaceWarp::setNormalization is a no-op (bpvo/disparity_space_warp.h:87-90)
    launch_normalization(s, tab, NF, count, p.maxTestLevel, c->L, c->dspace ? 0 : p.withNormalization);
  }
  // one read-back of the point counts: the host needs them to size the template-build and GN grids
  launch_gather_counts(s, tab, NF, count, p.maxTestLevel, c->L, d_ints);
  FR_CK(c, fr, hipMemcpyAsync(h_ints, d_ints, sizeof(int) * kMaxLevels * (size_t) count, hipMemcpyDeviceToHost, s));
  if(fr.selected_ev) FR_CK(c, fr, hipEventRecord(fr.selected_ev, s));
  if(fr.on_selected) fr.on_selected();
  FR_CK(c, fr, hipStreamSynchronize(s));
  std::vector<int> max_n(c->L, 0);
  double pts = 0;
  for(int i = 0; i < count; ++i) {
    FrameSlot& f = c->frames[first + i * stride];
    for(int l = 0; l < c->L; ++l) {
      f.n_host[l] = (l >= p.maxTestLevel) ? h_ints[(size_t) i * kMaxLevels + l] : 0;
      max_n[l] = std::max(max_n[l], f.n_host[l]);
      pts += f.n_host[l];
    }
  }
  if(c->profiling) {
    std::lock_guard<std::mutex> lk(c->units_mu);
    c->kc_units[KC_TEMPLATE] += pts;
    c->kc_units[KC_NORMALIZATION] += pts;
  }
  for(int l = c->L - 1; l >= p.maxTestLevel; --l) {
    ScopedTimer t(c, KC_TEMPLATE, 0.0, fr.ln);
    launch_template_build(s, tab + (size_t) l * NF, c->C, max_n[l], count, p.gradientEstimation == BPVO_GRAD_CD5);
  }
  if(!fr.own_thread) {      // (a lane thread goes straight on to its estimation on the same stream)
    FR_CK(c, fr, hipStreamSynchronize(s));
    FR_CK(c, fr, hipGetLastError());
    resolve_events(c);
  }
  for(int i = 0; i < count; ++i) c->frames[first + i * stride].has_template = true;
  return BPVO_OK;
}
int frames_set_template(bpvo_hip_ctx* c, int first, int stride, int count)
{
  if(count <= 0) return BPVO_OK;
  if(first < 0 || stride < 1 || first + (count - 1) * stride >= c->n_frames) return fail(c, BPVO_ERR_INVALID_ARG, "bad frame slot range");
  for(int i = 0; i < count; ++i) {
    if(!c->frames[first + i * stride].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");   // vo_frame.cc:63
    if(!c->frames[first + i * stride].has_disp) return fail(c, BPVO_ERR_NO_DATA, "no disparity in frame (the current frame of a pair batch)");
  }
  return frames_set_template(c, first, stride, count, ctx_run(c));
}

// ---- estimatePose ---------------------------------------------------------------------------------------------------
// (an early return must not leave work in flight that still reads the lane's pinned staging: drain the stream first)
#define LANE_CK(ln_, expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if(e_ != hipSuccess) {                                                                  \
      (ln_)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
      (void) hipStreamSynchronize((ln_)->stream);                                           \
      return BPVO_ERR_DEVICE;                                                               \
    }                                                                                       \
  } while(0)

// VisualOdometryPoseEstimator::estimatePose (reference: bpvo/vo_pose_estimator.cc:63-93) for a group of `n` workspaces on one
// lane.  wss[i]: workspace, refs[i] / curs[i]: frame slots.  T_init host [n][16] or null (Identity).
// Does a group of n pairs take the team-persistent kernel?  (kLinear, the f64 formulation, C = 8 or 1 like gn_persistent_kernel; not
// while per-kernel timings are being collected: there are no kernels to time)
bool team_serves(const bpvo_hip_ctx* c, int n)
{
  const bool size_ok = (n >= 2 && n > c->persist_max_ws && n <= c->team_max_pairs) || (n == 1 && c->team_single);
  return c->team_mode && c->persistent && !c->persistent_failed.load() && size_ok &&
         (c->C == 8 || c->C == 1) && c->params.interp == BPVO_INTERP_LINEAR && !c->fast_warp && !c->profile_all && !c->profile_k6_all &&
         c->num_cus >= 2 && g_live_ctx[c->device & 63].load() <= 1;
}

// allow_persistent: only a group that has the device to itself (a batch on ONE lane) may take the persistent kernel — two
// hand-barrier grids of concurrent lanes must not be co-scheduled.
int estimate_group(bpvo_hip_ctx* c, Lane* ln, int n, const int* wss, const int* refs, const int* curs, const float* T_init,
                   float* poses, bpvo_hip_stats* stats, float* d_records_out, bool allow_persistent)
{
  if(n <= 0) return BPVO_OK;
  (void) hipSetDevice(c->device);
  const bpvo_hip_params& p = c->params;
  const int NP = c->n_pairs;
  // (the pinned staging of a lane is free here: every call that uses it ends with a synchronisation of the lane's stream)
  std::vector<int> max_pts(c->L, 0);
  for(int l = 0; l < c->L; ++l)
    for(int i = 0; i < n; ++i) {
      PairJob& pj = ln->h_pjobs[(size_t) l * NP + i];
      pj = make_pair_job(c, wss[i], refs[i], curs[i], l);
      // Dense levels (no non-maximum suppression: most pixels are template points) gather their taps straight from the descriptor:
      // neighbouring points share three quarters of their footprints, so the 32-byte records are fetched about once per pixel
      // from HBM, where the per-point tap cache reads 128 bytes per point whatever the neighbours do.  The cache pays at the
      // sparse levels (one point in ~25 pixels: every footprint its own two or three lines).  Batches only: the persistent
      // single-pair kernel keeps its (L2-resident) cache.
      if((c->C == 8 || c->C == 1) && n > c->persist_max_ws && (double) pj.n > c->tapcache_max_density * (double) c->geom[l].npix) {
        pj.tapcache_on = 0;
      }
      max_pts[l] = std::max(max_pts[l], pj.n);
    }
  if(c->ctl_by_kernel.load()) launch_copy_rows(ln->stream, ln->d_pjobs, ln->h_pjobs, sizeof(PairJob) * (size_t) c->L * NP, sizeof(PairJob) * (size_t) c->L * NP, 1);
  else LANE_CK(ln, hipMemcpyAsync(ln->d_pjobs, ln->h_pjobs, sizeof(PairJob) * (size_t) c->L * NP, hipMemcpyHostToDevice, ln->stream));
  const float* dT = nullptr;
  if(T_init) {
    std::memcpy(ln->h_T, T_init, sizeof(float) * 16 * n);
    if(c->ctl_by_kernel.load()) launch_copy_rows(ln->stream, ln->d_Tinit, ln->h_T, sizeof(float) * 16 * n, sizeof(float) * 16 * n, 1);
    else LANE_CK(ln, hipMemcpyAsync(ln->d_Tinit, ln->h_T, sizeof(float) * 16 * n, hipMemcpyHostToDevice, ln->stream));
    dT = ln->d_Tinit;
  }
  launch_set_pose(ln->stream, ln->d_pjobs + (size_t) (c->L - 1) * NP, dT, n);
  const bool pk_group = allow_persistent && c->persistent && !c->persistent_failed.load() && n <= c->persist_max_ws && !c->profile_all;
  bool persistent = pk_group;
  if(persistent) LANE_CK(ln, hipMemsetAsync(ln->d_pk_ctl, 0, sizeof(unsigned) * kPkCtlWords * kMaxLevels, ln->stream));

  // PoseEstimatorParameters(AlgorithmParameters) (bpvo/pose_estimator_params.cc:27-33): maxFuncEvals stays 6*200 (Q4);
  // the low-res parameter set equals the full-res one (Q3).
  const int max_fun_evals = 6 * 200;
  // Small batches: the whole level loop in ONE launch, a team of workgroups per pair (gn_team_kernel)
  bool team_ran = false;
  if(allow_persistent && ln == &c->lanes[0] && team_serves(c, n)) {
    GNTeamLaunch t;
    t.jobs_all = ln->d_pjobs; t.job_pitch = NP; t.n_pairs = n; t.level_hi = c->L - 1; t.level_lo = p.maxTestLevel;
    t.C = c->C; t.loss = p.lossFunction; t.fuse_frozen = c->fuse_frozen;
    t.scale_is_moot = (p.lossFunction == BPVO_LOSS_L2 && c->C == 8 && c->fuse_frozen) ? 1 : 0;
    // one workgroup per CU: teams of CUs / pairs workgroups (at most 64: the single-pair kernel's size), as many teams as fit
    const int slots = c->num_cus;
    int ts = c->team_size_env > 0 ? c->team_size_env : std::max(1, std::min(64, slots / n));
    ts = std::max(1, std::min(ts, slots));
    t.team_size = ts;
    t.n_teams = std::max(1, std::min(std::min(n, slots / ts), kMaxTeams));
    t.ctl = ln->d_team_ctl;
    t.timeout_ticks = c->persist_timeout;
    LANE_CK(ln, hipMemsetAsync(ln->d_team_ctl, 0, sizeof(unsigned) * (size_t) gn_team_ctl_words(t.n_teams), ln->stream));
    const hipError_t te = launch_gn_team(ln->stream, t, p.maxIterations, max_fun_evals, p.parameterTolerance, p.functionTolerance, p.gradientTolerance);
    if(te == hipSuccess) {
      team_ran = true;
      c->team_launches.fetch_add(1);
      LANE_CK(ln, hipMemcpyAsync(ln->h_team_ctl, ln->d_team_ctl, sizeof(unsigned) * 32, hipMemcpyDeviceToHost, ln->stream));
    } else {
      (void) hipGetLastError();
      c->persistent_failed.store(true);      // degrade to the chain, now and for later calls
    }
  }
  for(int l = c->L - 1; l >= p.maxTestLevel && !team_ran; --l) {
    GNLaunch g;
    g.jobs = ln->d_pjobs + (size_t) l * NP;
    g.npairs = n;
    g.max_points = max_pts[l];
    g.C = c->C;
    g.loss = p.lossFunction;
    g.fast_warp = c->fast_warp;
    g.interp = p.interp;
    g.fuse_frozen = c->fuse_frozen;
    // kL2: the weights are 1 whatever the robust scale — with the fused path every linearisation is irls_reduce + gn_step only
    const bool l2_moot = p.lossFunction == BPVO_LOSS_L2 && c->C == 8 && c->fuse_frozen && !c->fast_warp && p.interp == BPVO_INTERP_LINEAR;
    launch_level_begin(ln->stream, g.jobs, n, g.max_points, l, l2_moot ? 1 : 0);    // (and the tap-cache keys of the level)
    if(g.max_points <= 0) continue;
    if(persistent && gn_persistent_serves(g)) {
      // the whole level in one launch
      const hipError_t pe = launch_gn_persistent(ln->stream, g, p.maxIterations, max_fun_evals, p.parameterTolerance, p.functionTolerance, p.gradientTolerance,
                                                 ln->d_pk_ctl + (size_t) l * kPkCtlWords, gn_persistent_grid(g, c->persist_grid), c->persist_timeout);
      if(pe == hipSuccess) {
        c->persistent_levels.fetch_add(1);
        continue;
      }
      // the device cannot grant the kernel its LDS / residency (or the launch failed): degrade to the four-kernel chain — this level,
      // the rest of the pyramid and every later call of the context — instead of failing the estimate
      (void) hipGetLastError();
      c->persistent_failed.store(true);
    }
    persistent = false;     // (a level the kernel does not serve: the rest of the pyramid takes the chain as well)
    // At most maxIterations + 2 linearisations per level (pose_estimator_base.h:373-393); the state machine on the device
    // enforces the limits, the host queues rounds of kItersPerSync iterations until the device reports no active workspace.
    // Every round ends with a compaction of the list of still-active workspaces (ActiveSet, kernels.h) and the copy of its
    // count.  The rounds are PIPELINED: round r + 1 is queued with the list and count that came out of round r - 1, as soon
    // as those have landed — the device never waits for the host (a synchronisation per round was a ~30 us bubble: 7 % of a
    // round at 128 pairs, 11 % for a single pair).  Workspaces that finished in between are still dispatched for one more
    // round (their workgroups exit on the first load), and the level ends with one round of empty launches.
    const int max_lin = std::min(p.maxIterations + 2, max_fun_evals);
    const int kItersPerSync = 4;
    const int max_rounds = (max_lin + kItersPerSync - 1) / kItersPerSync + 2;
    constexpr unsigned kProfileEvery = 5;   // co-prime with kItersPerSync: no phase lock with the host round trips
    int* const lists[3] = {ln->d_list, ln->d_list + NP, ln->d_list + 2 * (size_t) NP};
    int n_cur = n;
    g.active.list = nullptr;                // first rounds: every workspace of the group, in order
    // Once NO active workspace of the list estimates its robust scale any more (a frozen scale stays frozen for the level, and the
    // list only shrinks), the median has nothing to do and — with the fused path, where irls_reduce recomputes the residuals of
    // frozen workspaces itself — neither has warp_residual: their launches are dropped for the rest of the level.  (Each would
    // still cost its floor of ~5 us per iteration in the tail of a level.)  l2_moot: true from the first linearisation.
    const bool fused_path = c->C == 8 && c->fuse_frozen && !c->fast_warp && p.interp == BPVO_INTERP_LINEAR;
    bool none_moving = l2_moot && c->skip_frozen_launches;
    for(int round = 0; round < max_rounds; ++round) {
      g.npairs = n_cur;
      g.merge_irls = n_cur < c->irls_merge_below ? 1 : 0;
      const bool launch_median_k = !none_moving, launch_warp_k = !(none_moving && fused_path);
      for(int k = 0; k < kItersPerSync; ++k) {
        // level 1 brackets every kProfileEvery-th warp_residual launch of the lane with events (a running counter, so the
        // sampled launches rotate through all iterations and levels): an event pair costs a few µs of dispatch gap
        if(launch_warp_k) {
          const bool sampled = c->profile_all || c->profile_k6_all || (ln->k6_seq++ % kProfileEvery) == 0;
          ScopedTimer t(c, KC_WARP_RESIDUAL, 0.0, ln, sampled);
          launch_warp_residual(ln->stream, g);
        }
        // level 2 times every kernel; level 3 (bench.py's single-lane roofline pass) every warp_residual AND every irls_reduce launch
        { ScopedTimer t(c, KC_MEDIAN, 0.0, ln, c->profile_all && launch_median_k); if(launch_median_k) launch_median(ln->stream, g); }
        { ScopedTimer t(c, KC_IRLS_REDUCE, 0.0, ln, c->profile_all || c->profile_k6_all); launch_irls_reduce(ln->stream, g); }
        { ScopedTimer t(c, KC_GN_STEP, 0.0, ln, c->profile_all);
          launch_gn_step(ln->stream, g, 0, p.maxIterations, max_fun_evals, p.parameterTolerance, p.functionTolerance, p.gradientTolerance); }
      }
      const int slot = round % 3;
      launch_compact_active(ln->stream, g.jobs, g.active, n_cur, lists[slot], ln->d_active + 2 * slot);
      LANE_CK(ln, hipMemcpyAsync(ln->h_active + 2 * slot, ln->d_active + 2 * slot, 2 * sizeof(int), hipMemcpyDeviceToHost, ln->stream));
      LANE_CK(ln, hipEventRecord(ln->round_ev[slot], ln->stream));
      if(c->sync_rounds) {                  // A/B: one synchronisation per round, the round's own list feeds the next
        LANE_CK(ln, hipStreamSynchronize(ln->stream));
        if(ln->h_active[2 * slot] <= 0) break;
        n_cur = ln->h_active[2 * slot];
        none_moving = none_moving || (c->skip_frozen_launches && ln->h_active[2 * slot + 1] == 0);
        g.active.list = lists[slot];
        continue;
      }
      if(round == 0) continue;              // nothing to learn yet: queue the second round behind the first
      const int prev = (round - 1) % 3;
      LANE_CK(ln, hipEventSynchronize(ln->round_ev[prev]));
      const int n_prev = ln->h_active[2 * prev];
      if(n_prev <= 0) break;                // (the round just queued runs empty)
      n_cur = n_prev;
      none_moving = none_moving || (c->skip_frozen_launches && ln->h_active[2 * prev + 1] == 0);
      g.active.list = lists[prev];
    }
  }
  launch_pack_records(ln->stream, ln->d_pjobs + (size_t) (c->L - 1) * NP, n, c->L, d_records_out);
  bool frac_queued = false;
  if(n == 1 && c->prefetch_frac_thr >= 0.0f && ln == &c->lanes[0]) {
    // fraction_good of this workspace at the level the estimate ended on, from the job already on the device
    const PairJob* job = ln->d_pjobs + (size_t) p.maxTestLevel * NP;
    const int npts = ln->h_pjobs[(size_t) p.maxTestLevel * NP].n;
    if(npts > 0) {
      GNLaunch gr;
      gr.jobs = job; gr.npairs = 1; gr.max_points = npts; gr.C = c->C;
      launch_refresh_residuals(ln->stream, gr);
      LANE_CK(ln, hipMemsetAsync(c->d_count, 0, sizeof(unsigned int), ln->stream));
      launch_count_good(ln->stream, job, npts, c->C, p.lossFunction, c->prefetch_frac_thr, c->d_count);
      LANE_CK(ln, hipMemcpyAsync(c->h_ints, c->d_count, sizeof(unsigned int), hipMemcpyDeviceToHost, ln->stream));
      frac_queued = true;
      c->frac_n = npts;
    }
  }
  // only this group's states: other lanes may still be writing theirs
  int ws_lo = wss[0], ws_hi = wss[0];
  for(int i = 1; i < n; ++i) { ws_lo = std::min(ws_lo, wss[i]); ws_hi = std::max(ws_hi, wss[i]); }
  LANE_CK(ln, hipMemcpyAsync(ln->h_states + ws_lo, c->d_states + ws_lo, sizeof(GNState) * (size_t) (ws_hi - ws_lo + 1), hipMemcpyDeviceToHost, ln->stream));
  if(pk_group)
    LANE_CK(ln, hipMemcpyAsync(ln->h_pk_ctl, ln->d_pk_ctl, sizeof(unsigned) * kPkCtlWords * kMaxLevels, hipMemcpyDeviceToHost, ln->stream));
  LANE_CK(ln, hipStreamSynchronize(ln->stream));
  LANE_CK(ln, hipGetLastError());
  if(frac_queued) { c->frac_valid = true; c->frac_ws = wss[0]; c->frac_thr = c->prefetch_frac_thr; c->frac_cnt = (unsigned) c->h_ints[0]; }
  if(pk_group && std::getenv("BPVO_HIP_PK_TIMING")) {     // library built with -DBPVO_PK_TIMING: per-phase ticks (10 ns) of workgroup 0
    static const char* names[6] = {"warp", "barrier1", "median", "irls", "barrier2", "step"};
    for(int l = c->L - 1; l >= 0; --l) {
      const unsigned* t = ln->h_pk_ctl + (size_t) l * kPkCtlWords;
      if(!t[15]) continue;
      std::fprintf(stderr, "pk level %d: %u iterations;", l, t[15]);
      for(int k = 0; k < 6; ++k) std::fprintf(stderr, " %s %.2f", names[k], 0.01 * t[8 + k] / t[15]);
      std::fprintf(stderr, " | step: sum_partials %.2f unpack %.2f solve %.2f pose %.2f tests %.2f", 0.01 * t[20] / t[15], 0.01 * t[16] / t[15], 0.01 * t[17] / t[15],
                   0.01 * t[18] / t[15], 0.01 * t[19] / t[15]);
      std::fprintf(stderr, " us per iteration\n");
    }
  }
  if(team_ran && std::getenv("BPVO_HIP_PK_TIMING")) {       // library built with -DBPVO_PK_TIMING: phases of team 0's first workgroup
    static const char* names[6] = {"warp", "barrier1", "median", "irls", "barrier2", "step"};
    for(int l = c->L - 1; l >= 0 && l < 4; --l) {
      const unsigned* t = ln->h_team_ctl + 4 + 7 * l;
      if(!t[6]) continue;
      std::fprintf(stderr, "team level %d: %u iterations;", l, t[6]);
      for(int k = 0; k < 6; ++k) std::fprintf(stderr, " %s %.2f", names[k], 0.01 * t[k] / t[6]);
      std::fprintf(stderr, " us per iteration\n");
    }
  }
  if(team_ran && ln->h_team_ctl[1] != 0) {
    // a team barrier timed out (teams not co-resident): rerun the group through the four-kernel chain and stay on it
    c->persistent_failed.store(true);
    return estimate_group(c, ln, n, wss, refs, curs, T_init, poses, stats, d_records_out, false);
  }
  if(pk_group) {
    bool gave_up = false;
    for(int l = 0; l < c->L; ++l) gave_up = gave_up || ln->h_pk_ctl[(size_t) l * kPkCtlWords + 1] != 0;
    if(gave_up) {
      // a barrier timed out: the states of that level were not written back.  Rerun the group through the four-kernel chain
      // (same results) and keep this context on it.
      c->persistent_failed.store(true);
      return estimate_group(c, ln, n, wss, refs, curs, T_init, poses, stats, d_records_out, false);
    }
  }
  for(int i = 0; i < n; ++i) {
    const GNState& st = ln->h_states[wss[i]];
    if(poses) std::memcpy(poses + 16 * (size_t) i, st.T_out, 16 * sizeof(float));
    if(stats)
      for(int l = 0; l < c->L; ++l) stats[(size_t) i * c->L + l] = st.stats[l];
  }
  return BPVO_OK;
}

int estimate_batch(bpvo_hip_ctx* c, int n, const int* wss, const int* refs, const int* curs, const float* T_init, float* poses,
                   bpvo_hip_stats* stats)
{
  if(n <= 0) return BPVO_OK;
  if(n > c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "more pairs than workspaces");
  const bpvo_hip_params& p = c->params;
  for(int i = 0; i < n; ++i) {
    if(wss[i] < 0 || wss[i] >= c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "bad workspace");
    if(refs[i] < 0 || refs[i] >= c->n_frames || curs[i] < 0 || curs[i] >= c->n_frames) return fail(c, BPVO_ERR_INVALID_ARG, "bad frame slot");
    if(!c->frames[refs[i]].has_template) return fail(c, BPVO_ERR_NO_TEMPLATE, "reference frame has no template");
    if(!c->frames[curs[i]].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");
  }
  const int lanes_ok = g_live_ctx[c->device & 63].load() > 1 ? 1 : std::min((int) c->lanes.size(), c->max_lanes_now);
  int nl = std::max(1, std::min(lanes_ok, n / kMinPairsPerLane));
  if(team_serves(c, n)) nl = 1;      // the team-persistent kernel takes the whole chip
  // frame stages run on the ctx stream: the other lanes' streams start from a quiet device.  A single lane IS the ctx stream — its
  // launches simply queue behind the frame stage (sequential addFrame: ~30 us of idle device per frame otherwise).
  if(nl > 1) HIP_CK(c, hipStreamSynchronize(c->stream));
  std::vector<int> rcs(nl, BPVO_OK);
  c->frac_valid = false;      // (on the API thread: the lane threads only read the context's settings)
  auto run = [&](int k) {
    const int lo = (int) ((long long) n * k / nl), hi = (int) ((long long) n * (k + 1) / nl);
    rcs[k] = estimate_group(c, &c->lanes[k], hi - lo, wss + lo, refs + lo, curs + lo, T_init ? T_init + 16 * (size_t) lo : nullptr,
                            poses ? poses + 16 * (size_t) lo : nullptr, stats ? stats + (size_t) lo * c->L : nullptr,
                            c->d_records + (size_t) kRecordFloats * lo, nl == 1);
  };
  if(nl == 1) {
    run(0);
  } else {
    std::vector<std::thread> th;
    for(int k = 1; k < nl; ++k) th.emplace_back(run, k);
    run(0);
    for(auto& t : th) t.join();
  }
  for(int k = 0; k < nl; ++k)
    if(rcs[k]) { c->err = c->lanes[k].err; return rcs[k]; }
  resolve_events(c);
  for(int i = 0; i < n; ++i) {
    Workspace& w = c->ws[wss[i]];
    w.last_ref = refs[i];
    w.last_cur = curs[i];
    w.last_level = p.maxTestLevel;
  }
  return BPVO_OK;
}

// tiled device layout (types.h tile_index) -> reference channel-major layout: out[(ch*n + i)*E + e] for records of
// C*E floats per point cut in V-float pieces
void detile_to_channel_major(const float* src, int n, int C, int E, int V, float* out)
{
  const int W = C * E, pieces = W / V;
  for(int i = 0; i < n; ++i)
    for(int w = 0; w < W; ++w) {
      const int piece = w / V, within = w - piece * V;
      const float v = src[(((size_t) (i / kTile) * pieces + piece) * kTile + (size_t) (i % kTile)) * V + within];
      const int ch = w / E, e = w - ch * E;
      out[((size_t) ch * n + i) * E + e] = v;
    }
}
size_t tiled_floats(int n, int floats_per_point) { return (size_t) ((n + kTile - 1) / kTile) * kTile * floats_per_point; }

int refresh_counters(bpvo_hip_ctx* c)
{
  // per-workspace counters (PairJob::cnt), summed here
  std::vector<unsigned long long> all(kWsCounters * (size_t) c->n_pairs);
  HIP_CK(c, hipMemcpy(all.data(), c->d_counters, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  unsigned long long h[kWsCounters] = {};
  for(int w = 0; w < c->n_pairs; ++w)
    for(int k = 0; k < kWsCounters; ++k) h[k] += all[kWsCounters * (size_t) w + k];
  c->median_bracketed = h[2];
  c->median_full = h[3];
  for(int k = 0; k < 4; ++k) c->tap_counts[k] = h[5 + k];
  c->total_lin = h[1];
  // units of the GN kernels = points linearised (device-side count: only the pairs still active in a launch count)
  // warp_residual at profiling level 1 is timed on a 1-in-kProfileEvery sample of its launches: its units are scaled to
  // the sampled launches so that units / launches stays the points of an average launch
  double all_k6 = 0;
  for(const auto& ln : c->lanes) all_k6 += ln.k6_seq;
  const bool sampled = c->profiling && !c->profile_all && !c->profile_k6_all && all_k6 > 0;
  // (h[4]: the points warp_residual itself processed; workspaces with a frozen scale go through irls_reduce's fused path)
  c->kc_units[KC_WARP_RESIDUAL] = sampled ? (double) h[4] * (double) c->kc_launches[KC_WARP_RESIDUAL] / all_k6 : (double) h[4];
  c->points_fused = (double) h[10];
  c->kc_units[KC_IRLS_REDUCE] = (double) h[0];
  c->kc_units[KC_MEDIAN] = (double) h[0];
  c->kc_units[KC_GN_STEP] = (double) h[1];
  return BPVO_OK;
}

int upload_single_job(bpvo_hip_ctx* c, int ws, int ref, int cur, int level)
{
  c->lanes[0].h_pjobs[0] = make_pair_job(c, ws, ref, cur, level);
  HIP_CK(c, hipMemcpyAsync(c->d_job1, c->lanes[0].h_pjobs, sizeof(PairJob), hipMemcpyHostToDevice, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));   // h_pjobs is reused by the next call
  return BPVO_OK;
}

void trajectory_push(bpvo_hip_ctx* c, const M44& T)   // Trajectory::push_back + InvertPose (bpvo/trajectory.cc:30-50)
{
  M44 Ti = m44_identity();
  for(int i = 0; i < 3; ++i)
    for(int j = 0; j < 3; ++j) Ti.m[i * 4 + j] = T.m[j * 4 + i];
  for(int i = 0; i < 3; ++i) {
    float s = Ti.m[0 * 4 + i] * T.m[3];
    s += Ti.m[1 * 4 + i] * T.m[7];
    s += Ti.m[2 * 4 + i] * T.m[11];
    Ti.m[i * 4 + 3] = -s;
  }
  if(!c->trajectory.empty()) c->trajectory.push_back(m44_mul(c->trajectory.back(), Ti));
  else c->trajectory.push_back(Ti);
}

// Fused path of the estimate loops: the residual / valid buffers of a workspace may lag behind its last linearisation
// (GNState::r_stale).  Everything that reads them goes through here first; the check itself happens on the device.
int ensure_residuals(bpvo_hip_ctx* c, int ws)
{
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0 || c->C != 8) return BPVO_OK;
  int rc = upload_single_job(c, ws, w.last_ref, w.last_cur, w.last_level);
  if(rc) return rc;
  GNLaunch g;
  g.jobs = c->d_job1; g.npairs = 1; g.max_points = c->frames[w.last_ref].n_host[w.last_level]; g.C = c->C;
  launch_refresh_residuals(c->stream, g);
  return BPVO_OK;
}

int fraction_good(bpvo_hip_ctx* c, int ws, float thr, float* frac)
{
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n = c->frames[w.last_ref].n_host[w.last_level];
  if(c->frac_valid && c->frac_ws == ws && c->frac_thr == thr && c->frac_n == n) {      // queued behind the estimate by addFrame
    *frac = c->frac_cnt / static_cast<float>((size_t) n * c->C);
    return BPVO_OK;
  }
  int rc = ensure_residuals(c, ws);
  if(rc) return rc;
  rc = upload_single_job(c, ws, w.last_ref, w.last_cur, w.last_level);
  if(rc) return rc;
  HIP_CK(c, hipMemsetAsync(c->d_count, 0, sizeof(unsigned int), c->stream));
  launch_count_good(c->stream, c->d_job1, n, c->C, c->params.lossFunction, thr, c->d_count);
  unsigned int cnt = 0;
  HIP_CK(c, hipMemcpyAsync(c->h_ints, c->d_count, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  cnt = (unsigned int) c->h_ints[0];
  *frac = cnt / static_cast<float>((size_t) n * c->C);   // vo_pose_estimator.cc:105-106
  return BPVO_OK;
}

int get_weights_host(bpvo_hip_ctx* c, int ws, std::vector<float>& w_cm, int* n_out)
{
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n = c->frames[w.last_ref].n_host[w.last_level];
  const int C = c->C;
  int rc = ensure_residuals(c, ws);
  if(rc) return rc;
  rc = upload_single_job(c, ws, w.last_ref, w.last_cur, w.last_level);
  if(rc) return rc;
  launch_weights(c->stream, c->d_job1, n, C, c->params.lossFunction, c->d_wtmp);
  std::vector<float> pm((size_t) n * C);
  HIP_CK(c, hipMemcpyAsync(pm.data(), c->d_wtmp, pm.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  w_cm.resize(pm.size());
  for(int i = 0; i < n; ++i)
    for(int ch = 0; ch < C; ++ch) w_cm[(size_t) ch * n + i] = pm[(size_t) i * C + ch];
  *n_out = n;
  return BPVO_OK;
}

}  // namespace

#define CHECK_CTX(c) if(!(c)) return BPVO_ERR_INVALID_ARG
#define CHECK_SLOT(c, s) if((s) < 0 || (s) >= (c)->n_frames) return fail(c, BPVO_ERR_INVALID_ARG, "bad frame slot")
#define CHECK_WS(c, w) if((w) < 0 || (w) >= (c)->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "bad workspace")
#define CHECK_LEVEL(c, l) if((l) < (c)->params.maxTestLevel || (l) >= (c)->L) return fail(c, BPVO_ERR_INVALID_ARG, "bad level")

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
    if(nb > 4) return unsupported("latchNumBytes: 1, 2 and 4 (8, 16 and 32 channels) are on the device path");
    if(c->params.latchHalfSsdSize < 0 || c->params.latchHalfSsdSize > 8) return unsupported("latchHalfSsdSize: 0 .. 8 are on the device path");
  }
  if(c->params.descriptor == BPVO_DESC_CENTRAL_DIFFERENCE) {
    if(c->params.centralDifferenceRadius <= 0) { g_create_error = "invalid radius"; return BPVO_ERR_INVALID_ARG; }   // central_difference_descriptor.cc:19
    if(c->params.centralDifferenceRadius > 3) return unsupported("centralDifferenceRadius: 1, 2 and 3 (8, 24 and 48 channels) are on the device path");
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
  const bool grad_smoothed = c->params.descriptor == BPVO_DESC_INTENSITY_AND_GRADIENT && c->params.sigmaPriorToCensusTransform > 0.0f;
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
  for(auto& w : cp->ws) {
    CREATE_CK(hipMalloc((void**) &w.r, sizeof(float) * (size_t) cp->cap_max * cp->C));
    CREATE_CK(hipMalloc((void**) &w.valid, (size_t) cp->cap_max));
    CREATE_CK(hipMalloc((void**) &w.cand, sizeof(uint32_t) * (size_t) cp->cap_max * cp->C));
    CREATE_CK(hipMalloc((void**) &w.med_blk, sizeof(uint32_t) * 4 * nblk_max));
    if(cp->C == 8 || cp->C == 1) {      // tap cache of warp_residual: 4 taps x C floats per point
      CREATE_CK(hipMalloc((void**) &w.tapkey, sizeof(uint32_t) * (size_t) cp->cap_max));
      CREATE_CK(hipMalloc((void**) &w.tapcache, sizeof(float) * 4 * cp->C * (size_t) cp->cap_max));
    }
    // two buffers of tile partials (the persistent kernels double-buffer them by iteration parity, kernels_gn.hip pk_partials)
    CREATE_CK(hipMalloc((void**) &w.partials, sizeof(float) * (size_t) gn_partials_entries(cp->cap_max, cp->C) * kPartialStride));
  }
  CREATE_CK(hipMalloc((void**) &cp->d_states, sizeof(GNState) * n_pairs));
  CREATE_CK(hipMemset(cp->d_states, 0, sizeof(GNState) * n_pairs));
  CREATE_CK(hipMalloc((void**) &cp->d_fjobs, 2 * sizeof(FrameJob) * (size_t) cp->L * n_frames));
  CREATE_CK(hipMalloc((void**) &cp->d_job1, sizeof(PairJob)));
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
    int max_lanes = cp->C == 8 ? kDefaultLanes : kDefaultLanesNarrow;
    if(const char* e = std::getenv("BPVO_HIP_LANES")) max_lanes = std::max(1, std::min(8, std::atoi(e)));
    if(const char* e = std::getenv("BPVO_HIP_FUSE_FROZEN")) cp->fuse_frozen = std::atoi(e) != 0;
    if(const char* e = std::getenv("BPVO_HIP_IRLS_MERGE_BELOW")) cp->irls_merge_below = std::max(0, std::atoi(e));
    if(const char* e = std::getenv("BPVO_HIP_SYNC_ROUNDS")) cp->sync_rounds = std::atoi(e) != 0;
    if(const char* e = std::getenv("BPVO_HIP_STAGGER")) cp->stagger = std::atoi(e) != 0;
    if(const char* e = std::getenv("BPVO_HIP_SKIP_FROZEN")) cp->skip_frozen_launches = std::atoi(e) != 0;
    if(const char* e = std::getenv("BPVO_HIP_TAPCACHE_MAX_DENSITY")) cp->tapcache_max_density = std::atof(e);
    if(const char* e = std::getenv("BPVO_HIP_SPLIT_CENSUS")) cp->split_census = std::atoi(e) != 0;
    if(const char* e = std::getenv("BPVO_HIP_PERSISTENT")) cp->persistent = std::atoi(e) != 0;
    if(const char* e = std::getenv("BPVO_HIP_PERSIST_MAX_WS")) cp->persist_max_ws = std::max(1, std::min(kPersistMaxWs, std::atoi(e)));
    if(const char* e = std::getenv("BPVO_HIP_PERSIST_GRID")) cp->persist_grid = std::max(1, std::min(128, std::atoi(e)));
    if(const char* e = std::getenv("BPVO_HIP_PERSIST_TIMEOUT_TICKS")) cp->persist_timeout = std::max(1ll, std::atoll(e));   // tests of the give-up path
    if(const char* e = std::getenv("BPVO_HIP_UPLOAD_WORKERS")) cp->up_workers = std::max(0, std::min(32, std::atoi(e)));
    if(const char* e = std::getenv("BPVO_HIP_UPLOAD_GROUP")) cp->up_group = std::max(1, std::min(64, std::atoi(e)));
    if(const char* e = std::getenv("BPVO_HIP_UPLOAD_SUBBATCHES")) cp->up_subbatches = std::max(1, std::min(8, std::atoi(e)));
    if(const char* e = std::getenv("BPVO_HIP_UPLOAD_STREAMS")) cp->up_streams_n = std::max(1, std::min(32, std::atoi(e)));
    if(const char* e = std::getenv("BPVO_HIP_UPLOAD_PLAN")) {
      double f0 = 0, f1 = 0;
      if(std::sscanf(e, "%lf%*[,:]%lf", &f0, &f1) == 2 && f0 >= 0.0 && f1 > 0.0 && f0 + f1 < 1.0) { cp->up_plan[0] = f0; cp->up_plan[1] = f1; }
    }
    if(const char* e = std::getenv("BPVO_HIP_CTL_KERNEL_COPY")) cp->ctl_kernel_mode = std::max(0, std::min(2, std::atoi(e)));
    cp->ctl_by_kernel = cp->ctl_kernel_mode == 2;
    if(const char* e = std::getenv("BPVO_HIP_TEAM")) cp->team_mode = std::atoi(e) != 0;
    if(const char* e = std::getenv("BPVO_HIP_TEAM_MAX_PAIRS")) cp->team_max_pairs = std::max(0, std::atoi(e));
    if(const char* e = std::getenv("BPVO_HIP_TEAM_SINGLE")) cp->team_single = std::atoi(e) != 0;
    if(const char* e = std::getenv("BPVO_HIP_TEAM_SIZE")) cp->team_size_env = std::max(0, std::min(256, std::atoi(e)));
    {
      hipDeviceProp_t prop;
      if(hipGetDeviceProperties(&prop, device) == hipSuccess) cp->num_cus = prop.multiProcessorCount;
      if(const char* e = std::getenv("BPVO_HIP_TEAM_CUS")) cp->num_cus = std::max(1, std::atoi(e));      // (tests: fewer teams than pairs)
    }
    cp->lanes.resize(std::max(1, std::min(max_lanes, n_pairs / kMinPairsPerLane)));
  }
  for(size_t k = 0; k < cp->lanes.size(); ++k) {
    Lane& ln = cp->lanes[k];
    if(k == 0) { ln.stream = cp->stream; ln.owns_stream = false; }
    else { CREATE_CK(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking)); ln.owns_stream = true; }
    CREATE_CK(hipMalloc((void**) &ln.d_pjobs, sizeof(PairJob) * (size_t) cp->L * n_pairs));
    CREATE_CK(hipMalloc((void**) &ln.d_Tinit, sizeof(float) * 16 * n_pairs));
    CREATE_CK(hipMalloc((void**) &ln.d_active, 8 * sizeof(int)));
    CREATE_CK(hipMalloc((void**) &ln.d_list, 3 * sizeof(int) * (size_t) n_pairs));
    for(auto& e : ln.round_ev) CREATE_CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CREATE_CK(hipEventCreateWithFlags(&ln.selected_ev, hipEventDisableTiming));
    for(auto& e : ln.staging_ev) CREATE_CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CREATE_CK(hipHostMalloc((void**) &ln.h_pjobs, sizeof(PairJob) * (size_t) cp->L * n_pairs));
    CREATE_CK(hipHostMalloc((void**) &ln.h_T, sizeof(float) * 16 * n_pairs));
    CREATE_CK(hipHostMalloc((void**) &ln.h_active, 8 * sizeof(int)));
    CREATE_CK(hipMalloc((void**) &ln.d_pk_ctl, sizeof(unsigned) * kPkCtlWords * kMaxLevels));
    CREATE_CK(hipHostMalloc((void**) &ln.h_pk_ctl, sizeof(unsigned) * kPkCtlWords * kMaxLevels));
    if(k == 0) {
      CREATE_CK(hipMalloc((void**) &ln.d_team_ctl, sizeof(unsigned) * (size_t) gn_team_ctl_words(kMaxTeams)));
      CREATE_CK(hipHostMalloc((void**) &ln.h_team_ctl, sizeof(unsigned) * 32));
      std::memset(ln.h_team_ctl, 0, sizeof(unsigned) * 32);
    }
    CREATE_CK(hipHostMalloc((void**) &ln.h_states, sizeof(GNState) * n_pairs));
  }
  CREATE_CK(hipMalloc((void**) &cp->d_records, sizeof(float) * kRecordFloats * n_pairs));
  CREATE_CK(hipMalloc((void**) &cp->d_wtmp, sizeof(float) * (size_t) cp->cap_max * cp->C));
  CREATE_CK(hipMalloc((void**) &cp->d_count, sizeof(unsigned int)));
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
  (void) hipFree(c->d_states); (void) hipFree(c->d_fjobs); (void) hipFree(c->d_job1); (void) hipFree(c->d_latch_off);
  (void) hipFree(c->d_records); (void) hipFree(c->d_wtmp);
  (void) hipFree(c->d_count); (void) hipFree(c->d_counters); (void) hipFree(c->d_trace);
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

int bpvo_hip_frame_set_data(bpvo_hip_ctx* c, int slot, const uint8_t* image, const float* disparity)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  (void) hipSetDevice(c->device);
  int rc = frames_set_data(c, slot, 1, 1, image, disparity, false);
  if(rc) return rc;
  HIP_CK(c, hipStreamSynchronize(c->stream));   // the caller may reuse its buffers on return (vo_frame.cc:50-51)
  resolve_events(c);
  return BPVO_OK;
}
int bpvo_hip_frame_set_data_device(bpvo_hip_ctx* c, int slot, const uint8_t* d_image, const float* d_disparity)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  (void) hipSetDevice(c->device);
  int rc = frames_set_data(c, slot, 1, 1, d_image, d_disparity, true);
  if(rc) return rc;
  HIP_CK(c, hipStreamSynchronize(c->stream));
  resolve_events(c);
  return BPVO_OK;
}
int bpvo_hip_frames_set_data(bpvo_hip_ctx* c, int first_slot, int slot_stride, int count, const uint8_t* images,
                             const float* disparities, int on_device)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  int rc = frames_set_data(c, first_slot, slot_stride, count, images, disparities, on_device != 0);
  if(rc) return rc;
  HIP_CK(c, hipStreamSynchronize(c->stream));
  resolve_events(c);
  return BPVO_OK;
}
int bpvo_hip_frame_set_template(bpvo_hip_ctx* c, int slot)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  (void) hipSetDevice(c->device);
  return frames_set_template(c, slot, 1, 1);
}
int bpvo_hip_frames_set_template(bpvo_hip_ctx* c, int first_slot, int slot_stride, int count)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  return frames_set_template(c, first_slot, slot_stride, count);
}
int bpvo_hip_frame_clear(bpvo_hip_ctx* c, int slot)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  c->frames[slot].has_data = false;
  c->frames[slot].has_template = false;
  return BPVO_OK;
}
int bpvo_hip_frame_state(const bpvo_hip_ctx* c, int slot, int* has_data, int* has_template)
{
  if(!c || slot < 0 || slot >= c->n_frames) return BPVO_ERR_INVALID_ARG;
  *has_data = c->frames[slot].has_data;
  *has_template = c->frames[slot].has_template;
  return BPVO_OK;
}

// ---- accessors ------------------------------------------------------------------------------------------------------
int bpvo_hip_get_image(bpvo_hip_ctx* c, int slot, int level, uint8_t* out)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  if(level < 0 || level >= c->L) return fail(c, BPVO_ERR_INVALID_ARG, "bad level");
  if(!c->frames[slot].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipMemcpyAsync(out, c->frames[slot].img[level], c->geom[level].npix, hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
int bpvo_hip_get_descriptor_channel(bpvo_hip_ctx* c, int slot, int level, int channel, float* out)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot); CHECK_LEVEL(c, level);
  if(channel < 0 || channel >= c->C) return fail(c, BPVO_ERR_INVALID_ARG, "bad channel");
  if(!c->frames[slot].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");
  (void) hipSetDevice(c->device);
  const size_t npix = c->geom[level].npix;
  // de-interleave one channel: 2-D copy with a source pitch of C floats
  HIP_CK(c, hipMemcpy2DAsync(out, sizeof(float), c->frames[slot].desc[level] + channel, sizeof(float) * c->C, sizeof(float), npix,
                             hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
int bpvo_hip_get_saliency(bpvo_hip_ctx* c, int slot, int level, float* out)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot); CHECK_LEVEL(c, level);
  if(!c->frames[slot].has_template) return fail(c, BPVO_ERR_NO_TEMPLATE, "no template");
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipMemcpyAsync(out, c->frames[slot].sal[level], c->geom[level].npix * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
#define TMPL(c, slot, level)                                                                    \
  CHECK_CTX(c); CHECK_SLOT(c, slot); CHECK_LEVEL(c, level);                                     \
  if(!(c)->frames[slot].has_template) return fail(c, BPVO_ERR_NO_TEMPLATE, "no template");      \
  (void) hipSetDevice((c)->device);                                                             \
  FrameSlot& f = (c)->frames[slot];                                                             \
  const int n = f.n_host[level]

int bpvo_hip_num_points(bpvo_hip_ctx* c, int slot, int level, int* n_out) { TMPL(c, slot, level); *n_out = n; return BPVO_OK; }
int bpvo_hip_get_points(bpvo_hip_ctx* c, int slot, int level, float* xyzw)
{
  TMPL(c, slot, level);
  if(n) HIP_CK(c, hipMemcpyAsync(xyzw, f.pts[level], sizeof(float4) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
int bpvo_hip_get_point_indices(bpvo_hip_ctx* c, int slot, int level, int* inds)
{
  TMPL(c, slot, level);
  if(n) HIP_CK(c, hipMemcpyAsync(inds, f.inds[level], sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
int bpvo_hip_get_pixels(bpvo_hip_ctx* c, int slot, int level, float* pixels)
{
  TMPL(c, slot, level);
  const int C = c->C;
  std::vector<float> t(tiled_floats(n, C));
  if(n) HIP_CK(c, hipMemcpyAsync(t.data(), f.pix[level], t.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  detile_to_channel_major(t.data(), n, C, 1, C == 8 ? 4 : C, pixels);
  return BPVO_OK;
}
int bpvo_hip_get_jacobians(bpvo_hip_ctx* c, int slot, int level, float* J)
{
  TMPL(c, slot, level);
  if(n == 0) return BPVO_OK;
  // the rows are not stored: evaluate them on the device from (point, Ix, Iy) exactly like irls_reduce does
  const size_t bytes = sizeof(float) * 6 * (size_t) n * c->C;
  float* d_out = nullptr;
  HIP_CK(c, hipMalloc((void**) &d_out, bytes));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  c->h_fjobs[0] = make_frame_job(c, f, level);
  hipError_t e = hipMemcpyAsync(c->d_fjobs, c->h_fjobs, sizeof(FrameJob), hipMemcpyHostToDevice, c->stream);
  if(e == hipSuccess) {
    launch_export_jacobians(c->stream, c->d_fjobs, c->C, n, d_out);
    e = hipMemcpyAsync(J, d_out, bytes, hipMemcpyDeviceToHost, c->stream);
  }
  if(e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void) hipFree(d_out);
  HIP_CK(c, e);
  return BPVO_OK;
}
int bpvo_hip_get_normalization(bpvo_hip_ctx* c, int slot, int level, float T[16], float T_inv[16])
{
  TMPL(c, slot, level);
  M44 t = m44_identity(), ti = m44_identity();
  // no normalisation was set (withNormalization off, an empty level, or DisparitySpaceWarp, whose setNormalization is a
  // no-op): the warp keeps the Identity it was constructed with (bpvo/rigid_body_warp.cc:27-28), not [1, -1 * 0]
  if(!c->params.withNormalization || c->dspace || n == 0) {
    std::memcpy(T, t.m, 64);
    std::memcpy(T_inv, ti.m, 64);
    return BPVO_OK;
  }
  float nrm[4];
  HIP_CK(c, hipMemcpyAsync(nrm, f.nrm + 4 * level, sizeof(nrm), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  t.m[0] = t.m[5] = t.m[10] = nrm[0];
  t.m[3] = -nrm[0] * nrm[1]; t.m[7] = -nrm[0] * nrm[2]; t.m[11] = -nrm[0] * nrm[3];
  ti.m[0] = ti.m[5] = ti.m[10] = 1.0f / nrm[0];
  ti.m[3] = nrm[1]; ti.m[7] = nrm[2]; ti.m[11] = nrm[3];
  std::memcpy(T, t.m, 64);
  std::memcpy(T_inv, ti.m, 64);
  return BPVO_OK;
}

// ---- operator-level seam --------------------------------------------------------------------------------------------
static int linearize_impl(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, int level, const float T[16], int reset_scale, float given_scale,
                          float H[36], float G[6], float* f_norm, float* sigma, int* num_valid)
{
  CHECK_CTX(c); CHECK_WS(c, ws); CHECK_SLOT(c, ref_slot); CHECK_SLOT(c, cur_slot); CHECK_LEVEL(c, level);
  if(!c->frames[ref_slot].has_template) return fail(c, BPVO_ERR_NO_TEMPLATE, "reference frame has no template");
  if(!c->frames[cur_slot].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");
  if(c->frames[ref_slot].n_host[level] <= 0) return fail(c, BPVO_ERR_NO_TEMPLATE, "you should call setData before calling computeResiduals");
  c->frac_valid = false;
  (void) hipSetDevice(c->device);
  int rc = upload_single_job(c, ws, ref_slot, cur_slot, level);
  if(rc) return rc;
  Lane& l0 = c->lanes[0];
  std::memcpy(l0.h_T, T, 16 * sizeof(float));
  HIP_CK(c, hipMemcpyAsync(l0.d_Tinit, l0.h_T, 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  launch_prepare_linearize(c->stream, c->d_job1, l0.d_Tinit, reset_scale, level, given_scale);
  GNLaunch g;
  g.jobs = c->d_job1; g.npairs = 1; g.max_points = c->frames[ref_slot].n_host[level]; g.C = c->C; g.loss = c->params.lossFunction;
  g.fast_warp = c->fast_warp;
  g.interp = c->params.interp;
  launch_reset_tapkeys(c->stream, g);
  { ScopedTimer t(c, KC_WARP_RESIDUAL, 0.0); launch_warp_residual(c->stream, g); }
  { ScopedTimer t(c, KC_MEDIAN, 0.0); launch_median(c->stream, g); }
  { ScopedTimer t(c, KC_IRLS_REDUCE, 0.0); launch_irls_reduce(c->stream, g); }
  { ScopedTimer t(c, KC_GN_STEP, 0.0); launch_gn_step(c->stream, g, 1, 0, 0, 0, 0, 0); }
  HIP_CK(c, hipMemcpyAsync(l0.h_states, c->d_states + ws, sizeof(GNState), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  HIP_CK(c, hipGetLastError());
  resolve_events(c);
  const GNState& st = l0.h_states[0];
  std::memcpy(H, st.H, sizeof(st.H));
  std::memcpy(G, st.G, sizeof(st.G));
  *f_norm = st.f_norm;
  if(sigma) *sigma = st.scale;
  *num_valid = (int) st.n_valid;
  c->ws[ws].last_ref = ref_slot; c->ws[ws].last_cur = cur_slot; c->ws[ws].last_level = level;
  return BPVO_OK;
}
int bpvo_hip_linearize(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, int level, const float T[16], int reset_scale,
                       float H[36], float G[6], float* f_norm, float* sigma, int* num_valid)
{
  return linearize_impl(c, ws, ref_slot, cur_slot, level, T, reset_scale ? 1 : 0, 0.0f, H, G, f_norm, sigma, num_valid);
}
int bpvo_hip_linearize_at_scale(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, int level, const float T[16], float sigma,
                                float H[36], float G[6], float* f_norm, int* num_valid)
{
  if(c && !(sigma > 0.0f)) return fail(c, BPVO_ERR_INVALID_ARG, "sigma must be positive");
  return linearize_impl(c, ws, ref_slot, cur_slot, level, T, 2, sigma, H, G, f_norm, nullptr, num_valid);
}

int bpvo_hip_get_residuals(bpvo_hip_ctx* c, int ws, float* r, size_t* n_out)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n = c->frames[w.last_ref].n_host[w.last_level], C = c->C;
  if(n_out) *n_out = (size_t) n * C;
  if(!r) return BPVO_OK;
  (void) hipSetDevice(c->device);
  { int rc = ensure_residuals(c, ws); if(rc) return rc; }
  std::vector<float> t(tiled_floats(n, C));
  if(n) HIP_CK(c, hipMemcpyAsync(t.data(), w.r, t.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  detile_to_channel_major(t.data(), n, C, 1, C == 8 ? 4 : C, r);
  return BPVO_OK;
}
int bpvo_hip_get_valid(bpvo_hip_ctx* c, int ws, uint16_t* v, size_t* n_out)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n = c->frames[w.last_ref].n_host[w.last_level];
  if(n_out) *n_out = (size_t) n;
  if(!v) return BPVO_OK;
  (void) hipSetDevice(c->device);
  { int rc = ensure_residuals(c, ws); if(rc) return rc; }
  std::vector<uint8_t> b((size_t) n);
  HIP_CK(c, hipMemcpyAsync(b.data(), w.valid, b.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  for(int i = 0; i < n; ++i) v[i] = b[i];
  return BPVO_OK;
}
int bpvo_hip_get_weights(bpvo_hip_ctx* c, int ws, float* w, size_t* n_out)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  Workspace& wk = c->ws[ws];
  if(wk.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n0 = c->frames[wk.last_ref].n_host[wk.last_level];
  if(n_out) *n_out = (size_t) n0 * c->C;
  if(!w) return BPVO_OK;
  (void) hipSetDevice(c->device);
  std::vector<float> cm;
  int n = 0;
  int rc = get_weights_host(c, ws, cm, &n);
  if(rc) return rc;
  std::memcpy(w, cm.data(), cm.size() * sizeof(float));
  return BPVO_OK;
}
int bpvo_hip_fraction_good(bpvo_hip_ctx* c, int ws, float threshold, float* frac)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  (void) hipSetDevice(c->device);
  return fraction_good(c, ws, threshold, frac);
}

// TemplateData::computeResiduals throws on an empty template (reference: bpvo/template_data.cc:177).  The single-pair entry
// points mirror that; the batch entry points skip such levels of the affected pair (its statistics keep kSolverError).
static int check_template_not_empty(bpvo_hip_ctx* c, int ref_slot)
{
  if(ref_slot < 0 || ref_slot >= c->n_frames || !c->frames[ref_slot].has_template) return BPVO_OK;   // reported elsewhere
  for(int l = c->params.maxTestLevel; l < c->L; ++l)
    if(c->frames[ref_slot].n_host[l] <= 0) return fail(c, BPVO_ERR_NO_TEMPLATE, "you should call setData before calling computeResiduals");
  return BPVO_OK;
}

int bpvo_hip_set_warp_formulation(bpvo_hip_ctx* c, int mode)
{
  CHECK_CTX(c);
  if(mode != BPVO_WARP_PHOTO_ERROR_F64 && mode != BPVO_WARP_PROJECT_POINTS_F32 && mode != BPVO_WARP_DISPARITY_SPACE_F32)
    return fail(c, BPVO_ERR_INVALID_ARG, "unknown warp formulation");
  if(mode != BPVO_WARP_PHOTO_ERROR_F64 && c->params.interp != BPVO_INTERP_LINEAR)
    return fail(c, BPVO_ERR_UNSUPPORTED, "the f32 formulations are kLinear only (bpvo/photo_error.cc:118-214)");
  const int dspace = (mode == BPVO_WARP_DISPARITY_SPACE_F32) ? 1 : 0;
  if(dspace != c->dspace) {
    // templates hold the points / gradients of the other warp: they have to be rebuilt (frame data stays)
    for(auto& f : c->frames) f.has_template = false;
  }
  c->dspace = dspace;
  c->fast_warp = (mode != BPVO_WARP_PHOTO_ERROR_F64) ? 1 : 0;
  return BPVO_OK;
}

int bpvo_hip_estimate_pose(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, const float T_init[16], float T_est[16],
                           bpvo_hip_stats* stats)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  if(!T_init || !T_est) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr pose");
  if(int rc = check_template_not_empty(c, ref_slot)) return rc;
  (void) hipSetDevice(c->device);
  return estimate_batch(c, 1, &ws, &ref_slot, &cur_slot, T_init, T_est, stats);
}

int bpvo_hip_estimate_pose_trace(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, const float T_init[16], float T_est[16],
                                 bpvo_hip_stats* stats, float* records, int max_records, int* n_records)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  if(!T_init || !T_est || !n_records || max_records < 0 || (max_records > 0 && !records)) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr pose / records");
  if(int rc = check_template_not_empty(c, ref_slot)) return rc;
  (void) hipSetDevice(c->device);
  // at most min(maxIterations + 2, maxFuncEvals) linearisations per level (pose_estimator_base.h:373-393)
  const int cap = c->L * (std::min(std::max(c->params.maxIterations, 0) + 2, 6 * 200) + 1);
  if(cap > c->trace_cap) {
    HIP_CK(c, hipStreamSynchronize(c->stream));
    (void) hipFree(c->d_trace);
    c->d_trace = nullptr; c->trace_cap = 0;
    HIP_CK(c, hipMalloc((void**) &c->d_trace, sizeof(float) * kTraceFloats * (size_t) cap));
    c->trace_cap = cap;
  }
  c->trace_ws = ws;
  const int rc = estimate_batch(c, 1, &ws, &ref_slot, &cur_slot, T_init, T_est, stats);
  c->trace_ws = -1;
  if(rc) return rc;
  const int n = c->lanes[0].h_states[ws].trace_n;
  *n_records = n;
  const int ncopy = std::min(std::min(n, max_records), c->trace_cap);
  if(ncopy > 0) HIP_CK(c, hipMemcpy(records, c->d_trace, sizeof(float) * kTraceFloats * (size_t) ncopy, hipMemcpyDeviceToHost));
  return BPVO_OK;
}

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
  const int lvl = c->params.maxTestLevel;
  FrameSlot& ref = c->frames[c->vo_ref];
  const int n = ref.n_host[lvl];
  std::vector<float> w_cm;
  int nw = 0;
  int rc = get_weights_host(c, 0, w_cm, &nw);
  if(rc) return rc;
  if((size_t) n > w_cm.size()) return fail(c, BPVO_ERR_INVALID_ARG, "size mismatch");
  std::vector<float> pts((size_t) n * 4);
  std::vector<uint8_t> img(c->geom[0].npix);
  if(n) HIP_CK(c, hipMemcpyAsync(pts.data(), ref.pts[lvl], pts.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipMemcpyAsync(img.data(), ref.img[0], img.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  c->cloud.resize(n);
  const float* Kl = c->geom[lvl].K;
  for(int i = 0; i < n; ++i) {
    const float* X = pts.data() + 4 * (size_t) i;
    float x[3];
    for(int r = 0; r < 3; ++r) {   // getImagePoint (bpvo/rigid_body_warp.h:123-128)
      float s = Kl[r * 3 + 0] * X[0];
      s += Kl[r * 3 + 1] * X[1];
      s += Kl[r * 3 + 2] * X[2];
      x[r] = s;
    }
    const float z_i = 1.0f / x[2];
    float u = z_i * x[0], v = z_i * x[1];
    if(c->dspace) { u = X[0] + Kl[2]; v = X[1] + Kl[5]; }   // DisparitySpaceWarp::getImagePoint (disparity_space_warp.h:73-76)
    uint8_t col = 0;
    if(v >= 0 && v < c->rows && u >= 0 && u < c->cols) col = img[(size_t) ((int) v) * c->cols + (int) u];
    bpvo_hip_point_with_info& pw = c->cloud[i];
    std::memset(&pw, 0, sizeof(pw));
    std::memcpy(pw.xyzw, X, 4 * sizeof(float));
    pw.rgba[0] = col; pw.rgba[1] = col; pw.rgba[2] = col; pw.rgba[3] = 255;
    pw.weight = w_cm[i];
  }
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
    if(sp->windowRadius < 0 || sp->windowRadius > 7 || c->rows <= sp->windowRadius) return fail(c, BPVO_ERR_UNSUPPORTED, "SGM: windowRadius 0..7 (and fewer than image rows) are on the device path");
    // int16 path costs: the sums of four paths stay clear of saturation for penalties below this (the original saturates silently)
    if(sp->smoothnessPenaltyLarge > 4000) return fail(c, BPVO_ERR_UNSUPPORTED, "SGM: smoothnessPenaltyLarge <= 4000 on the device path");
    return BPVO_OK;
  }
  if(sp->algorithm != BPVO_STEREO_BLOCK_MATCHING) return fail(c, BPVO_ERR_UNSUPPORTED, "StereoAlgorithm: BlockMatching and SGM are on the device path (SGBM is OpenCV's, RSGM is not built)");
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
  c->cloud.clear();                 // the point cloud belongs to one Result (bpvo/types.h:549-563)
  c->cloud_pose = I;

  int rc = frames_set_data(c, c->vo_cur, 1, 1, image, disparity, on_device);   // _cur_frame->setData (vo.cc:131)
  if(rc) return rc;
  HIP_CK(c, hipStreamSynchronize(c->stream));

  if(!c->frames[c->vo_ref].has_template) {            // first frame (vo.cc:133-139)
    std::swap(c->vo_ref, c->vo_cur);
    rc = frames_set_template(c, c->vo_ref, 1, 1);
    if(rc) return rc;
    trajectory_push(c, c->T_kf);
    ret->isKeyFrame = 1;
    ret->keyFramingReason = BPVO_KF_FIRST_FRAME;
    return BPVO_OK;
  }

  M44 T_est;
  const int ws0 = 0;
  rc = check_template_not_empty(c, c->vo_ref);
  if(rc) return rc;
  c->prefetch_frac_thr = c->params.goodPointThreshold;      // should_keyframe's fraction of good points rides behind the estimate
  rc = estimate_batch(c, 1, &ws0, &c->vo_ref, &c->vo_cur, c->T_kf.m, T_est.m, ret->optimizerStatistics);
  c->prefetch_frac_thr = -1.0f;
  if(rc) return rc;
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
      rc = frames_set_template(c, c->vo_ref, 1, 1);
      if(rc) return rc;
      rc = estimate_batch(c, 1, &ws0, &c->vo_ref, &c->vo_cur, I.m, T_est.m, ret->optimizerStatistics);
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
  if(n) *n = c->cloud.size();
  if(pts) std::memcpy(pts, c->cloud.data(), c->cloud.size() * sizeof(bpvo_hip_point_with_info));
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

// ---- batches --------------------------------------------------------------------------------------------------------
int bpvo_hip_batch_estimate(bpvo_hip_ctx* c, int n_pairs, const float* T_init, float* poses, bpvo_hip_stats* stats)
{
  CHECK_CTX(c);
  if(n_pairs < 0 || 2 * n_pairs > c->n_frames || n_pairs > c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "batch exceeds ctx capacity");
  (void) hipSetDevice(c->device);
  std::vector<int> wss(n_pairs), refs(n_pairs), curs(n_pairs);
  for(int p = 0; p < n_pairs; ++p) { wss[p] = p; refs[p] = 2 * p; curs[p] = 2 * p + 1; }
  return estimate_batch(c, n_pairs, wss.data(), refs.data(), curs.data(), T_init, poses, stats);
}
// ---- upload pipeline of host-buffer batches ---------------------------------------------------------------------------------
// The caller's buffers are pageable: a hipMemcpyAsync from them is staged by the runtime through ONE thread's memcpy (a few GB/s) on the
// stream that should be computing.  Here up_workers threads copy chunks of kUploadChunkPairs pairs (both images, the disparity of the
// template frame A only) into pinned slots of their own and hand them to the copy engines on their own streams; a lane's frame stage
// takes the chunks of its pairs as they land (one stream-wait per chunk) and ingests them from the device staging area.  The first chunk
// is all the device ever waits for; the rest of the upload runs under the compute of the chunks before it.
constexpr int kUploadChunkPairs = 16;
struct UploadRun {
  bpvo_hip_ctx* c = nullptr;
  int n_pairs = 0;
  std::vector<std::pair<int, int>> chunks;     // [first pair, count), lane after lane
  std::vector<int> recorded;                   // 1: the chunk's event has been recorded (or the worker failed: error set)
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::thread> workers;
  std::string err;
  ~UploadRun() { for(auto& t : workers) if(t.joinable()) t.join(); }
};

int upload_prepare(bpvo_hip_ctx* c, int n_pairs)
{
  const size_t npix = c->geom[0].npix;
  if((int) c->up_streams.size() < c->up_workers) {
    for(int w = (int) c->up_streams.size(); w < c->up_workers; ++w) {
      hipStream_t st = nullptr;
      HIP_CK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      c->up_streams.push_back(st);
      c->up_slot_bytes = (size_t) kUploadChunkPairs * npix * (2 + 4);
      uint8_t* pin = nullptr;
      HIP_CK(c, hipHostMalloc((void**) &pin, 2 * c->up_slot_bytes));
      c->up_pinned.push_back(pin);
      for(int sl = 0; sl < 2; ++sl) {
        hipEvent_t e = nullptr;
        HIP_CK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->up_slot_free.push_back(e);
      }
    }
  }
  if(n_pairs > c->up_cap_pairs) {
    HIP_CK(c, hipDeviceSynchronize());
    (void) hipFree(c->up_d_img); (void) hipFree(c->up_d_disp);
    c->up_d_img = nullptr; c->up_d_disp = nullptr; c->up_cap_pairs = 0;
    HIP_CK(c, hipMalloc((void**) &c->up_d_img, (size_t) 2 * n_pairs * npix));
    HIP_CK(c, hipMalloc((void**) &c->up_d_disp, (size_t) n_pairs * npix * sizeof(float)));
    c->up_cap_pairs = n_pairs;
  }
  return BPVO_OK;
}

// The pairs of a host batch are cut into nl * nsub groups of consecutive pairs, uploaded in that order; lane k runs the groups k, k + nl,
// k + 2 nl, ... one after the other, each end to end (frame stage as its chunks land, template, estimate).  With nsub = 1 a lane's first
// kernel of the Gauss-Newton stage waits for HALF the batch (2 lanes) to cross the bus and the second lane for all of it: 46 ms of a
// 196 ms step were exposed upload (profiles/r03_host_buffers_first.txt).  With nsub = 2 the first group is a quarter of the batch and
// every later group has landed long before its lane gets to it.
void host_groups(int n_pairs, int nl, int nsub, std::vector<std::pair<int, int>>& groups)
{
  const int ng = nl * nsub;
  groups.clear();
  for(int g = 0; g < ng; ++g) {
    const int lo = (int) ((long long) n_pairs * g / ng), hi = (int) ((long long) n_pairs * (g + 1) / ng);
    groups.emplace_back(lo, hi);
  }
}
// Two lanes, three groups in upload order: [0, a) -> lane 0, [a, a + b) -> lane 1, the rest -> lane 0 again (group 3, lane 1's second, is
// empty).  With two equal groups nothing but frame kernels runs for the first 30 ms of a 1024-pair step (lane 0's half has to land
// first: profiles/r03_host_timeline.txt); a first group of a fifth of the batch has landed after 10 ms, and what it loses as a small
// batch is less than the 17 ms it gains.
void host_groups_plan(const bpvo_hip_ctx* c, int n_pairs, std::vector<std::pair<int, int>>& groups)
{
  auto chunks = [](double pairs) { return (int) std::lround(pairs / kUploadChunkPairs) * kUploadChunkPairs; };
  const int a = std::max(4 * kUploadChunkPairs, chunks(c->up_plan[0] * n_pairs));
  const int b = std::max(4 * kUploadChunkPairs, std::min(n_pairs - a - 4 * kUploadChunkPairs, chunks(c->up_plan[1] * n_pairs)));
  groups.clear();
  groups.emplace_back(0, a);
  groups.emplace_back(a, a + b);
  groups.emplace_back(a + b, n_pairs);
  groups.emplace_back(n_pairs, n_pairs);
}
// starts the workers; chunks are cut inside the groups, in group order
int upload_start(bpvo_hip_ctx* c, UploadRun& u, int n_pairs, const std::vector<std::pair<int, int>>& groups, const uint8_t* images, const float* disparities)
{
  int rc = upload_prepare(c, n_pairs);
  if(rc) return rc;
  u.c = c; u.n_pairs = n_pairs;
  for(const auto& g : groups)
    for(int p0 = g.first; p0 < g.second; p0 += kUploadChunkPairs) u.chunks.emplace_back(p0, std::min(kUploadChunkPairs, g.second - p0));
  const int nchunks = (int) u.chunks.size();
  while((int) c->up_chunk_done.size() < nchunks) {
    hipEvent_t e = nullptr;
    HIP_CK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->up_chunk_done.push_back(e);
  }
  u.recorded.assign(nchunks, 0);
  const size_t npix = c->geom[0].npix;
  const int T = std::min(c->up_workers, nchunks);
  const auto t_start = std::chrono::steady_clock::now();
  c->up_last_bytes = (size_t) n_pairs * npix * (2 + 4);
  auto done_count = std::make_shared<std::atomic<int>>(0);
  for(int w = 0; w < T; ++w) {
    u.workers.emplace_back([c, &u, w, T, nchunks, npix, images, disparities, t_start, done_count] {
      (void) hipSetDevice(c->device);
      hipStream_t st = c->up_streams[w % std::max(1, std::min(c->up_streams_n, (int) c->up_streams.size()))];
      int turn = 0;
      for(int k = w; k < nchunks; k += T, ++turn) {
        const int p0 = u.chunks[k].first, np = u.chunks[k].second, sl = turn & 1;
        hipError_t e = hipEventSynchronize(c->up_slot_free[2 * w + sl]);       // the copy that last read this slot has finished
        uint8_t* pin_img = c->up_pinned[w] + (size_t) sl * c->up_slot_bytes;
        float* pin_disp = reinterpret_cast<float*>(pin_img + (size_t) kUploadChunkPairs * npix * 2);
        std::memcpy(pin_img, images + (size_t) 2 * p0 * npix, (size_t) 2 * np * npix);
        for(int i = 0; i < np; ++i) std::memcpy(pin_disp + (size_t) i * npix, disparities + (size_t) 2 * (p0 + i) * npix, npix * sizeof(float));
        if(e == hipSuccess) e = hipMemcpyAsync(c->up_d_img + (size_t) 2 * p0 * npix, pin_img, (size_t) 2 * np * npix, hipMemcpyHostToDevice, st);
        if(e == hipSuccess) e = hipMemcpyAsync(c->up_d_disp + (size_t) p0 * npix, pin_disp, (size_t) np * npix * sizeof(float), hipMemcpyHostToDevice, st);
        if(e == hipSuccess) e = hipEventRecord(c->up_slot_free[2 * w + sl], st);
        if(e == hipSuccess) e = hipEventRecord(c->up_chunk_done[k], st);
        {
          std::lock_guard<std::mutex> lk(u.mu);
          if(e != hipSuccess && u.err.empty()) u.err = std::string("upload pipeline: ") + hipGetErrorString(e);
          u.recorded[k] = 1;
        }
        u.cv.notify_all();
      }
      (void) hipStreamSynchronize(st);
      if(done_count->fetch_add(1) + 1 == T)
        c->up_last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    });
  }
  return BPVO_OK;
}

// setData of the pairs [lo, hi) of a lane from the staging area as the chunks land, up_group chunks per frame-stage launch (the first
// group of a lane is a single chunk: the device starts as soon as anything is there)
int upload_consume(bpvo_hip_ctx* c, UploadRun& u, int lo, int hi, const FrameRun& fr_lane)
{
  const size_t npix = c->geom[0].npix;
  std::vector<size_t> mine;
  for(size_t k = 0; k < u.chunks.size(); ++k)
    if(u.chunks[k].first >= lo && u.chunks[k].first < hi) mine.push_back(k);
  for(size_t i = 0; i < mine.size();) {
    const size_t take = std::min(mine.size() - i, (size_t) (i == 0 ? 1 : c->up_group));
    for(size_t q = i; q < i + take; ++q) {
      const size_t k = mine[q];
      {
        std::unique_lock<std::mutex> lk(u.mu);
        u.cv.wait(lk, [&] { return u.recorded[k] != 0; });
        if(!u.err.empty()) { (fr_lane.own_thread ? fr_lane.ln->err : c->err) = u.err; return BPVO_ERR_DEVICE; }
      }
      FR_CK(c, fr_lane, hipStreamWaitEvent(fr_lane.stream, c->up_chunk_done[k], 0));
    }
    const int p0 = u.chunks[mine[i]].first;
    int np = 0;
    for(size_t q = i; q < i + take; ++q) np += u.chunks[mine[q]].second;      // (the chunks of a lane are contiguous)
    FrameRun fr = fr_lane;
    fr.tab = 2 * p0;
    fr.selected_ev = nullptr; fr.on_selected = nullptr;
    // skip_odd_disp = 2: the staging area holds the A frames' disparities only, packed
    int rc = frames_set_data(c, 2 * p0, 1, 2 * np, c->up_d_img + (size_t) 2 * p0 * npix, c->up_d_disp + (size_t) p0 * npix, true, fr, 2);
    if(rc) return rc;
    i += take;
  }
  return BPVO_OK;
}

// Staggered lanes (round 2).  A batch whose frame stage ran as a whole before any estimation starts every lane at the coarsest
// pyramid level at the same moment: for the first two levels (a few hundred points per pair) every launch is latency-bound and the
// chip idles, whatever the number of lanes.  Here each lane runs ITS pairs end to end on its own stream — setData, setTemplate,
// estimatePose — and lane k's frame stage is queued behind lane k-1's selection: the chip-filling frame kernels of one lane run
// under the narrow coarse-level iterations of the previous one, and the coarse levels of lane k under the fine levels of lane k-1.
// Same kernels on the same data per pair: results are bit-identical to the one-stage-at-a-time form (BPVO_HIP_STAGGER=0).
int batch_run_staggered(bpvo_hip_ctx* c, int n_pairs, int nl, const uint8_t* images, const float* disparities, bool on_device, float* poses,
                        bpvo_hip_stats* stats, UploadRun* pipe, const std::vector<std::pair<int, int>>* host_group_list)
{
  HIP_CK(c, hipStreamSynchronize(c->stream));
  c->frac_valid = false;
  const size_t npix = c->geom[0].npix;
  std::vector<int> rcs(nl, BPVO_OK);
  std::mutex mu;
  std::condition_variable cv;
  std::vector<int> selected(nl, 0);     // 1: the lane recorded its selected_ev (or failed before: nobody waits for ever)
  // the groups of consecutive pairs a lane runs one after the other: one per lane, or (host buffers) nsub per lane in upload order
  std::vector<std::pair<int, int>> groups;
  if(host_group_list) groups = *host_group_list;
  else host_groups(n_pairs, nl, 1, groups);
  const int nsub = (int) groups.size() / nl;
  auto run = [&](int k) {
    Lane* ln = &c->lanes[k];
    (void) hipSetDevice(c->device);
    auto release_next = [&mu, &cv, &selected, k] { { std::lock_guard<std::mutex> lk(mu); selected[k] = 1; } cv.notify_all(); };
    struct Release { std::function<void()> f; ~Release() { f(); } } always{release_next};   // whatever happens, nobody waits for ever
    // device-resident inputs: lane k's frame stage starts behind lane k - 1's selection (stagger); host inputs arrive staggered anyway
    if(k > 0 && !pipe) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return selected[k - 1] != 0; }); }
      if(hipStreamWaitEvent(ln->stream, c->lanes[k - 1].selected_ev, 0) != hipSuccess) { ln->err = "hipStreamWaitEvent"; rcs[k] = BPVO_ERR_DEVICE; return; }
    }
    for(int sub = 0; sub < nsub; ++sub) {
      const int lo = groups[(size_t) sub * nl + k].first, hi = groups[(size_t) sub * nl + k].second, n = hi - lo;
      if(n <= 0) continue;
      FrameRun fr{ln->stream, ln, 2 * lo, true, nullptr, nullptr};
      int rc = pipe ? upload_consume(c, *pipe, lo, hi, fr)
                    : frames_set_data(c, 2 * lo, 1, 2 * n, images + (size_t) 2 * lo * npix, disparities + (size_t) 2 * lo * npix, on_device, fr, 1);
      if(rc) { rcs[k] = rc; return; }
      if(sub == 0) {
        fr.selected_ev = ln->selected_ev;     // recorded, and the next lane released, before the template stage waits for its point counts
        fr.on_selected = release_next;
      }
      rc = frames_set_template(c, 2 * lo, 2, n, fr);
      if(rc) { rcs[k] = rc; return; }
      std::vector<int> wss(n), refs(n), curs(n);
      for(int i = 0; i < n; ++i) { wss[i] = lo + i; refs[i] = 2 * (lo + i); curs[i] = 2 * (lo + i) + 1; }
      rc = estimate_group(c, ln, n, wss.data(), refs.data(), curs.data(), nullptr, poses ? poses + 16 * (size_t) lo : nullptr,
                          stats ? stats + (size_t) lo * c->L : nullptr, c->d_records + (size_t) kRecordFloats * lo, false);
      if(rc) { rcs[k] = rc; return; }
    }
  };
  {
    std::vector<std::thread> th;
    for(int k = 1; k < nl; ++k) th.emplace_back(run, k);
    run(0);
    for(auto& t : th) t.join();
  }
  for(int k = 0; k < nl; ++k)
    if(rcs[k]) { c->err = c->lanes[k].err; return rcs[k]; }
  resolve_events(c);
  for(int i = 0; i < n_pairs; ++i) {
    Workspace& w = c->ws[i];
    w.last_ref = 2 * i;
    w.last_cur = 2 * i + 1;
    w.last_level = c->params.maxTestLevel;
  }
  return BPVO_OK;
}

int bpvo_hip_batch_run(bpvo_hip_ctx* c, int n_pairs, const uint8_t* images, const float* disparities, int on_device,
                       float* poses, bpvo_hip_stats* stats)
{
  CHECK_CTX(c);
  if(n_pairs < 0 || 2 * n_pairs > c->n_frames || n_pairs > c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "batch exceeds ctx capacity");
  (void) hipSetDevice(c->device);
  if(n_pairs > 0 && (!images || !disparities)) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr image/disparity");
  const int lanes_ok = g_live_ctx[c->device & 63].load() > 1 ? 1 : std::min((int) c->lanes.size(), c->max_lanes_now);
  int nl = std::max(1, std::min(lanes_ok, n_pairs / kMinPairsPerLane));
  if(team_serves(c, n_pairs)) nl = 1;
  // host buffers: batches of at least two chunks go through the upload pipeline
  const bool use_pipe = !on_device && c->up_workers > 0 && n_pairs >= 2 * kUploadChunkPairs;
  if(c->stagger && nl > 1 && !c->profile_all) {
    if(!use_pipe) return batch_run_staggered(c, n_pairs, nl, images, disparities, on_device != 0, poses, stats, nullptr, nullptr);
    // (groups of at least 64 pairs: smaller ones cost more in launch floors than their earlier start is worth)
    std::vector<std::pair<int, int>> groups;
    const int nsub = std::max(1, std::min(c->up_subbatches, n_pairs / (64 * nl)));
    const bool plan = nl == 2 && nsub == 1 && c->up_plan[0] > 0.0 && n_pairs >= 32 * kUploadChunkPairs;
    if(plan) host_groups_plan(c, n_pairs, groups);
    else host_groups(n_pairs, nl, nsub, groups);
    UploadRun pipe;
    int rcp = upload_start(c, pipe, n_pairs, groups, images, disparities);
    if(rcp) return rcp;
    if(c->ctl_kernel_mode == 1) c->ctl_by_kernel = true;
    rcp = batch_run_staggered(c, n_pairs, nl, images, disparities, false, poses, stats, &pipe, &groups);
    if(c->ctl_kernel_mode == 1) c->ctl_by_kernel = false;
    return rcp;
  }
  int rc;
  if(use_pipe) {
    UploadRun pipe;
    std::vector<std::pair<int, int>> groups;
    host_groups(n_pairs, 1, 1, groups);
    rc = upload_start(c, pipe, n_pairs, groups, images, disparities);
    if(rc) return rc;
    rc = upload_consume(c, pipe, 0, n_pairs, ctx_run(c));
  } else {
    rc = frames_set_data(c, 0, 1, 2 * n_pairs, images, disparities, on_device != 0, 1);
  }
  if(rc) return rc;
  rc = frames_set_template(c, 0, 2, n_pairs);
  if(rc) return rc;
  return bpvo_hip_batch_estimate(c, n_pairs, nullptr, poses, stats);
}
int bpvo_hip_batch_result_records_device(bpvo_hip_ctx* c, const float** d_records, int* floats_per_pair)
{
  CHECK_CTX(c);
  *d_records = c->d_records;
  *floats_per_pair = kRecordFloats;
  return BPVO_OK;
}

int bpvo_hip_batch_copy_records_device(bpvo_hip_ctx* c, float* d_dst, int n_pairs)
{
  CHECK_CTX(c);
  if(!d_dst || n_pairs < 0 || n_pairs > c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "bad record copy");
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipMemcpyAsync(d_dst, c->d_records, sizeof(float) * kRecordFloats * (size_t) n_pairs, hipMemcpyDeviceToDevice, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}

// ---- measurement ----------------------------------------------------------------------------------------------------
int bpvo_hip_profiling(bpvo_hip_ctx* c, int enable)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  resolve_events(c);
  c->profiling = enable != 0;
  c->profile_all = enable == 2;
  c->profile_k6_all = enable == 3;
  for(int k = 0; k < KC_COUNT; ++k) { c->kc_ms[k] = 0; c->kc_units[k] = 0; c->kc_launches[k] = 0; }
  HIP_CK(c, hipMemset(c->d_counters, 0, kWsCounters * sizeof(unsigned long long) * c->n_pairs));
  c->total_lin = 0;
  for(auto& ln : c->lanes) ln.k6_seq = 0;
  return BPVO_OK;
}
int bpvo_hip_get_kernel_stats(bpvo_hip_ctx* c, bpvo_hip_kernel_stat* out, int max_out, int* n_out)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  resolve_events(c);
  int rc = refresh_counters(c);
  if(rc) return rc;
  // algorithmic bytes per unit: unit = template point for the GN kernels, pixel otherwise
  const double C = c->C;
  const double bpu[KC_COUNT] = {2.0, 1.0 + 4.0 * C, 4.0 * C + 4.0 + 4.0, 32.0, 16.0 + 4.0 + 5.0 * 4.0 * C + 28.0 * C, 18.0 + 24.0 * C,
                                4.0 * C, 2.0 + 28.0 * C, 0.0};
  int n = 0;
  for(int k = 0; k < KC_COUNT && n < max_out; ++k, ++n) {
    std::memset(&out[n], 0, sizeof(out[n]));
    std::snprintf(out[n].name, sizeof(out[n].name), "%s", kKernelNames[k]);
    out[n].launches = c->kc_launches[k];
    out[n].total_ms = c->kc_ms[k];
    out[n].units = c->kc_units[k];
    out[n].bytes_per_unit = bpu[k];
    if(k == KC_IRLS_REDUCE && c->kc_units[k] > 0)   // fused points carry warp_residual's bytes as well
      out[n].bytes_per_unit = bpu[k] + bpu[KC_WARP_RESIDUAL] * c->points_fused / c->kc_units[k];
  }
  *n_out = n;
  return BPVO_OK;
}
int bpvo_hip_fused_point_counts(bpvo_hip_ctx* c, uint64_t* fused, uint64_t* total)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  int rc = refresh_counters(c);
  if(rc) return rc;
  *fused = (uint64_t) c->points_fused;
  *total = (uint64_t) c->kc_units[KC_IRLS_REDUCE];
  return BPVO_OK;
}
int bpvo_hip_persistent_counts(bpvo_hip_ctx* c, uint64_t* levels, int* gave_up)
{
  if(!c) return BPVO_ERR_INVALID_ARG;
  if(levels) *levels = c->persistent_levels.load();
  if(gave_up) *gave_up = c->persistent_failed.load() ? 1 : 0;
  return BPVO_OK;
}

int bpvo_hip_upload_stats(bpvo_hip_ctx* c, double* seconds, uint64_t* bytes)
{
  if(!c) return BPVO_ERR_INVALID_ARG;
  if(seconds) *seconds = c->up_last_seconds;
  if(bytes) *bytes = (uint64_t) c->up_last_bytes;
  return BPVO_OK;
}
int bpvo_hip_team_counts(bpvo_hip_ctx* c, uint64_t* launches)
{
  if(!c || !launches) return BPVO_ERR_INVALID_ARG;
  *launches = c->team_launches.load();
  return BPVO_OK;
}

int bpvo_hip_median_path_counts(bpvo_hip_ctx* c, uint64_t* bracketed, uint64_t* full)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  int rc = refresh_counters(c);
  if(rc) return rc;
  *bracketed = c->median_bracketed;
  *full = c->median_full;
  return BPVO_OK;
}
int bpvo_hip_tap_cache_counts(bpvo_hip_ctx* c, uint64_t out[4])
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  int rc = refresh_counters(c);
  if(rc) return rc;
  for(int k = 0; k < 4; ++k) out[k] = c->tap_counts[k];
  return BPVO_OK;
}
// diagnostics: the Gauss-Newton state of a workspace after its last call — out[0..15] T, [16..51] H, [52..57] G, [58..63] dp,
// [64] f_norm, [65] scale, [66] delta_scale, [67] g_norm, [68..83] T_lin (pose of the last linearisation)
int bpvo_hip_debug_gn_state(bpvo_hip_ctx* c, int ws, float out[84])
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  (void) hipSetDevice(c->device);
  GNState st;
  HIP_CK(c, hipStreamSynchronize(c->stream));
  HIP_CK(c, hipMemcpy(&st, c->d_states + ws, sizeof(st), hipMemcpyDeviceToHost));
  std::memcpy(out, st.T, 64); std::memcpy(out + 16, st.H, 144); std::memcpy(out + 52, st.G, 24); std::memcpy(out + 58, st.dp, 24);
  out[64] = st.f_norm; out[65] = st.scale; out[66] = st.delta_scale; out[67] = st.g_norm;
  std::memcpy(out + 68, st.T_lin, 64);
  return BPVO_OK;
}
int bpvo_hip_set_max_lanes(bpvo_hip_ctx* c, int n)
{
  CHECK_CTX(c);
  c->max_lanes_now = n >= 1 ? n : (1 << 30);
  return BPVO_OK;
}
int bpvo_hip_total_linearizations(bpvo_hip_ctx* c, uint64_t* n)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipStreamSynchronize(c->stream));
  int rc = refresh_counters(c);
  if(rc) return rc;
  *n = c->total_lin;
  return BPVO_OK;
}

}  // extern "C"
