// libbpvo_hip, host side: the context and what its translation units share (internal header; the interface is include/bpvo_hip/c_api.h).
//   context.hip   create / destroy, storage, job tables, per-context options            frames.hip    setData / setTemplate stages + accessors
//   estimate.hip  estimatePose: the Gauss-Newton drivers (chain, persistent, team)       batch.hip     pair batches, the upload pipeline
//   vo.hip        VisualOdometry::addFrame, point cloud, trajectory, stereo front-end    measure.hip   profiling, counters, diagnostics
// The host keeps bpvo's object model (frames with a descriptor pyramid and a template pyramid, a pose estimator with per-level
// Gauss-Newton runs, the VisualOdometry keyframe state machine) but every O(pixels) / O(points) array lives in HBM; per GN iteration the
// host sees one 4-byte "pairs still active" counter.  See DESIGN.md.
#pragma once
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
#include "latch_table.h"

namespace bpvo_hip_host {

using namespace bpvo_hip;


// live contexts per device: the estimation lanes of one context are streams = hardware queues, of which a process has a handful;
// a second context's lanes end up multiplexed on the same queues and LOSE (measured: 505 k instead of 666 k GN it/s for a
// 128-pair batch next to a second context), so batches only fan out over lanes while theirs is the only context on the device
extern std::atomic<int> g_live_ctx[64];
extern thread_local std::string g_create_error;   // bpvo_hip_last_error(nullptr): the failed create of THIS thread (contexts are created
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
  bool lazy[kMaxLevels] = {};     // levels whose descriptor records were not stored (FrameJob::lazy): census bytes + channel 0 only
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
static const char* const kKernelNames[KC_COUNT] = {"pyramid", "descriptor", "saliency_select", "normalization", "template_build", "warp_residual",
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
constexpr int kDefaultLanes = 3;          // C = 8: +1 ... +2.6 % over two at 128 - 512 pairs and at 640x480, +0.5 % at 1024 (profiles/r04_small_batch_model.txt); C = 1: -2 %
constexpr int kDefaultLanesNarrow = 2;
constexpr int kMinPairsPerLane = 8;
constexpr int kPkCtlWords = 32;    // one 128-byte line per level
constexpr int kMaxTeams = 1024;

}  // namespace bpvo_hip_host

using namespace bpvo_hip;
using namespace bpvo_hip_host;

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
  int scratch_planes = kDfPlanes;   // work planes of FrameSlot::scratch
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
  unsigned* d_tickets = nullptr;             // [n_pairs] PairJob::ticket
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
  bpvo_hip_point_with_info* d_cloud = nullptr;   // the last key frame's point cloud: built on the device (vo.hip build_point_cloud), copied out when asked for
  size_t d_cloud_cap = 0;
  size_t cloud_n = 0;
  M44 cloud_pose;
  // measurement
  double points_fused = 0;     // points linearised through the fused path since the last counter reset
  int fast_warp = 0;           // bpvo_hip_set_warp_formulation
  int dspace = 0;              // BPVO_WARP_DISPARITY_SPACE_F32: DisparitySpaceWarp as the warp (implies fast_warp)
  int fuse_frozen = 1;         // estimate loops: fused residual + reduction once a workspace's scale is frozen (bit-identical,
                               // +3 % GN iterations/s).  Option "fuse_frozen".  (C = 8: ONE irls_reduce launch serves the plain and
                               // the fused workspaces with a per-workspace branch; the two-launch form measured slower at every batch size, round 2)
  int G = 1, Cg = 0;           // channel groups of a wide descriptor (C > 48: C = G x Cg, Cg one of the channel counts the per-point kernels are built for;
                               // types.h PairJob::pitch); G = 1: none
  int reference_reduction = 0; // option "reference_reduction" (validation mode): H, G and the squared norm summed in the reference's index order in f32
                               // (kernels_gn_ref.hip) — the four-kernel chain only, no fused path, no step inside the reduction, no persistent / team kernel
  int step_in_reduce_max = 128; // groups of up to this many pairs run the four-kernel chain as three: the last tile of a workspace in irls_reduce takes
                               // the Gauss-Newton step (gn_step.h gn_last_tile; same sums in the same order).  Worth +2 % at 64 pairs per lane (the
                               // 128-pair shard), nothing at 128, -3 % at 512: the first wave of every tile waits for its write-through store and its
                               // ticket.  Option "step_in_reduce_max_pairs" (0: never)
  // Groups of at most persist_max_ws workspaces (a single pair: sequential addFrame) run every pyramid level in ONE persistent
  // launch (kernels_gn.hip, gn_persistent_kernel) instead of rounds of four kernels per iteration; bit-identical results.
  // Options "persistent" (0 turns it off), "persist_max_ws", "persist_grid" size it.  persistent_failed: a launch gave up at a barrier
  // (workgroups not co-resident) — the context stays on the four-kernel chain from then on.
  int persistent = 1, persist_max_ws = 1, persist_grid = 64;
  int persist_max_points = 32768;     // option "persist_max_points": a level with more template points than this — and the finer levels behind it — takes the
                               // four-kernel chain even for a single pair: the persistent kernel's grid (64 workgroups) wins on the 6 - 26 k points of a
                               // template with non-maximum suppression and loses on dense ones (8 channels: 30.7 against 32.1 us per linearisation at 16 k
                               // points, 59.8 against 49.0 at 38 k, 220 against 134 at 280 k; one channel: equal at 40 k; scripts/persist_crossover.py).
                               // Default 32768 for eight channels, 65536 for one (set at creation).
  // option "dense_candidates_from": chain launches over a level of at least this many template points keep the exact median's candidates
  // in ONE run per workspace (kernels.h, GNLaunch::dense_candidates) — the finish of a 300 k-point level walked 1172 segments, 120 us where the
  // run takes 10.  Not for channel groups (their jobs hold their own counters).
  int dense_candidates_from = 32768;
  long long persist_timeout = 50000000ll;   // ticks of the 100 MHz wall clock a grid barrier waits before it gives up (0.5 s)
  // Batches of 2 .. team_max_pairs pairs run their whole Gauss-Newton stage in ONE launch of the team-persistent kernel
  // (kernels_gn.hip, gn_team_kernel): teams of team_size workgroups, one workgroup per CU, a pair per team at a time.
  // Options "team" (0 turns it off), "team_max_pairs", "team_size" (0 = CUs / pairs), "team_cus" (CUs the grid may claim: tests).
  int team_mode = 1, team_max_pairs = 128, team_full_pairs = 80, team_size_env = 0, num_cus = 0, device_cus = 0;   // (team_full_pairs: up to here whatever the fill)
  int team_join = 2;             // option "team_join": workgroups of a team that has run out of pairs join the teams still at work (kernels_gn_team.hip
                                 // pk_join_team): 0 never, 1 teams on the workgroup's own XCD, 2 any team
  int merge_levels_max_frames = 8;   // option "levels_in_one_launch_max_frames": frame stages of at most this many frames run the levels of the
                                 // bit-planes, selection and template-build kernels in one launch each (frames.hip)
  int small_batch_fused = 1;     // option "small_batch_fused": contexts of a few pairs — job table + poses in one launch, states copied out by pack_records (estimate.hip)
  int nrm_dpp_asm = 4;           // option "normalization_form": the sequential Hartley sums (kernels_frame.hip) as 1: hand-scheduled DPP add chains (170 us for a
                                 // 1241x376 template, 2.28 ms for a dense 640x480 one), 0: the compiler's DPP form (281 us / 3.6 ms), 2: broadcast LDS reads +
                                 // plain adds, no cross-lane traffic and no asm (311 us / 4.0 ms: the compiler's loop does not keep the adds back to back), 3: the
                                 // same reads with the adds as asm blocks of plain v_add_f32 on a wave that does nothing else (170 us / 2.04 ms), 4 (the default):
                                 // 3 for launches of at most 1024 workgroups, 1 for larger ones
  int nrm_side_stream = 1;       // option "normalization_side_stream": frames.hip frames_set_template
  int nrm_defer = 1;             // option "normalization_deferred": ... and the sums of the levels below the coarsest run on under the coarsest level's iterations
  int team_split_max_pairs = 4;  // option "team_split_max_pairs": team batches of up to this many pairs run the coarsest level in a launch of its own, the deferred
                                 // normalisation under it (estimate.hip)
  // the normalisation's streams (created at their first use) and their events: [0] fork, [1] coarsest level done (or all), [2] the levels between the
  // coarsest and the finest done, [3] the finest level done.  Deferred form: the finest level — the long one: 2.3 ms of dependent adds for a dense
  // 640x480 template — has a stream of its own, so that the levels above it are ready, and their Gauss-Newton iterations run, while it is still adding.
  hipStream_t side_stream = nullptr, side_stream2 = nullptr;
  hipStream_t copy_stream = nullptr;      // addFrame's disparity upload (frames.hip upload_disparity); created at its first use, with its event
  hipEvent_t copy_ev = nullptr;
  // run once by the estimation of a group on the context's stream right before its final synchronisation, i.e. with every kernel of the estimate
  // queued (estimate.hip): addFrame's disparity upload
  std::function<int()> before_final_sync;
  int vo_disparity_late = 1;      // option "vo_disparity_late": addFrame uploads a host frame's disparity under the queued estimate (1) or in its data stage (0)
  hipEvent_t side_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t nrm_pending = nullptr;   // non-null: recorded behind the normalisation of the levels between the coarsest and the finest of the template stage just
                                 // queued; whoever reads those levels' (scale, centroid) next makes its stream wait for it (estimate.hip) and clears it
  hipEvent_t nrm_pending_finest = nullptr;   // ... and behind the finest level's
  int team_spares = 1;           // option "team_spares": the team kernel's grid fills the chip, the workgroups beyond the teams join them (growing form only)
  int team_join_from_pairs = 48; // option "team_join_from_pairs": smaller batches run the fixed-size team kernel (A/B: profiles/r05_team_join.txt)
  int team_local_barriers = 1;   // option "team_local_barriers": kernels_gn_team.hip pk_team_barrier mode 2 for teams on one XCD (0: agent-scope fences always)
  std::atomic<uint64_t> team_launches{0};
  std::atomic<uint64_t> team_joins{0};            // workgroups that left a team without pairs and joined one at work (measurement)
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
  // kUploadChunkPairs pairs in pinned memory and copy them on a stream of their own into a device staging area, chunk after chunk in lane
  // order, while the lanes already work on the chunks that have landed (upload_pipeline below).  Option "upload_workers" (0 = off).
  // Fixed by measurement (profiles/r03_host_buffers_tuning.txt, r03_host_timeline.txt): a lane's frame stage takes kUploadGroup chunks at
  // once (16-pair launches are too small to fill the chip), one group per lane, and ONE copy stream shared by the workers — a process has a
  // handful of hardware queues and HIP streams are multiplexed onto them: with a stream per worker the lanes' kernels queued behind other
  // workers' copies and nothing ran until the last chunk had landed.  While a pipeline runs the lanes' small control tables are copied by
  // a kernel from their pinned rows (a hipMemcpyAsync waits its turn behind 45 MB chunk copies in the DMA engines).
  int up_workers = 6;
  std::atomic<bool> ctl_by_kernel{false};
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
  double up_plan[2] = {0.19, 0.50};      // options "upload_plan_first" / "upload_plan_second"; first = 0: two equal groups.  Measured: profiles/r03_host_buffers_plan.txt
  int lazy_template = 1;                 // option "lazy_template_descriptor": the template frames (A) of a pair batch keep census bytes + channel 0 at the NMS
                                         // levels instead of 32-byte records nobody reads (bit-planes, CD3); template_build forms its stencils from the census
  int keep_current_disparity = 0;        // option: pair batches store the disparity of the CURRENT frames (B) too, so that a B slot can be made
                                         // a template later (frames_set_template); off by default: 1.9 GB per 1024-pair step never read
  double up_last_seconds = 0.0;          // wall time the workers of the last call needed for all chunks (measurement)
  size_t up_last_bytes = 0;
  bool counted_live = false;   // this context is in g_live_ctx
  // addFrame: the fraction of good points (should_keyframe's last criterion) is queued right behind the estimation, before the host
  // waits for the pose, instead of in a second round trip; frac_* hold it for fraction_good (same kernels, same count)
  float prefetch_frac_thr = -1.0f;    // >= 0 while bpvo_hip_add_frame runs its estimate
  bool frac_valid = false; int frac_ws = -1; float frac_thr = 0.0f; unsigned frac_cnt = 0; int frac_n = 0;
  double tapcache_max_density = 0.5;   // option: levels with more template points per pixel than this run without the tap cache (batches)
  bool stagger = true;         // option (0: batches run stage by stage over all pairs instead of lane by lane, batch_run_staggered)
  int stagger_min_pairs = 192; // ... for batches of at least this many pairs: below, the lanes' short frame stages are not worth their serialisation
                               // (128 pairs on three lanes: +4.5 % without, +1.4 % with).  Option "stagger_min_pairs"
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

namespace bpvo_hip_host {

#define HIP_CK(ctx_, expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if(e_ != hipSuccess) {                                                                  \
      (ctx_)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
      return BPVO_ERR_DEVICE;                                                               \
    }                                                                                       \
  } while(0)

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

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

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

// HIP events around a kernel class on a lane's stream (measurement; resolved by resolve_events)
hipEvent_t take_event(Lane* ln);
struct ScopedTimer {
  Lane* ln;
  EventPair ep;
  bool on;
  hipStream_t st;      // the stream the timed launches go to: the lane's own, or another one that is joined into it before the lane is synchronised (the side stream)
  ScopedTimer(bpvo_hip_ctx* c_, int kc, double units, Lane* lane = nullptr, bool sampled = true, hipStream_t stream = nullptr)
      : ln(lane ? lane : &c_->lanes[0]), on(c_->profiling && sampled), st(stream ? stream : (lane ? lane : &c_->lanes[0])->stream)
  {
    if(!on) return;
    ep.kc = kc; ep.units = units;
    ep.a = take_event(ln); ep.b = take_event(ln);
    (void) hipEventRecord(ep.a, st);
  }
  ~ScopedTimer()
  {
    if(!on) return;
    (void) hipEventRecord(ep.b, st);
    ln->ev_pending.push_back(ep);
  }
};

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
  // a template stage whose estimation follows on the same stream inside the same call (bpvo_hip_batch_run on one lane):
  bool defer_finest_nrm = false;       // the normalisation of every level but the coarsest stays on the side streams: ctx->nrm_pending / nrm_pending_finest, joined by the estimation
  bool no_final_sync = false;          // no host synchronisation at the end of the stage
  // frames_set_data from host buffers (addFrame): the disparity — which nothing of the data stage or of the estimate reads — is not copied by the stage;
  // the caller uploads it (upload_disparity, frames.hip) once the estimate is queued, so that the copy's hold on the host lies under the GPU's work
  bool skip_disparity_upload = false;
};
#define FR_CK(c_, fr_, expr)                                                                \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if(e_ != hipSuccess) {                                                                  \
      ((fr_).own_thread ? (fr_).ln->err : (c_)->err) = std::string(#expr) + ": " + hipGetErrorString(e_); \
      return BPVO_ERR_DEVICE;                                                               \
    }                                                                                       \
  } while(0)

// ---- shared between the translation units (definitions: see the table at the top)
int fail(bpvo_hip_ctx* c, int code, const char* msg);
void gaussian_kernel5(double sigma, float k[3]);
void gaussian_taps(int n, double sigma, GaussTaps* g);
int imsmooth_taps(float sigma);
int auto_gauss_taps_f32(float sigma);
void carve_frame_data(bpvo_hip_ctx* c, FrameSlot& f, unsigned char* base, size_t* total);
void carve_frame_tmpl(bpvo_hip_ctx* c, FrameSlot& f, unsigned char* base, size_t* total);
int ensure_template_storage(bpvo_hip_ctx* c, FrameSlot& f);
FrameJob make_frame_job(bpvo_hip_ctx* c, FrameSlot& f, int l);
PairJob make_pair_job(bpvo_hip_ctx* c, int ws, int ref, int cur, int l);
void resolve_events(bpvo_hip_ctx* c);
FrameRun ctx_run(bpvo_hip_ctx* c);
int upload_frame_jobs(bpvo_hip_ctx* c, int first, int stride, int count, const FrameRun& fr, int which, const FrameJob** tab);
int frames_set_data(bpvo_hip_ctx* c, int first, int stride, int count, const uint8_t* images, const float* disps, bool on_device, const FrameRun& fr,
                    int skip_odd_disp = 0);
int frames_set_data(bpvo_hip_ctx* c, int first, int stride, int count, const uint8_t* images, const float* disps, bool on_device, int skip_odd_disp = 0);
int frames_set_template(bpvo_hip_ctx* c, int first, int stride, int count, const FrameRun& fr);
int upload_disparity(bpvo_hip_ctx* c, int slot, const float* disparity);
int frames_set_template(bpvo_hip_ctx* c, int first, int stride, int count);
bool team_serves(const bpvo_hip_ctx* c, int n);
PairJob group_pair_job(const bpvo_hip_ctx* c, const PairJob& whole, int k);
inline int job_tables(const bpvo_hip_ctx* c) { return c->G > 1 ? 1 + c->G : 1; }      // job tables of a lane: the whole jobs, then one table per channel group
// The per-point kernels of a linearisation (warp + residual, the reduction): once — or, for a wide descriptor, once per channel group, on the
// group's job table (`stride` jobs behind the table before it) with the group's channel count.
template <class F>
inline void for_each_group(const bpvo_hip_ctx* c, const bpvo_hip::GNLaunch& g, size_t stride, F&& f)
{
  if(c->G <= 1) { f(g); return; }
  for(int k = 0; k < c->G; ++k) {
    bpvo_hip::GNLaunch gg = g;
    gg.jobs = g.jobs + (size_t) (1 + k) * stride;
    gg.C = c->Cg;
    f(gg);
  }
}
// the exact median of a wide descriptor: the WHOLE job (it walks the bracket segments of every group), instantiated for the group's channel count
// (the size of a segment)
// entry of Workspace::med_blk (uint4 units, a multiple of 8 = one 128-byte line) where the dense form's totals begin
inline size_t med_totals_at(const bpvo_hip_ctx* c) { return (((size_t) ((c->cap_max + bpvo_hip::kChunkPoints - 1) / bpvo_hip::kChunkPoints) * (size_t) c->G) + 7) / 8 * 8; }
// every deferred normalisation joins `s` (errors aside, the estimation has done that level by level: estimate.hip)
inline hipError_t join_pending_normalization(bpvo_hip_ctx* c, hipStream_t s)
{
  hipError_t e = hipSuccess;
  if(c->nrm_pending) { e = hipStreamWaitEvent(s, c->nrm_pending, 0); c->nrm_pending = nullptr; }
  if(c->nrm_pending_finest) { const hipError_t e2 = hipStreamWaitEvent(s, c->nrm_pending_finest, 0); c->nrm_pending_finest = nullptr; if(e == hipSuccess) e = e2; }
  return e;
}
// can a template of this context have a level of more points than the persistent kernels are given (persist_max_points)?  Only a level without
// non-maximum suppression selects (nearly) every pixel; known from the parameters alone, before any template exists
inline bool templates_may_be_dense(const bpvo_hip_ctx* c)
{
  for(int l = c->params.maxTestLevel; l < c->L; ++l)
    if(c->geom[l].nms_radius <= 0 && c->geom[l].npix > (size_t) std::max(0, c->persist_max_points)) return true;
  return false;
}
// will the estimate of ONE pair be a handful of launches — the persistent kernel, a launch per pyramid level — queued in one go?  (Then the host is free
// while the GPU works, and addFrame uploads its frame's disparity in that time: vo.hip.  On the chain the host paces the rounds, and the upload's hold on
// it costs more behind the last round than in front of the first: conf/tsukuba.cfg's parameters 2.9 -> 3.2 ms per frame.)
inline bool single_pair_is_queued_at_once(const bpvo_hip_ctx* c)
{
  return c->persistent && !c->persistent_failed.load() && !c->profile_all && !c->reference_reduction && (c->C == 8 || c->C == 1) &&
         c->params.interp == BPVO_INTERP_LINEAR && !c->fast_warp && !templates_may_be_dense(c);
}
inline int dense_candidates(const bpvo_hip_ctx* c, int max_points) { return (c->G == 1 && max_points >= c->dense_candidates_from) ? 1 : 0; }
inline bpvo_hip::GNLaunch median_launch(const bpvo_hip_ctx* c, bpvo_hip::GNLaunch g) { if(c->G > 1) g.C = c->Cg; return g; }
int estimate_group(bpvo_hip_ctx* c, Lane* ln, int n, const int* wss, const int* refs, const int* curs, const float* T_init, float* poses,
                   bpvo_hip_stats* stats, float* d_records_out, bool allow_persistent);
int estimate_batch(bpvo_hip_ctx* c, int n, const int* wss, const int* refs, const int* curs, const float* T_init, float* poses, bpvo_hip_stats* stats);
void detile_to_channel_major(const float* src, int n, int C, int E, int V, float* out);
size_t tiled_floats(int n, int floats_per_point);
int refresh_counters(bpvo_hip_ctx* c);
int upload_single_job(bpvo_hip_ctx* c, int ws, int ref, int cur, int level);
void trajectory_push(bpvo_hip_ctx* c, const M44& T);
int ensure_residuals(bpvo_hip_ctx* c, int ws);
int fraction_good(bpvo_hip_ctx* c, int ws, float thr, float* frac);
int get_weights_host(bpvo_hip_ctx* c, int ws, std::vector<float>& w_cm, int* n_out);
int check_template_not_empty(bpvo_hip_ctx* c, int ref_slot);
int ensure_lanes(bpvo_hip_ctx* c, int n);
int lanes_for(bpvo_hip_ctx* c, int n, int cap);
int ensure_dense_descriptor(bpvo_hip_ctx* c, int slot);      // a slot with lazy levels gets its full records (accessors, a template frame used as current)
int set_option(bpvo_hip_ctx* c, const std::string& key, double v);
int apply_options_string(bpvo_hip_ctx* c, const char* str);
struct OptionDef { const char* key; double lo, hi; std::function<double(bpvo_hip_ctx*)> get; std::function<int(bpvo_hip_ctx*, double)> set; };
const std::vector<OptionDef>& option_table();

}  // namespace bpvo_hip_host

#define CHECK_CTX(c) if(!(c)) return BPVO_ERR_INVALID_ARG
#define CHECK_SLOT(c, s) if((s) < 0 || (s) >= (c)->n_frames) return fail(c, BPVO_ERR_INVALID_ARG, "bad frame slot")
#define CHECK_WS(c, w) if((w) < 0 || (w) >= (c)->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "bad workspace")
#define CHECK_LEVEL(c, l) if((l) < (c)->params.maxTestLevel || (l) >= (c)->L) return fail(c, BPVO_ERR_INVALID_ARG, "bad level")
