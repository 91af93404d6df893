// Device-resident data model of libbpvo_hip.  Plain structs shared by the host driver and the kernels.
#pragma once

#include <stdint.h>

#include "../../include/bpvo_hip/c_api.h"
#include "device_math.h"

namespace bpvo_hip {

constexpr int kMaxLevels = BPVO_HIP_MAX_LEVELS;
constexpr int kHistBins1 = 2048;   // radix-select pass 1: |r| bits [30:20]
constexpr int kReduceVals = 28;    // 21 upper-triangular H + 6 G + sum w r^2
constexpr int kPartialStride = 32;
constexpr int kWsCounters = 12;
constexpr int kRecordFloats = 32;  // packed per-pair result record (see bpvo_hip_batch_result_records_device)
constexpr int kTraceFloats = BPVO_HIP_TRACE_FLOATS;   // one record per linearisation (c_api.h)
constexpr int kMaxGroups = 16;     // channel groups of a wide descriptor (more than 48 channels): LATCH with 64 bytes is 16 groups of 32

// TILED per-point layout.  Template pixels, Jacobians and residuals are records of W floats per point
// (W = C, 6*C, C).  A record is cut into V-float vector pieces (V = 4 for C = 8; 1 or 2 for C = 1) and points are grouped
// in tiles of kTile; inside a tile the layout is piece-major:
//     vec[((i / kTile) * PIECES + piece) * kTile + (i % kTile)]
// so that lane l of a wave reading piece k of its point touches one contiguous 64*V*4-byte segment: every wave-level
// load/store of these arrays is fully coalesced (a plain point-major record layout makes each 16-byte load of a
// wave hit 64 different cache lines).  Tile size, measured on the 1024-pair bench (warp_residual, algorithmic TB/s):
// 64: 5.25, 256: 5.27, 1024: 5.40, 2048: 5.66, 4096: 5.66, 8192: 5.27, 16384: 5.26, 65536: 5.13 — HBM likes long runs
// per stream (a piece of a 2048-point tile is a 32 KiB run) as long as the pieces of one point stay within ~0.5 MiB.
#ifndef BPVO_TILE_VALUE
#define BPVO_TILE_VALUE 2048
#endif
constexpr int kTile = BPVO_TILE_VALUE;   // points per tile (a power of two, multiple of the wavefront size)
template <int PIECES>
__host__ __device__ inline size_t tile_index(int i, int piece)
{
  return ((size_t) (i / kTile) * PIECES + piece) * kTile + (size_t) (i % kTile);
}


// One row of the template Jacobian (reference: RigidBodyWarp::computeJacobian, bpvo/rigid_body_warp.cc:60-315; same formulas
// as the scalar jacobian() of bpvo/rigid_body_warp.h:94-106).  The reference's SSE code divides with
// div_ps(a, b) = _mm_mul_ps(a, _mm_rcp_ps(b)) (rigid_body_warp.cc:47-58): a multiply by an APPROXIMATE (12-bit, vendor
// specific) reciprocal.  Here the operation structure is kept — a * (1/b) — with the correctly rounded reciprocal
// 1.0f / b in place of _mm_rcp_ps: deterministic, and only three reciprocals per point (1/z, 1/z^2,
// 1/(z*s)).  Ix, Iy are the channel gradients already multiplied by fx, fy.
// Deterministic IEEE arithmetic: evaluating a row again in irls_reduce gives bit-identical values to evaluating it once
// at template-build time, so the 24-byte rows are never stored.
struct JacPoint { float x, y, rz, rz2, rzs, xc1, yc2, zc3, s_i; };
__host__ __device__ inline JacPoint jac_point(float x, float y, float z, const float* nrm /* s, c1, c2, c3 */)
{
  JacPoint p;
  p.x = x; p.y = y;
  p.rz = 1.0f / z;
  p.rz2 = 1.0f / (z * z);
  p.rzs = 1.0f / (z * nrm[0]);
  p.xc1 = x - nrm[1]; p.yc2 = y - nrm[2]; p.zc3 = z - nrm[3];
  p.s_i = (float) (1.0 / (double) nrm[0]);
  return p;
}
__host__ __device__ inline void jac_row(const JacPoint& p, float Ix, float Iy, float* J)
{
  const float xIx_yIy = p.x * Ix + p.y * Iy;
  J[0] = (-((Iy * p.zc3) * p.rz)) - ((xIx_yIy * p.yc2) * p.rz2);     // rigid_body_warp.cc:80-96
  J[1] = ((Ix * p.zc3) * p.rz) + ((xIx_yIy * p.xc1) * p.rz2);        // :131-137
  J[2] = ((Iy * p.xc1) - (Ix * p.yc2)) * p.rz;                       // :173-178
  J[3] = Ix * p.rzs;                                                 // :205-206
  J[4] = Iy * p.rzs;                                                 // :232-233
  J[5] = -((p.s_i * xIx_yIy) * p.rz2);                               // :295-299
}

// The same for DisparitySpaceWarp (reference: DisparitySpaceWarp::jacobian, bpvo/disparity_space_warp.h:40-64 — scalar f32,
// C++ evaluation order; the point is (x - cx, y - cy, d, 1), Ix / Iy are the RAW channel gradients).  fx_i, fy_i, b_i are
// the constructor's 1.0f / fx, 1.0f / fy, 1.0f / b (bpvo/disparity_space_warp.cc:31-33).
__host__ __device__ inline void dspace_jac_row(float x, float y, float d, float fx, float fy, float fx_i, float fy_i, float b_i,
                                               float Ix, float Iy, float* J)
{
  const float t2 = x * Ix, t3 = y * Iy, t4 = t2 + t3;
  J[0] = ((-Iy) * fy) - ((t4 * fy_i) * y);
  J[1] = (Ix * fx) + ((t4 * fx_i) * x);
  J[2] = (((Iy * fy) * fx_i) * x) - (((Ix * fx) * fy_i) * y);
  J[3] = (Ix * d) * b_i;
  J[4] = (((Iy * d) * fy) * fx_i) * b_i;
  J[5] = (((-d) * t4) * fx_i) * b_i;
}

// phases of the device-side PoseEstimatorBase::run state machine (gn_step kernel)
enum { PHASE_FIRST = 0, PHASE_LOOP = 1, PHASE_DONE = 2 };

// PoseEstimatorBase + PoseEstimatorData_ + AutoScaleEstimator state of ONE estimation workspace
// (reference: bpvo/pose_estimator_base.h:67-151,190-206; bpvo/mestimator.h:62-88), device resident.
struct GNState {
  float T[16];           // data.T, row-major
  float H[36];
  float G[6];
  float dp[6];
  float f_norm, f_norm_prev, dp_norm_prev, g_tol, g_norm;
  float scale, delta_scale;                 // AutoScaleEstimator::_scale, _delta_scale
  int   num_fun_evals, num_iterations, status, phase, active, has_converged;
  uint32_t n_valid;                          // valid points of the last linearisation
  int   level;
  // bracketed median selection (kernels_gn.hip K7): bracket [lo_key, hi_key) on the bit pattern of |r| around the
  // previous median of the level, per-block counters filled by warp_residual (bracket_block) and consumed by median_finish_kernel
  float last_median;
  uint32_t lo_key, hi_key;
  int   median_valid;
  bpvo_hip_stats stats[kMaxLevels];
  float T_out[16];                           // pose handed back (T in/out of run())
  // Fused path for frozen scales (irls_reduce recomputes the residuals, warp_residual skips the workspace): the
  // residual / valid buffers are then NOT those of the last linearisation.  T_lin is the pose of the last linearisation
  // and r_stale says that the buffers have to be refreshed from it before anything reads them (done on demand).
  float T_lin[16];
  int   r_stale;
  int   trace_n;                             // records written to PairJob::trace since set_pose (bpvo_hip_estimate_pose_trace)
};

// PoseEstimatorParameters as the device-side state machine reads them (gn_step.h)
struct GNParams { int max_iterations, max_fun_evals; float p_tol, f_tol, g_tol; };

// Streaming (non-temporal) 16-byte accesses for data that is read or written exactly once per launch and is far larger
// than the caches: keeps such streams from evicting each other in L2 (measured +15 % on the access pattern of
// warp_residual, scripts/micro/streams.hip).
#if defined(__HIPCC__)
typedef float bpvo_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_stream(const float4* p)
{
  const bpvo_v4f v = __builtin_nontemporal_load(reinterpret_cast<const bpvo_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_stream(float4* p, const float4& a)
{
  bpvo_v4f v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w;
  __builtin_nontemporal_store(v, reinterpret_cast<bpvo_v4f*>(p));
}
// ... and the choice as a template parameter: launches that are latency-bound on a working set that fits the L2s (the persistent
// single-pair kernel: 6 MB at the finest level of a 1241x376 pair) want the lines kept — non-temporal accesses there cost 8 % of an
// estimatePose (profiles/r02_persistent_phases.txt)
template <bool NT>
__device__ __forceinline__ float4 load_v4(const float4* p) { if constexpr(NT) return load_stream(p); else return *p; }
template <bool NT>
__device__ __forceinline__ void store_v4(float4* p, const float4& a) { if constexpr(NT) store_stream(p, a); else *p = a; }
#endif

// Pointer members of the job tables.  A pointer a kernel loads from memory is a GENERIC pointer to the compiler (it cannot know the
// address space), and every access through it becomes a flat_load / flat_store: 64-bit address arithmetic in vector registers, a wait on
// both memory counters, the LDS aperture check.  Declared as global-address-space pointers in the device pass (same eight bytes, same
// layout as the host's plain pointer) the accesses are global_load / global_store — the whole Gauss-Newton and frame code was flat.
template <class T>
struct GPtr {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef T __attribute__((address_space(1)))* raw_t;
#else
  typedef T* raw_t;
#endif
  raw_t p;
  __host__ __device__ operator T*() const { return (T*) p; }
  __host__ __device__ T* get() const { return (T*) p; }
  __host__ __device__ T* operator->() const { return (T*) p; }
  __host__ __device__ GPtr& operator=(T* q) { p = (raw_t) q; return *this; }
};

static_assert(sizeof(GPtr<float>) == sizeof(float*) && alignof(GPtr<float>) == alignof(float*), "GPtr must be layout-compatible with a plain pointer: the host fills the tables the device reads");

// everything a kernel needs to know about one (workspace, level) linearisation
struct PairJob {
  // template (reference frame) at this level
  GPtr<const float4> pts;      // [N] (X,Y,Z,1)
  GPtr<const float> pix;      // [N][C] tiled (see tile_index)
  GPtr<const float> grad;     // [N][2][C] tiled: (fx*Ix[c]), (fy*Iy[c]) — the 1x6 Jacobians are recomputed from these
  GPtr<const float> nrm;      // (s, c1, c2, c3) Hartley normalisation of the level
  int           n;        // number of points (multiple of 16)
  // current frame descriptor at this level, pixel-interleaved [rows*cols][C]
  GPtr<const float> desc;
  int           rows, cols;
  float         K[9];     // level intrinsics (K * 0.5^l, K(2,2) = 1)
  float         b;        // level baseline (b * 2^l)
  int           dspace;   // 1: DisparitySpaceWarp (BPVO_WARP_DISPARITY_SPACE_F32): pts = (x - cx, y - cy, d, 1), grad = raw (Ix, Iy)
  // workspace
  GPtr<float> r;        // [N][C] residuals, tiled
  GPtr<uint8_t> valid;    // [N]
  GPtr<uint32_t> tapkey;   // [N] (yi << 16 | xi) of the footprint held in tapcache, 0xffffffff = none (C = 8 and C = 1)
  int           tapcache_on; // 0: this (workspace, level) gathers its taps straight from the descriptor (dense levels of a batch); the keys are
                           // still reset at the start of the level, so that a later call that does use the cache finds no stale entry
  GPtr<float> tapcache; // C = 8: [N][32] tiled, the 4 taps x 8 channels of the footprint last gathered for the point; C = 1: [N] float4
  GPtr<uint32_t> cand;     // [N*C] candidate keys of the bracketed median selection, one 256*C segment per block
  GPtr<uint32_t> med_blk;  // [ceil(N/256)][4] per-block {below, inside, valid points, tap-cache hits} of the bracket pass
  GPtr<float> partials; // [nblocks][kPartialStride]
  GPtr<unsigned long long> cnt; // [kWsCounters] per-workspace measurement counters: [0] points linearised, [1] linearisations,
                           // [2] bracketed / [3] full median selections, [4] points processed by warp_residual (the rest
                           // went through the fused path of irls_reduce), [5] tap-cache hits / [6] lookups (= valid points),
                           // [7] / [8] the same over the first 8 linearisations of a level, [10] points linearised through the
                           // fused path; written by one thread each: no atomics
  GPtr<GNState> st;
  GPtr<unsigned> ticket;   // tiles of the running irls_reduce launch that have stored their partial (the last one takes the step; back to 0 by then)
  // per-linearisation trace of the Gauss-Newton run (bpvo_hip_estimate_pose_trace; null otherwise): trace_cap records of
  // kTraceFloats floats, written by the thread that runs the serial step — the table the reference prints per iteration at
  // kIteration verbosity (bpvo/pose_estimator_base.h:231-247), with the pose, H, G and dp added
  GPtr<float> trace;
  int           trace_cap;
  // Descriptors of more than 48 channels run their per-point kernels (warp + residual, reduction) once per channel GROUP: a group's job is the
  // whole job with desc / pix / grad / r advanced to the group's first channel and cand / med_blk / partials to the group's segments, and
  // `pitch` — floats between the records of consecutive pixels / points (2 x pitch between gradient records) — still the descriptor's channel
  // count.  The whole job (what the median, the step and every per-workspace kernel take) has n_groups = the number of groups: its median walks
  // n_groups x chunks bracket segments, its step sums n_groups x tiles partials.  pitch = C, n_groups = 1 everywhere else.
  int           pitch;
  int           n_groups;
  int           med_tot;   // entry of med_blk (uint4 units) that holds the four totals of the dense bracket form (bracket_chunk<C, true>): behind every chunk's counters
};

// selection / template-build job for one (frame, level)
constexpr int kDfPlanes = 7;
struct FrameJob {
  GPtr<const uint8_t> img;       // level image u8
  GPtr<uint8_t> cen;       // census scratch u8 (BitPlanes)
  GPtr<float> desc;      // [rows*cols][C]
  GPtr<float> ch0;       // [rows*cols] copy of descriptor channel 0 (C = 8): the saliency map needs little else (Q7), and a
                            // compact plane spares it a strided pass over the 32-byte records
  GPtr<float> scratch;   // descriptor fields: kDfPlanes work planes of the level-0 size
  GPtr<float> sal;       // [rows*cols]
  GPtr<uint8_t> flag;      // [rows*cols] candidate flags (NMS radius > 1)
  GPtr<unsigned long long> words;   // [rows * ceil(cols / 64)] candidate bits, the same storage (NMS radius <= 1)
  GPtr<int> blk_count; // per 256-pixel chunk / per word: count, then exclusive offset
  GPtr<int> n_out;     // device: number of points kept (multiple of 16)
  GPtr<const float> disp;      // full-resolution disparity
  GPtr<float4> pts;
  GPtr<int> inds;
  GPtr<float> pix;
  GPtr<float> grad;
  GPtr<float> nrm;       // (s, c1, c2, c3)
  int            rows, cols, level, disp_cols;
  int            cap;       // capacity of pts/inds
  int            nms_radius;   // <= 0: NMS off for this level
  float          K[9];
  float          b;
  int            dspace;    // 1: DisparitySpaceWarp points / raw gradients (see PairJob)
  int            lazy;      // 1 (C = 8 bit-planes, template frames of a pair batch at the NMS levels): the level's 32-byte records are NOT
                            // stored — `cen` holds the census bytes and `ch0` channel 0 (all the selection needs); template_build
                            // forms the records of its stencils from the census bytes (kernels_frame.hip)
};

}  // namespace bpvo_hip
