// Persistent Gauss-Newton kernels (gfx950): a whole pyramid level of ONE pair (gn_persistent_kernel) or the whole Gauss-Newton stage
// of a small BATCH (gn_team_kernel: a team of workgroups per pair) in one launch.  The phases call the device functions of the
// four-kernel chain (gn_warp.h, gn_median.h, gn_irls.h, gn_step.h) with the chain's chunk and tile indices: bit-identical results.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "gn_common.h"
#include "gn_warp.h"
#include "gn_median.h"
#include "gn_irls.h"
#include "gn_step.h"

namespace bpvo_hip {

// ------------------------------------------------------------------------------------------------------------------
// Persistent Gauss-Newton kernel for SMALL groups (a single pair: sequential addFrame; up to kPersistMaxWs pairs): a whole
// pyramid level — every linearisation, median, reduction, solve and pose update until the last workspace of the group has
// finished — in ONE launch.  The four-kernel chain spends a single pair's iteration on four dependent launches of 5 - 9 us
// each, every one of which re-reads the job and the state from HBM; here
//   * the state of every workspace lives in LDS for the whole level, one copy per workgroup, all copies identical: the serial
//     steps (robust scale, 6x6 solve, pose update, convergence tests) are executed REDUNDANTLY by every workgroup on its own
//     copy — deterministic arithmetic on identical inputs — so nothing has to be broadcast and two of the four
//     synchronisation points of an iteration disappear;
//   * the two that remain (all residual chunks before the median, all tile partials before the solve) are grid barriers: one
//     agent-scope release + arrive + poll + acquire per workgroup (guide: "barrier-counter"), a frozen robust scale needs only
//     the second;
//   * a 512-thread workgroup works as two 256-thread chunks of warp_residual / tiles of irls_reduce side by side, calling the
//     very device functions of the four kernels (warp_point, bracket_chunk, median_block, irls_tile, gn_sum_partials,
//     gn_serial_step) with the same chunk / tile indices, so every value — residuals, median, partials, their f64 sum — is
//     bit-identical to the chain's.
// Residency: the grid (at most kPersistMaxGrid workgroups, one per CU: 123 KB of LDS) is far below the chip's 256 CUs and the
// launcher checks the occupancy query; should the workgroups still not become co-resident (another process holding the CUs), the
// poll of a barrier gives up after `timeout` ticks of the 100 MHz wall clock, raises ctl[1] and every workgroup leaves WITHOUT
// writing the states back — the host then reruns the group through the four-kernel chain (bpvo_hip.hip).  The GPU cannot hang.
// 512 threads: two waves per SIMD, i.e. 256 VGPRs — a 1024-thread workgroup leaves 128, and the fused irls_tile (136 as a kernel)
// then spills inside its point loop (measured: 23 us per iteration for that phase instead of 8)
constexpr int PK_THREADS = 512;
constexpr int PK_VB = PK_THREADS / 256;      // 256-thread chunks / tiles per workgroup
static_assert(K6_BLOCK == 256 && GN_BLOCK == 256, "the persistent kernel's virtual blocks are 256 threads");
static_assert(PK_THREADS / 64 >= kPersistMaxWs, "pk_step_phase: one wave per workspace");



// The phases are separate NON-inlined functions: inlined into one body the compiler hoists every workspace's addresses and job
// fields across all of them and spills hundreds of bytes per lane; as functions each gets its own register allocation.  Their
// LDS is declared at namespace scope for that reason.
constexpr int kStateWords = (int) (sizeof(GNState) / sizeof(uint32_t));
__shared__ uint32_t pk_state[kPersistMaxWs][kStateWords];
__shared__ float pk_sum[kPersistMaxWs][kPartialStride];
__shared__ float pk_nrm[kPersistMaxWs][8];
__shared__ SolveScratch pk_scratch[kPersistMaxWs];
__shared__ BracketLds pk_br[PK_VB];
__shared__ IrlsPartLds pk_part[PK_VB];
__shared__ int pk_ok;
// this workgroup's place in its group: member index and number of workgroups that share a pair's chunks / tiles and meet at its
// barriers (gn_persistent_kernel: blockIdx.x of gridDim.x; gn_team_kernel: see team_geometry)
__shared__ int pk_member, pk_nwg;
__device__ __forceinline__ GNState* pk_st(int ws) { return reinterpret_cast<GNState*>(pk_state[ws]); }

// warp_residual (+ bracket step) of workspace ws: chunk c goes to workgroup c % nwg, virtual block (c / nwg) % PK_VB
template <int C>
__device__ __attribute__((noinline)) void pk_warp_phase(const PairJob* __restrict__ jobs, int ws, bool stats_wg)
{
  const int tid = threadIdx.x, vsub = tid >> 8, vtid = tid & 255;
  const int nwg = pk_nwg;
  const GNState* st = pk_st(ws);
  const PairJob& j = jobs[ws];
  const int n = j.n;
  const int nchunks = (n + K6_BLOCK - 1) / K6_BLOCK;
  float P[12];
  projection_matrix(j, st->T, P);
  const bool bracket = (st->delta_scale > 1e-6f) && st->median_valid;
  const unsigned lo_key = st->lo_key, hi_key = st->hi_key;
  if(stats_wg && tid == 0) j.cnt[4] += (unsigned long long) n;
  for(int base = 0; base < nchunks; base += nwg * PK_VB) {
    const int chunk = base + vsub * nwg + pk_member;
    const bool has = chunk < nchunks;
    const int i_raw = chunk * K6_BLOCK + vtid;
    const bool in_block = has && i_raw < n;
    const int i = in_block ? i_raw : n - 1;
    float res[C];
    bool hit;
    const bool valid = warp_point<C, false, false, false>(j, P, i, in_block, res, hit);      // cached (not streaming) accesses
    if(in_block) {
      j.valid[i] = valid ? 1 : 0;
      if constexpr(C == 8) {
        float4* o = reinterpret_cast<float4*>(j.r.get());
        o[tile_index<2>(i, 0)] = make_float4(res[0], res[1], res[2], res[3]);
        o[tile_index<2>(i, 1)] = make_float4(res[4], res[5], res[6], res[7]);
      } else {
#pragma unroll
        for(int c = 0; c < C; ++c) j.r[(size_t) i * C + c] = res[c];
      }
    }
    if(bracket) bracket_chunk<C>(j, lo_key, hi_key, valid && in_block, hit && valid && in_block, res, (unsigned) chunk, vtid >> 6, pk_br[vsub], has);
  }
}

// The same phase for the TEAM kernel (C = 8), where all CUs of the chip run teams at once and a memory round trip takes 2 - 3 us instead
// of under one: with one point per thread the phase is a chain of dependent round trips (point -> projection -> key -> taps) at 8 waves
// per CU, and it stretched from 28 to 80 - 110 us per iteration at the finest level of a 128-pair batch
// (profiles/r03_team_phases_under_load_before.txt).  Here a thread carries U points — one from each of U chunks — through the phase in
// stages: everything whose address depends on the point index only (point, tap-cache key, the eight cached tap vectors, template pixels) is
// requested for all U points at once, then the U projections, then the gathers of the misses (at dense levels, which run without the
// cache: of all points) for all U at once.  Same expressions as warp_point, operation for operation: same bits.  The cached taps are loaded
// speculatively, as irls_tile_lat does (3 % of them are discarded at the sparse levels).
struct WarpStage {
  float4 X, t[8], px[2];
  double xf, yf;
  unsigned key;
  int i, xi, yi, chunk;
  bool has, in_block, valid, hit;
};
__shared__ BracketLds pk_br_u[PK_VB][4];
template <int U, bool NT>
__device__ __attribute__((noinline)) void pk_warp_phase_staged(const PairJob* __restrict__ jobs, int ws, bool stats_wg)
{
  static_assert(U >= 1 && U <= 4, "pk_br_u");
  const int tid = threadIdx.x, vsub = tid >> 8, vtid = tid & 255;
  const int nwg = pk_nwg;
  const GNState* st = pk_st(ws);
  const PairJob& j = jobs[ws];
  const int n = j.n, W = j.cols, R = j.rows;
  const int nchunks = (n + K6_BLOCK - 1) / K6_BLOCK;
  float P[12];
  projection_matrix(j, st->T, P);
  const bool bracket = (st->delta_scale > 1e-6f) && st->median_valid;
  const unsigned lo_key = st->lo_key, hi_key = st->hi_key;
  const bool cached = j.tapcache_on != 0;
  float4* const tc = reinterpret_cast<float4*>(j.tapcache.get());
  const float4* const p0 = reinterpret_cast<const float4*>(j.pix.get());
  if(stats_wg && tid == 0) j.cnt[4] += (unsigned long long) n;
  for(int base = 0; base < nchunks; base += nwg * PK_VB * U) {
    WarpStage s[U];
    // stage A: everything addressed by the point index
#pragma unroll
    for(int u = 0; u < U; ++u) {
      s[u].chunk = base + (u * PK_VB + vsub) * nwg + pk_member;
      s[u].has = s[u].chunk < nchunks;
      const int i_raw = s[u].chunk * K6_BLOCK + vtid;
      s[u].in_block = s[u].has && i_raw < n;
      const int i = s[u].in_block ? i_raw : n - 1;
      s[u].i = i;
      s[u].X = load_v4<NT>(j.pts + i);
      s[u].px[0] = load_v4<NT>(p0 + tile_index<2>(i, 0));
      s[u].px[1] = load_v4<NT>(p0 + tile_index<2>(i, 1));
      if(cached) {
        s[u].key = j.tapkey[i];
#pragma unroll
        for(int k = 0; k < 8; ++k) s[u].t[k] = load_v4<NT>(tc + tile_index<8>(i, k));
      } else {
        s[u].key = 0xffffffffu;
      }
    }
    // stage B: projection, validity (warp_point), and the gathers of the footprints the cache does not hold
#pragma unroll
    for(int u = 0; u < U; ++u) {
      const float4 X = s[u].X;
      const double X0 = (double) X.x, X1 = (double) X.y, X2 = (double) X.z, X3 = (double) X.w;
      double uu[3];
#pragma unroll
      for(int r = 0; r < 3; ++r) {
        double a = (double) P[r * 4 + 0] * X0;
        a += (double) P[r * 4 + 1] * X1;
        a += (double) P[r * 4 + 2] * X2;
        a += (double) P[r * 4 + 3] * X3;
        uu[r] = a;
      }
      const double zi = 1.0 / uu[2];
      const double x = zi * uu[0], y = zi * uu[1];
      const bool in_range = (x > -2147483648.0) && (x < 2147483648.0) && (y > -2147483648.0) && (y < 2147483648.0);
      int xi = 0, yi = 0;
      if(in_range) {
        xi = (int) x; xi -= (xi > x);
        yi = (int) y; yi -= (yi > y);
      }
      s[u].valid = in_range && xi >= 0 && xi < W - 1 && yi >= 0 && yi < R - 1;
      s[u].xi = xi; s[u].yi = yi;
      s[u].xf = x - (double) xi; s[u].yf = y - (double) yi;
      const unsigned key = ((unsigned) yi << 16) | (unsigned) xi;
      s[u].hit = s[u].valid && cached && s[u].key == key;
      s[u].key = key;
      if(s[u].valid && !s[u].hit) {
        const float4* q0 = reinterpret_cast<const float4*>(j.desc + ((size_t) yi * W + xi) * 8);
        const float4* q1 = q0 + (size_t) W * 2;
        s[u].t[0] = q0[0]; s[u].t[1] = q0[1]; s[u].t[2] = q0[2]; s[u].t[3] = q0[3];
        s[u].t[4] = q1[0]; s[u].t[5] = q1[1]; s[u].t[6] = q1[2]; s[u].t[7] = q1[3];
      }
    }
    // stage C: residuals, stores, cache update, bracket step
#pragma unroll
    for(int u = 0; u < U; ++u) {
      const int i = s[u].i;
      float res[8];
      if(s[u].valid) {
        const double xf = s[u].xf, yf = s[u].yf, wx = 1.0 - xf, wy = 1.0 - yf;
        const float4* t = s[u].t;
        // pieces 0, 1: I00 of channels 0-3 / 4-7; 2, 3: I01; 4, 5: I10; 6, 7: I11 (warp_point)
        const float i00[8] = {t[0].x, t[0].y, t[0].z, t[0].w, t[1].x, t[1].y, t[1].z, t[1].w};
        const float i01[8] = {t[2].x, t[2].y, t[2].z, t[2].w, t[3].x, t[3].y, t[3].z, t[3].w};
        const float i10[8] = {t[4].x, t[4].y, t[4].z, t[4].w, t[5].x, t[5].y, t[5].z, t[5].w};
        const float i11[8] = {t[6].x, t[6].y, t[6].z, t[6].w, t[7].x, t[7].y, t[7].z, t[7].w};
        const float i0[8] = {s[u].px[0].x, s[u].px[0].y, s[u].px[0].z, s[u].px[0].w, s[u].px[1].x, s[u].px[1].y, s[u].px[1].z, s[u].px[1].w};
#pragma unroll
        for(int c = 0; c < 8; ++c) {
          const double Iw = wy * ((double) i00[c] * wx + (double) i01[c] * xf) + yf * ((double) i10[c] * wx + (double) i11[c] * xf);
          res[c] = (float) (Iw - (double) i0[c]);
        }
        if(!s[u].hit && s[u].in_block && cached) {
#pragma unroll
          for(int k = 0; k < 8; ++k) store_v4<NT>(tc + tile_index<8>(i, k), t[k]);
          j.tapkey[i] = s[u].key;
        }
      } else {
#pragma unroll
        for(int c = 0; c < 8; ++c) res[c] = 0.0f;
      }
      if(s[u].in_block) {
        j.valid[i] = s[u].valid ? 1 : 0;
        float4* o = reinterpret_cast<float4*>(j.r.get());
        store_v4<NT>(o + tile_index<2>(i, 0), make_float4(res[0], res[1], res[2], res[3]));
        store_v4<NT>(o + tile_index<2>(i, 1), make_float4(res[4], res[5], res[6], res[7]));
      }
      if(bracket)
        bracket_chunk<8>(j, lo_key, hi_key, s[u].valid && s[u].in_block, s[u].hit && s[u].valid && s[u].in_block, res, (unsigned) s[u].chunk, vtid >> 6,
                         pk_br_u[vsub][u], s[u].has);
    }
  }
}

template <int C>
__device__ __attribute__((noinline)) void pk_median_phase(const PairJob* __restrict__ jobs, int ws, bool stats_wg)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];     // median_block's histograms and key cache
  median_block<C, PK_THREADS>(jobs[ws], pk_st(ws), smem_raw, stats_wg);
  __syncthreads();
}

// irls_reduce of workspace ws: tile t goes to workgroup t % nwg, virtual block (t / nwg) % PK_VB
// (inlined into the kernel, unlike the other phases: as a function it uses all 256 VGPRs and would save and restore ~110
// callee-saved registers per call through scratch — 250 KB per workgroup each way: measured 12.6 instead of 9.2 us per iteration.
// Splitting a tile's points over the workgroup's two virtual blocks, contributions exchanged through LDS and added in point
// order, was measured as well: 10.2 us — the exchange costs more than the halved arithmetic saves.)
// The tile partials are DOUBLE-BUFFERED by iteration parity.  An iteration whose active workspaces all have a frozen scale (fused
// path) or a moot one (kL2) has no warp / median phase and hence no grid barrier between the step of iteration k and the reduction
// of iteration k + 1: a workgroup that finishes its step early would overwrite partials a slower workgroup is still summing (every
// workgroup sums all tiles for its own copy of the state).  With two buffers the writes of iteration k + 1 go to the other one; the
// buffer of iteration k is written again in iteration k + 2 at the earliest, i.e. after the barrier of iteration k + 1, which every
// workgroup only reaches after its step of iteration k.  The second buffer starts right behind the ntiles entries of the first (the
// allocation holds gn_partials_entries() = 2 * ceil(cap / pts_per_block) entries).
__device__ __forceinline__ float* pk_partials(const PairJob& j, int pts_per_block, unsigned parity)
{
  const int ntiles = (j.n + pts_per_block - 1) / pts_per_block;
  return j.partials + (size_t) (parity & 1u) * (size_t) ntiles * kPartialStride;
}
template <int C, int LOSS, bool FUSED>
__device__ __forceinline__ void pk_irls_phase(const PairJob* __restrict__ jobs, int ws, int pts_per_block, unsigned parity)
{
  const int tid = threadIdx.x, vsub = tid >> 8, vtid = tid & 255;
  const int nwg = pk_nwg;
  const PairJob& j = jobs[ws];
  const int ntiles = (j.n + pts_per_block - 1) / pts_per_block;
  float* const partials = pk_partials(j, pts_per_block, parity);
  for(int base = 0; base < ntiles; base += nwg * PK_VB) {
    const int tile = base + vsub * nwg + pk_member;
    // (the partials are stored THROUGH the caches and read past them by the step: the barrier between the two phases then needs no
    // release / acquire pair — an L2 write-back and an invalidation per workgroup and iteration for 30 floats per tile)
    if constexpr(C == 8) irls_tile_lat<LOSS, FUSED>(j, pk_st(ws), pts_per_block, tile, vtid, pk_part[vsub], tile < ntiles, partials, true);
    else irls_tile<C, LOSS, FUSED>(j, pk_st(ws), pts_per_block, tile, vtid, pk_part[vsub], tile < ntiles, partials, true);
    __syncthreads();
  }
}

// gn_step: wave w sums the partials of workspace w, its lane 0 runs the serial step on this workgroup's copy of the state
__device__ __attribute__((noinline)) void pk_step_phase(const PairJob* __restrict__ jobs, int nws, int pts_per_block, GNParams prm, int fuse, bool stats_wg,
                                                        unsigned parity)
{
  const int ws = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool mine = ws < nws && pk_st(ws < nws ? ws : 0)->active;
#ifdef BPVO_PK_TIMING
  long long sub_t = wall_clock64();
#endif
  if(mine) gn_sum_partials<true>(jobs[ws], pts_per_block, lane, pk_sum[ws], pk_partials(jobs[ws], pts_per_block, parity));
  __syncthreads();
#ifdef BPVO_PK_TIMING
  if(threadIdx.x == 0) GN_SUBTICK(4);
#endif
  if(mine && lane == 0)
    gn_serial_step(jobs[ws], pk_st(ws), pk_nrm[ws], pk_sum[ws], &pk_scratch[ws], 0, prm.max_iterations, prm.max_fun_evals, prm.p_tol, prm.f_tol,
                   prm.g_tol, fuse, stats_wg);
  __syncthreads();
}

// returns false when the barrier gave up (timeout, or another workgroup's abort)
// `light`: what crosses the barrier was stored through the caches and will be read past them (the reduction's partials): arrival and
// departure only, no release / acquire of the L2
__device__ __attribute__((noinline)) bool pk_grid_barrier(unsigned* ctl, unsigned epoch, long long timeout, bool light = false)
{
  // EVERY wave waits for its own global stores to have left (the vector L1 is write-through: vmcnt(0) = they are in the L2) before
  // the workgroup barrier: s_barrier alone does not wait for vmcnt, and thread 0's wait below covers thread 0's wave only.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if(threadIdx.x == 0) {
    if(!light) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned target = epoch * (unsigned) pk_nwg;
    __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    int ok = 1;
    unsigned spins = 0;
    while(__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if((++spins & 63u) == 0u || timeout < 64) {     // (tiny budgets: the tests of this path)
        if(__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if(wall_clock64() - t0 > timeout) { __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
      }
    }
    if(!light) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    pk_ok = ok;
  }
  __syncthreads();
  return pk_ok != 0;
}

template <int C, int LOSS>
__global__ __launch_bounds__(PK_THREADS) void gn_persistent_kernel(const PairJob* __restrict__ jobs, int nws, int pts_per_block, GNParams prm,
                                                                   int fuse_frozen, unsigned* ctl, long long timeout, int begin_level, int begin_moot,
                                                                   const PairJob* __restrict__ next_jobs)
{
  constexpr bool kCanFuse = (C == 8);
  const int tid = threadIdx.x;
  const bool stats_wg = blockIdx.x == 0;
  const bool fuse = kCanFuse && fuse_frozen;
  if(tid == 0) { pk_member = (int) blockIdx.x; pk_nwg = (int) gridDim.x; }

  for(int ws = 0; ws < nws; ++ws) {
    const uint32_t* g = reinterpret_cast<const uint32_t*>(jobs[ws].st.get());
    for(int i = tid; i < kStateWords; i += PK_THREADS) pk_state[ws][i] = g[i];
    if(tid < 4) pk_nrm[ws][tid] = jobs[ws].nrm[tid];
    if(tid == 4) pk_nrm[ws][4] = jobs[ws].dspace ? 1.0f : 0.0f;
  }
  __syncthreads();
  // begin_level >= 0: the level's start (level_begin_kernel's reset of the state) is taken here, by every workgroup on its own copy — the
  // tap-cache keys of this level were invalidated by the epilogue of the kernel of the level before (next_jobs there): no launch between two levels
  if(begin_level >= 0) {
    if(tid < nws) gn_level_reset(pk_st(tid), begin_level, begin_moot, jobs[tid].n);
    __syncthreads();
  }

  unsigned epoch = 0, epoch_it = 0;     // grid barriers passed; iterations done (parity of the partials buffer)
  bool ok = true;
  // BPVO_PK_TIMING: workgroup 0 accumulates the 100 MHz wall-clock ticks of every phase in ctl[8..13] and the iterations in ctl[15]
#ifdef BPVO_PK_TIMING
  long long tk = wall_clock64();
  unsigned acc_t[6] = {0, 0, 0, 0, 0, 0}, iters = 0;
  if(tid < 8) pk_sub[tid] = 0;
  __syncthreads();
#define PK_TICK(k) do { __syncthreads(); const long long t_ = wall_clock64(); acc_t[k] += (unsigned) (t_ - tk); tk = t_; } while(0)
#else
#define PK_TICK(k) do { } while(0)
#endif
  for(;;) {
    // who does what in this iteration: the same answer in every workgroup (identical state copies)
    bool any_active = false, any_warp = false;
    for(int ws = 0; ws < nws; ++ws) {
      const GNState* st = pk_st(ws);
      if(!st->active) continue;
      any_active = true;
      if(!fuse || st->delta_scale > 1e-6f) any_warp = true;
    }
    if(!any_active) break;
#ifdef BPVO_PK_TIMING
    tk = wall_clock64(); ++iters;
#endif

    if(any_warp) {
      // warp_residual of the workspaces whose robust scale still moves (all of them without the fused path) ...
      for(int ws = 0; ws < nws; ++ws) {
        const GNState* st = pk_st(ws);
        if(st->active && (!fuse || st->delta_scale > 1e-6f)) pk_warp_phase<C>(jobs, ws, stats_wg);
      }
      PK_TICK(0);
      ok = pk_grid_barrier(ctl, ++epoch, timeout);
      PK_TICK(1);
      if(!ok) break;
      // ... and their exact median + robust scale, every workgroup on its own copy of the state
      for(int ws = 0; ws < nws; ++ws) {
        const GNState* st = pk_st(ws);
        if(st->active && st->delta_scale > 1e-6f) pk_median_phase<C>(jobs, ws, stats_wg);
      }
      PK_TICK(2);
    }
    // weights + normal equations per tile (frozen scale with the fused path: residuals recomputed there)
    for(int ws = 0; ws < nws; ++ws) {
      const GNState* st = pk_st(ws);
      if(!st->active) continue;
      if constexpr(kCanFuse) {
        if(fuse && !(st->delta_scale > 1e-6f)) pk_irls_phase<C, LOSS, true>(jobs, ws, pts_per_block, epoch_it);
        else pk_irls_phase<C, LOSS, false>(jobs, ws, pts_per_block, epoch_it);
      } else {
        pk_irls_phase<C, LOSS, false>(jobs, ws, pts_per_block, epoch_it);
      }
    }
    PK_TICK(3);
    ok = pk_grid_barrier(ctl, ++epoch, timeout, true);
    PK_TICK(4);
    if(!ok) break;
    pk_step_phase(jobs, nws, pts_per_block, prm, fuse ? 1 : 0, stats_wg, epoch_it);
    PK_TICK(5);
    ++epoch_it;
  }
#ifdef BPVO_PK_TIMING
  if(blockIdx.x == 0 && tid == 0) {
    for(int k = 0; k < 6; ++k) ctl[8 + k] = acc_t[k];
    ctl[15] = iters;
    for(int k = 0; k < 5; ++k) ctl[16 + k] = pk_sub[k];
  }
#endif

  if(ok && blockIdx.x == 0) {
    for(int ws = 0; ws < nws; ++ws) {
      uint32_t* g = reinterpret_cast<uint32_t*>(jobs[ws].st.get());
      for(int i = tid; i < kStateWords; i += PK_THREADS) g[i] = pk_state[ws][i];
    }
  }
  // the tap-cache keys of the NEXT level's points (one key array per workspace, shared by the levels): nobody reads a key after the last
  // barrier of the last iteration, the next level's kernel starts behind this one
  if(ok && next_jobs) {
    for(int ws = 0; ws < nws; ++ws) {
      const PairJob& nj = next_jobs[ws];
      if(!nj.tapkey) continue;
      for(int i = (int) blockIdx.x * PK_THREADS + tid; i < nj.n; i += (int) gridDim.x * PK_THREADS) nj.tapkey[i] = 0xffffffffu;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// TEAM-persistent Gauss-Newton kernel for small BATCHES (2 .. 64 pairs).
// The four-kernel chain pays a floor per launch (ramp, drain, the list -> job -> state chain of dependent loads: ~10 us) that a
// 1024-pair batch amortises and a 128-pair batch does not: 4 launches x ~220 iterations x 10 us is a third of its Gauss-Newton time,
// and every level lasts as long as its slowest pair (profiles/r02_pipe/, profiles/r03_persistent_grid_probe.txt).  Here the
// workgroups of the grid form TEAMS, blockIdx.y = team, gridDim.x = workgroups per team (one per CU, all teams co-resident: the launcher
// sizes the grid to the CUs).  A team runs ONE pair at a time through ALL its pyramid levels and all their iterations with the
// phases of gn_persistent_kernel — the same device functions, chunk and tile indices as the chain, so every value is
// bit-identical — synchronised by barriers of its own (a counter per team): no launch between iterations, no host round trip
// between levels, no pair ever waits for another.  Pairs are handed out dynamically (one agent-scope counter), so a batch larger
// than the number of teams balances itself.  Teams desynchronise, which is the point: the memory-bound phases of some overlap the
// latency-bound ones (median, solve) of others.
// Barrier that cannot complete (teams not co-resident): the poll gives up after `timeout`, raises the abort word and every workgroup
// leaves; the host reruns the group on the chain, as for gn_persistent_kernel.
#ifndef TEAM_WARP_U_VALUE
#define TEAM_WARP_U_VALUE 2
#endif
#ifndef TEAM_NT_VALUE
#define TEAM_NT_VALUE 0
#endif
constexpr bool TEAM_NT = TEAM_NT_VALUE != 0;        // streaming (non-temporal) accesses in the team kernel's warp phase
constexpr int TEAM_WARP_U = TEAM_WARP_U_VALUE;      // points a thread of the team kernel's warp phase carries at once
constexpr int kTeamCtlWords = 32;       // one 128-byte line per team: [0] arrivals, [1] next pair broadcast slot; global line 0: [1] abort, [2] next pair
__shared__ int pk_next_pair;
__shared__ unsigned pk_team_xccs;      // XCDs the team's workgroups run on (bit mask)

// mode 0: release / acquire at agent scope — the L2 of the workgroup's XCD written back before the arrival, invalidated after the
//   departure: what crosses the barrier may be read by a workgroup on another XCD through plain loads.
// mode 1 (light): what crosses was stored through the caches and is read past them (the reduction's partials): arrival and departure only.
// mode 2 (local): every workgroup of the team sits on ONE XCD (verified at the start of the kernel from the hardware's XCC id): they share
//   its L2, so the writers' plain stores only have to have arrived there (the explicit s_waitcnt vmcnt(0) every wave executes ahead of
//   the workgroup barrier below: the vector L1 is write-through) — no L2 write-back — and the readers drop what their L1 holds with the acquire's own invalidation (buffer_inv sc1).
//   (The workgroup-scope form, buffer_inv sc0, was 5 % faster still and is NOT enough: the one-channel team test read stale residuals.)
// `target`: the value the team's arrival counter reaches when every member of the team has arrived at THIS barrier — the running sum of the
// team sizes over its barriers so far (pk_arrivals: a team grows when idle workgroups join it, see pk_join_team).
__shared__ unsigned pk_arrivals;
__shared__ int pk_size_seen;      // the team's size word — size | (iteration it applies to) << 16, published by the leader — as read with the count that completed the last barrier
// What the team kernel needs now and then (barrier budget, solver tolerances, the shape of the launch) lives in LDS, not in scalar registers
// held across the reduction phase: that phase is inlined and takes every register there is — with the join logic's operands alive across
// it the kernel spilled 55 vector registers (324 B of scratch per lane) and lost 2 % (profiles/r05_team_join.txt).
struct TeamCfg { long long timeout; unsigned* abort_word; GNParams prm; int n_pairs, n_teams, team_size, level_lo, scale_is_moot, local_ok, join_mode; unsigned xcc_bit; };
__shared__ TeamCfg pk_cfg;
__device__ __attribute__((noinline)) bool pk_team_barrier(unsigned* team_ctl, unsigned target, int mode = 0)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores in the L2 before the arrival (see pk_grid_barrier)
  __syncthreads();
  if(threadIdx.x == 0) {
    if(mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(team_ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned* const abort_word = pk_cfg.abort_word;      // (read behind the arrival: off the path of the workgroups that wait for this one)
    const long long timeout = pk_cfg.timeout;
    const long long t0 = wall_clock64();
    int ok = 1;
    unsigned spins = 0;
    // the arrival counter and the team size its leader has published sit in one aligned 8-byte word: ONE load per poll reads both, and the
    // size read together with the count that completes the barrier is the one the leader wrote ahead of its own arrival (pk_publish_admission)
    unsigned long long both;
    while((unsigned) (both = __hip_atomic_load(reinterpret_cast<unsigned long long*>(team_ctl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
      __builtin_amdgcn_s_sleep(1);
      if((++spins & 63u) == 0u || timeout < 64) {
        if(__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if(wall_clock64() - t0 > timeout) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
      }
    }
    if(mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if(mode == 2) asm volatile("buffer_inv sc1" ::: "memory");
    pk_ok = ok;
    pk_size_seen = (int) (both >> 32);
  }
  __syncthreads();
  return pk_ok != 0;
}

// PoseEstimatorBase::reset + the head of run() on this workgroup's LDS copy (level_begin_kernel's body), and this workgroup's share
// of the tap-cache keys of the level
__device__ __forceinline__ void pk_level_begin(const PairJob& j, int level, int scale_is_moot)
{
  const int nthreads_team = pk_nwg * PK_THREADS;
  if(j.tapkey)
    for(int i = pk_member * PK_THREADS + (int) threadIdx.x; i < j.n; i += nthreads_team) j.tapkey[i] = 0xffffffffu;
  if(threadIdx.x < 4) pk_nrm[0][threadIdx.x] = j.nrm[threadIdx.x];
  if(threadIdx.x == 4) pk_nrm[0][4] = j.dspace ? 1.0f : 0.0f;
  if(threadIdx.x == 0) {
    GNState* st = pk_st(0);
    st->scale = 1.0f;
    st->delta_scale = scale_is_moot ? 0.0f : 1e10f;
    st->f_norm_prev = 0.0f;
    st->g_tol = 0.0f;
    st->g_norm = 0.0f;
    st->num_fun_evals = 0;
    st->num_iterations = 0;
    st->status = BPVO_STATUS_MAX_ITERATIONS;
    st->phase = PHASE_FIRST;
    st->has_converged = 0;
    st->level = level;
    st->median_valid = 0;
    st->last_median = 0.0f;
    for(int i = 0; i < 16; ++i) st->T[i] = st->T_out[i];
    for(int i = 0; i < 6; ++i) st->dp[i] = 0.0f;
    st->active = (j.n > 0) ? 1 : 0;
  }
}

// ---- the team kernel with teams of FIXED size (rounds 3 - 4), kept for the launches in which no team can grow — teams that start at the
// admission cap (batches of up to 16 pairs), a single team — and for the batches below kTeamJoinFromPairs, where the growing form's
// bookkeeping costs more than the little imbalance of a few large teams gives back (8 / 16 / 32 pairs: 3 - 4 % slower, 64: + 2 %, 80: + 5 %,
// 128: + 7 %; profiles/r05_team_join.txt).  Same phases, same chunk and tile indices.  Its barrier counts epochs of a constant team size.
__device__ __attribute__((noinline)) bool pk_team_barrier_fixed(unsigned* team_ctl, unsigned* abort_word, unsigned epoch, long long timeout, int mode = 0)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores in the L2 before the arrival (see pk_grid_barrier)
  __syncthreads();
  if(threadIdx.x == 0) {
    if(mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned target = epoch * (unsigned) pk_nwg;
    __hip_atomic_fetch_add(team_ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    int ok = 1;
    unsigned spins = 0;
    while(__hip_atomic_load(team_ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if((++spins & 63u) == 0u || timeout < 64) {
        if(__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if(wall_clock64() - t0 > timeout) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
      }
    }
    if(mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if(mode == 2) asm volatile("buffer_inv sc1" ::: "memory");
    pk_ok = ok;
  }
  __syncthreads();
  return pk_ok != 0;
}


template <int C, int LOSS>
__global__ __launch_bounds__(PK_THREADS) void gn_team_fixed_kernel(const PairJob* __restrict__ jobs_all /*[levels][job_pitch]*/, int job_pitch, int n_pairs,
                                                             int team_size, int n_teams, int level_hi, int level_lo, int pts_per_block,
                                                             GNParams prm, int fuse_frozen, int scale_is_moot, unsigned* ctl, long long timeout, int local_ok)
{
  constexpr bool kCanFuse = (C == 8);
  const int tid = threadIdx.x;
  int team, member;
  {
    const int b = (int) blockIdx.x;
    if((n_teams & 7) == 0) {
      const int xcd = b & 7, slot = b >> 3;            // slot-th workgroup of its XCD
      team = xcd * (n_teams >> 3) + slot / team_size;
      member = slot % team_size;
    } else {
      team = b / team_size;
      member = b % team_size;
    }
  }
  // (the second launch of a split run — estimate.hip — behind a first one that gave up: its abort word was carried over, nothing to do here)
  if(__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  if(tid == 0) { pk_member = member; pk_nwg = team_size; }
  const bool stats_wg = member == 0;
  const bool fuse = kCanFuse && fuse_frozen;
  unsigned* const global_ctl = ctl;                                           // [1] abort, [2] next pair to hand out
  unsigned* const team_ctl = ctl + (size_t) (1 + team) * kTeamCtlWords;       // [0] arrivals, [1] pair slot
  unsigned epoch = 0, epoch_it = 0;
  // which XCD this workgroup runs on (HW_REG_XCC_ID, bits 3:0), registered in the team's line [2] before the first barrier; behind it every
  // member knows whether the team shares one L2 (team_mode 2: pk_team_barrier) — whatever the dispatcher did with the grid
  if(tid == 0) (void) __hip_atomic_fetch_or(team_ctl + 2, 1u << (__builtin_amdgcn_s_getreg(63508) & 0xf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int team_mode = 0;
  // BPVO_PK_TIMING: workgroup 0 of team 0 accumulates the 100 MHz ticks of its phases, per pyramid level, in ctl[4 .. 31]: 7 words per level
  // {warp, barrier1, median, irls, barrier2, step, iterations}
#ifdef BPVO_PK_TIMING
  long long tk = wall_clock64();
  unsigned acc_t[kMaxLevels][7];
  for(int l = 0; l < kMaxLevels; ++l) for(int k = 0; k < 7; ++k) acc_t[l][k] = 0;
#define TEAM_TICK(k) do { __syncthreads(); const long long t_ = wall_clock64(); acc_t[level][k] += (unsigned) (t_ - tk); tk = t_; } while(0)
#else
#define TEAM_TICK(k) do { } while(0)
#endif

  for(;;) {
    // next pair of this team: its workgroup 0 draws, the barrier publishes the draw to the others
    // (every workgroup reads the slot right after this barrier and before it arrives at the next one, which the drawing workgroup
    // must pass before it can draw again: one slot is enough)
    if(stats_wg && tid == 0) {
      const unsigned p = __hip_atomic_fetch_add(global_ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(team_ctl + 1, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if(!pk_team_barrier_fixed(team_ctl, global_ctl + 1, ++epoch, timeout)) return;
    if(tid == 0) {
      pk_next_pair = (int) __hip_atomic_load(team_ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pk_team_xccs = __hip_atomic_load(team_ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    team_mode = (local_ok && __popc(pk_team_xccs) == 1) ? 2 : 0;
    const int pair = pk_next_pair;
    if(pair >= n_pairs) return;

    {   // the pair's state: HBM -> this workgroup's LDS copy (set_pose_kernel has run: T_out, statistics defaults)
      const uint32_t* g = reinterpret_cast<const uint32_t*>(jobs_all[(size_t) level_hi * job_pitch + pair].st.get());
      for(int i = tid; i < kStateWords; i += PK_THREADS) pk_state[0][i] = g[i];
    }
    __syncthreads();

    for(int level = level_hi; level >= level_lo; --level) {
      const PairJob* __restrict__ jobs = jobs_all + (size_t) level * job_pitch + pair;      // jobs[0]: this pair at this level
      __syncthreads();      // every wave has read `active` of the level it leaves before thread 0 rewrites the LDS state
      pk_level_begin(jobs[0], level, scale_is_moot);
      if(!pk_team_barrier_fixed(team_ctl, global_ctl + 1, ++epoch, timeout, team_mode)) return;         // keys reset before any phase reads them
      for(;;) {
        const GNState* st = pk_st(0);
        if(!st->active) break;
        const bool moving = st->delta_scale > 1e-6f;
#ifdef BPVO_PK_TIMING
        tk = wall_clock64(); acc_t[level][6] += 1;
#endif
        if(!fuse || moving) {
          if constexpr(C == 8) pk_warp_phase_staged<TEAM_WARP_U, TEAM_NT>(jobs, 0, stats_wg);
          else pk_warp_phase<C>(jobs, 0, stats_wg);
          TEAM_TICK(0);
          if(!pk_team_barrier_fixed(team_ctl, global_ctl + 1, ++epoch, timeout, team_mode)) return;
          TEAM_TICK(1);
          if(moving) pk_median_phase<C>(jobs, 0, stats_wg);
          TEAM_TICK(2);
        }
        if constexpr(kCanFuse) {
          if(fuse && !(pk_st(0)->delta_scale > 1e-6f)) pk_irls_phase<C, LOSS, true>(jobs, 0, pts_per_block, epoch_it);   // (after the median: the chain's rule)
          else pk_irls_phase<C, LOSS, false>(jobs, 0, pts_per_block, epoch_it);
        } else {
          pk_irls_phase<C, LOSS, false>(jobs, 0, pts_per_block, epoch_it);
        }
        TEAM_TICK(3);
        if(!pk_team_barrier_fixed(team_ctl, global_ctl + 1, ++epoch, timeout, 1)) return;      // (only the partials cross: light)
        TEAM_TICK(4);
        pk_step_phase(jobs, 1, pts_per_block, prm, fuse ? 1 : 0, stats_wg, epoch_it);
        TEAM_TICK(5);
        ++epoch_it;
      }
    }
#ifdef BPVO_PK_TIMING
    if(team == 0 && member == 0 && tid == 0)
      for(int l = 0; l < 4; ++l) for(int k = 0; k < 7; ++k) global_ctl[4 + l * 7 + k] += acc_t[l][k];      // words 4 .. 31 of the global line, summed over the team's pairs
#endif
    // the pair is done: its state back to HBM (one copy; the others are identical)
    if(stats_wg) {
      uint32_t* g = reinterpret_cast<uint32_t*>(jobs_all[(size_t) level_hi * job_pitch + pair].st.get());
      for(int i = tid; i < kStateWords; i += PK_THREADS) g[i] = pk_state[0][i];
    }
    __syncthreads();
  }
}


// ---- teams that GROW: idle workgroups join the teams that are still working -------------------------------------------------------
// A batch ends when its slowest pair does, and the pairs of a batch are far from equal (128 KITTI pairs: mean 17.4 ms per pair on its
// two CUs, slowest 21.0: a fifth of the chip's time idle behind the last pairs, profiles/r05_team_join.txt).  When a team draws no more
// pair, each of its workgroups looks for a team that is still at work, takes a ticket on that team's control line and waits; the
// team's first workgroup (the leader) looks at the tickets once per iteration, right before it arrives at the barrier between
// reduction and step, and publishes the new team size; after the step old members and newcomers meet at one ADMISSION barrier
// (full release / acquire at agent scope: whatever the team wrote so far — tap cache, keys, state — becomes visible to the newcomers'
// XCDs), the leader's copy of the state — all copies are identical — goes through HBM to the newcomers, and the next iteration deals
// chunks and tiles over the larger team.  Chunk and tile indices, and with them every value, do not depend on the team size
// (test_team_kernel_shapes); the barriers count arrivals against the running sum of the team sizes (pk_arrivals).
// Team control line: [0] arrivals  [1] team size, published by the leader (0: not yet; one 8-byte word with [0])  [2] XCDs of the members (bit mask)  [11] pair slot
// [4] tickets taken  [5] closed (the team has drawn its last pair and dissolved)  [6] arrival target of the pending admission barrier
// [7] pair  [8] level  [9] iteration parity counter for the newcomers  [10] level the team works on (for the choice of a team)
#ifndef TEAM_ADMIT_EVERY_VALUE
#define TEAM_ADMIT_EVERY_VALUE 1
#endif
#ifndef TEAM_MAX_SIZE_VALUE
#define TEAM_MAX_SIZE_VALUE 16
#endif
constexpr unsigned kTeamAdmitEvery = TEAM_ADMIT_EVERY_VALUE;      // the leader looks at the tickets every so many iterations (a power of two): each look is an L2 round trip for every member (128 pairs: 976 k GN it/s with 4, 985 k with 2, 988 k with 1; profiles/r05_team_join.txt)
constexpr int kTeamMaxSize = TEAM_MAX_SIZE_VALUE;       // a team stops admitting here: the serial phases and the barriers grow with it (team_size sweep, profiles/r05_team_join.txt)
struct TeamSeat { int team, member, nwg, pair, level; unsigned epoch_it, arrivals; };
__shared__ TeamSeat pk_seat;
__shared__ int pk_joined;
// thread 0 of an idle workgroup: find a team, take a ticket, wait for the admission; false: nothing left to join (or abort / timeout)
__device__ __attribute__((noinline)) bool pk_join_team(unsigned* ctl, int own_team, TeamSeat* seat)
{
  unsigned* const abort_word = ctl + 1;
  const int n_teams = pk_cfg.n_teams, own_team_size = pk_cfg.team_size;
  const unsigned my_xcc_bit = pk_cfg.xcc_bit;
  const bool same_xcd_only = pk_cfg.join_mode == 1;
  const long long timeout = pk_cfg.timeout;
  const long long t0 = wall_clock64();
  for(;;) {
    // choice: the open team on the coarsest level (most of its work ahead), then the smallest; teams on this workgroup's XCD first
    int best = -1, best_score = -1;
    bool unstarted = false;
    for(int k = 1; k <= n_teams; ++k) {
      const int t = (own_team + k) % n_teams;
      unsigned* line = ctl + (size_t) (1 + t) * kTeamCtlWords;
      if(__hip_atomic_load(line + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) continue;       // dissolved
      const unsigned size = __hip_atomic_load(line + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu;      // (the upper half: the iteration tag, pk_publish_admission)
      if(size == 0u) { unstarted = true; continue; }      // (its leader has not said hello yet: a spare workgroup at the start of the launch)
      const unsigned tickets = __hip_atomic_load(line + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned crowd = (unsigned) own_team_size + tickets;      // (every team starts with own_team_size members)
      if(crowd >= (unsigned) kTeamMaxSize) continue;
      const unsigned xccs = __hip_atomic_load(line + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool local = xccs == my_xcc_bit;
      if(same_xcd_only && !local) continue;
      const int level = (int) __hip_atomic_load(line + 10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int score = (local ? 1 << 20 : 0) + (level << 8) + (kTeamMaxSize - (int) crowd);
      if(score > best_score) { best_score = score; best = t; }
    }
    if(best < 0) {
      if(!unstarted) return false;
      __builtin_amdgcn_s_sleep(32);
      if(__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      if(wall_clock64() - t0 > timeout) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
      continue;
    }
    unsigned* line = ctl + (size_t) (1 + best) * kTeamCtlWords;
    const unsigned ticket = __hip_atomic_fetch_add(line + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned my_index = (unsigned) own_team_size + ticket;
    if(my_index >= (unsigned) kTeamMaxSize) continue;      // (lost a race for the last seat: the ticket stays unused — the leader never admits beyond the cap)
    (void) __hip_atomic_fetch_or(line + 2, my_xcc_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // wait for the leader's word: admitted (size > my index) or dissolved
    unsigned spins = 0;
    for(;;) {
      const unsigned size = __hip_atomic_load(line + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu;
      if(size > my_index) {
        seat->team = best; seat->member = (int) my_index; seat->nwg = (int) size;
        seat->arrivals = __hip_atomic_load(line + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seat->pair = (int) __hip_atomic_load(line + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seat->level = (int) __hip_atomic_load(line + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seat->epoch_it = __hip_atomic_load(line + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
      }
      if(__hip_atomic_load(line + 5, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        // dissolved; an admission that included this ticket would have been published before that
        if((__hip_atomic_load(line + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu) > my_index) continue;
        break;      // look for another team
      }
      __builtin_amdgcn_s_sleep(8);
      if((++spins & 63u) == 0u) {
        if(__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if(wall_clock64() - t0 > timeout) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
      }
    }
  }
}

// The whole workgroup: find a team and wait for the admission (thread 0, pk_join_team), wait until the team's barrier ahead of the admission
// barrier is complete, meet the team at the admission barrier, take over the leader's state.  Leaves the seat in pk_seat and the team's
// XCDs in pk_team_xccs; false: nothing left to join, abort or timeout.
__device__ __attribute__((noinline)) bool pk_take_seat(unsigned* ctl, const PairJob* __restrict__ jobs_all, int job_pitch, int level_hi, int own_team)
{
  const int tid = threadIdx.x;
  if(!pk_cfg.join_mode) return false;
  if(tid == 0) pk_joined = pk_join_team(ctl, own_team, &pk_seat) ? 1 : 0;
  __syncthreads();
  if(!pk_joined) return false;
  unsigned* const team_ctl = ctl + (size_t) (1 + pk_seat.team) * kTeamCtlWords;
  if(tid == 0) {
    pk_member = pk_seat.member; pk_nwg = pk_seat.nwg;
    pk_arrivals = pk_seat.arrivals;      // the admission barrier's target: where the team's count stands once this workgroup is in
    (void) __hip_atomic_fetch_add(ctl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Not before the team's own barrier between reduction and step is complete: the arrival counter is one running sum, and an
    // arrival ahead of that barrier would be counted for it.  Its target is the admission barrier's minus the new team size.
    const unsigned before = pk_seat.arrivals - (unsigned) pk_seat.nwg;
    const long long t0 = wall_clock64();
    unsigned spins = 0;
    int ok = 1;
    while((int) (__hip_atomic_load(team_ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - before) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if((++spins & 63u) == 0u) {
        if(__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if(wall_clock64() - t0 > pk_cfg.timeout) { __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
      }
    }
    pk_joined = ok;
  }
  __syncthreads();
  if(!pk_joined) return false;
  // the admission barrier (full release / acquire), then the state the leader put into HBM ahead of it
  if(!pk_team_barrier(team_ctl, pk_seat.arrivals, 0)) return false;
  {
    const int pair = pk_seat.pair;
    const uint32_t* g = reinterpret_cast<const uint32_t*>(jobs_all[(size_t) level_hi * job_pitch + pair].st.get());
    for(int i = tid; i < kStateWords; i += PK_THREADS) pk_state[0][i] = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const PairJob& jl = jobs_all[(size_t) pk_seat.level * job_pitch + pair];
    if(tid < 4) pk_nrm[0][tid] = jl.nrm[tid];
    if(tid == 4) pk_nrm[0][4] = jl.dspace ? 1.0f : 0.0f;
  }
  if(tid == 0) pk_team_xccs = __hip_atomic_load(team_ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return true;
}

// The leader (thread 0), every kTeamAdmitEvery-th iteration, right before it arrives at the barrier between reduction and step: newcomers
// are told the size of the team from the next iteration on, the arrival target of the admission barrier behind this iteration's step,
// and where the team stands — all of it ahead of the leader's own arrival, so that every old member reads the new size right behind
// that barrier.
__device__ __attribute__((noinline)) void pk_publish_admission(unsigned* team_ctl, int pair, int level, unsigned epoch_it, unsigned arrivals)
{
  const unsigned tickets = __hip_atomic_load(team_ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned want = min((unsigned) pk_cfg.team_size + tickets, (unsigned) kTeamMaxSize);
  if(want <= (unsigned) pk_nwg) return;
  __hip_atomic_store(team_ctl + 6, arrivals + (unsigned) pk_nwg + want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(team_ctl + 7, (unsigned) pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(team_ctl + 8, (unsigned) level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(team_ctl + 9, epoch_it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // The size is TAGGED with the iteration it applies to.  In iterations with a single team barrier (frozen scale, fused path) a fast leader can be
  // a whole iteration ahead of a member that has arrived at barrier k but not yet made the poll that completes it; that member would read the
  // count of barrier k together with the size published for k + 1, enter the admission barrier one iteration early — counted as its arrival at
  // barrier k + 1 — and skip that iteration's partials.  With the tag it ignores a size meant for a later iteration (pk_admit) and meets it at
  // that iteration's barrier.  (Sizes are at most kTeamMaxSize; the tag is the iteration modulo 2^16.)
  __hip_atomic_store(team_ctl + 1, want | ((epoch_it & 0xffffu) << 16), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// Every member, behind the step of such an iteration: has the team grown?  Then old members and newcomers meet once, with the full fences;
// the leader's state goes ahead through HBM.  false: the barrier gave up.
__device__ __attribute__((noinline)) bool pk_admit(unsigned* team_ctl, GNState* g_state, bool leader, unsigned iteration)
{
  const int tid = threadIdx.x;
  // (read with the count that completed the barrier between reduction and step: pk_team_barrier; a size tagged for another iteration — the
  // initial one, one already adopted, one a leader that runs ahead has published for the NEXT iteration — changes nothing now)
  const unsigned word = (unsigned) pk_size_seen;
  const int new_nwg = ((word >> 16) == (iteration & 0xffffu) && (word & 0xffffu) != 0u) ? (int) (word & 0xffffu) : pk_nwg;
  if(new_nwg <= pk_nwg) return true;
  if(leader) {
    uint32_t* g = reinterpret_cast<uint32_t*>(g_state);
    for(int i = tid; i < kStateWords; i += PK_THREADS) __hip_atomic_store(g + i, pk_state[0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const unsigned target = pk_arrivals + (unsigned) new_nwg;
  __syncthreads();
  if(tid == 0) { pk_nwg = new_nwg; pk_arrivals = target; }
  if(!pk_team_barrier(team_ctl, target, 0)) return false;
  if(tid == 0) pk_team_xccs = __hip_atomic_load(team_ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return true;
}

// Grid: 1-D, n_teams * team_size workgroups.  Workgroups are dealt to the XCDs round robin by their linear index (guide: block b runs on
// XCD b % 8 — observed, for speed only); when the teams divide evenly over the 8 XCDs a team's workgroups are taken from ONE XCD so that
// the pair's taps, residuals and partials stay in that XCD's L2 between phases.  Any mapping is correct (the barriers are agent-scope).
// (A WIDE form of this kernel — every phase under 128 VGPRs and 68 KB of LDS, two workgroups per CU — was built and measured in round 4:
// slower at every batch size, profiles/r04_team_wide_rejected.txt.)
// One pair from `level` down, on this workgroup's team.  A function of its own, NOT inlined into the kernel's pair / join loop: the
// reduction phase inside takes every register there is, and with the loop's bookkeeping (which team, leader or not, the seat of a
// newcomer) alive across it the kernel spilled 50 vector registers; here those are plain arguments, fixed for the call, and the
// call's save / restore is paid once per pair.  resume: this workgroup has just joined the team in the middle of `level`.
// Returns false when a barrier gave up.
struct TeamPairArgs { const PairJob* jobs_all; unsigned* team_ctl; int job_pitch, level_hi, pts_per_block, fuse, pair, level, leader, team_mode, resume; unsigned epoch_it; };
__shared__ unsigned pk_epoch_it_out;
template <int C, int LOSS, bool JOIN>
__device__ __forceinline__ bool team_run_pair_body(const TeamPairArgs a)
{
  constexpr bool kCanFuse = (C == 8);
  const int tid = threadIdx.x;
  // Arguments of a non-inlined device function arrive in VECTOR registers: the compiler no longer knows that they are the same in every
  // lane, and everything derived from them — the job's fields, every address of the phases — would be computed and loaded per lane (the
  // first form of this function: 3 - 5 % slower than the inlined kernel at 8 - 32 pairs).  readfirstlane says it.
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto uni_ptr = [&](const void* q) {
    const unsigned long long u = (unsigned long long) q;
    return (unsigned long long) (unsigned) uni((int) (unsigned) u) | ((unsigned long long) (unsigned) uni((int) (unsigned) (u >> 32)) << 32);
  };
  const PairJob* __restrict__ const jobs_all = reinterpret_cast<const PairJob*>(uni_ptr(a.jobs_all));
  unsigned* const team_ctl = reinterpret_cast<unsigned*>(uni_ptr(a.team_ctl));
  const bool stats_wg = uni(a.leader) != 0, fuse = uni(a.fuse) != 0;
  const int pair = uni(a.pair), pts_per_block = uni(a.pts_per_block), job_pitch = uni(a.job_pitch), level_hi = uni(a.level_hi);
  int team_mode = uni(a.team_mode);
  bool resume = uni(a.resume) != 0;
  unsigned epoch_it = (unsigned) uni((int) a.epoch_it);
  // arrivals: what the team's counter reads once every member has arrived at the barrier passed last (the running sum of the team
  // sizes over its barriers); nwg: the team's size.  Register copies of pk_arrivals / pk_nwg, which the admission code updates.
  unsigned arrivals = (unsigned) uni((int) pk_arrivals);
  int nwg = uni(pk_nwg);
#define TEAM_BARRIER(mode_) pk_team_barrier(team_ctl, arrivals += (unsigned) nwg, mode_)
#ifdef BPVO_PK_TIMING
  long long tk = wall_clock64();
  unsigned acc_t[kMaxLevels][7];
  for(int l = 0; l < kMaxLevels; ++l) for(int k = 0; k < 7; ++k) acc_t[l][k] = 0;
#define TEAM_TICK(k) do { __syncthreads(); const long long t_ = wall_clock64(); acc_t[level][k] += (unsigned) (t_ - tk); tk = t_; } while(0)
#else
#define TEAM_TICK(k) do { } while(0)
#endif
  const int level_lo = uni(pk_cfg.level_lo), scale_is_moot = uni(pk_cfg.scale_is_moot), join_mode = JOIN ? uni(pk_cfg.join_mode) : 0;
  for(int level = uni(a.level); level >= level_lo; --level) {
    const PairJob* __restrict__ jobs = jobs_all + (size_t) level * job_pitch + pair;      // jobs[0]: this pair at this level
    if(!resume) {
      __syncthreads();      // every wave has read `active` of the level it leaves before thread 0 rewrites the LDS state
      pk_level_begin(jobs[0], level, scale_is_moot);
      if(stats_wg && tid == 0) __hip_atomic_store(team_ctl + 10, (unsigned) level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if(!TEAM_BARRIER(team_mode)) return false;         // keys reset before any phase reads them
    }
    resume = false;
    for(;;) {
      const GNState* st = pk_st(0);
      if(!st->active) break;
      const bool moving = st->delta_scale > 1e-6f;
#ifdef BPVO_PK_TIMING
      tk = wall_clock64(); acc_t[level][6] += 1;
#endif
      if(!fuse || moving) {
        if constexpr(C == 8) pk_warp_phase_staged<TEAM_WARP_U, TEAM_NT>(jobs, 0, stats_wg);
        else pk_warp_phase<C>(jobs, 0, stats_wg);
        TEAM_TICK(0);
        if(!TEAM_BARRIER(team_mode)) return false;
        TEAM_TICK(1);
        if(moving) pk_median_phase<C>(jobs, 0, stats_wg);
        TEAM_TICK(2);
      }
      if constexpr(kCanFuse) {
        if(fuse && !(pk_st(0)->delta_scale > 1e-6f)) pk_irls_phase<C, LOSS, true>(jobs, 0, pts_per_block, epoch_it);   // (after the median: the chain's rule)
        else pk_irls_phase<C, LOSS, false>(jobs, 0, pts_per_block, epoch_it);
      } else {
        pk_irls_phase<C, LOSS, false>(jobs, 0, pts_per_block, epoch_it);
      }
      TEAM_TICK(3);
      const bool admission_turn = JOIN && join_mode && (epoch_it & (kTeamAdmitEvery - 1u)) == 0u;      // (every member has the same epoch_it)
      if(admission_turn && stats_wg && tid == 0) pk_publish_admission(team_ctl, pair, level, epoch_it, arrivals);
      if(!TEAM_BARRIER(1)) return false;      // (only the partials cross: light)
      TEAM_TICK(4);
      pk_step_phase(jobs, 1, pts_per_block, pk_cfg.prm, fuse ? 1 : 0, stats_wg, epoch_it);
      TEAM_TICK(5);
      ++epoch_it;
      if(admission_turn) {
        if(tid == 0) pk_arrivals = arrivals;
        if(!pk_admit(team_ctl, jobs_all[(size_t) level_hi * job_pitch + pair].st.get(), stats_wg, epoch_it - 1u)) return false;
        arrivals = (unsigned) uni((int) pk_arrivals); nwg = uni(pk_nwg);
        team_mode = (pk_cfg.local_ok && __popc(pk_team_xccs) == 1) ? 2 : 0;
      }
    }
  }
#ifdef BPVO_PK_TIMING
  if(stats_wg && team_ctl == pk_cfg.abort_word - 1 + kTeamCtlWords && tid == 0)      // (team 0's leader)
    for(int l = 0; l < 4; ++l) for(int k = 0; k < 7; ++k) (pk_cfg.abort_word - 1)[4 + l * 7 + k] += acc_t[l][k];      // words 4 .. 31 of the global line, summed over the team's pairs
#endif
  if(tid == 0) { pk_epoch_it_out = epoch_it; pk_arrivals = arrivals; }
  // the pair is done: its state back to HBM (one copy; the others are identical)
  if(stats_wg) {
    uint32_t* g = reinterpret_cast<uint32_t*>(jobs_all[(size_t) level_hi * job_pitch + pair].st.get());
    for(int i = tid; i < kStateWords; i += PK_THREADS) g[i] = pk_state[0][i];
  }
  __syncthreads();
  return true;
}

template <int C, int LOSS>
__device__ __attribute__((noinline)) bool team_run_pair(const TeamPairArgs a) { return team_run_pair_body<C, LOSS, true>(a); }

// JOIN = false: the launches in which nobody can join anybody (teams that start at the admission cap — batches of up to 16 pairs — or a
// single team): the pair's loops inlined into the kernel as they were before teams could grow, without the admission turns (3 % at 8 pairs,
// where an iteration lasts 12 us).
template <int C, int LOSS, bool JOIN>
__global__ __launch_bounds__(PK_THREADS) void gn_team_kernel(const PairJob* __restrict__ jobs_all /*[levels][job_pitch]*/, int job_pitch, int n_pairs,
                                                             int team_size, int n_teams, int level_hi, int level_lo, int pts_per_block,
                                                             GNParams prm, int fuse_frozen, int scale_is_moot, unsigned* ctl, long long timeout, int local_ok,
                                                             int join_mode)
{
  constexpr bool kCanFuse = (C == 8);
  const int tid = threadIdx.x;
  int team, member;
  // SPARE workgroups: the grid may be larger than n_teams * team_size (the launcher fills the chip: 96 pairs as teams of 2 leave 64 CUs
  // over) — what is beyond the teams starts as a helper and joins a team at its first admission
  const bool spare = (int) blockIdx.x >= n_teams * team_size;
  {
    const int b = (int) blockIdx.x;
    if(spare) {
      team = b % n_teams;       // (where its search for a team starts)
      member = -1;
    } else if((n_teams & 7) == 0) {
      const int xcd = b & 7, slot = b >> 3;            // slot-th workgroup of its XCD
      team = xcd * (n_teams >> 3) + slot / team_size;
      member = slot % team_size;
    } else {
      team = b / team_size;
      member = b % team_size;
    }
  }
  // (the second launch of a split run behind a first one that gave up: its abort word was carried over, nothing to do here)
  if(__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  if(tid == 0) {
    pk_member = member; pk_nwg = team_size; pk_arrivals = 0u;
    pk_cfg.timeout = timeout; pk_cfg.abort_word = ctl + 1; pk_cfg.prm = prm; pk_cfg.n_pairs = n_pairs; pk_cfg.n_teams = n_teams; pk_cfg.team_size = team_size;
    pk_cfg.level_lo = level_lo; pk_cfg.scale_is_moot = scale_is_moot; pk_cfg.local_ok = local_ok; pk_cfg.join_mode = join_mode;
    pk_cfg.xcc_bit = 1u << (__builtin_amdgcn_s_getreg(63508) & 0xf);
  }
  __syncthreads();
  bool leader = member == 0 && !spare;       // the team's first workgroup: draws pairs, keeps the counters, admits newcomers
  unsigned* const global_ctl = ctl;                                     // [1] abort, [2] next pair to hand out, [3] workgroups that joined another team
  unsigned* team_ctl = ctl + (size_t) (1 + team) * kTeamCtlWords;
  // which XCD this workgroup runs on (HW_REG_XCC_ID, bits 3:0), registered in the team's line [2] before the first barrier; behind it every
  // member knows whether the team shares one L2 (team_mode 2: pk_team_barrier) — whatever the dispatcher did with the grid
  if(tid == 0 && !spare) {
    (void) __hip_atomic_fetch_or(team_ctl + 2, pk_cfg.xcc_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(leader) __hip_atomic_store(team_ctl + 1, (unsigned) team_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  TeamPairArgs a;
  a.jobs_all = jobs_all; a.job_pitch = job_pitch; a.level_hi = level_hi; a.pts_per_block = pts_per_block; a.fuse = (kCanFuse && fuse_frozen) ? 1 : 0;
  a.epoch_it = 0;
  bool idle = spare;       // a workgroup without a team: looks for one before anything else
  for(;;) {
    // next pair of this team: its leader draws, the barrier publishes the draw to the others
    // (every workgroup reads the slot right after this barrier and before it arrives at the next one, which the drawing workgroup
    // must pass before it can draw again: one slot is enough)
    if(!idle) {
    if(leader && tid == 0) {
      const unsigned p = __hip_atomic_fetch_add(global_ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(team_ctl + 11, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // no pair left: the team dissolves — said BEFORE the barrier, behind every admission this leader ever published
      if(p >= (unsigned) n_pairs) __hip_atomic_store(team_ctl + 5, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    {
      const unsigned target = pk_arrivals + (unsigned) pk_nwg;
      __syncthreads();
      if(tid == 0) pk_arrivals = target;
      if(!pk_team_barrier(team_ctl, target)) return;
    }
    if(tid == 0) {
      pk_next_pair = (int) __hip_atomic_load(team_ctl + 11, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pk_team_xccs = __hip_atomic_load(team_ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    a.pair = pk_next_pair;
    }
    a.resume = 0;
    a.level = level_hi;
    if(idle || a.pair >= n_pairs) {
      idle = false;
      // idle: join a team that still works (join_mode 0: leave; 1: teams on this workgroup's XCD only; 2: any)
      if constexpr(!JOIN) return;
      if(!pk_take_seat(ctl, jobs_all, job_pitch, level_hi, team)) return;
      team = pk_seat.team;
      team_ctl = ctl + (size_t) (1 + team) * kTeamCtlWords;
      leader = false;
      a.pair = pk_seat.pair; a.level = pk_seat.level; a.epoch_it = pk_seat.epoch_it;
      a.resume = 1;
    } else {
      // the pair's state: HBM -> this workgroup's LDS copy (set_pose_kernel has run: T_out, statistics defaults)
      const uint32_t* g = reinterpret_cast<const uint32_t*>(jobs_all[(size_t) level_hi * job_pitch + a.pair].st.get());
      for(int i = tid; i < kStateWords; i += PK_THREADS) pk_state[0][i] = g[i];
    }
    __syncthreads();
    a.team_ctl = team_ctl; a.leader = leader ? 1 : 0;
    a.team_mode = (local_ok && __popc(pk_team_xccs) == 1) ? 2 : 0;
    if constexpr(JOIN) { if(!team_run_pair<C, LOSS>(a)) return; }
    else { if(!team_run_pair_body<C, LOSS, false>(a)) return; }
    a.epoch_it = pk_epoch_it_out;
  }
}

// ---- persistent kernel for small groups
bool gn_persistent_serves(const GNLaunch& g)
{
  return (g.C == 8 || g.C == 1) && g.interp == BPVO_INTERP_LINEAR && !g.fast_warp && g.npairs >= 1 && g.npairs <= kPersistMaxWs && !g.active.list;
}
int gn_persistent_grid(const GNLaunch& g, int max_grid)
{
  const int chunks = (g.max_points + K6_BLOCK - 1) / K6_BLOCK;
  return std::max(1, std::min(max_grid, (chunks + PK_VB - 1) / PK_VB));
}
template <int C>
static hipError_t launch_gn_persistent_c(hipStream_t s, const GNLaunch& g, const GNParams& prm, unsigned* ctl, int grid, long long timeout)
{
  const int ppb = gn_pts_per_block(C);
  const int fuse = (C == 8 && g.fuse_frozen) ? 1 : 0;
  auto go = [&](auto kern) -> hipError_t {
    // once per kernel and device (the lanes' host threads may race here): the opt-in for the 123 KB of median_block's LDS and the
    // residency check — one workgroup per CU must fit, the grid itself (<= 128) is far below the number of CUs
    static std::once_flag once[64];
    static hipError_t status[64];
    int dev = 0;
    (void) hipGetDevice(&dev);
    dev &= 63;
    std::call_once(once[dev], [&] {
      status[dev] = hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) kMedianLds);
      int per_cu = 0;
      if(status[dev] == hipSuccess) status[dev] = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, PK_THREADS, kMedianLds);
      if(status[dev] == hipSuccess && per_cu < 1) status[dev] = hipErrorLaunchOutOfResources;
    });
    if(status[dev] != hipSuccess) return status[dev];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PK_THREADS), kMedianLds, s, g.jobs, g.npairs, ppb, prm, fuse, ctl, timeout, g.begin_level, g.begin_moot, g.next_jobs);
    return hipGetLastError();
  };
  switch(g.loss) {
    case BPVO_LOSS_HUBER: return go(gn_persistent_kernel<C, BPVO_LOSS_HUBER>);
    case BPVO_LOSS_TUKEY: return go(gn_persistent_kernel<C, BPVO_LOSS_TUKEY>);
    default: return go(gn_persistent_kernel<C, BPVO_LOSS_L2>);
  }
}
hipError_t launch_gn_persistent(hipStream_t s, const GNLaunch& g, int max_iterations, int max_fun_evals, float p_tol, float f_tol, float g_tol,
                                unsigned* ctl, int grid, long long timeout_ticks)
{
  if(g.max_points <= 0) return hipSuccess;
  GNParams prm;
  prm.max_iterations = max_iterations; prm.max_fun_evals = max_fun_evals; prm.p_tol = p_tol; prm.f_tol = f_tol; prm.g_tol = g_tol;
  if(g.C == 8) return launch_gn_persistent_c<8>(s, g, prm, ctl, grid, timeout_ticks);
  return launch_gn_persistent_c<1>(s, g, prm, ctl, grid, timeout_ticks);
}
// ---- team-persistent kernel for small batches
int gn_team_ctl_words(int n_teams) { return (1 + n_teams) * kTeamCtlWords; }
// between the two launches of a split run: the control words back to zero — EXCEPT the abort word, so that a second launch behind a first one
// that gave up at a barrier leaves at once instead of running every remaining level on states that were never written back
__global__ void team_ctl_reset_keep_abort_kernel(unsigned* ctl, int words)
{
  for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x)
    if(i != 1) ctl[i] = 0u;
}
void launch_team_ctl_reset_keep_abort(hipStream_t s, unsigned* ctl, int n_teams)
{
  const int words = gn_team_ctl_words(n_teams);
  hipLaunchKernelGGL(team_ctl_reset_keep_abort_kernel, dim3(std::max(1, std::min(64, (words + 255) / 256))), dim3(256), 0, s, ctl, words);
}
int gn_team_max_size() { return kTeamMaxSize; }
template <int C>
static hipError_t launch_gn_team_c(hipStream_t s, const GNTeamLaunch& t, const GNParams& prm)
{
  const int ppb = gn_pts_per_block(C);
  const int fuse = (C == 8 && t.fuse_frozen) ? 1 : 0;
  constexpr size_t lds = kMedianLds;
  constexpr int need_per_cu = 1;
  auto go = [&](auto kern) -> hipError_t {
    static std::once_flag once[64];
    static hipError_t status[64];
    int dev = 0;
    (void) hipGetDevice(&dev);
    dev &= 63;
    std::call_once(once[dev], [&] {
      status[dev] = hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
      int per_cu = 0;
      if(status[dev] == hipSuccess) status[dev] = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, PK_THREADS, lds);
      if(status[dev] == hipSuccess && per_cu < need_per_cu) status[dev] = hipErrorLaunchOutOfResources;
    });
    if(status[dev] != hipSuccess) return status[dev];
    hipLaunchKernelGGL(kern, dim3(t.team_size * t.n_teams + t.spare_workgroups), dim3(PK_THREADS), lds, s, t.jobs_all, t.job_pitch, t.n_pairs, t.team_size, t.n_teams, t.level_hi,
                       t.level_lo, ppb, prm, fuse, t.scale_is_moot, t.ctl, t.timeout_ticks, t.local_barriers, t.join_mode);
    return hipGetLastError();
  };
  if(t.join_mode) {
    switch(t.loss) {
      case BPVO_LOSS_HUBER: return go(gn_team_kernel<C, BPVO_LOSS_HUBER, true>);
      case BPVO_LOSS_TUKEY: return go(gn_team_kernel<C, BPVO_LOSS_TUKEY, true>);
      default: return go(gn_team_kernel<C, BPVO_LOSS_L2, true>);
    }
  }
  auto go_fixed = [&](auto kern) -> hipError_t {
    static std::once_flag once[64];
    static hipError_t status[64];
    int dev = 0;
    (void) hipGetDevice(&dev);
    dev &= 63;
    std::call_once(once[dev], [&] {
      status[dev] = hipFuncSetAttribute((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
      int per_cu = 0;
      if(status[dev] == hipSuccess) status[dev] = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, PK_THREADS, lds);
      if(status[dev] == hipSuccess && per_cu < need_per_cu) status[dev] = hipErrorLaunchOutOfResources;
    });
    if(status[dev] != hipSuccess) return status[dev];
    hipLaunchKernelGGL(kern, dim3(t.team_size * t.n_teams), dim3(PK_THREADS), lds, s, t.jobs_all, t.job_pitch, t.n_pairs, t.team_size, t.n_teams, t.level_hi,
                       t.level_lo, ppb, prm, fuse, t.scale_is_moot, t.ctl, t.timeout_ticks, t.local_barriers);
    return hipGetLastError();
  };
  switch(t.loss) {
    case BPVO_LOSS_HUBER: return go_fixed(gn_team_fixed_kernel<C, BPVO_LOSS_HUBER>);
    case BPVO_LOSS_TUKEY: return go_fixed(gn_team_fixed_kernel<C, BPVO_LOSS_TUKEY>);
    default: return go_fixed(gn_team_fixed_kernel<C, BPVO_LOSS_L2>);
  }
}
hipError_t launch_gn_team(hipStream_t s, const GNTeamLaunch& t, int max_iterations, int max_fun_evals, float p_tol, float f_tol, float g_tol)
{
  GNParams prm;
  prm.max_iterations = max_iterations; prm.max_fun_evals = max_fun_evals; prm.p_tol = p_tol; prm.f_tol = f_tol; prm.g_tol = g_tol;
  if(t.C == 8) return launch_gn_team_c<8>(s, t, prm);
  return launch_gn_team_c<1>(s, t, prm);
}
}  // namespace bpvo_hip
