// Gauss-Newton kernels, shared device code: workgroup shapes, the active list, the bracket step of the exact median (K7a).
// Included by kernels_gn.hip (the four-kernel chain) and kernels_gn_team.hip (the persistent kernels): both call the SAME device
// functions with the same chunk / tile indices, which is what makes their results bit-identical.
#pragma once
#include <float.h>

#include <type_traits>

#include "kernels.h"

namespace bpvo_hip {


constexpr int GN_BLOCK = 256;
// workgroup size of the warp + residual kernels = chunk of the fused bracket step (candidate segments, med_blk entries).
// 64-thread workgroups stream a little better (the bare access pattern: 6.48 against 6.24 TB/s, scripts/micro/streams.hip;
// warp_residual: +1 %), but median_finish then walks four times as many candidate segments: 256 measured 3 % faster overall.
#ifndef K6_BLOCK_VALUE
#define K6_BLOCK_VALUE 256
#endif
constexpr int K6_BLOCK = K6_BLOCK_VALUE;
static_assert(K6_BLOCK == kChunkPoints, "the host sizes bracket segments and group offsets with kChunkPoints");
constexpr int K6_WAVES = K6_BLOCK / 64;

// Sums of N <= 32 per-lane accumulators over the 64 lanes of a wavefront, every one with the pairing of
//   for(o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o);
// as lane 0 sees it — (i, i+32), then (i, i+16), (i, i+8) ... (i, i+1): the same additions in the same order, bit for bit — but as a
// reduce-scatter on the VALU.  __shfl_down is ds_bpermute, an LDS round trip: 6 per accumulator, 180 per wave for the 30 sums of the
// normal equations.  Here v_permlane32_swap folds the two halves of the wave for TWO accumulators per instruction (the lower half
// goes on with accumulator k, the upper half with k + H), v_permlane16_swap folds the rows of 16 the same way, and the last four
// steps are DPP row shifts inside the rows: 78 VALU instructions and no LDS for N = 30.  Afterwards lane 16 * r of a wave holds, in
// out[k], the sum of accumulator wave_tree_index<N>(r, k) (-1: nothing).
template <int N>
__host__ __device__ constexpr int wave_tree_index(int row, int k)
{
  constexpr int H = (N + 1) / 2, Q = (H + 1) / 2;
  const int ka = k + (row & 1) * Q;
  const int a = ka + (row >> 1) * H;
  return (ka < H && a < N) ? a : -1;
}
template <int N>
__device__ __forceinline__ void wave_tree_sums(const float (&acc)[N], float (&out)[(((N + 1) / 2) + 1) / 2])
{
  static_assert(N >= 2 && N <= 32, "four rows of up to eight sums");
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  constexpr int H = (N + 1) / 2, Q = (H + 1) / 2;
  float t[H];
#pragma unroll
  for(int k = 0; k < H; ++k) {
    const u2 s = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[k]), __float_as_uint(acc[k + H < N ? k + H : k]), false, false);
    t[k] = __uint_as_float(s.x) + __uint_as_float(s.y);
  }
#pragma unroll
  for(int k = 0; k < Q; ++k) {
    const u2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(t[k]), __float_as_uint(t[k + Q < H ? k + Q : k]), false, false);
    float v = __uint_as_float(s.x) + __uint_as_float(s.y);
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x108, 0xf, 0xf, false));   // row_shl:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104, 0xf, 0xf, false));   // row_shl:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x102, 0xf, 0xf, false));   // row_shl:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x101, 0xf, 0xf, false));   // row_shl:1
    out[k] = v;
  }
}
// ... and into the per-wave row of an LDS table: part[a] = sum of accumulator a
template <int N>
__device__ __forceinline__ void wave_tree_sums_to(const float (&acc)[N], int lane, float* part)
{
  constexpr int Q = (((N + 1) / 2) + 1) / 2;
  float out[Q];
  wave_tree_sums<N>(acc, out);
  if((lane & 15) == 0) {
    const int row = lane >> 4;
#pragma unroll
    for(int k = 0; k < Q; ++k) {
      const int a = wave_tree_index<N>(row, k);
      if(a >= 0) part[a] = out[k];
    }
  }
}

// 32-bit integer inclusive scan / sum over the 64 lanes of a wavefront on the VALU: row_shr 1, 2, 4, 8 inside the rows of 16 (lanes without
// a source add 0), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  The sum is the scan's last lane, read with
// v_readlane (uniform over the wave).
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned x)
{
  int v = (int) x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, true);
  return (unsigned) v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned x) { return (unsigned) __builtin_amdgcn_readlane((int) wave_incl_scan_u32(x), 63); }

// workspace of a workgroup: k-th entry of the active list, or k itself without a list
__device__ __forceinline__ int active_workspace(const ActiveSet& a, int k) { return a.list ? a.list[k] : k; }

// K7a (fused into warp_residual): bracket counting + candidate compaction for the exact median of the NEXT kernel.
// The median moves little between GN iterations, so while the residuals are still in registers every block counts its
// keys (bit patterns of |r| of valid points) below the bracket [lo, hi) around the previous median and compacts the keys
// inside it.  No global atomics: block b of a workspace owns med_blk[b] = {#below, #inside, #valid points} and the
// candidate segment cand[b * 256 * C ...]; the in-block compaction is a wave scan + LDS offsets.  All 256 threads of the
// block must call it.
// Generalised form: `blk` is the chunk (the blockIdx.x of warp_residual), `wave` the wavefront inside the 256-thread chunk, `s` the
// chunk's LDS scratch; `write` = false for a chunk past the end that only keeps its threads in step (persistent kernel).  All
// threads of the WORKGROUP must call it (it holds a __syncthreads).
// DENSE (chain launches over templates of many points, GNLaunch::dense_candidates): chunk b belongs to run b mod 64 of the workspace; its
// leader adds the chunk's four counts to the run's totals (two 64-bit adds: {inside | below << 32}, {valid points | tap-cache hits << 32};
// every run's totals on a 128-byte line of their own, entry med_tot + 8 * run of med_blk) and the returning add gives the chunk its place
// in the run's contiguous candidates, cand[run * dense_run_cap ...].  The finish then reads 64 totals and 64 arrays instead of a
// thousand counters and segments.  (One run for all chunks: a thousand returning adds on one address, 50 us.)  The order inside a run
// differs from launch to launch; the order statistics selected from the candidates do not.
__device__ __forceinline__ unsigned dense_run_cap(int n, int C) { return (unsigned) ((((n + K6_BLOCK - 1) / K6_BLOCK) + kDenseRuns - 1) / kDenseRuns) * (unsigned) (K6_BLOCK * C); }
struct BracketLds { unsigned in[K6_WAVES], below[K6_WAVES], valid[K6_WAVES], base; };
template <int C, bool DENSE = false>
__device__ __forceinline__ void bracket_chunk(const PairJob& j, unsigned lo, unsigned hi, bool v, bool hit, const float (&res)[C],
                                              unsigned blk, int wave, BracketLds& s, bool write)
{
  // one "inside the bracket" bit per channel: 64 bits once a point has more than 32 channels (central difference, 48)
  using mask_t = typename std::conditional<(C > 32), unsigned long long, unsigned>::type;
  static_assert(C <= 64, "bracket_block keeps one mask bit per channel");
  unsigned keys[C];
  unsigned below = 0, cnt = 0;
  mask_t mask = 0;
#pragma unroll
  for(int c = 0; c < C; ++c) {
    const unsigned k = __float_as_uint(res[c]) & 0x7fffffffu;
    keys[c] = k;
    const bool in = v && (k >= lo) && (k < hi);
    below += (v && k < lo) ? 1u : 0u;
    cnt += in ? 1u : 0u;
    mask |= (mask_t) (in ? 1u : 0u) << c;
  }
  const int lane = threadIdx.x & 63;
  // the inclusive scan of the counts and the two wave sums on the VALU (DPP row shifts + the two row broadcasts: 18 ds_bpermute round
  // trips as __shfl_up / __shfl_down ladders, per 64 points); integer sums — any order
  const unsigned incl = wave_incl_scan_u32(cnt);
  const unsigned sum_below = wave_sum_u32(below);
  const unsigned sum_valid = wave_sum_u32((v ? 1u : 0u) | (hit ? 0x10000u : 0u));      // valid points | tap-cache hits << 16 (wave-uniform results)
  unsigned woff = 0;
  static_assert(!DENSE || K6_WAVES > 1, "the dense form hands the run's offset through the chunk's LDS scratch");
  if constexpr(K6_WAVES == 1) {     // one wavefront per workgroup: no LDS, no barrier
    const unsigned t_in = (unsigned) __builtin_amdgcn_readlane((int) incl, 63);
    if(lane == 0 && write) reinterpret_cast<uint4*>(j.med_blk.get())[blk] = make_uint4(sum_below, t_in, sum_valid & 0xffffu, sum_valid >> 16);
  } else {
    if(lane == 63) s.in[wave] = incl;
    if(lane == 0) { s.below[wave] = sum_below; s.valid[wave] = sum_valid; }
    __syncthreads();
    for(int w = 0; w < wave; ++w) woff += s.in[w];
    if(wave == 0 && lane == 0 && write) {
      uint4 o = make_uint4(0u, 0u, 0u, 0u);
      for(int w = 0; w < K6_WAVES; ++w) { o.x += s.below[w]; o.y += s.in[w]; o.z += s.valid[w] & 0xffffu; o.w += s.valid[w] >> 16; }
      if constexpr(DENSE) {
        const unsigned run = blk & (unsigned) (kDenseRuns - 1);
        unsigned long long* tot = reinterpret_cast<unsigned long long*>(j.med_blk.get() + 4 * ((size_t) j.med_tot + 8 * run));
        const unsigned long long old = atomicAdd(tot, (unsigned long long) o.y | ((unsigned long long) o.x << 32));
        atomicAdd(tot + 1, (unsigned long long) o.z | ((unsigned long long) o.w << 32));
        s.base = run * dense_run_cap(j.n, C) + (unsigned) old;
      } else {
        reinterpret_cast<uint4*>(j.med_blk.get())[blk] = o;
      }
    }
  }
  if constexpr(DENSE) __syncthreads();      // (the leader's returning add)
  if(cnt) {
    unsigned* seg = DENSE ? j.cand + s.base : j.cand + (size_t) blk * K6_BLOCK * C;
    unsigned pos = woff + incl - cnt;
#pragma unroll
    for(int c = 0; c < C; ++c)
      if(mask & ((mask_t) 1u << c)) seg[pos++] = keys[c];
  }
}

// the form warp_residual uses: one 256-thread workgroup = one chunk
template <int C>
__device__ __forceinline__ void bracket_block(const PairJob& j, unsigned lo, unsigned hi, bool v, bool hit, const float (&res)[C], bool dense)
{
  __shared__ BracketLds s;
  if(dense) bracket_chunk<C, true>(j, lo, hi, v, hit, res, blockIdx.x, (int) (threadIdx.x >> 6), s, true);
  else bracket_chunk<C, false>(j, lo, hi, v, hit, res, blockIdx.x, (int) (threadIdx.x >> 6), s, true);
}

}  // namespace bpvo_hip
