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
constexpr int K6_WAVES = K6_BLOCK / 64;

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
struct BracketLds { unsigned in[K6_WAVES], below[K6_WAVES], valid[K6_WAVES]; };
template <int C>
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
  unsigned incl = cnt, sum_below = below, sum_valid = (v ? 1u : 0u) | (hit ? 0x10000u : 0u);   // valid points | tap-cache hits << 16
#pragma unroll
  for(int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o);
    if(lane >= o) incl += t;
  }
#pragma unroll
  for(int o = 32; o >= 1; o >>= 1) {
    sum_below += __shfl_down(sum_below, o);
    sum_valid += __shfl_down(sum_valid, o);
  }
  unsigned woff = 0;
  if constexpr(K6_WAVES == 1) {     // one wavefront per workgroup: no LDS, no barrier
    const unsigned t_in = __shfl(incl, 63);
    if(lane == 0 && write) reinterpret_cast<uint4*>(j.med_blk.get())[blk] = make_uint4(sum_below, t_in, sum_valid & 0xffffu, sum_valid >> 16);
  } else {
    if(lane == 63) s.in[wave] = incl;
    if(lane == 0) { s.below[wave] = sum_below; s.valid[wave] = sum_valid; }
    __syncthreads();
    for(int w = 0; w < wave; ++w) woff += s.in[w];
    if(wave == 0 && lane == 0 && write) {
      uint4 o = make_uint4(0u, 0u, 0u, 0u);
      for(int w = 0; w < K6_WAVES; ++w) { o.x += s.below[w]; o.y += s.in[w]; o.z += s.valid[w] & 0xffffu; o.w += s.valid[w] >> 16; }
      reinterpret_cast<uint4*>(j.med_blk.get())[blk] = o;
    }
  }
  if(cnt) {
    unsigned* seg = j.cand + (size_t) blk * K6_BLOCK * C;
    unsigned pos = woff + incl - cnt;
#pragma unroll
    for(int c = 0; c < C; ++c)
      if(mask & ((mask_t) 1u << c)) seg[pos++] = keys[c];
  }
}

// the form warp_residual uses: one 256-thread workgroup = one chunk
template <int C>
__device__ __forceinline__ void bracket_block(const PairJob& j, unsigned lo, unsigned hi, bool v, bool hit, const float (&res)[C])
{
  __shared__ BracketLds s;
  bracket_chunk<C>(j, lo, hi, v, hit, res, blockIdx.x, (int) (threadIdx.x >> 6), s, true);
}

}  // namespace bpvo_hip
