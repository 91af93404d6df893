// Gauss-Newton / IRLS kernels (gfx950), batched over estimation workspaces (independent frame pairs) with blockIdx.y:
//   warp_residual  K6  project template points with the current pose (f64), valid mask, bilinear gather of the
//                      current frame's pixel-interleaved descriptor, residuals                     (18 + 24*C B/point)
//   median         K7  exact median of |r| over valid entries: 3-pass radix select on the IEEE bit pattern with
//                      LDS histograms, one workgroup per pair; robust scale + freeze rule               (4*C B/point)
//   irls_reduce    K8  M-estimator weights fused with the J^T W J / J^T W r / sum w r^2 reduction:
//                      Jacobian rows recomputed from (point, Ix, Iy); in-thread accumulation, wavefront shuffle tree,
//                      LDS across waves, per-block partials            (algorithmic 2 + 28*C B/point; moved 18 + 12*C)
//   gn_step        K9  deterministic f64 sum of the partials, 6x6 LDLT solve, SE(3) update and the convergence /
//                      iteration bookkeeping of PoseEstimatorBase::run, all on the device (no host round trip of H, G)
// No MFMA anywhere: ~1 flop/byte, HBM-bound gather + rank-1 accumulate.
#include <float.h>

#include <algorithm>
#include <cstdlib>

#include <mutex>

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

// ------------------------------------------------------------------------------------------------------------------
// K6 warp_residual.  reference: TemplateData::computeResiduals (bpvo/template_data.cc:174-189) =
//   RigidBodyWarp::setPose (bpvo/rigid_body_warp.h:111-114): P = K * T[0:3,:] in f32, index-order sums
//   PhotoError::Impl::init (bpvo/photo_error.cc:344-363): x = normHomog(P.cast<double>() * X.cast<double>()),
//       Floor (:255-265), valid = 0 <= xi < W-1 && 0 <= yi < R-1 (kLinear)
//   PhotoError::Impl::run kLinear (bpvo/photo_error.cc:365-389,446-449): Iw in f64, r = float(Iw - I0); invalid -> 0
// One thread per template point; all C channels of the point are handled by the same thread because the descriptor is
// pixel-interleaved: the 4 taps are 2 x (2*C floats) contiguous, fetched as 16-byte loads.
// FAST selects the reference's alternative all-f32 formulation (inactive there, PHOTO_ERROR_OPT = 0): projectPoints
// (bpvo/project_points.cc:180-214: x = P*X in f32, w = 1.0f/x2, xi = (int) xf — truncation, not floor — valid =
// 0 <= xi < W-1 && 0 <= yi < R-1, coefficients C = [xf*yf - yf - xf + 1, xf - xf*yf, yf - xf*yf, xf*yf]) followed by
// PhotoError::Impl::operator() / run of that branch (bpvo/photo_error.cc:118-214; same arithmetic as BilinearInterp,
// bpvo/interp_util.h:49-71,93-96,184-203): Iw = dp_ps(C, [I00, I01, I10, I11]) = (C0*I00 + C1*I01) + (C2*I10 + C3*I11),
// r = Iw - I0, and for an invalid point Iw = 0, i.e. r = -I0.
// P = K * T[0:3,:] in f32, index-order sums (RigidBodyWarp::setPose, bpvo/rigid_body_warp.h:111-114)
__device__ __forceinline__ void projection_matrix(const PairJob& j, const float* __restrict__ T, float (&P)[12])
{
#pragma unroll
  for(int r = 0; r < 3; ++r)
#pragma unroll
    for(int c = 0; c < 4; ++c) {
      float s = j.K[r * 3 + 0] * T[0 * 4 + c];
      s += j.K[r * 3 + 1] * T[1 * 4 + c];
      s += j.K[r * 3 + 2] * T[2 * 4 + c];
      P[r * 4 + c] = s;
    }
}

// DisparitySpaceWarp::setPose (bpvo/disparity_space_warp.h:36): H = G * T * G_inv in f32, the two fixed 4x4 products left
// to right, G / G_inv as the constructor fills them (bpvo/disparity_space_warp.cc:26-47).  P <- rows 0, 1, 3 of H: with
// them operator() (:66-71) is the projectPoints form below plus the principal point (x = pw0 * (1 / pw3) + cx).
__device__ __forceinline__ void dspace_matrix(const PairJob& j, const float* __restrict__ T, float (&P)[12])
{
  const float fx = j.K[0], fy = j.K[4];
  M44 G, Gi, Tm;
  for(int i = 0; i < 16; ++i) { G.m[i] = 0.0f; Gi.m[i] = 0.0f; Tm.m[i] = T[i]; }
  G.m[0] = fx; G.m[5] = fy; G.m[11] = fx * j.b; G.m[14] = 1.0f;
  Gi.m[0] = (float) (1.0 / (double) fx); Gi.m[5] = (float) (1.0 / (double) fy); Gi.m[11] = 1.0f;
  Gi.m[14] = (float) (1.0 / (double) (fx * j.b));
  const M44 H = m44_mul(m44_mul(G, Tm), Gi);
#pragma unroll
  for(int c = 0; c < 4; ++c) { P[c] = H.m[c]; P[4 + c] = H.m[4 + c]; P[8 + c] = H.m[12 + c]; }
}

// One template point of warp_residual: projection, validity, (cached) bilinear taps, residuals of all C channels.
// `in_block` gates the tap-cache update (lanes past the end of a block redo the last point, loads only).  Returns valid.
// HALF (C = 8, f64 formulation): the taps are fetched and consumed in two groups of four channels, which halves the
// registers they occupy — for the fused path of irls_reduce, where the 29 accumulators are live as well.
template <int C, bool FAST, bool HALF = false, bool NT = true>
__device__ __forceinline__ bool warp_point(const PairJob& j, const float (&P)[12], int i, bool in_block, float (&res)[C], bool& cache_hit)
{
  cache_hit = false;
  const int W = j.cols, R = j.rows;
  const float4 X = load_v4<NT>(j.pts + i);
  // C = 1: the launches are short and latency-bound, so the key, the cached taps and the template pixel are requested
  // together with the point instead of after the projection (16 speculative bytes per point; for C = 8 the same
  // speculation costs 128 bytes and was measured slower)
  unsigned spec_key = 0; float4 spec_taps = make_float4(0.0f, 0.0f, 0.0f, 0.0f); float spec_pix = 0.0f;
  if constexpr(C == 1) {
    if(j.tapcache_on) {      // (uniform over the workspace: dense levels run without the cache)
      spec_key = j.tapkey[i];
      spec_taps = load_v4<NT>(reinterpret_cast<const float4*>(j.tapcache.get()) + i);
    }
    spec_pix = j.pix[i];
  }
  int xi = 0, yi = 0;
  bool valid;
  double xf = 0.0, yf = 0.0;       // fractional parts (standard formulation)
  float cf[4] = {0, 0, 0, 0};      // interpolation coefficients (FAST formulation)
  if constexpr(!FAST) {
    const double X0 = (double) X.x, X1 = (double) X.y, X2 = (double) X.z, X3 = (double) X.w;
    double u[3];
#pragma unroll
    for(int r = 0; r < 3; ++r) {
      double s = (double) P[r * 4 + 0] * X0;
      s += (double) P[r * 4 + 1] * X1;
      s += (double) P[r * 4 + 2] * X2;
      s += (double) P[r * 4 + 3] * X3;
      u[r] = s;
    }
    const double zi = 1.0 / u[2];
    const double x = zi * u[0], y = zi * u[1];
    // Floor(): static_cast<int> then -(i > v).  x86 yields INT_MIN for NaN / out-of-range doubles, which can never be a
    // valid pixel; the explicit range test gives the same verdict without relying on v_cvt_i32_f64 saturation.
    const bool in_range = (x > -2147483648.0) && (x < 2147483648.0) && (y > -2147483648.0) && (y < 2147483648.0);
    if(in_range) {
      xi = (int) x; xi -= (xi > x);
      yi = (int) y; yi -= (yi > y);
    }
    valid = in_range && xi >= 0 && xi < W - 1 && yi >= 0 && yi < R - 1;
    xf = x - (double) xi; yf = y - (double) yi;
  } else {
    float u[3];
#pragma unroll
    for(int r = 0; r < 3; ++r) {
      float s = P[r * 4 + 0] * X.x;
      s += P[r * 4 + 1] * X.y;
      s += P[r * 4 + 2] * X.z;
      s += P[r * 4 + 3] * X.w;
      u[r] = s;
    }
    const float w_i = 1.0f / u[2];
    float fx = w_i * u[0], fy = w_i * u[1];
    if(j.dspace) { fx = fx + j.K[2]; fy = fy + j.K[5]; }   // DisparitySpaceWarp::operator() (disparity_space_warp.h:66-71)
    // (int) xf: cvttss2si gives INT_MIN for NaN / out-of-range, never a valid pixel
    const bool in_range = (fx > -2147483648.0f) && (fx < 2147483648.0f) && (fy > -2147483648.0f) && (fy < 2147483648.0f);
    if(in_range) { xi = (int) fx; yi = (int) fy; }
    valid = in_range && xi >= 0 && xi < W - 1 && yi >= 0 && yi < R - 1;
    fx -= (float) xi; fy -= (float) yi;
    const float xfyf = fx * fy;
    cf[0] = xfyf - fy - fx + 1.0f; cf[1] = fx - xfyf; cf[2] = fy - xfyf; cf[3] = xfyf;
  }

  if constexpr(HALF && C == 8 && !FAST) {
    if(valid) {
      const double wx = 1.0 - xf, wy = 1.0 - yf;
      const float4* q0 = reinterpret_cast<const float4*>(j.desc + ((size_t) yi * W + xi) * 8);
      const float4* q1 = q0 + (size_t) W * 2;
      const unsigned key = ((unsigned) yi << 16) | (unsigned) xi;
      const bool cached = j.tapcache_on != 0;       // (uniform over the workspace) dense levels gather straight from the descriptor
      const bool hit = cached && j.tapkey[i] == key;
      cache_hit = hit;
      float4* tc = reinterpret_cast<float4*>(j.tapcache.get());
      const float4* p0 = reinterpret_cast<const float4*>(j.pix.get());
#pragma unroll
      for(int h = 0; h < 2; ++h) {
        float4 a, b, c, d;      // I00, I01, I10, I11 of channels 4h .. 4h+3
        if(hit) {
          a = load_v4<NT>(tc + tile_index<8>(i, h)); b = load_v4<NT>(tc + tile_index<8>(i, 2 + h));
          c = load_v4<NT>(tc + tile_index<8>(i, 4 + h)); d = load_v4<NT>(tc + tile_index<8>(i, 6 + h));
        } else {
          a = q0[h]; b = q0[2 + h]; c = q1[h]; d = q1[2 + h];
          if(in_block && cached) {
            store_v4<NT>(tc + tile_index<8>(i, h), a); store_v4<NT>(tc + tile_index<8>(i, 2 + h), b);
            store_v4<NT>(tc + tile_index<8>(i, 4 + h), c); store_v4<NT>(tc + tile_index<8>(i, 6 + h), d);
          }
        }
        const float4 t = load_v4<NT>(p0 + tile_index<2>(i, h));
        const float i00[4] = {a.x, a.y, a.z, a.w}, i01[4] = {b.x, b.y, b.z, b.w}, i10[4] = {c.x, c.y, c.z, c.w},
                    i11[4] = {d.x, d.y, d.z, d.w}, i0[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for(int k = 0; k < 4; ++k) {
          const double Iw = wy * ((double) i00[k] * wx + (double) i01[k] * xf) + yf * ((double) i10[k] * wx + (double) i11[k] * xf);
          res[4 * h + k] = (float) (Iw - (double) i0[k]);
        }
        if(h == 0) __builtin_amdgcn_sched_barrier(0);   // keep the second group's loads behind the first group's arithmetic
      }
      if(!hit && in_block && cached) j.tapkey[i] = key;
    } else {
#pragma unroll
      for(int c = 0; c < 8; ++c) res[c] = 0.0f;
    }
    return valid;
  }

  if(valid) {
    const double wx = 1.0 - xf, wy = 1.0 - yf;
    const float* __restrict__ d0 = j.desc + ((size_t) yi * W + xi) * C;
    const float* __restrict__ d1 = d0 + (size_t) W * C;
    float I00[C], I01[C], I10[C], I11[C], I0[C];
    if constexpr(C == 8) {
      // Tap cache: the integer footprint (xi, yi) of a point rarely changes between consecutive GN iterations of a level
      // (sub-pixel pose updates), and then the four taps are the same 128 bytes.  They are kept per point in a tiled,
      // fully coalesced buffer keyed by (yi << 16 | xi): a hit replaces the gather — two 64-byte segments that cost
      // 2.3 128-byte HBM lines on average (profiles/r01_pmc_summary.txt) — by one coalesced 128-byte read.
      const unsigned key = ((unsigned) yi << 16) | (unsigned) xi;
      const bool cached = j.tapcache_on != 0;       // (uniform over the workspace) dense levels gather straight from the descriptor
      const bool hit = cached && j.tapkey[i] == key;
      cache_hit = hit;
      float4 a0, a1, a2, a3, b0, b1, b2, b3;
      float4* tc = reinterpret_cast<float4*>(j.tapcache.get());
      if(hit) {
        a0 = load_v4<NT>(tc + tile_index<8>(i, 0)); a1 = load_v4<NT>(tc + tile_index<8>(i, 1));
        a2 = load_v4<NT>(tc + tile_index<8>(i, 2)); a3 = load_v4<NT>(tc + tile_index<8>(i, 3));
        b0 = load_v4<NT>(tc + tile_index<8>(i, 4)); b1 = load_v4<NT>(tc + tile_index<8>(i, 5));
        b2 = load_v4<NT>(tc + tile_index<8>(i, 6)); b3 = load_v4<NT>(tc + tile_index<8>(i, 7));
      } else {
        const float4* q0 = reinterpret_cast<const float4*>(d0);
        const float4* q1 = reinterpret_cast<const float4*>(d1);
        a0 = q0[0]; a1 = q0[1]; a2 = q0[2]; a3 = q0[3];
        b0 = q1[0]; b1 = q1[1]; b2 = q1[2]; b3 = q1[3];
        if(in_block && cached) {
          store_v4<NT>(tc + tile_index<8>(i, 0), a0); store_v4<NT>(tc + tile_index<8>(i, 1), a1);
          store_v4<NT>(tc + tile_index<8>(i, 2), a2); store_v4<NT>(tc + tile_index<8>(i, 3), a3);
          store_v4<NT>(tc + tile_index<8>(i, 4), b0); store_v4<NT>(tc + tile_index<8>(i, 5), b1);
          store_v4<NT>(tc + tile_index<8>(i, 6), b2); store_v4<NT>(tc + tile_index<8>(i, 7), b3);
          j.tapkey[i] = key;
        }
      }
      const float4* p0 = reinterpret_cast<const float4*>(j.pix.get());
      const float4 t0 = load_v4<NT>(p0 + tile_index<2>(i, 0)), t1 = load_v4<NT>(p0 + tile_index<2>(i, 1));
      I00[0] = a0.x; I00[1] = a0.y; I00[2] = a0.z; I00[3] = a0.w; I00[4] = a1.x; I00[5] = a1.y; I00[6] = a1.z; I00[7] = a1.w;
      I01[0] = a2.x; I01[1] = a2.y; I01[2] = a2.z; I01[3] = a2.w; I01[4] = a3.x; I01[5] = a3.y; I01[6] = a3.z; I01[7] = a3.w;
      I10[0] = b0.x; I10[1] = b0.y; I10[2] = b0.z; I10[3] = b0.w; I10[4] = b1.x; I10[5] = b1.y; I10[6] = b1.z; I10[7] = b1.w;
      I11[0] = b2.x; I11[1] = b2.y; I11[2] = b2.z; I11[3] = b2.w; I11[4] = b3.x; I11[5] = b3.y; I11[6] = b3.z; I11[7] = b3.w;
      I0[0] = t0.x; I0[1] = t0.y; I0[2] = t0.z; I0[3] = t0.w; I0[4] = t1.x; I0[5] = t1.y; I0[6] = t1.z; I0[7] = t1.w;
    } else if constexpr(C == 1) {
      // the same tap cache for single-channel descriptors: the four taps of a point are one 16-byte record.  The gather
      // costs two (mostly distinct) HBM lines per point at the sparse levels for 16 useful bytes; a hit is one coalesced load.
      const unsigned key = ((unsigned) yi << 16) | (unsigned) xi;
      float4* tc = reinterpret_cast<float4*>(j.tapcache.get());
      float4 t = spec_taps;
      const bool cached = j.tapcache_on != 0;
      cache_hit = cached && spec_key == key;
      if(!cache_hit) {
        t = make_float4(d0[0], d0[1], d1[0], d1[1]);
        if(in_block && cached) { store_v4<NT>(tc + i, t); j.tapkey[i] = key; }
      }
      I00[0] = t.x; I01[0] = t.y; I10[0] = t.z; I11[0] = t.w;
      I0[0] = spec_pix;
    } else {
#pragma unroll
      for(int c = 0; c < C; ++c) {
        I00[c] = d0[c]; I01[c] = d0[C + c]; I10[c] = d1[c]; I11[c] = d1[C + c];
        I0[c] = j.pix[(size_t) i * C + c];
      }
    }
#pragma unroll
    for(int c = 0; c < C; ++c) {
      if constexpr(!FAST) {
        const double Iw = wy * ((double) I00[c] * wx + (double) I01[c] * xf) + yf * ((double) I10[c] * wx + (double) I11[c] * xf);
        res[c] = (float) (Iw - (double) I0[c]);
      } else {
        const float Iw = (cf[0] * I00[c] + cf[1] * I01[c]) + (cf[2] * I10[c] + cf[3] * I11[c]);
        res[c] = Iw - I0[c];
      }
    }
  } else {
    if constexpr(!FAST) {
#pragma unroll
      for(int c = 0; c < C; ++c) res[c] = 0.0f;
    } else {   // operator() returns 0 for an invalid point and run() still subtracts I0 (photo_error.cc:203-210)
      if constexpr(C == 8) {
        const float4* p0 = reinterpret_cast<const float4*>(j.pix.get());
        const float4 t0 = p0[tile_index<2>(i, 0)], t1 = p0[tile_index<2>(i, 1)];
        res[0] = 0.0f - t0.x; res[1] = 0.0f - t0.y; res[2] = 0.0f - t0.z; res[3] = 0.0f - t0.w;
        res[4] = 0.0f - t1.x; res[5] = 0.0f - t1.y; res[6] = 0.0f - t1.z; res[7] = 0.0f - t1.w;
      } else {
#pragma unroll
        for(int c = 0; c < C; ++c) res[c] = 0.0f - j.pix[(size_t) i * C + c];
      }
    }
  }
  return valid;
}

// The work of one 256-thread chunk of warp_residual on workspace j: points [chunk * 256, chunk * 256 + 256), `s` the chunk's LDS
// scratch of the bracket step.  All 256 threads must call it.
// mode 0: every active workspace.  mode 1 (estimate loops with the fused path): skip workspaces whose scale is frozen
// for the rest of the level — no median is needed and irls_reduce recomputes their residuals itself.  mode 2: refresh
// the residual / valid buffers of workspaces marked r_stale from the pose of their last linearisation (T_lin).
template <int C, bool FAST>
__device__ __forceinline__ void warp_chunk(const PairJob& j, int mode, unsigned chunk, BracketLds& s)
{
  const GNState* __restrict__ st = j.st;
  if(mode == 2) { if(!st->r_stale) return; }
  else {
    if(!st->active) return;
    if(mode == 1 && !(st->delta_scale > 1e-6f)) return;
  }
  const int n = j.n;
  if((int) (chunk * K6_BLOCK) >= n) return;

  if(mode != 2 && chunk == 0 && threadIdx.x == 0) j.cnt[4] += (unsigned long long) n;   // points this kernel processes

  float P[12];
  if(FAST && j.dspace) dspace_matrix(j, mode == 2 ? st->T_lin : st->T, P);
  else projection_matrix(j, mode == 2 ? st->T_lin : st->T, P);

  // lanes past the end of the last block redo the last point (loads only) so that the whole block reaches the
  // block-level bracket step below; their stores are masked
  const int i_raw = chunk * K6_BLOCK + threadIdx.x;
  const bool in_block = i_raw < n;
  const int i = in_block ? i_raw : n - 1;
  float res[C];
  bool hit;
  const bool valid = warp_point<C, FAST>(j, P, i, in_block, res, hit);
  if(in_block) j.valid[i] = valid ? 1 : 0;
  if(in_block) {
    if constexpr(C == 8) {     // tiled residual record: two fully coalesced 16-byte stores per lane
      float4* o = reinterpret_cast<float4*>(j.r.get());
      store_stream(o + tile_index<2>(i, 0), make_float4(res[0], res[1], res[2], res[3]));
      store_stream(o + tile_index<2>(i, 1), make_float4(res[4], res[5], res[6], res[7]));
    } else {      // generic C: point-major records [N][C]
#pragma unroll
      for(int c = 0; c < C; ++c) j.r[(size_t) i * C + c] = res[c];
    }
  }
  // bracket pass of the exact median (see bracket_chunk) while the residuals are in registers
  if(mode != 2 && (st->delta_scale > 1e-6f) && st->median_valid)
    bracket_chunk<C>(j, st->lo_key, st->hi_key, valid && in_block, hit && valid && in_block, res, chunk, (int) (threadIdx.x >> 6), s, true);
}

template <int C, bool FAST>
__global__ __launch_bounds__(K6_BLOCK) void warp_residual_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int mode)
{
  __shared__ BracketLds s;
  warp_chunk<C, FAST>(jobs[active_workspace(act, blockIdx.y)], mode, blockIdx.x, s);
}

// clears r_stale after a refresh launch (one thread per workspace)
__global__ void clear_stale_kernel(const PairJob* jobs, int n)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if(p < n) jobs[p].st->r_stale = 0;
}

// ------------------------------------------------------------------------------------------------------------------
// K6' warp_residual for the other interpolation types of the standard PhotoError branch (bpvo/photo_error.cc:391-444):
// kCosine (2 x 2 taps, f32 coefficients from a double cosine), kCubic (4 x 4 taps, OpenCV-style cubic with A = -0.5)
// and kCubicHermite (4 x 4 taps, Bourke's Hermite form, bias = tension = 0).  The projection, Floor and the validity
// test are the f64 ones of kLinear with (border_lo, border_hi) = (0, 1) for kCosine and (1, 3) for the cubic ones
// (:347-348); the interpolation itself is f32.  The 4-tap forms read columns xi .. xi+3 and rows yi-1 .. yi+2 exactly as
// the reference addresses them; row yi+2 can be one past the image there (yi < rows-1 is all `valid` guarantees) —
// that row index is clamped to rows-1 here and in the oracle (Q21).  No tap cache: these are operator variants, not the
// benchmarked configuration; the taps are gathered straight from the pixel-interleaved descriptor.
__device__ __forceinline__ void interp_cosine(float x, float (&c)[2])
{
  const double m = (1.0 - cos((double) x * 3.14159265358979323846)) / 2.0;
  c[0] = (float) (1.0 - m);
  c[1] = (float) m;
}
__device__ __forceinline__ void interp_cubic(float x, float (&c)[4])
{
  const float A = -0.5f;
  c[0] = ((A * (x + 1.0f) - 5.0f * A) * (x + 1.0f) + 8.0f * A) * (x + 1.0f) - 4.0f * A;
  c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
  c[2] = ((A + 2.0f) * (1.0f - x) - (A + 3.0f)) * (1.0f - x) * (1.0f - x) + 1.0f;
  c[3] = 1.0f - c[0] - c[1] - c[2];
}
__device__ __forceinline__ float interp_hermite(float y0, float y1, float y2, float y3, float mu)
{
  const float mu2 = mu * mu;
  const float mu3 = mu * mu2;
  const float m0 = (float) (((double) (y1 - y0) / 2.0) + ((double) (y2 - y1) / 2.0));
  const float m1 = (float) (((double) (y2 - y1) / 2.0) + ((double) (y3 - y2) / 2.0));
  const float a0 = 2 * mu3 - 3 * mu2 + 1;
  const float a1 = mu3 - 2 * mu2 + mu;
  const float a2 = mu3 - mu2;
  const float a3 = -2 * mu3 + 3 * mu2;
  return a0 * y1 + a1 * m0 + a2 * m1 + a3 * y2;
}
// Eigen 3.2 fixed 4-float dot: one packet product reduced with haddps twice -> (a0 + a1) + (a2 + a3)
__device__ __forceinline__ float dot4(float a0, float a1, float a2, float a3, const float (&b)[4])
{
  return (a0 * b[0] + a1 * b[1]) + (a2 * b[2] + a3 * b[3]);
}

template <int C>
__global__ __launch_bounds__(K6_BLOCK) void warp_residual_interp_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int interp)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.y)];
  const GNState* __restrict__ st = j.st;
  if(!st->active) return;
  const int n = j.n;
  if((int) (blockIdx.x * K6_BLOCK) >= n) return;

  float P[12];
#pragma unroll
  for(int r = 0; r < 3; ++r)
#pragma unroll
    for(int c = 0; c < 4; ++c) {
      float s = j.K[r * 3 + 0] * st->T[0 * 4 + c];
      s += j.K[r * 3 + 1] * st->T[1 * 4 + c];
      s += j.K[r * 3 + 2] * st->T[2 * 4 + c];
      P[r * 4 + c] = s;
    }

  const int i_raw = blockIdx.x * K6_BLOCK + threadIdx.x;
  const bool in_block = i_raw < n;
  const int i = in_block ? i_raw : n - 1;
  const int W = j.cols, R = j.rows;
  const float4 X = j.pts[i];
  const bool two_tap = interp == BPVO_INTERP_COSINE;
  const int border_lo = two_tap ? 0 : 1, border_hi = two_tap ? 1 : 3;
  int xi = 0, yi = 0;
  const double X0 = (double) X.x, X1 = (double) X.y, X2 = (double) X.z, X3 = (double) X.w;
  double u[3];
#pragma unroll
  for(int r = 0; r < 3; ++r) {
    double s = (double) P[r * 4 + 0] * X0;
    s += (double) P[r * 4 + 1] * X1;
    s += (double) P[r * 4 + 2] * X2;
    s += (double) P[r * 4 + 3] * X3;
    u[r] = s;
  }
  const double zi = 1.0 / u[2];
  const double x = zi * u[0], y = zi * u[1];
  const bool in_range = (x > -2147483648.0) && (x < 2147483648.0) && (y > -2147483648.0) && (y < 2147483648.0);
  if(in_range) {
    xi = (int) x; xi -= (xi > x);
    yi = (int) y; yi -= (yi > y);
  }
  const bool valid = in_range && xi >= border_lo && xi < W - border_hi && yi >= border_lo && yi < R - 1;
  const float xf = (float) (x - (double) xi), yf = (float) (y - (double) yi);
  if(in_block) j.valid[i] = valid ? 1 : 0;

  float res[C];
#pragma unroll
  for(int c = 0; c < C; ++c) res[c] = 0.0f;
  if(valid) {
    float I0[C];
    if constexpr(C == 8) {
      const float4* p0 = reinterpret_cast<const float4*>(j.pix.get());
      const float4 t0 = p0[tile_index<2>(i, 0)], t1 = p0[tile_index<2>(i, 1)];
      I0[0] = t0.x; I0[1] = t0.y; I0[2] = t0.z; I0[3] = t0.w; I0[4] = t1.x; I0[5] = t1.y; I0[6] = t1.z; I0[7] = t1.w;
    } else {
#pragma unroll
      for(int c = 0; c < C; ++c) I0[c] = j.pix[(size_t) i * C + c];
    }
    if(two_tap) {
      float Cx[2], Cy[2];
      interp_cosine(xf, Cx);
      interp_cosine(yf, Cy);
      const float* __restrict__ d0 = j.desc + ((size_t) yi * W + xi) * C;
      const float* __restrict__ d1 = d0 + (size_t) W * C;
#pragma unroll
      for(int c = 0; c < C; ++c) {
        const float e1 = d0[c] * Cx[0] + d0[C + c] * Cx[1];
        const float e2 = d1[c] * Cx[0] + d1[C + c] * Cx[1];
        res[c] = (Cy[0] * e1 + Cy[1] * e2) - I0[c];
      }
    } else {
      const float* __restrict__ rowp[4];
#pragma unroll
      for(int k = 0; k < 4; ++k) rowp[k] = j.desc + ((size_t) min(yi - 1 + k, R - 1) * W + xi) * C;
      if(interp == BPVO_INTERP_CUBIC) {
        float Cx[4], Cy[4];
        interp_cubic(xf, Cx);
        interp_cubic(yf, Cy);
#pragma unroll
        for(int c = 0; c < C; ++c) {
          float d[4];
#pragma unroll
          for(int k = 0; k < 4; ++k) d[k] = dot4(rowp[k][c], rowp[k][C + c], rowp[k][2 * C + c], rowp[k][3 * C + c], Cx);
          res[c] = dot4(Cy[0], Cy[1], Cy[2], Cy[3], d) - I0[c];
        }
      } else {
#pragma unroll
        for(int c = 0; c < C; ++c) {
          float V[4];
#pragma unroll
          for(int k = 0; k < 4; ++k) V[k] = interp_hermite(rowp[k][c], rowp[k][C + c], rowp[k][2 * C + c], rowp[k][3 * C + c], xf);
          res[c] = interp_hermite(V[0], V[1], V[2], V[3], yf) - I0[c];
        }
      }
    }
  }
  if(in_block) {
    if constexpr(C == 8) {
      float4* o = reinterpret_cast<float4*>(j.r.get());
      o[tile_index<2>(i, 0)] = make_float4(res[0], res[1], res[2], res[3]);
      o[tile_index<2>(i, 1)] = make_float4(res[4], res[5], res[6], res[7]);
    } else {
#pragma unroll
      for(int c = 0; c < C; ++c) j.r[(size_t) i * C + c] = res[c];
    }
  }
  if((st->delta_scale > 1e-6f) && st->median_valid) bracket_block<C>(j, st->lo_key, st->hi_key, valid && in_block, false, res);
}

// ------------------------------------------------------------------------------------------------------------------
// K7 median + robust scale.  reference: AutoScaleEstimator::estimateScale / ScaleEstimator (bpvo/mestimator.cc:452-490)
// and median() (bpvo/utils.h:224-252): sigma = (1.4826f * (1 + 5/(n-6))) * median(|r| : valid), n = C * #valid (size_t
// arithmetic), sigma < 1e-6 -> 1, recomputed only while |sigma - sigma_prev| > 1e-6 (Q5, Q6).
//
// The order statistics x[n/2] (and x[n/2-1] for even n) are EXACT; they are found by MSB radix selection on the bit
// pattern of |r| (monotone for non-negative floats), with two cursors (lo, hi) refined in lock-step.  Two paths:
//
//  bracketed (every linearisation of a level but the first): the median moves little between GN iterations, so the
//    bracket step fused into warp_residual (bracket_block) only COUNTS the keys below a bracket [lo, hi) around the
//    previous median and compacts the few keys inside it into per-block candidate segments; K7b (median_finish_kernel)
//    then selects among the candidates only.  If the wanted ranks fall outside the bracket the full path runs — the
//    result is exact either way; the bracket width adapts to the last observed change.
//  full (first linearisation of a level, bracket miss): 3 passes over all keys, bits [30:20], [19:9], [8:0], one
//    workgroup per workspace (1024 threads, or 512 in launches wider than the chip: median_finish_kernel) with LDS histograms
//    (privatised copies in pass 1 to cut same-bin atomic serialisation); keys surviving pass 1 are cached in LDS so pass 3
//    never touches HBM again.
constexpr int MED_THREADS = 1024;     // median_finish_kernel; the persistent kernel runs the same code with 512 (template parameter NT)
constexpr int MED_COPIES = 4;
constexpr int MED_BINS = 2048;
constexpr int MED_CACHE = 20480;

struct MedCursor { unsigned prefix; unsigned rank; };

template <int NT = 1024>
__device__ __forceinline__ unsigned block_excl_scan_1024(unsigned v, unsigned* s_wave /*[16]*/, unsigned& total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = v;
#pragma unroll
  for(int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o);
    if(lane >= o) incl += t;
  }
  __syncthreads();
  if(lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned woff = 0, tot = 0;
#pragma unroll
  for(int w = 0; w < NT / 64; ++w) {
    const unsigned t = s_wave[w];
    if(w < wave) woff += t;
    tot += t;
  }
  total = tot;
  return woff + incl - v;
}

// every thread owns BPT = 2048 / NT consecutive bins (2t, 2t+1 for 1024 threads) of a (<= 2048)-bin histogram, h[] their counts,
// excl the number of keys in the bins before them: find the bins holding ranks k_lo / k_hi
template <int BPT>
__device__ __forceinline__ void find_ranks(const unsigned (&h)[BPT], unsigned excl, unsigned k_lo, unsigned k_hi, MedCursor* out /*[2]*/)
{
  unsigned b = (unsigned) BPT * threadIdx.x, e = excl;
#pragma unroll
  for(int q = 0; q < BPT; ++q) {
    if(k_lo >= e && k_lo < e + h[q]) { out[0].prefix = b + q; out[0].rank = k_lo - e; }
    if(k_hi >= e && k_hi < e + h[q]) { out[1].prefix = b + q; out[1].rank = k_hi - e; }
    e += h[q];
  }
}

template <int C, int NT, typename F>
__device__ __forceinline__ void for_each_valid_key(const PairJob& j, F f)
{
  const int n = j.n;
  if constexpr(C == 8) {
    const float4* q = reinterpret_cast<const float4*>(j.r.get());
    constexpr int U = 4;   // points in flight per thread: all loads of a round are issued before any is consumed
    for(int base = threadIdx.x; base < n; base += NT * U) {
      unsigned char v[U];
      float4 a[U], b[U];
#pragma unroll
      for(int u = 0; u < U; ++u) {
        const int pt = base + u * NT;
        const bool in = pt < n;
        v[u] = in ? j.valid[pt] : (unsigned char) 0;
        a[u] = in ? load_stream(q + tile_index<2>(pt, 0)) : make_float4(0, 0, 0, 0);
        b[u] = in ? load_stream(q + tile_index<2>(pt, 1)) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for(int u = 0; u < U; ++u) {
        if(!v[u]) continue;
        const int pt = base + u * NT;
        f(__float_as_uint(a[u].x) & 0x7fffffffu, pt); f(__float_as_uint(a[u].y) & 0x7fffffffu, pt);
        f(__float_as_uint(a[u].z) & 0x7fffffffu, pt); f(__float_as_uint(a[u].w) & 0x7fffffffu, pt);
        f(__float_as_uint(b[u].x) & 0x7fffffffu, pt); f(__float_as_uint(b[u].y) & 0x7fffffffu, pt);
        f(__float_as_uint(b[u].z) & 0x7fffffffu, pt); f(__float_as_uint(b[u].w) & 0x7fffffffu, pt);
      }
    }
  } else if constexpr(C != 1) {   // generic C: point-major records
    for(int pt = threadIdx.x; pt < n; pt += NT) {
      if(!j.valid[pt]) continue;
#pragma unroll
      for(int c = 0; c < C; ++c) f(__float_as_uint(j.r[(size_t) pt * C + c]) & 0x7fffffffu, pt);
    }
  } else {
    for(int p4 = threadIdx.x * 4; p4 < n; p4 += NT * 4) {   // n is a multiple of 16
      const uchar4 v = *reinterpret_cast<const uchar4*>(j.valid + p4);
      const float4 a = *reinterpret_cast<const float4*>(j.r + p4);
      if(v.x) f(__float_as_uint(a.x) & 0x7fffffffu, p4 + 0);
      if(v.y) f(__float_as_uint(a.y) & 0x7fffffffu, p4 + 1);
      if(v.z) f(__float_as_uint(a.z) & 0x7fffffffu, p4 + 2);
      if(v.w) f(__float_as_uint(a.w) & 0x7fffffffu, p4 + 3);
    }
  }
}

// One refinement pass of the two-cursor radix select.  A key takes part in cursor X iff its bits above (shift + width)
// equal X.prefix; its digit is (key >> shift) & (2^width - 1), width <= 11.  On return the cursors carry the extended
// prefix and the rank inside the selected digit bin.  Block-wide (1024 threads); `src(f)` calls f(key) for every key.
template <int NT, typename Src>
__device__ __forceinline__ void refine_pass(Src&& src, unsigned shift, unsigned width, MedCursor& lo, MedCursor& hi, unsigned* hist_lo,
                                            unsigned* hist_hi, unsigned* s_wave, MedCursor* cur)
{
  const int tid = threadIdx.x;
  const bool split = lo.prefix != hi.prefix;
  const unsigned nbins = 1u << width, up = shift + width;
  __syncthreads();
  for(unsigned i = tid; i < nbins; i += NT) { hist_lo[i] = 0; hist_hi[i] = 0; }
  __syncthreads();
  src([&](unsigned key) {
    const unsigned top = (up >= 32u) ? 0u : (key >> up);
    const unsigned dg = (key >> shift) & (nbins - 1u);
    if(top == lo.prefix) atomicAdd(&hist_lo[dg], 1u);
    else if(split && top == hi.prefix) atomicAdd(&hist_hi[dg], 1u);
  });
  __syncthreads();
  unsigned dummy;
  constexpr int BPT = MED_BINS / NT;
  const bool own = (unsigned) BPT * tid < nbins;        // nbins is a power of two >= BPT or smaller than it: bins past nbins read as 0
  unsigned ha[BPT], hb[BPT], sa = 0, sb = 0;
#pragma unroll
  for(int q = 0; q < BPT; ++q) { ha[q] = (own && (unsigned) BPT * tid + q < nbins) ? hist_lo[BPT * tid + q] : 0u; sa += ha[q]; }
  const unsigned ea = block_excl_scan_1024<NT>(sa, s_wave, dummy);
  unsigned eb = ea;
  if(split) {
#pragma unroll
    for(int q = 0; q < BPT; ++q) { hb[q] = (own && (unsigned) BPT * tid + q < nbins) ? hist_hi[BPT * tid + q] : 0u; sb += hb[q]; }
    eb = block_excl_scan_1024<NT>(sb, s_wave, dummy);
  }
  MedCursor tmp[2];
  tmp[0].prefix = 0xffffffffu; tmp[1].prefix = 0xffffffffu; tmp[0].rank = tmp[1].rank = 0;
  if(own) {
    find_ranks<BPT>(ha, ea, lo.rank, split ? 0xffffffffu : hi.rank, tmp);
    if(tmp[0].prefix != 0xffffffffu) { cur[0].prefix = (lo.prefix << width) | tmp[0].prefix; cur[0].rank = tmp[0].rank; }
    if(!split && tmp[1].prefix != 0xffffffffu) { cur[1].prefix = (hi.prefix << width) | tmp[1].prefix; cur[1].rank = tmp[1].rank; }
    if(split) {
      tmp[1].prefix = 0xffffffffu;
      find_ranks<BPT>(hb, eb, 0xffffffffu, hi.rank, tmp);
      if(tmp[1].prefix != 0xffffffffu) { cur[1].prefix = (hi.prefix << width) | tmp[1].prefix; cur[1].rank = tmp[1].rank; }
    }
  }
  __syncthreads();
  lo = cur[0];
  hi = cur[1];
  __syncthreads();
}

// The work of one NT-thread workgroup on workspace j; `st` is the state it reads and (thread 0, at the very end) updates — the
// workspace's own in HBM, or a workgroup-local copy (persistent kernel, where every workgroup runs the selection redundantly and
// `stats` is true for one of them only).
// COPIES privatised pass-1 histograms (a power of two >= 2: the second one doubles as the segment-offset table of the bracketed path),
// CACHE words of key cache: the LDS footprint is ((COPIES + 1) * MED_BINS + CACHE + 24) words.
template <int C, int NT, int COPIES = MED_COPIES, int CACHE = MED_CACHE>
__device__ __forceinline__ void median_block(const PairJob& j, GNState* st, unsigned char* smem_raw, bool stats)
{
  unsigned* hist_lo = reinterpret_cast<unsigned*>(smem_raw);              // [COPIES][MED_BINS]
  unsigned* hist_hi = hist_lo + COPIES * MED_BINS;                    // [MED_BINS]
  unsigned* cache = hist_hi + MED_BINS;                                   // [CACHE]
  unsigned* s_wave = cache + CACHE;                                   // [16]
  MedCursor* cur = reinterpret_cast<MedCursor*>(s_wave + 16);             // [2]
  unsigned* s_misc = reinterpret_cast<unsigned*>(cur + 2);                // [0] cache count, [1] first valid point

  const int tid = threadIdx.x;
  float median = 0.0f;
  unsigned n_total = 0;
  bool done = false;
  unsigned tap_hits = 0, tap_lookups = 0;      // tap-cache statistics of this linearisation's warp_residual pass (bracket counters)

  // ---- bracketed path
  if(st->median_valid) {
    // totals of the per-block counters written by the bracket step of warp_residual (bracket_block)
    const int nblk = (j.n + K6_BLOCK - 1) / K6_BLOCK;
    unsigned c_below = 0, c_in = 0, c_valid = 0, c_hit = 0, cnt_first = 0;    // cnt_first: candidates of segment `tid`
    for(int b = tid; b < nblk; b += NT) {
      const uint4 o = reinterpret_cast<const uint4*>(j.med_blk.get())[b];
      if(b == tid) cnt_first = o.y;
      c_below += o.x; c_in += o.y; c_valid += o.z; c_hit += o.w;
    }
    unsigned t_below, t_in, t_valid;
    {   // four block sums with one LDS round
#pragma unroll
      for(int o = 32; o >= 1; o >>= 1) {
        c_below += __shfl_down(c_below, o);
        c_in += __shfl_down(c_in, o);
        c_valid += __shfl_down(c_valid, o);
        c_hit += __shfl_down(c_hit, o);
      }
      __syncthreads();
      if((tid & 63) == 0) { unsigned* w4 = cache + (tid >> 6) * 4; w4[0] = c_below; w4[1] = c_in; w4[2] = c_valid; w4[3] = c_hit; }
      __syncthreads();
      t_below = t_in = t_valid = 0;
#pragma unroll
      for(int w = 0; w < NT / 64; ++w) { t_below += cache[w * 4 + 0]; t_in += cache[w * 4 + 1]; t_valid += cache[w * 4 + 2]; tap_hits += cache[w * 4 + 3]; }
      tap_lookups = j.tapcache_on ? t_valid : 0u;      // (no tap cache at dense levels: nothing looked up)
      __syncthreads();
    }
    const unsigned nt = (unsigned) C * t_valid, below = t_below, m = t_in;
    const unsigned lo_key = st->lo_key, range = st->hi_key - st->lo_key;
    const unsigned k_hi = nt / 2, k_lo = (nt % 2 == 0 && nt > 0) ? k_hi - 1 : k_hi;
    if(nt >= 3 && k_lo >= below && k_hi < below + m && range > 0) {
      MedCursor lo, hi;
      lo.prefix = 0; hi.prefix = 0; lo.rank = k_lo - below; hi.rank = k_hi - below;
      const unsigned nbits = 32u - (unsigned) __clz((int) range);        // offsets d = key - lo_key are < range < 2^nbits
      // The candidates sit in per-block segments of 256*C slots.  They are first gathered into LDS as one dense run (offsets d): a
      // flat index f over all candidates is mapped to (segment, slot) through the exclusive scan of the segment counts, so that every
      // thread has several independent loads in flight — walking the segments one after the other costs two dependent global
      // latencies per segment and wave, twice (histogram pass, ranking pass), which was most of this path's time.  Too many
      // candidates or segments for the LDS areas: the segment walk from global memory (same keys, same result).
      unsigned* s_off = hist_lo + MED_BINS;                              // [nblk + 1] — refine_pass only uses the first MED_BINS words of hist_lo
      constexpr unsigned kListRoom = 2u * (unsigned) NT;                  // cache[0 .. 2 NT): lists of the ranking step
      unsigned* dense = cache + kListRoom;
      const bool in_lds = m <= (unsigned) CACHE - kListRoom && nblk < (COPIES - 1) * MED_BINS;
      if(in_lds) {
        unsigned run = 0;                                                  // running offset of the chunks of NT segments
        for(int b0 = 0; b0 < nblk; b0 += NT) {
          const int b = b0 + tid;
          const unsigned cnt = b0 == 0 ? cnt_first : (b < nblk ? reinterpret_cast<const uint4*>(j.med_blk.get())[b].y : 0u);
          unsigned total;
          const unsigned off = block_excl_scan_1024<NT>(cnt, s_wave, total);
          if(b < nblk) s_off[b] = run + off;
          run += total;
          __syncthreads();
        }
        if(tid == 0) s_off[nblk] = m;
        __syncthreads();
        constexpr int U = 4;
        int b = 0;                                                         // segment of the thread's current flat index (flat indices grow)
        for(unsigned f0 = tid; f0 < m; f0 += (unsigned) NT * U) {
          unsigned v[U];
#pragma unroll
          for(int u = 0; u < U; ++u) {
            const unsigned f = f0 + (unsigned) u * NT;
            if(f < m) {
              // first segment whose end lies beyond f: gallop, then bisect
              int step = 1, lo_b = b;
              while(lo_b + step < nblk && s_off[lo_b + step] <= f) { lo_b += step; step <<= 1; }
              int hi_b = min(lo_b + step, nblk);                           // s_off[lo_b] <= f < s_off[hi_b]
              while(hi_b - lo_b > 1) { const int mid = (lo_b + hi_b) >> 1; if(s_off[mid] <= f) lo_b = mid; else hi_b = mid; }
              b = lo_b;
              v[u] = j.cand[(size_t) b * K6_BLOCK * C + (f - s_off[b])];
            }
          }
#pragma unroll
          for(int u = 0; u < U; ++u) {
            const unsigned f = f0 + (unsigned) u * NT;
            if(f < m) dense[f] = v[u] - lo_key;
          }
        }
        __syncthreads();
      }
      auto src = [&](auto f) {
        if(in_lds) {
          for(unsigned i = tid; i < m; i += NT) f(dense[i]);
          return;
        }
        const int lane = tid & 63, wave = tid >> 6;
        for(int b = wave; b < nblk; b += NT / 64) {
          const unsigned mb = reinterpret_cast<const uint4*>(j.med_blk.get())[b].y;
          const unsigned* seg = j.cand + (size_t) b * K6_BLOCK * C;
          for(unsigned i = lane; i < mb; i += 64) f(seg[i] - lo_key);
        }
      };
      unsigned remaining = nbits;
      bool first = true;
      while(remaining > 0) {
        const unsigned width = remaining > 11u ? 11u : remaining;
        const bool was_split = lo.prefix != hi.prefix;
        remaining -= width;
        refine_pass<NT>(src, remaining, width, lo, hi, hist_lo, hist_hi, s_wave, cur);
        if(!first || remaining == 0) continue;
        first = false;
        // After the first digit the selected bins usually hold a handful of keys: finish by direct ranking (each thread
        // ranks one key of the bin by counting the smaller ones) instead of more histogram passes.
        const unsigned dmask = (1u << width) - 1u;
        const unsigned n_lo = hist_lo[lo.prefix & dmask];
        const unsigned n_hi = was_split ? hist_hi[hi.prefix & dmask] : hist_lo[hi.prefix & dmask];
        if(n_lo > (unsigned) NT || n_hi > (unsigned) NT) continue;
        unsigned* list_lo = cache;                    // [NT]
        unsigned* list_hi = cache + NT;               // [NT]
        const bool same_bin = lo.prefix == hi.prefix;
        __syncthreads();
        if(tid == 0) { s_misc[0] = 0; s_misc[1] = 0; }
        __syncthreads();
        const unsigned p_lo = lo.prefix, p_hi = hi.prefix, sh = remaining;
        src([&](unsigned d) {
          const unsigned top = d >> sh;
          if(top == p_lo) list_lo[atomicAdd(&s_misc[0], 1u)] = d;
          else if(!same_bin && top == p_hi) list_hi[atomicAdd(&s_misc[1], 1u)] = d;
        });
        __syncthreads();
        // rank of list[t] = #{smaller} + #{equal with smaller index}; exactly one element has the wanted rank
        auto pick = [&](const unsigned* list, unsigned cnt, unsigned want, unsigned* out) {
          if((unsigned) tid < cnt) {
            const unsigned mine = list[tid];
            unsigned rk = 0;
            for(unsigned q = 0; q < cnt; ++q) {
              const unsigned o = list[q];
              rk += (o < mine || (o == mine && q < (unsigned) tid)) ? 1u : 0u;
            }
            if(rk == want) *out = mine;
          }
        };
        pick(list_lo, n_lo, lo.rank, &cur[0].prefix);
        if(same_bin) pick(list_lo, n_lo, hi.rank, &cur[1].prefix);
        else pick(list_hi, n_hi, hi.rank, &cur[1].prefix);
        __syncthreads();
        lo.prefix = cur[0].prefix; hi.prefix = cur[1].prefix;     // full offsets d now
        __syncthreads();
        remaining = 0;
      }
      const float v_lo = __uint_as_float(lo_key + lo.prefix), v_hi = __uint_as_float(lo_key + hi.prefix);
      median = (nt % 2 != 0) ? v_hi : (float) (((double) (v_lo + v_hi)) / 2.0);
      n_total = nt;
      done = true;
    }
  }

  // ---- full path
  if(!done) {
    for(int i = tid; i < (COPIES + 1) * MED_BINS; i += NT) hist_lo[i] = 0;
    if(tid == 0) { s_misc[0] = 0; s_misc[1] = 0xffffffffu; }
    __syncthreads();
    {   // pass 1: bits [30:20], privatised histogram copies
      unsigned* h = hist_lo + (tid & (COPIES - 1)) * MED_BINS;
      unsigned first = 0xffffffffu;
      for_each_valid_key<C, NT>(j, [&](unsigned key, int pt) {
        atomicAdd(&h[key >> 20], 1u);
        first = min(first, (unsigned) pt);
      });
      if(C == 1 && first != 0xffffffffu) atomicMin(&s_misc[1], first);
    }
    __syncthreads();
    constexpr int BPT = MED_BINS / NT;
    unsigned hh[BPT], hsum = 0;
#pragma unroll
    for(int q = 0; q < BPT; ++q) {
      hh[q] = 0;
#pragma unroll
      for(int c = 0; c < COPIES; ++c) hh[q] += hist_lo[c * MED_BINS + BPT * tid + q];
      hsum += hh[q];
    }
    const unsigned excl = block_excl_scan_1024<NT>(hsum, s_wave, n_total);
    if(n_total >= 3) {
      const unsigned k_hi = n_total / 2, k_lo = (n_total % 2 == 0) ? k_hi - 1 : k_hi;
      find_ranks<BPT>(hh, excl, k_lo, k_hi, cur);
      __syncthreads();
      MedCursor lo = cur[0], hi = cur[1];
      __syncthreads();
      // pass 2: bits [19:9] of the keys in the selected pass-1 bucket(s); survivors cached in LDS
      const unsigned p_lo = lo.prefix, p_hi = hi.prefix;
      refine_pass<NT>([&](auto f) {
        for_each_valid_key<C, NT>(j, [&](unsigned key, int) {
          const unsigned top = key >> 20;
          if(top == p_lo || top == p_hi) {
            const unsigned idx = atomicAdd(&s_misc[0], 1u);
            if(idx < CACHE) cache[idx] = key;
          }
          f(key);
        });
      }, 9u, 11u, lo, hi, hist_lo, hist_hi, s_wave, cur);
      const unsigned ncache = s_misc[0];
      // pass 3: bits [8:0]
      refine_pass<NT>([&](auto f) {
        if(ncache <= CACHE) { for(unsigned i = tid; i < ncache; i += NT) f(cache[i]); }
        else for_each_valid_key<C, NT>(j, [&](unsigned key, int) { f(key); });
      }, 0u, 9u, lo, hi, hist_lo, hist_hi, s_wave, cur);
      const float v_lo = __uint_as_float(lo.prefix), v_hi = __uint_as_float(hi.prefix);
      median = (n_total % 2 != 0) ? v_hi : (float) (((double) (v_lo + v_hi)) / 2.0);   // (*m + *middle) / 2.0, utils.h:236
    } else if(n_total > 0) {
      // median(): data.size() < 3 -> data[0] = first valid entry in channel-major order (Q5); only reachable for C == 1
      __syncthreads();
      const unsigned first = s_misc[1];
      median = (first != 0xffffffffu) ? fabsf(j.r[(size_t) first * C]) : 0.0f;
    }
  }

  if(tid == 0) {
    if(stats) {
      j.cnt[done ? 2 : 3] += 1ull;                                          // measurement: bracketed vs full selections
      // tap cache: a linearisation without bracket counters is the first of a level (keys reset: no hits, every valid point looks up)
      if(!st->median_valid) tap_lookups = j.tapcache_on ? n_total / (unsigned) C : 0u;
      j.cnt[5] += tap_hits; j.cnt[6] += tap_lookups;
      if(st->num_fun_evals < 8) { j.cnt[7] += tap_hits; j.cnt[8] += tap_lookups; }
    }
    const unsigned long long nm6 = (unsigned long long) n_total - 6ull;     // size_t wrap for n < 6 (Q5)
    float s = (1.4826f * (1.0f + 5.0f / (float) nm6)) * median;
    if((double) s < 1e-6) s = 1.0f;
    st->delta_scale = fabsf(s - st->scale);
    st->scale = s;
    // bracket for the next linearisation of this level: centred on this median, as wide as 2.5x the last relative
    // change + 2 % (first use: 25 %), at most 50 %
    if(n_total >= 3 && median > 0.0f) {
      float rel = 0.25f;
      if(st->last_median > 0.0f) rel = fminf(0.5f, fmaxf(0.02f, 2.5f * fabsf(median - st->last_median) / st->last_median + 0.02f));
      st->last_median = median;
      st->lo_key = __float_as_uint(median * (1.0f - rel));
      st->hi_key = __float_as_uint(median * (1.0f + rel)) + 1u;
      st->median_valid = 1;
    } else {
      st->median_valid = 0;
    }
  }
}

// K7b: one workgroup per workspace — bracketed select among the candidates, or the full 3-pass select.
// Two shapes.  1024 threads with the full 123 KB (one workgroup per CU): the fastest single selection — launches of up to one
// workgroup per CU.  512 threads with 53 KB (three workgroups per CU): launches of MORE workgroups
// than CUs, which with the first shape run in waves of 256 workgroups at ~10 us each (1024 pairs: 41 / 32 / 22 / 12 us per launch as
// the pairs converge).  The selection is exact in either shape.
constexpr int MED_THREADS_B = 512, MED_COPIES_B = 2, MED_CACHE_B = 7168;   // 53 KB: three workgroups per CU
template <int C, int NT, int COPIES, int CACHE>
__global__ __launch_bounds__(NT) void median_finish_kernel(const PairJob* __restrict__ jobs, ActiveSet act)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.x)];
  GNState* st = j.st;
  if(!st->active) return;
  if(!(st->delta_scale > 1e-6f)) return;   // scale is stable: frozen for the rest of the level (mestimator.cc:472,485)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  median_block<C, NT, COPIES, CACHE>(j, st, smem_raw, true);
}

// ------------------------------------------------------------------------------------------------------------------
// K8 irls_reduce.  reference: MEstimator::ComputeWeights SIMD bodies (bpvo/mestimator.cc:242-282 Huber, :303-366 Tukey;
// `valid` ignored there, Q12/Q14) fused with LinearSystemBuilderReduction::rankUpdatePoint
// (bpvo/linear_system_builder.cc:140-205): w' = w * float(valid); H += (w' J_a) J_b (upper triangle); G += (w' r) J;
// e += (w' r) r.  Summation order differs from the reference's serial loop (Q15): per thread over channels and points,
// then a wavefront shuffle tree, then LDS across the 4 waves; per-block partials are combined in fixed order in f64 by
// gn_step, so the result is deterministic run to run.
constexpr int kNumAcc = 30;   // 21 H + 6 G + e + #valid points + tap-cache hits (fused path)

template <int LOSS>
__device__ __forceinline__ float mest_weight(float r, float sigma_inv)
{
  if(LOSS == BPVO_LOSS_HUBER) {
    const float k = 1.345f;
    const float x = fabsf(r * sigma_inv);
    return k / fmaxf(x, k);
  } else if(LOSS == BPVO_LOSS_TUKEY) {
    const float t = 4.685f;
    const float t_i = (float) (1.0 / 4.685f);
    const float x = r * sigma_inv;
    float q = x * t_i;
    q = 1.0f - q * q;
    q = q * q;
    return (fabsf(x) < t) ? q : 0.0f;
  }
  return 1.0f;
}

// the work of one workgroup of irls_reduce on workspace j
// `tile` is the run of pts_per_block points (the blockIdx.x of irls_reduce), `vtid` the thread's index among the 256 that share the
// tile, `s_part` their LDS scratch.  `has` = false: a tile past the end whose threads only keep in step (persistent kernel); all
// threads of the WORKGROUP must call the function (it holds a __syncthreads).
typedef float IrlsPartLds[4][kPartialStride];
template <int C, int LOSS, bool FUSED>
__device__ __forceinline__ void irls_tile(const PairJob& j, const GNState* __restrict__ st, int pts_per_block, int tile, int vtid,
                                          IrlsPartLds& s_part, bool has, float* __restrict__ partials)
{
  // Fused path (C = 8): the robust scale is frozen for the rest of the level, so nothing separates the residuals from
  // their weights any more — they are recomputed here exactly as warp_residual does (same warp_point, same tap cache) and
  // never written: the r write + read, the second point read and the valid byte (82 of 341 B per point and iteration)
  // disappear.  Same values, same accumulation order as the two-kernel form.
  constexpr bool fused = FUSED && (C == 8);
  float P[12];
  if constexpr(fused) {
    projection_matrix(j, st->T, P);
    // uniform over the workgroup: pin the 12 values to scalar registers (the vector budget decides the occupancy here)
#pragma unroll
    for(int k = 0; k < 12; ++k) P[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(P[k])));
  }
  const int n = j.n;
  const int p_begin = tile * pts_per_block;
  const int p_end = has ? min(n, p_begin + pts_per_block) : p_begin;
  const float sigma_inv = 1.0f / st->scale;
  const float s_nrm[4] = {j.nrm[0], j.nrm[1], j.nrm[2], j.nrm[3]};
  const bool dspace = j.dspace != 0;      // uniform over the launch
  const float ds_fx = j.K[0], ds_fy = j.K[4], ds_fx_i = 1.0f / j.K[0], ds_fy_i = 1.0f / j.K[4], ds_b_i = 1.0f / j.b;

  float acc[kNumAcc];
#pragma unroll
  for(int k = 0; k < kNumAcc; ++k) acc[k] = 0.0f;

  for(int i = p_begin + vtid; i < p_end; i += GN_BLOCK) {
    float rr[C], Ix[C], Iy[C];
    float v;
    if constexpr(fused) {
      bool hit;
      v = warp_point<8, false, true>(j, P, i, true, rr, hit) ? 1.0f : 0.0f;
      acc[29] += hit ? 1.0f : 0.0f;
    } else {
      v = (float) j.valid[i];
    }
    acc[28] += v;
    // per point: 16 B point + 2*C gradient floats + C residuals (all tiled / coalesced), ALL issued before the first use
    // (7 independent 16-byte loads in flight per lane for C = 8).
    //
    // Rank-2 structure: every channel's 1x6 Jacobian row at a point is J_c = Ix_c * A + Iy_c * B with A, B depending on
    // the point only (jac_row in types.h expanded in Ix, Iy).  Hence
    //    sum_c w_c J_c^T J_c = Sxx A A^T + Sxy (A B^T + B A^T) + Syy B B^T,   sum_c w_c r_c J_c^T = Gx A + Gy B
    // with the channel sums Sxx = sum w Ix^2, Sxy = sum w Ix Iy, Syy = sum w Iy^2, Gx = sum w r Ix, Gy = sum w r Iy.
    // Per (point, channel) that is 6 multiply-adds instead of the 27 of the reference's rankUpdatePoint; the 6x6 outer
    // products are formed once per point.  Algebraically identical, rounding differs at the 1e-7 level like any other
    // summation order (H, G are tolerance-compared).
    const float4 Pt = load_stream(j.pts + i);
    if constexpr(C == 8) {
      const float4* qr = reinterpret_cast<const float4*>(j.r.get());
      const float4* qg = reinterpret_cast<const float4*>(j.grad.get());
      if constexpr(!fused) {
        const float4 r0 = load_stream(qr + tile_index<2>(i, 0)), r1 = load_stream(qr + tile_index<2>(i, 1));
        rr[0] = r0.x; rr[1] = r0.y; rr[2] = r0.z; rr[3] = r0.w; rr[4] = r1.x; rr[5] = r1.y; rr[6] = r1.z; rr[7] = r1.w;
      }
      const float4 gx0 = load_stream(qg + tile_index<4>(i, 0)), gx1 = load_stream(qg + tile_index<4>(i, 1)),
                   gy0 = load_stream(qg + tile_index<4>(i, 2)), gy1 = load_stream(qg + tile_index<4>(i, 3));
      Ix[0] = gx0.x; Ix[1] = gx0.y; Ix[2] = gx0.z; Ix[3] = gx0.w; Ix[4] = gx1.x; Ix[5] = gx1.y; Ix[6] = gx1.z; Ix[7] = gx1.w;
      Iy[0] = gy0.x; Iy[1] = gy0.y; Iy[2] = gy0.z; Iy[3] = gy0.w; Iy[4] = gy1.x; Iy[5] = gy1.y; Iy[6] = gy1.z; Iy[7] = gy1.w;
    } else if constexpr(C == 1) {
      rr[0] = j.r[i];
      const float2 g2 = reinterpret_cast<const float2*>(j.grad.get())[i];
      Ix[0] = g2.x; Iy[0] = g2.y;
    } else {      // generic C: point-major r[N][C], grad[N][2][C]
#pragma unroll
      for(int c = 0; c < C; ++c) {
        rr[c] = j.r[(size_t) i * C + c];
        Ix[c] = j.grad[((size_t) i * 2 + 0) * C + c];
        Iy[c] = j.grad[((size_t) i * 2 + 1) * C + c];
      }
    }
    float Sxx = 0.0f, Sxy = 0.0f, Syy = 0.0f, Gx = 0.0f, Gy = 0.0f;
#pragma unroll
    for(int c = 0; c < C; ++c) {
      const float r = rr[c];
      const float w = mest_weight<LOSS>(r, sigma_inv) * v;
      const float wx = w * Ix[c], wy = w * Iy[c];
      Sxx += wx * Ix[c];
      Sxy += wx * Iy[c];
      Syy += wy * Iy[c];
      Gx += wx * r;
      Gy += wy * r;
      acc[27] += (w * r) * r;
    }
    float A[6], B[6];
    if(!dspace) {
      const JacPoint jp = jac_point(Pt.x, Pt.y, Pt.z, s_nrm);
      const float t_xz2 = jp.x * jp.rz2, t_yz2 = jp.y * jp.rz2;
      A[0] = -(t_xz2 * jp.yc2); A[1] = jp.zc3 * jp.rz + t_xz2 * jp.xc1; A[2] = -(jp.yc2 * jp.rz); A[3] = jp.rzs; A[4] = 0.0f; A[5] = -(jp.s_i * t_xz2);
      B[0] = -(jp.zc3 * jp.rz) - t_yz2 * jp.yc2; B[1] = t_yz2 * jp.xc1; B[2] = jp.xc1 * jp.rz; B[3] = 0.0f; B[4] = jp.rzs; B[5] = -(jp.s_i * t_yz2);
    } else {
      // DisparitySpaceWarp::jacobian (types.h dspace_jac_row) expanded in the raw gradients Ix, Iy; point = (x, y, d, 1)
      const float x = Pt.x, y = Pt.y, d = Pt.z;
      const float xfi = x * ds_fx_i, yfi = y * ds_fy_i, dbi = d * ds_b_i;
      A[0] = -(x * yfi); A[1] = ds_fx + x * xfi; A[2] = -(ds_fx * yfi); A[3] = dbi; A[4] = 0.0f; A[5] = -(dbi * xfi);
      B[0] = -ds_fy - y * yfi; B[1] = y * xfi; B[2] = ds_fy * xfi; B[3] = 0.0f; B[4] = dbi * (ds_fy * ds_fx_i); B[5] = -(dbi * (y * ds_fx_i));
    }
    {
      int idx = 0;
#pragma unroll
      for(int a = 0; a < 6; ++a) {
        const float pa = Sxx * A[a] + Sxy * B[a];      // coefficient of A[b]
        const float qa = Sxy * A[a] + Syy * B[a];      // coefficient of B[b]
#pragma unroll
        for(int b = a; b < 6; ++b) acc[idx++] += pa * A[b] + qa * B[b];
      }
#pragma unroll
      for(int a = 0; a < 6; ++a) acc[21 + a] += Gx * A[a] + Gy * B[a];
    }
  }

  // wavefront tree (64 lanes; the compiler lowers these shuffles to DPP adds — a reduce-scatter over ds_bpermute was 2.5x
  // slower), then LDS across the 4 waves
#pragma unroll
  for(int k = 0; k < kNumAcc; ++k) {
    float v = acc[k];
#pragma unroll
    for(int o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o);
    acc[k] = v;
  }
  const int lane = vtid & 63, wave = vtid >> 6;
  if(lane == 0) {
#pragma unroll
    for(int k = 0; k < kNumAcc; ++k) s_part[wave][k] = acc[k];
  }
  __syncthreads();
  if(vtid < kNumAcc && has) {
    const float v = (s_part[0][vtid] + s_part[1][vtid]) + (s_part[2][vtid] + s_part[3][vtid]);
    partials[(size_t) tile * kPartialStride + vtid] = v;
  }
}

// the form irls_reduce uses: one 256-thread workgroup = one tile
template <int C, int LOSS, bool FUSED>
__device__ __forceinline__ void irls_block(const PairJob& j, const GNState* __restrict__ st, int pts_per_block)
{
  if((int) blockIdx.x * pts_per_block >= j.n) return;
  __shared__ IrlsPartLds s_part;
  irls_tile<C, LOSS, FUSED>(j, st, pts_per_block, blockIdx.x, threadIdx.x, s_part, true, j.partials);
}

// irls_tile for LATENCY-bound launches (persistent kernel, C = 8): the same per-point arithmetic and the same accumulation order
// (a thread's points in ascending order, then the wave tree, then the four waves), but a thread handles its points two at a time
// and requests EVERYTHING both need — point, tap-cache key, the eight cached tap vectors, template pixels, gradients (fused
// path); point, valid byte, residuals, gradients (plain) — before the first use: one memory round trip per pair of points instead
// of four or five dependent ones per point (point -> projection -> key -> taps, in two halves).  The cached taps are loaded
// speculatively: on a miss (3 % of the lookups) they are discarded and the footprint is gathered as usual.  The throughput
// kernels do the opposite on purpose — there the speculative bytes cost more than the latency they hide.  Cached
// accesses instead of the streaming ones for the same reason: a single pair's working set stays in the L2s between iterations.
struct IrlsPointLat {
  float4 Pt, tc[8], px[2], g[4], r[2];
  unsigned key;
  float v;
};
template <bool FUSED>
__device__ __forceinline__ void irls_lat_load(const PairJob& j, int i, IrlsPointLat& d)
{
  d.Pt = load_v4<false>(j.pts + i);
  const float4* qg = reinterpret_cast<const float4*>(j.grad.get());
  if constexpr(FUSED) {
    if(j.tapcache_on) {      // (uniform over the workspace: the dense levels of a batch gather straight from the descriptor)
      d.key = j.tapkey[i];
      const float4* tc = reinterpret_cast<const float4*>(j.tapcache.get());
#pragma unroll
      for(int k = 0; k < 8; ++k) d.tc[k] = load_v4<false>(tc + tile_index<8>(i, k));
    } else {
      d.key = 0xffffffffu;
#pragma unroll
      for(int k = 0; k < 8; ++k) d.tc[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const float4* p0 = reinterpret_cast<const float4*>(j.pix.get());
    d.px[0] = load_v4<false>(p0 + tile_index<2>(i, 0)); d.px[1] = load_v4<false>(p0 + tile_index<2>(i, 1));
  } else {
    d.v = (float) j.valid[i];
    const float4* qr = reinterpret_cast<const float4*>(j.r.get());
    d.r[0] = load_v4<false>(qr + tile_index<2>(i, 0)); d.r[1] = load_v4<false>(qr + tile_index<2>(i, 1));
  }
#pragma unroll
  for(int k = 0; k < 4; ++k) d.g[k] = load_v4<false>(qg + tile_index<4>(i, k));
}

template <int LOSS, bool FUSED>
__device__ __forceinline__ void irls_tile_lat(const PairJob& j, const GNState* __restrict__ st, int pts_per_block, int tile, int vtid,
                                              IrlsPartLds& s_part, bool has, float* __restrict__ partials)
{
  float P[12];
  if constexpr(FUSED) {
    projection_matrix(j, st->T, P);
#pragma unroll
    for(int k = 0; k < 12; ++k) P[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(P[k])));
  }
  const int n = j.n, W = j.cols, R = j.rows;
  const int p_begin = tile * pts_per_block;
  const int p_end = has ? min(n, p_begin + pts_per_block) : p_begin;
  const float sigma_inv = 1.0f / st->scale;
  const float s_nrm[4] = {j.nrm[0], j.nrm[1], j.nrm[2], j.nrm[3]};
  const bool dspace = j.dspace != 0;
  const float ds_fx = j.K[0], ds_fy = j.K[4], ds_fx_i = 1.0f / j.K[0], ds_fy_i = 1.0f / j.K[4], ds_b_i = 1.0f / j.b;

  float acc[kNumAcc];
#pragma unroll
  for(int k = 0; k < kNumAcc; ++k) acc[k] = 0.0f;

  // one point: residuals (fused: warp_point's arithmetic on the preloaded taps), weights, rank-2 update — as in irls_tile
  auto point = [&](int i, const IrlsPointLat& d) {
    float rr[8], Ix[8], Iy[8];
    float v;
    if constexpr(FUSED) {
      const double X0 = (double) d.Pt.x, X1 = (double) d.Pt.y, X2 = (double) d.Pt.z, X3 = (double) d.Pt.w;
      double u[3];
#pragma unroll
      for(int r = 0; r < 3; ++r) {
        double s = (double) P[r * 4 + 0] * X0;
        s += (double) P[r * 4 + 1] * X1;
        s += (double) P[r * 4 + 2] * X2;
        s += (double) P[r * 4 + 3] * X3;
        u[r] = s;
      }
      const double zi = 1.0 / u[2];
      const double x = zi * u[0], y = zi * u[1];
      const bool in_range = (x > -2147483648.0) && (x < 2147483648.0) && (y > -2147483648.0) && (y < 2147483648.0);
      int xi = 0, yi = 0;
      if(in_range) {
        xi = (int) x; xi -= (xi > x);
        yi = (int) y; yi -= (yi > y);
      }
      const bool valid = in_range && xi >= 0 && xi < W - 1 && yi >= 0 && yi < R - 1;
      const double xf = x - (double) xi, yf = y - (double) yi;
      bool hit = false;
      if(valid) {
        const double wx = 1.0 - xf, wy = 1.0 - yf;
        const unsigned key = ((unsigned) yi << 16) | (unsigned) xi;
        const bool cached = j.tapcache_on != 0;
        hit = cached && d.key == key;
        float4 t[8];
#pragma unroll
        for(int k = 0; k < 8; ++k) t[k] = d.tc[k];
        if(!hit) {
          const float4* q0 = reinterpret_cast<const float4*>(j.desc + ((size_t) yi * W + xi) * 8);
          const float4* q1 = q0 + (size_t) W * 2;
          t[0] = q0[0]; t[1] = q0[1]; t[2] = q0[2]; t[3] = q0[3];
          t[4] = q1[0]; t[5] = q1[1]; t[6] = q1[2]; t[7] = q1[3];
          if(cached) {
            float4* tcw = reinterpret_cast<float4*>(j.tapcache.get());
#pragma unroll
            for(int k = 0; k < 8; ++k) store_v4<false>(tcw + tile_index<8>(i, k), t[k]);
            j.tapkey[i] = key;
          }
        }
        // pieces 0, 1: I00 of channels 0-3 / 4-7; 2, 3: I01; 4, 5: I10; 6, 7: I11 (warp_point)
        const float i00[8] = {t[0].x, t[0].y, t[0].z, t[0].w, t[1].x, t[1].y, t[1].z, t[1].w};
        const float i01[8] = {t[2].x, t[2].y, t[2].z, t[2].w, t[3].x, t[3].y, t[3].z, t[3].w};
        const float i10[8] = {t[4].x, t[4].y, t[4].z, t[4].w, t[5].x, t[5].y, t[5].z, t[5].w};
        const float i11[8] = {t[6].x, t[6].y, t[6].z, t[6].w, t[7].x, t[7].y, t[7].z, t[7].w};
        const float i0[8] = {d.px[0].x, d.px[0].y, d.px[0].z, d.px[0].w, d.px[1].x, d.px[1].y, d.px[1].z, d.px[1].w};
#pragma unroll
        for(int c = 0; c < 8; ++c) {
          const double Iw = wy * ((double) i00[c] * wx + (double) i01[c] * xf) + yf * ((double) i10[c] * wx + (double) i11[c] * xf);
          rr[c] = (float) (Iw - (double) i0[c]);
        }
      } else {
#pragma unroll
        for(int c = 0; c < 8; ++c) rr[c] = 0.0f;
      }
      v = valid ? 1.0f : 0.0f;
      acc[29] += hit ? 1.0f : 0.0f;
    } else {
      v = d.v;
      rr[0] = d.r[0].x; rr[1] = d.r[0].y; rr[2] = d.r[0].z; rr[3] = d.r[0].w; rr[4] = d.r[1].x; rr[5] = d.r[1].y; rr[6] = d.r[1].z; rr[7] = d.r[1].w;
    }
    acc[28] += v;
    Ix[0] = d.g[0].x; Ix[1] = d.g[0].y; Ix[2] = d.g[0].z; Ix[3] = d.g[0].w; Ix[4] = d.g[1].x; Ix[5] = d.g[1].y; Ix[6] = d.g[1].z; Ix[7] = d.g[1].w;
    Iy[0] = d.g[2].x; Iy[1] = d.g[2].y; Iy[2] = d.g[2].z; Iy[3] = d.g[2].w; Iy[4] = d.g[3].x; Iy[5] = d.g[3].y; Iy[6] = d.g[3].z; Iy[7] = d.g[3].w;
    float Sxx = 0.0f, Sxy = 0.0f, Syy = 0.0f, Gx = 0.0f, Gy = 0.0f;
#pragma unroll
    for(int c = 0; c < 8; ++c) {
      const float r = rr[c];
      const float w = mest_weight<LOSS>(r, sigma_inv) * v;
      const float wx = w * Ix[c], wy = w * Iy[c];
      Sxx += wx * Ix[c];
      Sxy += wx * Iy[c];
      Syy += wy * Iy[c];
      Gx += wx * r;
      Gy += wy * r;
      acc[27] += (w * r) * r;
    }
    const float4 Pt = d.Pt;
    float A[6], B[6];
    if(!dspace) {
      const JacPoint jp = jac_point(Pt.x, Pt.y, Pt.z, s_nrm);
      const float t_xz2 = jp.x * jp.rz2, t_yz2 = jp.y * jp.rz2;
      A[0] = -(t_xz2 * jp.yc2); A[1] = jp.zc3 * jp.rz + t_xz2 * jp.xc1; A[2] = -(jp.yc2 * jp.rz); A[3] = jp.rzs; A[4] = 0.0f; A[5] = -(jp.s_i * t_xz2);
      B[0] = -(jp.zc3 * jp.rz) - t_yz2 * jp.yc2; B[1] = t_yz2 * jp.xc1; B[2] = jp.xc1 * jp.rz; B[3] = 0.0f; B[4] = jp.rzs; B[5] = -(jp.s_i * t_yz2);
    } else {
      const float x = Pt.x, y = Pt.y, dd = Pt.z;
      const float xfi = x * ds_fx_i, yfi = y * ds_fy_i, dbi = dd * ds_b_i;
      A[0] = -(x * yfi); A[1] = ds_fx + x * xfi; A[2] = -(ds_fx * yfi); A[3] = dbi; A[4] = 0.0f; A[5] = -(dbi * xfi);
      B[0] = -ds_fy - y * yfi; B[1] = y * xfi; B[2] = ds_fy * xfi; B[3] = 0.0f; B[4] = dbi * (ds_fy * ds_fx_i); B[5] = -(dbi * (y * ds_fx_i));
    }
    int idx = 0;
#pragma unroll
    for(int a = 0; a < 6; ++a) {
      const float pa = Sxx * A[a] + Sxy * B[a];
      const float qa = Sxy * A[a] + Syy * B[a];
#pragma unroll
      for(int b = a; b < 6; ++b) acc[idx++] += pa * A[b] + qa * B[b];
    }
#pragma unroll
    for(int a = 0; a < 6; ++a) acc[21 + a] += Gx * A[a] + Gy * B[a];
  };

  for(int i0 = p_begin + vtid; i0 < p_end; i0 += 2 * GN_BLOCK) {
    const int i1 = i0 + GN_BLOCK;
    const bool has1 = i1 < p_end;
    IrlsPointLat d0, d1;
    irls_lat_load<FUSED>(j, i0, d0);
    irls_lat_load<FUSED>(j, has1 ? i1 : i0, d1);
    point(i0, d0);
    if(has1) point(i1, d1);
  }

#pragma unroll
  for(int k = 0; k < kNumAcc; ++k) {
    float v = acc[k];
#pragma unroll
    for(int o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o);
    acc[k] = v;
  }
  const int lane = vtid & 63, wave = vtid >> 6;
  if(lane == 0) {
#pragma unroll
    for(int k = 0; k < kNumAcc; ++k) s_part[wave][k] = acc[k];
  }
  __syncthreads();
  if(vtid < kNumAcc && has) {
    const float v = (s_part[0][vtid] + s_part[1][vtid]) + (s_part[2][vtid] + s_part[3][vtid]);
    partials[(size_t) tile * kPartialStride + vtid] = v;
  }
}

// Two instantiations share the work of a launch slot: FUSED = false handles the workspaces whose scale still moves, FUSED =
// true (136 VGPRs instead of 103: kept out of the plain kernel's register budget) the frozen ones.
template <int C, int LOSS, bool FUSED>
__global__ __launch_bounds__(GN_BLOCK) void irls_reduce_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int pts_per_block, int fuse_frozen)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.y)];
  const GNState* __restrict__ st = j.st;
  if(!st->active) return;
  if(fuse_frozen && (FUSED != !(st->delta_scale > 1e-6f))) return;
  irls_block<C, LOSS, FUSED>(j, st, pts_per_block);
}
// ... or ONE launch serves both kinds with a per-workspace branch (C = 8): every workgroup then runs at the fused form's
// register budget (3 waves per SIMD instead of 4), but small launches — the 128-pair shard of config 5, single pairs — do not
// pay a second, half-empty launch per iteration (each costs its ramp and drain: at 128 pairs the two launches took 59 us where
// the bytes are worth 38).
template <int LOSS>
__global__ __launch_bounds__(GN_BLOCK) void irls_reduce_both_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int pts_per_block)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.y)];
  const GNState* __restrict__ st = j.st;
  if(!st->active) return;
  if(st->delta_scale > 1e-6f) irls_block<8, LOSS, false>(j, st, pts_per_block);
  else irls_block<8, LOSS, true>(j, st, pts_per_block);
}

// ------------------------------------------------------------------------------------------------------------------
// K9 gn_step: PoseEstimatorBase::run as a device-side state machine (reference: bpvo/pose_estimator_base.h:324-407 with
// testConvergence :258-282, PoseEstimatorData_::solve :90-148, RigidBodyWarp::paramsToPose bpvo/rigid_body_warp.h:130-138).
// One wave per workspace: lanes 0..28 sum the per-block partials in block order in f64 (deterministic), lane 0 runs the
// 6x6 solve, pose update and bookkeeping — Q1 (pose updated again after convergence) and Q2 (iteration count) included.
__device__ __forceinline__ float inf_norm6(const float* g)
{
  float m = 0.0f;
  for(int i = 0; i < 6; ++i) m = fmaxf(m, fabsf(g[i]));
  return m;
}

__device__ void gn_update_pose(GNState* st, const float* nrm)
{
  float mdp[6];
  for(int i = 0; i < 6; ++i) mdp[i] = -st->dp[i];
  M44 T;
  for(int i = 0; i < 16; ++i) T.m[i] = st->T[i];
  // nrm[4] != 0: DisparitySpaceWarp::paramsToPose = TwistToMatrix(p), scalePose is the identity (disparity_space_warp.h:79-91)
  const M44 Tn = m44_mul(T, nrm[4] != 0.0f ? twist_to_matrix(mdp) : params_to_pose(nrm, mdp));
  for(int i = 0; i < 16; ++i) st->T[i] = Tn.m[i];
}

__device__ void gn_finalize(GNState* st)
{
  if(st->status != BPVO_STATUS_SOLVER_ERROR)
    for(int i = 0; i < 16; ++i) st->T_out[i] = st->T[i];
  st->num_iterations -= 1;
  bpvo_hip_stats& s = st->stats[st->level];
  s.numIterations = st->num_iterations;
  s.finalError = st->f_norm;
  s.firstOrderOptimality = st->g_norm;
  s.status = st->status;
  st->phase = PHASE_DONE;
  st->active = 0;
}

// the serial part of gn_step, executed by lane 0 on the LDS copy of the state; returns true if another linearisation
// is requested (the workspace stays active)
#ifdef BPVO_PK_TIMING
__shared__ unsigned pk_sub[8];      // timing build: 10-ns ticks of the serial step's parts (unpack, solve, pose update, tests), summed
#define GN_SUBTICK(k) do { const long long t_ = wall_clock64(); pk_sub[k] += (unsigned) (t_ - sub_t); sub_t = t_; } while(0)
#else
#define GN_SUBTICK(k) do { } while(0)
#endif
__device__ bool gn_logic(GNState* st, const float* nrm, const float* s_sum, SolveScratch* scratch, int mode, int max_iterations,
                         int max_fun_evals, float p_tol, float f_tol, float g_tol_param)
{
#ifdef BPVO_PK_TIMING
  long long sub_t = wall_clock64();
#endif
  // unpack: upper triangle -> symmetric H (toEigen + selfadjointView<Upper>, linear_system_builder.cc:207-221)
  {
    int idx = 0;
    for(int a = 0; a < 6; ++a)
      for(int b = a; b < 6; ++b) {
        st->H[a * 6 + b] = s_sum[idx];
        st->H[b * 6 + a] = s_sum[idx];
        ++idx;
      }
    for(int a = 0; a < 6; ++a) st->G[a] = s_sum[21 + a];
  }
  const float f_norm = sqrtf(s_sum[27]);               // LinearSystemBuilder::Run returns sqrt (:349)
  st->f_norm = f_norm;
  st->n_valid = (uint32_t) s_sum[28];
  st->num_fun_evals += 1;
  if(mode == 1) return true;
  GN_SUBTICK(0);

  const float sqrt_eps = sqrtf(FLT_EPSILON);

  if(st->phase == PHASE_FIRST) {
    const float g_norm = inf_norm6(st->G);
    st->g_norm = g_norm;
    st->g_tol = g_tol_param * fmaxf(g_norm, sqrt_eps);
    if(g_norm < st->g_tol) {                            // :343-354 initial value is optimal
      bpvo_hip_stats& s = st->stats[st->level];
      s.status = BPVO_STATUS_GRADIENT_TOL; s.finalError = f_norm; s.numIterations = 1; s.firstOrderOptimality = g_norm;
      st->status = BPVO_STATUS_GRADIENT_TOL;
      st->phase = PHASE_DONE; st->active = 0;
      return false;
    }
    if(!solve_system(st->H, st->G, st->dp, scratch)) {           // :356-362
      bpvo_hip_stats& s = st->stats[st->level];
      s.status = BPVO_STATUS_SOLVER_ERROR; s.finalError = f_norm; s.numIterations = 0; s.firstOrderOptimality = 0.0f;
      st->status = BPVO_STATUS_SOLVER_ERROR;
      st->phase = PHASE_DONE; st->active = 0;
      return false;
    }
    st->f_norm_prev = 0.0f;
    st->dp_norm_prev = 0.0f;
    st->has_converged = 0;
    gn_update_pose(st, nrm);                            // :371
  } else {
    // runIteration's solve (pose_estimator_gn.h:89-97)
    if(!solve_system(st->H, st->G, st->dp, scratch)) {
      st->status = BPVO_STATUS_SOLVER_ERROR;
      gn_finalize(st);                                  // `break`: no ++ on the way out
      return false;
    }
    GN_SUBTICK(1);
    gn_update_pose(st, nrm);                            // :390
    GN_SUBTICK(2);
    const bool cont = (st->num_iterations++ < max_iterations) && !st->has_converged && (st->num_fun_evals < max_fun_evals);
    if(!cont) { gn_finalize(st); return false; }
  }

  // top of the do-loop body (:374-383)
  float dp_norm = 0.0f;
  for(int i = 0; i < 6; ++i) dp_norm += st->dp[i] * st->dp[i];
  dp_norm = sqrtf(dp_norm);
  const float g_norm = inf_norm6(st->G);
  st->g_norm = g_norm;
  bool conv = false;
  if(dp_norm < p_tol || dp_norm < p_tol * (sqrt_eps + st->dp_norm_prev)) {
    st->status = BPVO_STATUS_PARAMETER_TOL; conv = true;
  } else if(f_norm < f_tol || f_norm < f_tol * (sqrt_eps + st->f_norm_prev) || fabsf(f_norm - st->f_norm_prev) < f_tol) {
    st->status = BPVO_STATUS_FUNCTION_TOL; conv = true;
  } else if(g_norm < st->g_tol) {
    st->status = BPVO_STATUS_GRADIENT_TOL; conv = true;
  }
  st->has_converged = conv ? 1 : 0;
  st->dp_norm_prev = dp_norm;
  st->f_norm_prev = f_norm;
  GN_SUBTICK(3);
  if(!conv) {
    st->phase = PHASE_LOOP;                             // next launch: linearize at the updated pose
    return true;
  }
  gn_update_pose(st, nrm);                              // Q1: applied again with the stale dp
  st->num_iterations++;                                 // the `numIterations++ <` of the failing while test
  gn_finalize(st);
  return false;
}

// lanes 0 .. kNumAcc-1 of one wave: deterministic sum (tile order, f64) of the tile partials of workspace j.
// The loads of 32 tiles are issued back to back, UNCONDITIONALLY (the tile index is clamped, the add is what the bound selects: a
// conditional load makes the compiler wait per branch), so a level costs one global-memory round trip per 32 tiles instead of one per
// 8: 2.2 -> 1.3 us of the serial step at the finest level of a 1241x376 pair (profiles/r02_persistent_phases.txt).  Same order of additions.
__device__ __forceinline__ void gn_sum_partials(const PairJob& j, int pts_per_block, int lane, float* s_sum /*[kPartialStride]*/,
                                                const float* __restrict__ partials)
{
  const int nblk = (j.n + pts_per_block - 1) / pts_per_block;
  if(lane < kNumAcc) {
    double s = 0.0;
    const float* __restrict__ pp = partials + lane;
    auto chunked = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      for(int b0 = 0; b0 < nblk; b0 += U) {
        float v[U];
#pragma unroll
        for(int u = 0; u < U; ++u) v[u] = pp[(size_t) min(b0 + u, nblk - 1) * kPartialStride];
#pragma unroll
        for(int u = 0; u < U; ++u) {
          const double t = s + (double) v[u];
          s = (b0 + u < nblk) ? t : s;
        }
      }
    };
    if(nblk <= 8) chunked(std::integral_constant<int, 8>());      // (coarse levels: no point in 32 loads for 6 tiles)
    else chunked(std::integral_constant<int, 32>());
    s_sum[lane] = (float) s;
  }
}

// one thread, on an LDS copy `st` of the state: the step that consumes the linearisation summed in s_sum.  `stats`: this copy
// is the one that keeps the workspace's measurement counters (the persistent kernel runs the step redundantly in every workgroup)
__device__ __forceinline__ void gn_serial_step(const PairJob& j, GNState* st, const float* s_nrm, const float* s_sum, SolveScratch* scratch,
                                               int mode, int max_iterations, int max_fun_evals, float p_tol, float f_tol, float g_tol_param,
                                               int fuse_frozen, bool stats)
{
  // the linearisation consumed here was taken at st->T; with the fused path its residuals were never written
  for(int i = 0; i < 16; ++i) st->T_lin[i] = st->T[i];
  const bool fused_lin = fuse_frozen && !(st->delta_scale > 1e-6f);
  st->r_stale = fused_lin ? 1 : 0;
  const bool again = gn_logic(st, s_nrm, s_sum, scratch, mode, max_iterations, max_fun_evals, p_tol, f_tol, g_tol_param);
  (void) again;   // who is still active is read from st->active (compact_active_kernel once per host round / the persistent loop)
  if(stats && mode == 0 && j.trace) {
    // bpvo_hip_estimate_pose_trace: the linearisation just consumed (pose, system, function value, scale, valid count) and the step
    // solved from it; record layout: BPVO_HIP_TRACE_FLOATS in c_api.h
    if(st->trace_n < j.trace_cap) {
      float* o = j.trace + (size_t) st->trace_n * kTraceFloats;
      for(int i = 0; i < 16; ++i) o[i] = st->T_lin[i];
      for(int i = 0; i < 36; ++i) o[16 + i] = st->H[i];
      for(int i = 0; i < 6; ++i) { o[52 + i] = st->G[i]; o[61 + i] = st->dp[i]; }
      o[58] = st->f_norm; o[59] = st->scale; o[60] = (float) st->n_valid; o[67] = (float) st->level;
    }
    st->trace_n += 1;
  }
  if(stats) {
    j.cnt[0] += (unsigned long long) j.n;     // measurement: points and linearisations processed (bench.py roofline)
    j.cnt[1] += 1ull;
    if(fused_lin) {                           // the fused path keeps its own tap-cache statistics (the others: median_finish)
      j.cnt[5] += (unsigned long long) s_sum[29]; j.cnt[6] += j.tapcache_on ? (unsigned long long) s_sum[28] : 0ull;
      j.cnt[10] += (unsigned long long) j.n;
    }
  }
}

__global__ __launch_bounds__(64) void gn_step_kernel(const PairJob* __restrict__ jobs, int pts_per_block, int mode,
                                                     int max_iterations, int max_fun_evals, float p_tol, float f_tol,
                                                     float g_tol_param, ActiveSet act, int fuse_frozen)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.x)];
  GNState* gst = j.st;
  if(!gst->active) return;

  // the state lives in HBM between launches; the serial bookkeeping runs on an LDS copy (global-memory round trips
  // would otherwise dominate this kernel: every field access is a dependent ~1 us load)
  constexpr int kWords = (int) (sizeof(GNState) / sizeof(uint32_t));
  static_assert(sizeof(GNState) % sizeof(uint32_t) == 0, "GNState must be word sized");
  __shared__ uint32_t s_state[kWords];
  __shared__ float s_sum[kPartialStride];
  __shared__ float s_nrm[5];
  __shared__ SolveScratch s_scratch;
  for(int i = threadIdx.x; i < kWords; i += 64) s_state[i] = reinterpret_cast<const uint32_t*>(gst)[i];
  if(threadIdx.x < 4) s_nrm[threadIdx.x] = j.nrm[threadIdx.x];
  if(threadIdx.x == 4) s_nrm[4] = j.dspace ? 1.0f : 0.0f;
  gn_sum_partials(j, pts_per_block, threadIdx.x, s_sum, j.partials);
  __syncthreads();
  if(threadIdx.x == 0)
    gn_serial_step(j, reinterpret_cast<GNState*>(s_state), s_nrm, s_sum, &s_scratch, mode, max_iterations, max_fun_evals, p_tol, f_tol,
                   g_tol_param, fuse_frozen, true);
  __syncthreads();
  for(int i = threadIdx.x; i < kWords; i += 64) reinterpret_cast<uint32_t*>(gst)[i] = s_state[i];
}

// ------------------------------------------------------------------------------------------------------------------
// Active list of the next host round: the still-active workspaces of the current list, in order (one 1024-thread
// workgroup, block scan per chunk of 1024 entries; no atomics, deterministic order).
__global__ __launch_bounds__(1024) void compact_active_kernel(const PairJob* __restrict__ jobs, ActiveSet in, int n_in,
                                                              int* __restrict__ out_list, int* __restrict__ out_count)
{
  // out_count[0]: entries of the new list; out_count[1]: how many of them still estimate their robust scale (delta_scale > 1e-6).  A
  // frozen scale stays frozen for the rest of the level, so once [1] is 0 the host stops launching warp_residual (fused path) and median.
  __shared__ unsigned s_wave[16];
  __shared__ unsigned s_base, s_moving;
  if(threadIdx.x == 0) { s_base = 0; s_moving = 0; }
  __syncthreads();
  unsigned moving = 0;
  for(int base = 0; base < n_in; base += 1024) {
    const int k = base + threadIdx.x;
    int ws = -1;
    if(k < n_in) {
      ws = in.list ? in.list[k] : k;
      const GNState* st = jobs[ws].st;
      if(!st->active) ws = -1;
      else if(st->delta_scale > 1e-6f) moving += 1u;
    }
    unsigned total;
    const unsigned off = block_excl_scan_1024(ws >= 0 ? 1u : 0u, s_wave, total);
    const unsigned b = s_base;
    if(ws >= 0) out_list[b + off] = ws;
    __syncthreads();
    if(threadIdx.x == 0) s_base = b + total;
    __syncthreads();
  }
  if(moving) atomicAdd(&s_moving, moving);
  __syncthreads();
  if(threadIdx.x == 0) { out_count[0] = (int) s_base; out_count[1] = (int) s_moving; }
}

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


struct GNParams { int max_iterations, max_fun_evals; float p_tol, f_tol, g_tol; };

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
__device__ __forceinline__ GNState* pk_st(int ws) { return reinterpret_cast<GNState*>(pk_state[ws]); }

// warp_residual (+ bracket step) of workspace ws: chunk c goes to workgroup c % nwg, virtual block (c / nwg) % PK_VB
template <int C>
__device__ __attribute__((noinline)) void pk_warp_phase(const PairJob* __restrict__ jobs, int ws, bool stats_wg)
{
  const int tid = threadIdx.x, vsub = tid >> 8, vtid = tid & 255;
  const int nwg = (int) gridDim.x;
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
    const int chunk = base + vsub * nwg + (int) blockIdx.x;
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
  const int nwg = (int) gridDim.x;
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
      s[u].chunk = base + (u * PK_VB + vsub) * nwg + (int) blockIdx.x;
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
// allocation holds cap / 256 entries, the reduction uses at most cap / 1024).
__device__ __forceinline__ float* pk_partials(const PairJob& j, int pts_per_block, unsigned parity)
{
  const int ntiles = (j.n + pts_per_block - 1) / pts_per_block;
  return j.partials + (size_t) (parity & 1u) * (size_t) ntiles * kPartialStride;
}
template <int C, int LOSS, bool FUSED>
__device__ __forceinline__ void pk_irls_phase(const PairJob* __restrict__ jobs, int ws, int pts_per_block, unsigned parity)
{
  const int tid = threadIdx.x, vsub = tid >> 8, vtid = tid & 255;
  const int nwg = (int) gridDim.x;
  const PairJob& j = jobs[ws];
  const int ntiles = (j.n + pts_per_block - 1) / pts_per_block;
  float* const partials = pk_partials(j, pts_per_block, parity);
  for(int base = 0; base < ntiles; base += nwg * PK_VB) {
    const int tile = base + vsub * nwg + (int) blockIdx.x;
    if constexpr(C == 8) irls_tile_lat<LOSS, FUSED>(j, pk_st(ws), pts_per_block, tile, vtid, pk_part[vsub], tile < ntiles, partials);
    else irls_tile<C, LOSS, FUSED>(j, pk_st(ws), pts_per_block, tile, vtid, pk_part[vsub], tile < ntiles, partials);
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
  if(mine) gn_sum_partials(jobs[ws], pts_per_block, lane, pk_sum[ws], pk_partials(jobs[ws], pts_per_block, parity));
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
__device__ __attribute__((noinline)) bool pk_grid_barrier(unsigned* ctl, unsigned epoch, long long timeout)
{
  __syncthreads();
  if(threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned target = epoch * gridDim.x;
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
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    pk_ok = ok;
  }
  __syncthreads();
  return pk_ok != 0;
}

template <int C, int LOSS>
__global__ __launch_bounds__(PK_THREADS) void gn_persistent_kernel(const PairJob* __restrict__ jobs, int nws, int pts_per_block, GNParams prm,
                                                                   int fuse_frozen, unsigned* ctl, long long timeout)
{
  constexpr bool kCanFuse = (C == 8);
  const int tid = threadIdx.x;
  const bool stats_wg = blockIdx.x == 0;
  const bool fuse = kCanFuse && fuse_frozen;

  for(int ws = 0; ws < nws; ++ws) {
    const uint32_t* g = reinterpret_cast<const uint32_t*>(jobs[ws].st.get());
    for(int i = tid; i < kStateWords; i += PK_THREADS) pk_state[ws][i] = g[i];
    if(tid < 4) pk_nrm[ws][tid] = jobs[ws].nrm[tid];
    if(tid == 4) pk_nrm[ws][4] = jobs[ws].dspace ? 1.0f : 0.0f;
  }
  __syncthreads();

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
    ok = pk_grid_barrier(ctl, ++epoch, timeout);
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

__device__ __attribute__((noinline)) bool pk_team_barrier(unsigned* team_ctl, unsigned* abort_word, unsigned epoch, long long timeout)
{
  __syncthreads();
  if(threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned target = epoch * gridDim.x;
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
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    pk_ok = ok;
  }
  __syncthreads();
  return pk_ok != 0;
}

// PoseEstimatorBase::reset + the head of run() on this workgroup's LDS copy (level_begin_kernel's body), and this workgroup's share
// of the tap-cache keys of the level
__device__ __forceinline__ void pk_level_begin(const PairJob& j, int level, int scale_is_moot)
{
  const int nthreads_team = (int) gridDim.x * PK_THREADS;
  if(j.tapkey)
    for(int i = (int) blockIdx.x * PK_THREADS + (int) threadIdx.x; i < j.n; i += nthreads_team) j.tapkey[i] = 0xffffffffu;
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

template <int C, int LOSS>
__global__ __launch_bounds__(PK_THREADS) void gn_team_kernel(const PairJob* __restrict__ jobs_all /*[levels][job_pitch]*/, int job_pitch, int n_pairs,
                                                             int level_hi, int level_lo, int pts_per_block, GNParams prm, int fuse_frozen,
                                                             int scale_is_moot, unsigned* ctl, long long timeout)
{
  constexpr bool kCanFuse = (C == 8);
  const int tid = threadIdx.x;
  const bool stats_wg = blockIdx.x == 0;
  const bool fuse = kCanFuse && fuse_frozen;
  unsigned* const global_ctl = ctl;                                           // [1] abort, [2] next pair to hand out
  unsigned* const team_ctl = ctl + (size_t) (1 + blockIdx.y) * kTeamCtlWords; // [0] arrivals, [1] pair slot
  unsigned epoch = 0, epoch_it = 0;
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
    if(!pk_team_barrier(team_ctl, global_ctl + 1, ++epoch, timeout)) return;
    if(tid == 0) pk_next_pair = (int) __hip_atomic_load(team_ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int pair = pk_next_pair;
    if(pair >= n_pairs) return;

    {   // the pair's state: HBM -> this workgroup's LDS copy (set_pose_kernel has run: T_out, statistics defaults)
      const uint32_t* g = reinterpret_cast<const uint32_t*>(jobs_all[(size_t) level_hi * job_pitch + pair].st.get());
      for(int i = tid; i < kStateWords; i += PK_THREADS) pk_state[0][i] = g[i];
    }
    __syncthreads();

    for(int level = level_hi; level >= level_lo; --level) {
      const PairJob* __restrict__ jobs = jobs_all + (size_t) level * job_pitch + pair;      // jobs[0]: this pair at this level
      pk_level_begin(jobs[0], level, scale_is_moot);
      if(!pk_team_barrier(team_ctl, global_ctl + 1, ++epoch, timeout)) return;         // keys reset before any phase reads them
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
          if(!pk_team_barrier(team_ctl, global_ctl + 1, ++epoch, timeout)) return;
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
        if(!pk_team_barrier(team_ctl, global_ctl + 1, ++epoch, timeout)) return;
        TEAM_TICK(4);
        pk_step_phase(jobs, 1, pts_per_block, prm, fuse ? 1 : 0, stats_wg, epoch_it);
        TEAM_TICK(5);
        ++epoch_it;
      }
    }
#ifdef BPVO_PK_TIMING
    if(blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
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

// ------------------------------------------------------------------------------------------------------------------
__global__ void set_pose_kernel(const PairJob* jobs, const float* T_init, int n)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if(p >= n) return;
  GNState* st = jobs[p].st;
  for(int i = 0; i < 16; ++i) st->T_out[i] = T_init ? T_init[p * 16 + i] : ((i % 5 == 0) ? 1.0f : 0.0f);
  st->trace_n = 0;
  for(int l = 0; l < kMaxLevels; ++l) {                 // OptimizerStatistics() defaults (bpvo/types.cc:306-310)
    st->stats[l].numIterations = 0;
    st->stats[l].finalError = -1.0f;
    st->stats[l].firstOrderOptimality = -1.0f;
    st->stats[l].status = BPVO_STATUS_SOLVER_ERROR;
  }
}

// PoseEstimatorBase::reset + the head of run() (bpvo/pose_estimator_base.h:287-293,327-335)
// scale_is_moot (kL2 with the fused path available): MEstimator::ComputeWeights gives w = 1 whatever the scale
// (bpvo/mestimator.cc:390-415), so the estimate loops never look at it: the level starts with the scale "frozen" at 1 and
// every linearisation takes the fused residual + reduction path — two launches per iteration, no median.  (The reference still
// runs estimateScale for kL2; its value is unobservable through estimatePose / addFrame.  bpvo_hip_linearize, which reports
// sigma, computes it.)
// One launch does both: workgroup (x, p) invalidates the tap-cache keys [256 x, 256 x + 256) of workspace p (reset_tapkeys_kernel below,
// which the linearize seam still uses on its own), thread 0 of workgroup (0, p) resets the state.
__global__ __launch_bounds__(GN_BLOCK) void level_begin_kernel(const PairJob* jobs, int npairs, int level, int scale_is_moot)
{
  const int p = blockIdx.y;
  const PairJob& j = jobs[p];
  {
    const int i = blockIdx.x * GN_BLOCK + threadIdx.x;
    if(i < j.n && j.tapkey) j.tapkey[i] = 0xffffffffu;
  }
  if(blockIdx.x != 0 || threadIdx.x != 0) return;
  GNState* st = j.st;
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

// invalidates the tap cache keys of every workspace of a launch (start of a level / of a linearize call)
__global__ __launch_bounds__(GN_BLOCK) void reset_tapkeys_kernel(const PairJob* jobs)
{
  const PairJob& j = jobs[blockIdx.y];
  const int i = blockIdx.x * GN_BLOCK + threadIdx.x;
  if(i < j.n && j.tapkey) j.tapkey[i] = 0xffffffffu;
}

// operator-level seam (bpvo_hip_linearize): pose in, optional AutoScaleEstimator::reset
// reset_scale 2: the scale is GIVEN (bpvo_hip_linearize_at_scale) — the estimator is left frozen at it, so the median kernel and
// the bracket step leave the workspace alone and irls_reduce weighs with exactly this value
__global__ void prepare_linearize_kernel(const PairJob* job, const float* T, int reset_scale, int level, float given_scale)
{
  if(threadIdx.x != 0 || blockIdx.x != 0) return;
  GNState* st = job->st;
  for(int i = 0; i < 16; ++i) st->T[i] = T[i];
  if(reset_scale == 2) { st->scale = given_scale; st->delta_scale = 0.0f; }
  else if(reset_scale) { st->scale = 1.0f; st->delta_scale = 1e10f; }
  if(reset_scale || st->level != level) { st->median_valid = 0; st->last_median = 0.0f; }
  st->level = level;
  st->active = 1;
}

// weights of the last linearisation, recomputed from r / valid / sigma on request
// (VisualOdometryPoseEstimator::getWeights, bpvo/vo_pose_estimator.cc:95-99; invalid entries have r = 0 -> w = 1, Q12)
template <int C, int LOSS>
__global__ __launch_bounds__(256) void weights_kernel(const PairJob* job, float* w_out /*[n][C] point-major*/)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if(i >= job->n) return;
  const float sigma_inv = 1.0f / job->st->scale;
  if constexpr(C == 8) {
    const float4* q = reinterpret_cast<const float4*>(job->r.get());
    const float4 a = q[tile_index<2>(i, 0)], b = q[tile_index<2>(i, 1)];
    float* o = w_out + (size_t) i * 8;
    o[0] = mest_weight<LOSS>(a.x, sigma_inv); o[1] = mest_weight<LOSS>(a.y, sigma_inv);
    o[2] = mest_weight<LOSS>(a.z, sigma_inv); o[3] = mest_weight<LOSS>(a.w, sigma_inv);
    o[4] = mest_weight<LOSS>(b.x, sigma_inv); o[5] = mest_weight<LOSS>(b.y, sigma_inv);
    o[6] = mest_weight<LOSS>(b.z, sigma_inv); o[7] = mest_weight<LOSS>(b.w, sigma_inv);
  } else {
#pragma unroll
    for(int c = 0; c < C; ++c) w_out[(size_t) i * C + c] = mest_weight<LOSS>(job->r[(size_t) i * C + c], sigma_inv);
  }
}

template <int C, int LOSS>
__global__ __launch_bounds__(256) void count_good_kernel(const PairJob* job, float thr, unsigned int* count)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned good = 0;
  if(i < job->n) {
    const float sigma_inv = 1.0f / job->st->scale;
    if constexpr(C == 8) {
      const float4* q = reinterpret_cast<const float4*>(job->r.get());
      const float4 a = q[tile_index<2>(i, 0)], b = q[tile_index<2>(i, 1)];
      good = (mest_weight<LOSS>(a.x, sigma_inv) > thr) + (mest_weight<LOSS>(a.y, sigma_inv) > thr) +
             (mest_weight<LOSS>(a.z, sigma_inv) > thr) + (mest_weight<LOSS>(a.w, sigma_inv) > thr) +
             (mest_weight<LOSS>(b.x, sigma_inv) > thr) + (mest_weight<LOSS>(b.y, sigma_inv) > thr) +
             (mest_weight<LOSS>(b.z, sigma_inv) > thr) + (mest_weight<LOSS>(b.w, sigma_inv) > thr);
    } else {
#pragma unroll
      for(int c = 0; c < C; ++c) good += mest_weight<LOSS>(job->r[(size_t) i * C + c], sigma_inv) > thr;
    }
  }
#pragma unroll
  for(int o = 32; o >= 1; o >>= 1) good += __shfl_down(good, o);
  if((threadIdx.x & 63) == 0 && good) atomicAdd(count, good);
}

// 32-float result record per pair for the RCCL gather: pose 3x4 (12), numIterations per level (8), status per level (8),
// total function evaluations are not kept per level so [28..31] = {finalError of the finest level, n_valid, 0, 0}
__global__ void pack_records_kernel(const PairJob* jobs, int n, int L, float* records)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if(p >= n) return;
  const GNState* st = jobs[p].st;
  float* o = records + (size_t) p * kRecordFloats;
  for(int i = 0; i < 12; ++i) o[i] = st->T_out[i];
  for(int l = 0; l < 8; ++l) {
    o[12 + l] = (l < L) ? (float) st->stats[l].numIterations : 0.0f;
    o[20 + l] = (l < L) ? (float) st->stats[l].status : 0.0f;
  }
  o[28] = st->stats[0].finalError;
  o[29] = (float) st->n_valid;
  o[30] = 0.0f;
  o[31] = 0.0f;
}

// ---- launchers ----------------------------------------------------------------------------------------------------
// fixed, so that a pair's block partials (and hence its rounding) do not depend on the size of the batch it is in
// (a function of C only, never of the batch).  C = 8: K8_PPB_VALUE.  C = 1: the per-point
// work is an eighth, so the 29-accumulator reduction tail of a workgroup dominates — 8 points per thread instead of 2.
#ifndef K8_PPB_VALUE
#define K8_PPB_VALUE 1024   // with 2048-point tiles: 256 -> 401 us, 512 -> 239, 1024 -> 228, 2048 -> 228 per 12.2 M-point launch
#endif
int gn_pts_per_block(int C) { return C == 8 ? K8_PPB_VALUE : 2048; }
// upper bound of the block-indexed buffers: bracket chunks of warp_residual (K6_BLOCK points) and reduction partials
int gn_num_blocks(int max_points) { return (max_points + K6_BLOCK - 1) / K6_BLOCK; }

void launch_set_pose(hipStream_t s, const PairJob* jobs, const float* T_init, int n)
{
  hipLaunchKernelGGL(set_pose_kernel, dim3((n + 63) / 64), dim3(64), 0, s, jobs, T_init, n);
}
void launch_level_begin(hipStream_t s, const PairJob* jobs, int npairs, int max_points, int level, int scale_is_moot)
{
  hipLaunchKernelGGL(level_begin_kernel, dim3(std::max(1, (max_points + GN_BLOCK - 1) / GN_BLOCK), npairs), dim3(GN_BLOCK), 0, s, jobs, npairs, level, scale_is_moot);
}
void launch_reset_tapkeys(hipStream_t s, const GNLaunch& g)
{
  if(g.max_points <= 0 || (g.C != 8 && g.C != 1)) return;
  hipLaunchKernelGGL(reset_tapkeys_kernel, dim3((g.max_points + GN_BLOCK - 1) / GN_BLOCK, g.npairs), dim3(GN_BLOCK), 0, s, g.jobs);
}
void launch_warp_residual(hipStream_t s, const GNLaunch& g)
{
  if(g.max_points <= 0) return;
  const dim3 grid((g.max_points + K6_BLOCK - 1) / K6_BLOCK, g.npairs);
  if(g.interp != BPVO_INTERP_LINEAR) {
    dispatch_channels(g.C, [&](auto c) {
      hipLaunchKernelGGL(warp_residual_interp_kernel<decltype(c)::value>, grid, dim3(K6_BLOCK), 0, s, g.jobs, g.active, g.interp);
    });
    return;
  }
  if(g.fast_warp) {
    dispatch_channels(g.C, [&](auto c) {
      hipLaunchKernelGGL((warp_residual_kernel<decltype(c)::value, true>), grid, dim3(K6_BLOCK), 0, s, g.jobs, g.active, 0);
    });
  } else {
    dispatch_channels(g.C, [&](auto c) {
      constexpr int CC = decltype(c)::value;
      hipLaunchKernelGGL((warp_residual_kernel<CC, false>), grid, dim3(K6_BLOCK), 0, s, g.jobs, g.active, (CC == 8 && g.fuse_frozen) ? 1 : 0);
    });
  }
}
// refresh the residual / valid buffers of the workspaces marked r_stale (fused path) from T_lin, then clear the marks
void launch_refresh_residuals(hipStream_t s, const GNLaunch& g)
{
  if(g.max_points <= 0 || g.C != 8) return;
  const dim3 grid((g.max_points + K6_BLOCK - 1) / K6_BLOCK, g.npairs);
  hipLaunchKernelGGL((warp_residual_kernel<8, false>), grid, dim3(K6_BLOCK), 0, s, g.jobs, ActiveSet(), 2);
  hipLaunchKernelGGL(clear_stale_kernel, dim3((g.npairs + 63) / 64), dim3(64), 0, s, g.jobs, g.npairs);
}
static constexpr size_t kMedianLds = ((MED_COPIES + 1) * MED_BINS + MED_CACHE + 16 + 4 + 4) * sizeof(unsigned);
static constexpr size_t kMedianLdsB = ((MED_COPIES_B + 1) * MED_BINS + MED_CACHE_B + 16 + 4 + 4) * sizeof(unsigned);
void launch_median(hipStream_t s, const GNLaunch& g)
{
  if(g.max_points <= 0) return;
  // the attribute is per device (a process may hold contexts on several) and the lanes' host threads race here
  static std::once_flag attr_once[64];
  int dev = 0;
  (void) hipGetDevice(&dev);
  std::call_once(attr_once[dev & 63], [] {
    for(int C : {1, 3, 5, 8, 10, 24, 48})
      dispatch_channels(C, [&](auto c) {
        constexpr int CC = decltype(c)::value;
        (void) hipFuncSetAttribute((const void*) median_finish_kernel<CC, MED_THREADS, MED_COPIES, MED_CACHE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) kMedianLds);
        (void) hipFuncSetAttribute((const void*) median_finish_kernel<CC, MED_THREADS_B, MED_COPIES_B, MED_CACHE_B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) kMedianLdsB);
      });
  });
  static const int wide_from = std::getenv("BPVO_HIP_MEDIAN_WIDE_FROM") ? std::atoi(std::getenv("BPVO_HIP_MEDIAN_WIDE_FROM")) : 257;   // A/B
  dispatch_channels(g.C, [&](auto c) {
    constexpr int CC = decltype(c)::value;
    if(g.npairs >= wide_from)
      hipLaunchKernelGGL((median_finish_kernel<CC, MED_THREADS_B, MED_COPIES_B, MED_CACHE_B>), dim3(g.npairs), dim3(MED_THREADS_B), kMedianLdsB, s, g.jobs, g.active);
    else
      hipLaunchKernelGGL((median_finish_kernel<CC, MED_THREADS, MED_COPIES, MED_CACHE>), dim3(g.npairs), dim3(MED_THREADS), kMedianLds, s, g.jobs, g.active);
  });
}

template <int C>
static void launch_irls_c(hipStream_t s, const GNLaunch& g, int ppb)
{
  const dim3 grid((g.max_points + ppb - 1) / ppb, g.npairs);
  const int fuse = (C == 8 && g.fuse_frozen && !g.fast_warp && g.interp == BPVO_INTERP_LINEAR) ? 1 : 0;
  if constexpr(C == 8) {
    if(fuse && g.merge_irls) {
      switch(g.loss) {
        case BPVO_LOSS_HUBER: hipLaunchKernelGGL((irls_reduce_both_kernel<BPVO_LOSS_HUBER>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb); break;
        case BPVO_LOSS_TUKEY: hipLaunchKernelGGL((irls_reduce_both_kernel<BPVO_LOSS_TUKEY>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb); break;
        default: hipLaunchKernelGGL((irls_reduce_both_kernel<BPVO_LOSS_L2>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb); break;
      }
      return;
    }
  }
  switch(g.loss) {
    case BPVO_LOSS_HUBER: hipLaunchKernelGGL((irls_reduce_kernel<C, BPVO_LOSS_HUBER, false>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, fuse); break;
    case BPVO_LOSS_TUKEY: hipLaunchKernelGGL((irls_reduce_kernel<C, BPVO_LOSS_TUKEY, false>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, fuse); break;
    default: hipLaunchKernelGGL((irls_reduce_kernel<C, BPVO_LOSS_L2, false>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, fuse); break;
  }
  if constexpr(C == 8) {
    if(!fuse) return;
    switch(g.loss) {
      case BPVO_LOSS_HUBER: hipLaunchKernelGGL((irls_reduce_kernel<8, BPVO_LOSS_HUBER, true>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, fuse); break;
      case BPVO_LOSS_TUKEY: hipLaunchKernelGGL((irls_reduce_kernel<8, BPVO_LOSS_TUKEY, true>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, fuse); break;
      default: hipLaunchKernelGGL((irls_reduce_kernel<8, BPVO_LOSS_L2, true>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, fuse); break;
    }
  }
}
void launch_irls_reduce(hipStream_t s, const GNLaunch& g)
{
  if(g.max_points <= 0) return;
  const int ppb = gn_pts_per_block(g.C);
  dispatch_channels(g.C, [&](auto c) { launch_irls_c<decltype(c)::value>(s, g, ppb); });
}
void launch_compact_active(hipStream_t s, const PairJob* jobs, ActiveSet in, int n_in, int* out_list, int* out_count)
{
  hipLaunchKernelGGL(compact_active_kernel, dim3(1), dim3(1024), 0, s, jobs, in, n_in, out_list, out_count);
}
void launch_gn_step(hipStream_t s, const GNLaunch& g, int mode, int max_iterations, int max_fun_evals, float p_tol,
                    float f_tol, float g_tol)
{
  const int ppb = gn_pts_per_block(g.C);
  const int fuse = (g.C == 8 && g.fuse_frozen && !g.fast_warp && g.interp == BPVO_INTERP_LINEAR) ? 1 : 0;
  hipLaunchKernelGGL(gn_step_kernel, dim3(g.npairs), dim3(64), 0, s, g.jobs, ppb, mode, max_iterations, max_fun_evals, p_tol,
                     f_tol, g_tol, g.active, fuse);
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PK_THREADS), kMedianLds, s, g.jobs, g.npairs, ppb, prm, fuse, ctl, timeout);
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
template <int C>
static hipError_t launch_gn_team_c(hipStream_t s, const GNTeamLaunch& t, const GNParams& prm)
{
  const int ppb = gn_pts_per_block(C);
  const int fuse = (C == 8 && t.fuse_frozen) ? 1 : 0;
  auto go = [&](auto kern) -> hipError_t {
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
    hipLaunchKernelGGL(kern, dim3(t.team_size, t.n_teams), dim3(PK_THREADS), kMedianLds, s, t.jobs_all, t.job_pitch, t.n_pairs, t.level_hi, t.level_lo, ppb,
                       prm, fuse, t.scale_is_moot, t.ctl, t.timeout_ticks);
    return hipGetLastError();
  };
  switch(t.loss) {
    case BPVO_LOSS_HUBER: return go(gn_team_kernel<C, BPVO_LOSS_HUBER>);
    case BPVO_LOSS_TUKEY: return go(gn_team_kernel<C, BPVO_LOSS_TUKEY>);
    default: return go(gn_team_kernel<C, BPVO_LOSS_L2>);
  }
}
hipError_t launch_gn_team(hipStream_t s, const GNTeamLaunch& t, int max_iterations, int max_fun_evals, float p_tol, float f_tol, float g_tol)
{
  GNParams prm;
  prm.max_iterations = max_iterations; prm.max_fun_evals = max_fun_evals; prm.p_tol = p_tol; prm.f_tol = f_tol; prm.g_tol = g_tol;
  if(t.C == 8) return launch_gn_team_c<8>(s, t, prm);
  return launch_gn_team_c<1>(s, t, prm);
}
void launch_prepare_linearize(hipStream_t s, const PairJob* job, const float* T, int reset_scale, int level, float given_scale)
{
  hipLaunchKernelGGL(prepare_linearize_kernel, dim3(1), dim3(64), 0, s, job, T, reset_scale, level, given_scale);
}
template <int C>
static void launch_weights_c(hipStream_t s, const PairJob* job, int n, int loss, float* w_out)
{
  const dim3 grid((n + 255) / 256);
  switch(loss) {
    case BPVO_LOSS_HUBER: hipLaunchKernelGGL((weights_kernel<C, BPVO_LOSS_HUBER>), grid, dim3(256), 0, s, job, w_out); break;
    case BPVO_LOSS_TUKEY: hipLaunchKernelGGL((weights_kernel<C, BPVO_LOSS_TUKEY>), grid, dim3(256), 0, s, job, w_out); break;
    default: hipLaunchKernelGGL((weights_kernel<C, BPVO_LOSS_L2>), grid, dim3(256), 0, s, job, w_out); break;
  }
}
void launch_weights(hipStream_t s, const PairJob* job, int n, int C, int loss, float* w_out)
{
  if(n <= 0) return;
  dispatch_channels(C, [&](auto c) { launch_weights_c<decltype(c)::value>(s, job, n, loss, w_out); });
}
template <int C>
static void launch_count_good_c(hipStream_t s, const PairJob* job, int n, int loss, float thr, unsigned int* count)
{
  const dim3 grid((n + 255) / 256);
  switch(loss) {
    case BPVO_LOSS_HUBER: hipLaunchKernelGGL((count_good_kernel<C, BPVO_LOSS_HUBER>), grid, dim3(256), 0, s, job, thr, count); break;
    case BPVO_LOSS_TUKEY: hipLaunchKernelGGL((count_good_kernel<C, BPVO_LOSS_TUKEY>), grid, dim3(256), 0, s, job, thr, count); break;
    default: hipLaunchKernelGGL((count_good_kernel<C, BPVO_LOSS_L2>), grid, dim3(256), 0, s, job, thr, count); break;
  }
}
void launch_count_good(hipStream_t s, const PairJob* job, int n, int C, int loss, float thr, unsigned int* count)
{
  if(n <= 0) return;
  dispatch_channels(C, [&](auto c) { launch_count_good_c<decltype(c)::value>(s, job, n, loss, thr, count); });
}
void launch_pack_records(hipStream_t s, const PairJob* jobs, int n, int L, float* records)
{
  hipLaunchKernelGGL(pack_records_kernel, dim3((n + 63) / 64), dim3(64), 0, s, jobs, n, L, records);
}

}  // namespace bpvo_hip
