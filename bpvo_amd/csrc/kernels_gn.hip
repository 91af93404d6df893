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

#include "gn_common.h"
#include "gn_warp.h"
#include "gn_median.h"
#include "gn_irls.h"
#include "gn_step.h"

namespace bpvo_hip {

template <int C, bool FAST>
__global__ __launch_bounds__(K6_BLOCK) void warp_residual_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int mode, int dense)
{
  __shared__ BracketLds s;
  warp_chunk<C, FAST>(jobs[active_workspace(act, blockIdx.y)], mode, blockIdx.x, s, dense != 0);
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

// Channels are handled in GROUPS of G (8 where C is a multiple of 8: the 32 bytes of a bit-planes pixel in one go; else 4, 2 (C = 10) or 1), and a
// group's 4 x 4 footprint ROW PAIR by row pair: the taps of two rows — 2 x 4 x G floats, every row one contiguous run of 4 x C floats that is requested
// once — are loaded, reduced along x to d[row][channel] (kCubic: dot4 with the x coefficients; kCubicHermite: interp_hermite at xf), and only then
// (sched_barrier) the next two rows are requested; the y pass runs on the four d's.  All channels and rows at once (the first form of this kernel)
// kept 16 x C tap registers live: 198 VGPRs / 2 waves per SIMD for C = 8, spills from C = 24 on.  Groups of FOUR channels over all four rows (tried
// first this round) fit the registers but requested every 128-byte row segment twice, half a record at a time: kCubic 113 -> 149 us per launch
// (profiles/r06_interp_kernel.txt).  Per channel the arithmetic and its order are unchanged.
template <int C> struct InterpGroup { static constexpr int G = (C % 8 == 0) ? 8 : (C % 4 == 0) ? 4 : (C % 2 == 0) ? 2 : 1; };
template <int G>
__device__ __forceinline__ void load_group(const float* __restrict__ p, float (&v)[G])
{
  if constexpr(G == 8) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else if constexpr(G == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else if constexpr(G == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
  else v[0] = p[0];
}

#ifndef ROW_BARRIER_AT
#define ROW_BARRIER_AT 1
#endif
template <int C, int INTERP>
__global__ __launch_bounds__(K6_BLOCK) void warp_residual_interp_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int dense)
{
  constexpr int interp = INTERP;
  const PairJob& j = jobs[active_workspace(act, blockIdx.y)];
  const GNState* __restrict__ st = j.st;
  if(!st->active) return;
  const int n = j.n;
  if((int) (blockIdx.x * K6_BLOCK) >= n) return;
  if(blockIdx.x == 0 && threadIdx.x == 0) j.cnt[4] += (unsigned long long) n;   // points this kernel processes (measurement)

  float P[12];
  projection_matrix(j, st->T, P);

  const int i_raw = blockIdx.x * K6_BLOCK + threadIdx.x;
  const bool in_block = i_raw < n;
  const int i = in_block ? i_raw : n - 1;
  const int W = j.cols, R = j.rows;
  const float4 X = j.pts[i];
  constexpr bool two_tap = interp == BPVO_INTERP_COSINE;
  constexpr int border_lo = two_tap ? 0 : 1, border_hi = two_tap ? 1 : 3;
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

  constexpr int G = InterpGroup<C>::G;
  const size_t PT = (C == 8 || C == 1) ? (size_t) C : (size_t) j.pitch;      // record pitch (a channel group of a wide descriptor: the whole channel count)
  // Tap cache (C = 8 and 1, as for kLinear: gn_warp.h): the point's whole footprint — 2 x 2 or 4 x 4 taps of C channels — kept per point in a
  // tiled, coalesced buffer keyed by the footprint's origin (yi << 16 | xi).  A template point moves by less than a pixel between iterations, so
  // at the sparse levels most lookups hit, and a hit replaces the gather — for 4 x 4 taps of eight channels four 128-byte row segments that
  // straddle ~7 HBM lines (914 B of traffic per point measured, profiles/r06_interp_kernel.txt) — by one coalesced read of 512 B.  Dense levels of
  // a batch run without it (tapcache_on = 0: neighbours share their lines there).
  constexpr bool kCache = (C == 8 || C == 1);
  constexpr int TAPS = two_tap ? 4 : 16;                 // taps per channel
  constexpr int PIECES = kCache ? (TAPS * C) / 4 : 1;    // 16-byte pieces of the cached footprint: C = 8: 8 / 32, C = 1: 1 / 4
  bool cached = false, hit = false;
  unsigned key = 0;
  float4* const tc = kCache ? reinterpret_cast<float4*>(j.tapcache.get()) : nullptr;
  if constexpr(kCache) {
    cached = j.tapcache_on != 0 && j.tapkey;             // (uniform over the workspace)
    key = ((unsigned) yi << 16) | (unsigned) xi;
    hit = cached && valid && j.tapkey[i] == key;
  }
  float res[C];
#pragma unroll
  for(int c = 0; c < C; ++c) res[c] = 0.0f;
  if(valid) {
    // coefficients of the point (f32; the cosine's from a double cosine) — once, not per channel
    float Cx[4] = {0, 0, 0, 0}, Cy[4] = {0, 0, 0, 0};
    if constexpr(two_tap) {
      float c2[2];
      interp_cosine(xf, c2); Cx[0] = c2[0]; Cx[1] = c2[1];
      interp_cosine(yf, c2); Cy[0] = c2[0]; Cy[1] = c2[1];
    } else if constexpr(interp == BPVO_INTERP_CUBIC) {
      interp_cubic(xf, Cx);
      interp_cubic(yf, Cy);
    }
    size_t row_off[4];      // float offset of (row, xi) for the 4 x 4 forms: rows yi-1 .. yi+2, the last clamped (Q21)
#pragma unroll
    for(int k = 0; k < 4; ++k) row_off[k] = ((size_t) min(yi - 1 + k, R - 1) * W + xi) * PT;
    const size_t off00 = ((size_t) yi * W + xi) * PT;
#pragma unroll
    for(int g = 0; g < C / G; ++g) {
      const int c0 = g * G;
      float I0[G];
      if constexpr(C == 8) {
        const float4* p0 = reinterpret_cast<const float4*>(j.pix.get());
        const float4 ta = p0[tile_index<2>(i, 0)], tb = p0[tile_index<2>(i, 1)];
        I0[0] = ta.x; I0[1] = ta.y; I0[2] = ta.z; I0[3] = ta.w; I0[4] = tb.x; I0[5] = tb.y; I0[6] = tb.z; I0[7] = tb.w;
      } else {
#pragma unroll
        for(int q = 0; q < G; ++q) I0[q] = j.pix[(size_t) i * PT + c0 + q];
      }
      if constexpr(two_tap) {
        float a[G], b[G], c[G], d[G];      // (yi, xi), (yi, xi+1), (yi+1, xi), (yi+1, xi+1)
        const float* __restrict__ d0 = j.desc + off00 + c0;
        if(kCache && hit) {
          if constexpr(C == 8) {      // pieces 0, 1: tap a; 2, 3: b; 4, 5: c; 6, 7: d (the layout of kLinear's cache)
            float4 q[8];
#pragma unroll
            for(int p = 0; p < 8; ++p) q[p] = tc[tile_index<8>(i, p)];
            const float4* qq = q;
            auto put = [&](float (&v)[G], int t) { v[0] = qq[2 * t].x; v[1] = qq[2 * t].y; v[2] = qq[2 * t].z; v[3] = qq[2 * t].w; v[4] = qq[2 * t + 1].x; v[5] = qq[2 * t + 1].y; v[6] = qq[2 * t + 1].z; v[7] = qq[2 * t + 1].w; };
            if constexpr(G == 8) { put(a, 0); put(b, 1); put(c, 2); put(d, 3); }
          } else if constexpr(C == 1) {
            const float4 q = tc[i];
            a[0] = q.x; b[0] = q.y; c[0] = q.z; d[0] = q.w;
          }
        } else {
          load_group<G>(d0, a); load_group<G>(d0 + PT, b);
          load_group<G>(d0 + (size_t) W * PT, c); load_group<G>(d0 + (size_t) W * PT + PT, d);
          if(kCache && cached && in_block) {
            if constexpr(C == 8 && G == 8) {
              auto get = [&](const float (&v)[G], int h) { return make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]); };
              tc[tile_index<8>(i, 0)] = get(a, 0); tc[tile_index<8>(i, 1)] = get(a, 1); tc[tile_index<8>(i, 2)] = get(b, 0); tc[tile_index<8>(i, 3)] = get(b, 1);
              tc[tile_index<8>(i, 4)] = get(c, 0); tc[tile_index<8>(i, 5)] = get(c, 1); tc[tile_index<8>(i, 6)] = get(d, 0); tc[tile_index<8>(i, 7)] = get(d, 1);
            } else if constexpr(C == 1) {
              tc[i] = make_float4(a[0], b[0], c[0], d[0]);
            }
          }
        }
#pragma unroll
        for(int q = 0; q < G; ++q) {
          const float e1 = a[q] * Cx[0] + b[q] * Cx[1];
          const float e2 = c[q] * Cx[0] + d[q] * Cx[1];
          res[c0 + q] = (Cy[0] * e1 + Cy[1] * e2) - I0[q];
        }
      } else {
        float d[4][G];      // the x pass of row k: kCubic dot4(taps, Cx), kCubicHermite interp_hermite(taps, xf)
#pragma unroll
        for(int row = 0; row < 4; ++row) {
          float t[4][G];      // [tap][channel of the group] of this row
          // ONE load path for a hit and a miss — the address is the cached piece or the descriptor's tap — so that the two cases do not hold two
          // sets of tap registers; cached footprint: piece (row * 4 + tap) * 2 + half for C = 8 (32 pieces), piece = row for C = 1 (its four taps)
          if constexpr(C == 8 && G == 8) {
            // the row's eight 16-byte loads from ONE per-lane base and two per-lane strides — cached: piece (row * 4 + tap) * 2 + half at
            // cb + piece * kTile; gathered: tap m at the row's record + 32 m bytes — formed row by row (the asm keeps the compiler from
            // materialising all 32 piece addresses of the footprint ahead of the first load: 64 registers)
            const char* rb = hit ? reinterpret_cast<const char*>(tc + tile_index<PIECES>(i, 0)) + (size_t) row * 8 * kTile * 16
                                 : reinterpret_cast<const char*>(j.desc + row_off[row]);
            asm volatile("" : "+v"(rb));
            const unsigned tap_stride = hit ? 2u * kTile * 16u : 32u, half_off = hit ? kTile * 16u : 16u;
#pragma unroll
            for(int m = 0; m < 4; ++m) {
              const float4 lo = *reinterpret_cast<const float4*>(rb + m * tap_stride), hi = *reinterpret_cast<const float4*>(rb + m * tap_stride + half_off);
              t[m][0] = lo.x; t[m][1] = lo.y; t[m][2] = lo.z; t[m][3] = lo.w; t[m][4] = hi.x; t[m][5] = hi.y; t[m][6] = hi.z; t[m][7] = hi.w;
            }
            if(cached && !hit && in_block) {
              char* wb = reinterpret_cast<char*>(tc + tile_index<PIECES>(i, 0)) + (size_t) row * 8 * kTile * 16;
#pragma unroll
              for(int m = 0; m < 4; ++m) {
                *reinterpret_cast<float4*>(wb + (size_t) (2 * m) * kTile * 16) = make_float4(t[m][0], t[m][1], t[m][2], t[m][3]);
                *reinterpret_cast<float4*>(wb + (size_t) (2 * m + 1) * kTile * 16) = make_float4(t[m][4], t[m][5], t[m][6], t[m][7]);
              }
            }
          } else if constexpr(C == 1) {
            const float* const cb = reinterpret_cast<const float*>(tc + tile_index<PIECES>(i, 0));      // row r of point i: cb + r * kTile * 4
#pragma unroll
            for(int m = 0; m < 4; ++m) t[m][0] = *(hit ? cb + (size_t) row * kTile * 4 + m : j.desc + row_off[row] + m);
            if(cached && !hit && in_block) (tc + tile_index<PIECES>(i, 0))[(size_t) row * kTile] = make_float4(t[0][0], t[1][0], t[2][0], t[3][0]);
          } else {
#pragma unroll
            for(int m = 0; m < 4; ++m) load_group<G>(j.desc + row_off[row] + (size_t) m * PT + c0, t[m]);
          }
#pragma unroll
          for(int q = 0; q < G; ++q) {
            if constexpr(interp == BPVO_INTERP_CUBIC) d[row][q] = dot4(t[0][q], t[1][q], t[2][q], t[3][q], Cx);
            else d[row][q] = interp_hermite(t[0][q], t[1][q], t[2][q], t[3][q], xf);
          }
          if(ROW_BARRIER_AT < 0 ? row < 3 : row == ROW_BARRIER_AT) __builtin_amdgcn_sched_barrier(0);      // the next rows are requested behind the x pass of these
        }
#pragma unroll
        for(int q = 0; q < G; ++q) {
          if constexpr(interp == BPVO_INTERP_CUBIC) {
            const float dd[4] = {d[0][q], d[1][q], d[2][q], d[3][q]};
            res[c0 + q] = dot4(Cy[0], Cy[1], Cy[2], Cy[3], dd) - I0[q];
          } else {
            res[c0 + q] = interp_hermite(d[0][q], d[1][q], d[2][q], d[3][q], yf) - I0[q];
          }
        }
      }
      if(g + 1 < C / G) __builtin_amdgcn_sched_barrier(0);      // the next group's loads stay behind this group's arithmetic
    }
  }
  if constexpr(kCache) {
    if(cached && valid && !hit && in_block) j.tapkey[i] = key;
  }
  if(in_block) {
    if constexpr(C == 8) {
      float4* o = reinterpret_cast<float4*>(j.r.get());
      o[tile_index<2>(i, 0)] = make_float4(res[0], res[1], res[2], res[3]);
      o[tile_index<2>(i, 1)] = make_float4(res[4], res[5], res[6], res[7]);
    } else {
#pragma unroll
      for(int c = 0; c < C; ++c) j.r[(size_t) i * PT + c] = res[c];
    }
  }
  if((st->delta_scale > 1e-6f) && st->median_valid) bracket_block<C>(j, st->lo_key, st->hi_key, valid && in_block, hit && in_block, res, dense != 0);
}

// K7b: one workgroup per workspace — bracketed select among the candidates, or the full 3-pass select.
// Two shapes.  1024 threads with the full 123 KB (one workgroup per CU): the fastest single selection — launches of up to one
// workgroup per CU.  512 threads with 53 KB (three workgroups per CU): launches of MORE workgroups
// than CUs, which with the first shape run in waves of 256 workgroups at ~10 us each (1024 pairs: 41 / 32 / 22 / 12 us per launch as
// the pairs converge).  The selection is exact in either shape.
template <int C, int NT, int COPIES, int CACHE>
__global__ __launch_bounds__(NT) void median_finish_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int dense)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.x)];
  GNState* st = j.st;
  if(!st->active) return;
  if(!(st->delta_scale > 1e-6f)) return;   // scale is stable: frozen for the rest of the level (mestimator.cc:472,485)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  median_block<C, NT, COPIES, CACHE>(j, st, smem_raw, true, dense != 0);
}

// Two instantiations share the work of a launch slot: FUSED = false handles the workspaces whose scale still moves, FUSED =
// true (136 VGPRs instead of 103: kept out of the plain kernel's register budget) the frozen ones.
// step.on (only without fuse_frozen, where one launch serves every workspace): the last tile of a workspace takes the Gauss-Newton
// step (gn_last_tile); a workspace without points still has its tile 0 for that.
struct GNStepArgs { int on; GNParams prm; };
// (waves per SIMD the reduction had on its own: left alone the compiler gives the step's serial code 180 registers, and the whole
// launch its occupancy)
constexpr int irls_min_waves(int C) { return C <= 10 ? 4 : C <= 24 ? 3 : 2; }
template <int C, int LOSS, bool FUSED>
__global__ __launch_bounds__(GN_BLOCK, irls_min_waves(C)) void irls_reduce_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int pts_per_block, int fuse_frozen,
                                                               GNStepArgs step)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.y)];
  const GNState* __restrict__ st = j.st;
  if(!st->active) return;
  if(fuse_frozen && (FUSED != !(st->delta_scale > 1e-6f))) return;
  if(!step.on) { irls_block<C, LOSS, FUSED>(j, st, pts_per_block); return; }
  const int tiles = max(1, (j.n + pts_per_block - 1) / pts_per_block);
  if((int) blockIdx.x >= tiles) return;
  irls_block<C, LOSS, FUSED>(j, st, pts_per_block, true);
  if(threadIdx.x >= 64) return;
  __shared__ GNStepLds s_step;
  if(gn_last_tile(j, tiles)) gn_step_wave<true>(j, s_step, pts_per_block, 0, step.prm, 0);
}
// ... or ONE launch serves both kinds with a per-workspace branch (C = 8): every workgroup then runs at the fused form's
// register budget (3 waves per SIMD instead of 4), but small launches — the 128-pair shard of config 5, single pairs — do not
// pay a second, half-empty launch per iteration (each costs its ramp and drain: at 128 pairs the two launches took 59 us where
// the bytes are worth 38).
#ifndef K8_BOTH_WAVES
#define K8_BOTH_WAVES 3   // the floor; with the fused multiply-adds of irls_mad the kernel takes 127 registers and gets four (FORCED to four before them, 12
                          // registers spilled: -1.5 % at 1024 pairs, -2 % at 128, profiles/r04_step_in_reduce.txt)
#endif
template <int LOSS>
__global__ __launch_bounds__(GN_BLOCK, K8_BOTH_WAVES) void irls_reduce_both_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int pts_per_block, GNStepArgs step)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.y)];
  const GNState* __restrict__ st = j.st;
  if(!st->active) return;
  const int tiles = max(1, (j.n + pts_per_block - 1) / pts_per_block);
  if(step.on && (int) blockIdx.x >= tiles) return;
  if(st->delta_scale > 1e-6f) irls_block<8, LOSS, false>(j, st, pts_per_block, step.on != 0);
  else irls_block<8, LOSS, true>(j, st, pts_per_block, step.on != 0);
  if(!step.on || threadIdx.x >= 64) return;
  __shared__ GNStepLds s_step;
  if(gn_last_tile(j, tiles)) gn_step_wave<true>(j, s_step, pts_per_block, 0, step.prm, 1);
}

__global__ __launch_bounds__(64, 4) void gn_step_kernel(const PairJob* __restrict__ jobs, int pts_per_block, int mode,
                                                     int max_iterations, int max_fun_evals, float p_tol, float f_tol,
                                                     float g_tol_param, ActiveSet act, int fuse_frozen)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.x)];
  GNState* gst = j.st;
  if(!gst->active) return;

  __shared__ GNStepLds s_step;
  const GNParams prm = {max_iterations, max_fun_evals, p_tol, f_tol, g_tol_param};
  gn_step_wave<false>(j, s_step, pts_per_block, mode, prm, fuse_frozen);
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
__global__ void set_pose_kernel(const PairJob* jobs, const float* T_init, int n, unsigned* clear, int clear_words)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  for(int i = p; i < clear_words; i += gridDim.x * blockDim.x) clear[i] = 0u;      // (the persistent kernel's control words: no memset launch of their own)
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

// A few pairs: the job table's upload, the initial poses and the cleared control words in ONE launch of one workgroup — the table is read
// straight from the pinned host rows (8-byte words), the poses are set through the HOST copy of the coarsest level's jobs (the device
// copy is being written by the neighbours).  A copy + a kernel otherwise: two 5 us stops of the stream in front of the first iteration.
__global__ __launch_bounds__(256) void set_pose_upload_kernel(unsigned long long* __restrict__ d_table, const unsigned long long* __restrict__ h_table,
                                                              size_t words, const PairJob* h_jobs_coarsest, const float* T_init, int n, unsigned* clear,
                                                              int clear_words)
{
  for(size_t i = threadIdx.x; i < words; i += 256) d_table[i] = h_table[i];
  for(int i = threadIdx.x; i < clear_words; i += 256) clear[i] = 0u;
  const int p = threadIdx.x;
  if(p >= n) return;
  GNState* st = h_jobs_coarsest[p].st;
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
  if(blockIdx.x != 0) return;
  if(threadIdx.x < kDenseRuns && j.med_blk) reinterpret_cast<uint4*>(j.med_blk.get())[j.med_tot + 8 * threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);      // the totals of the dense bracket form start a level at zero
  if(threadIdx.x != 0) return;
  gn_level_reset(j.st, level, scale_is_moot, j.n);
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

// ... and for descriptors of more than 48 channels (point-major records of C floats, run-time channel loop)
__device__ __forceinline__ float mest_weight_rt(int loss, float r, float sigma_inv)
{
  return loss == BPVO_LOSS_HUBER ? mest_weight<BPVO_LOSS_HUBER>(r, sigma_inv) : loss == BPVO_LOSS_TUKEY ? mest_weight<BPVO_LOSS_TUKEY>(r, sigma_inv) : 1.0f;
}
__global__ __launch_bounds__(256) void weights_wide_kernel(const PairJob* job, int C, int loss, float* w_out /*[n][C] point-major*/)
{
  const size_t k = (size_t) blockIdx.x * 256 + threadIdx.x;
  if(k >= (size_t) job->n * C) return;
  w_out[k] = mest_weight_rt(loss, job->r[k], 1.0f / job->st->scale);
}
__global__ __launch_bounds__(256) void count_good_wide_kernel(const PairJob* job, int C, int loss, float thr, unsigned int* count)
{
  const size_t total = (size_t) job->n * C;
  const float sigma_inv = 1.0f / job->st->scale;
  unsigned good = 0;
  for(size_t k = (size_t) blockIdx.x * 256 + threadIdx.x; k < total; k += (size_t) gridDim.x * 256) good += mest_weight_rt(loss, job->r[k], sigma_inv) > thr ? 1u : 0u;
  __shared__ unsigned s_good[4];
  good = wave_sum_u32(good);
  if((threadIdx.x & 63) == 0) s_good[threadIdx.x >> 6] = good;
  __syncthreads();
  if(threadIdx.x == 0) {
    const unsigned t = s_good[0] + s_good[1] + s_good[2] + s_good[3];
    if(t) atomicAdd(count, t);
  }
}

// getPointCloudFromRefFrame + GetColor (reference: bpvo/vo.cc:250-281) on the device: one 32-byte PointWithInfo per template point of the level the
// estimate ended on — the point, the key frame's grey value at its projection (getImagePoint, bpvo/rigid_body_warp.h:123-128: x = K X in f32, index
// order; DisparitySpaceWarp: disparity_space_warp.h:73-76), and weights[i] of the last linearisation, i.e. the weight of CHANNEL 0's residual
// (the reference indexes the channel-major weight array with the point index).  The records stay on the device until somebody asks for them.
struct CloudArgs { float K[9]; int rows, cols, dspace, C, loss; };
__global__ __launch_bounds__(256) void point_cloud_kernel(const PairJob* job, const uint8_t* __restrict__ img, CloudArgs a, bpvo_hip_point_with_info* __restrict__ out)
{
  const PairJob& j = *job;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if(i >= j.n) return;
  const float4 X = j.pts[i];
  float x[3];
#pragma unroll
  for(int r = 0; r < 3; ++r) {
    float s = a.K[r * 3 + 0] * X.x;
    s += a.K[r * 3 + 1] * X.y;
    s += a.K[r * 3 + 2] * X.z;
    x[r] = s;
  }
  const float z_i = 1.0f / x[2];
  float u = z_i * x[0], v = z_i * x[1];
  if(a.dspace) { u = X.x + a.K[2]; v = X.y + a.K[5]; }
  uint8_t col = 0;
  if(v >= 0 && v < a.rows && u >= 0 && u < a.cols) col = img[(size_t) ((int) v) * a.cols + (int) u];
  const float r0 = a.C == 8 ? j.r[tile_index<2>(i, 0) * 4] : j.r[(size_t) i * (a.C == 1 ? 1 : j.pitch)];
  bpvo_hip_point_with_info pw;
  pw.xyzw[0] = X.x; pw.xyzw[1] = X.y; pw.xyzw[2] = X.z; pw.xyzw[3] = X.w;
  pw.rgba[0] = col; pw.rgba[1] = col; pw.rgba[2] = col; pw.rgba[3] = 255;
  pw.weight = mest_weight_rt(a.loss, r0, 1.0f / j.st->scale);
#pragma unroll
  for(int k = 0; k < 8; ++k) pw.pad[k] = 0;
  out[i] = pw;
}

// (grid-stride over the points and ONE add per workgroup: a thousand workgroups of four waves adding to one word took 55 us on a
// 300 k-point template — the adds serialise at the L2 — where the points take 5)
template <int C, int LOSS>
__global__ __launch_bounds__(256) void count_good_kernel(const PairJob* job, float thr, unsigned int* count)
{
  unsigned good = 0;
  const int n = job->n;
  const float sigma_inv = 1.0f / job->st->scale;
  for(int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    if constexpr(C == 8) {
      const float4* q = reinterpret_cast<const float4*>(job->r.get());
      const float4 a = q[tile_index<2>(i, 0)], b = q[tile_index<2>(i, 1)];
      good += (mest_weight<LOSS>(a.x, sigma_inv) > thr) + (mest_weight<LOSS>(a.y, sigma_inv) > thr) +
              (mest_weight<LOSS>(a.z, sigma_inv) > thr) + (mest_weight<LOSS>(a.w, sigma_inv) > thr) +
              (mest_weight<LOSS>(b.x, sigma_inv) > thr) + (mest_weight<LOSS>(b.y, sigma_inv) > thr) +
              (mest_weight<LOSS>(b.z, sigma_inv) > thr) + (mest_weight<LOSS>(b.w, sigma_inv) > thr);
    } else {
#pragma unroll
      for(int c = 0; c < C; ++c) good += mest_weight<LOSS>(job->r[(size_t) i * C + c], sigma_inv) > thr;
    }
  }
  __shared__ unsigned s_good[4];
  good = wave_sum_u32(good);
  if((threadIdx.x & 63) == 0) s_good[threadIdx.x >> 6] = good;
  __syncthreads();
  if(threadIdx.x == 0) {
    const unsigned t = s_good[0] + s_good[1] + s_good[2] + s_good[3];
    if(t) atomicAdd(count, t);
  }
}

// 32-float result record per pair for the RCCL gather: pose 3x4 (12), numIterations per level (8), status per level (8),
// total function evaluations are not kept per level so [28..31] = {finalError of the finest level, n_valid, 0, 0}
// h_states (pinned host, or null): the states — and h_ctl: the persistent kernel's control words — written to the host by this launch as well
// (a few pairs: two copies of a few hundred bytes are two more stops of the stream behind the last iteration)
__global__ void pack_records_kernel(const PairJob* jobs, int n, int L, float* records, const GNState* d_states, GNState* h_states, const unsigned* d_ctl,
                                    unsigned* h_ctl, int ctl_words, unsigned* zero)
{
  if(zero && blockIdx.x == 0 && threadIdx.x == 0) *zero = 0u;      // (the counter of the count_good launch that follows: no memset launch of its own)
  if(h_states) {
    constexpr int kWords = (int) (sizeof(GNState) / 4);
    for(int q = 0; q < n; ++q) {
      const GNState* st = jobs[q].st;
      const unsigned* src = reinterpret_cast<const unsigned*>(st);
      unsigned* dst = reinterpret_cast<unsigned*>(h_states + (st - d_states));
      for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < kWords; i += gridDim.x * blockDim.x) dst[i] = src[i];
    }
    for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < ctl_words; i += gridDim.x * blockDim.x) h_ctl[i] = d_ctl[i];
  }
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
// entries of kPartialStride floats a workspace's partials need: ceil(cap / points per tile) tiles, twice (pk_partials puts the
// odd-parity buffer right behind the ntiles entries of the even one) — whatever the tile size of the build and however small the level
int gn_partials_entries(int cap, int C)
{
  const int ppb = gn_pts_per_block(C);
  return 2 * std::max(1, (cap + ppb - 1) / ppb);
}

void launch_set_pose(hipStream_t s, const PairJob* jobs, const float* T_init, int n, unsigned* clear, int clear_words)
{
  hipLaunchKernelGGL(set_pose_kernel, dim3((n + 63) / 64), dim3(64), 0, s, jobs, T_init, n, clear, clear_words);
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
      constexpr int CC = decltype(c)::value;
      switch(g.interp) {
        case BPVO_INTERP_COSINE: hipLaunchKernelGGL((warp_residual_interp_kernel<CC, BPVO_INTERP_COSINE>), grid, dim3(K6_BLOCK), 0, s, g.jobs, g.active, g.dense_candidates); break;
        case BPVO_INTERP_CUBIC: hipLaunchKernelGGL((warp_residual_interp_kernel<CC, BPVO_INTERP_CUBIC>), grid, dim3(K6_BLOCK), 0, s, g.jobs, g.active, g.dense_candidates); break;
        default: hipLaunchKernelGGL((warp_residual_interp_kernel<CC, BPVO_INTERP_CUBIC_HERMITE>), grid, dim3(K6_BLOCK), 0, s, g.jobs, g.active, g.dense_candidates); break;
      }
    });
    return;
  }
  if(g.fast_warp) {
    dispatch_channels(g.C, [&](auto c) {
      hipLaunchKernelGGL((warp_residual_kernel<decltype(c)::value, true>), grid, dim3(K6_BLOCK), 0, s, g.jobs, g.active, 0, g.dense_candidates);
    });
  } else {
    dispatch_channels(g.C, [&](auto c) {
      constexpr int CC = decltype(c)::value;
      hipLaunchKernelGGL((warp_residual_kernel<CC, false>), grid, dim3(K6_BLOCK), 0, s, g.jobs, g.active, (CC == 8 && g.fuse_frozen) ? 1 : 0, g.dense_candidates);
    });
  }
}
// refresh the residual / valid buffers of the workspaces marked r_stale (fused path) from T_lin, then clear the marks
void launch_refresh_residuals(hipStream_t s, const GNLaunch& g)
{
  if(g.max_points <= 0 || g.C != 8) return;
  const dim3 grid((g.max_points + K6_BLOCK - 1) / K6_BLOCK, g.npairs);
  hipLaunchKernelGGL((warp_residual_kernel<8, false>), grid, dim3(K6_BLOCK), 0, s, g.jobs, ActiveSet(), 2, 0);
  hipLaunchKernelGGL(clear_stale_kernel, dim3((g.npairs + 63) / 64), dim3(64), 0, s, g.jobs, g.npairs);
}
void launch_median(hipStream_t s, const GNLaunch& g)
{
  if(g.max_points <= 0) return;
  // the attribute is per device (a process may hold contexts on several) and the lanes' host threads race here
  static std::once_flag attr_once[64];
  int dev = 0;
  (void) hipGetDevice(&dev);
  std::call_once(attr_once[dev & 63], [] {
    for(int C : {1, 3, 5, 8, 10, 16, 24, 32, 48})
      dispatch_channels(C, [&](auto c) {
        constexpr int CC = decltype(c)::value;
        (void) hipFuncSetAttribute((const void*) median_finish_kernel<CC, MED_THREADS, MED_COPIES, MED_CACHE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) kMedianLds);
        (void) hipFuncSetAttribute((const void*) median_finish_kernel<CC, MED_THREADS_B, MED_COPIES_B, MED_CACHE_B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) kMedianLdsB);
      });
  });
  constexpr int wide_from = 257;      // launches of more workgroups than CUs take the 512-thread shape (profiles/r02_median_shapes.txt, r03_shard_ab_lanes_median.txt)
  dispatch_channels(g.C, [&](auto c) {
    constexpr int CC = decltype(c)::value;
    if(g.npairs >= wide_from)
      hipLaunchKernelGGL((median_finish_kernel<CC, MED_THREADS_B, MED_COPIES_B, MED_CACHE_B>), dim3(g.npairs), dim3(MED_THREADS_B), kMedianLdsB, s, g.jobs, g.active, g.dense_candidates);
    else
      hipLaunchKernelGGL((median_finish_kernel<CC, MED_THREADS, MED_COPIES, MED_CACHE>), dim3(g.npairs), dim3(MED_THREADS), kMedianLds, s, g.jobs, g.active, g.dense_candidates);
  });
}

template <int C>
static void launch_irls_c(hipStream_t s, const GNLaunch& g, int ppb)
{
  const dim3 grid((g.max_points + ppb - 1) / ppb, g.npairs);
  const int fuse = (C == 8 && g.fuse_frozen && !g.fast_warp && g.interp == BPVO_INTERP_LINEAR) ? 1 : 0;
  GNStepArgs step;
  step.on = g.step_in_reduce;
  step.prm = g.step_prm;
  if constexpr(C == 8) {
    if(fuse) {      // one launch serves the workspaces with a moving scale and the frozen ones (per-workspace branch)
      switch(g.loss) {
        case BPVO_LOSS_HUBER: hipLaunchKernelGGL((irls_reduce_both_kernel<BPVO_LOSS_HUBER>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, step); break;
        case BPVO_LOSS_TUKEY: hipLaunchKernelGGL((irls_reduce_both_kernel<BPVO_LOSS_TUKEY>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, step); break;
        default: hipLaunchKernelGGL((irls_reduce_both_kernel<BPVO_LOSS_L2>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, step); break;
      }
      return;
    }
  }
  switch(g.loss) {
    case BPVO_LOSS_HUBER: hipLaunchKernelGGL((irls_reduce_kernel<C, BPVO_LOSS_HUBER, false>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, fuse, step); break;
    case BPVO_LOSS_TUKEY: hipLaunchKernelGGL((irls_reduce_kernel<C, BPVO_LOSS_TUKEY, false>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, fuse, step); break;
    default: hipLaunchKernelGGL((irls_reduce_kernel<C, BPVO_LOSS_L2, false>), grid, dim3(GN_BLOCK), 0, s, g.jobs, g.active, ppb, fuse, step); break;
  }
}
void launch_irls_reduce(hipStream_t s, const GNLaunch& g)
{
  if(g.max_points <= 0) return;
  if(g.reference_reduction) { launch_reference_reduce(s, g); return; }
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
  // (reference_reduction: ONE partial per workspace whatever its size — a tile that holds every point)
  const int ppb = g.reference_reduction ? (1 << 30) : gn_pts_per_block(g.C);
  const int fuse = (g.C == 8 && g.fuse_frozen && !g.fast_warp && g.interp == BPVO_INTERP_LINEAR && !g.reference_reduction) ? 1 : 0;
  hipLaunchKernelGGL(gn_step_kernel, dim3(g.npairs), dim3(64), 0, s, g.jobs, ppb, mode, max_iterations, max_fun_evals, p_tol,
                     f_tol, g_tol, g.active, fuse);
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
  if(C > 48) { hipLaunchKernelGGL(weights_wide_kernel, dim3((unsigned) (((size_t) n * C + 255) / 256)), dim3(256), 0, s, job, C, loss, w_out); return; }
  dispatch_channels(C, [&](auto c) { launch_weights_c<decltype(c)::value>(s, job, n, loss, w_out); });
}
template <int C>
static void launch_count_good_c(hipStream_t s, const PairJob* job, int n, int loss, float thr, unsigned int* count)
{
  const dim3 grid(std::min((n + 255) / 256, 256));
  switch(loss) {
    case BPVO_LOSS_HUBER: hipLaunchKernelGGL((count_good_kernel<C, BPVO_LOSS_HUBER>), grid, dim3(256), 0, s, job, thr, count); break;
    case BPVO_LOSS_TUKEY: hipLaunchKernelGGL((count_good_kernel<C, BPVO_LOSS_TUKEY>), grid, dim3(256), 0, s, job, thr, count); break;
    default: hipLaunchKernelGGL((count_good_kernel<C, BPVO_LOSS_L2>), grid, dim3(256), 0, s, job, thr, count); break;
  }
}
void launch_point_cloud(hipStream_t s, const PairJob* job, int n, int C, int loss, const uint8_t* img, int rows, int cols, const float K[9], int dspace,
                        bpvo_hip_point_with_info* out)
{
  if(n <= 0) return;
  CloudArgs a;
  for(int k = 0; k < 9; ++k) a.K[k] = K[k];
  a.rows = rows; a.cols = cols; a.dspace = dspace; a.C = C; a.loss = loss;
  hipLaunchKernelGGL(point_cloud_kernel, dim3((n + 255) / 256), dim3(256), 0, s, job, img, a, out);
}
void launch_count_good(hipStream_t s, const PairJob* job, int n, int C, int loss, float thr, unsigned int* count)
{
  if(n <= 0) return;
  if(C > 48) { hipLaunchKernelGGL(count_good_wide_kernel, dim3((unsigned) std::min<size_t>(((size_t) n * C + 255) / 256, 1024)), dim3(256), 0, s, job, C, loss, thr, count); return; }
  dispatch_channels(C, [&](auto c) { launch_count_good_c<decltype(c)::value>(s, job, n, loss, thr, count); });
}
void launch_pack_records(hipStream_t s, const PairJob* jobs, int n, int L, float* records, const GNState* d_states, GNState* h_states, const unsigned* d_ctl,
                         unsigned* h_ctl, int ctl_words, unsigned* zero)
{
  static_assert(sizeof(GNState) % 4 == 0, "pack_records_kernel moves 4-byte words");
  hipLaunchKernelGGL(pack_records_kernel, dim3((n + 63) / 64), dim3(h_states ? 256 : 64), 0, s, jobs, n, L, records, d_states, h_states, d_ctl, h_ctl,
                     h_ctl ? ctl_words : 0, zero);
}
void launch_set_pose_upload(hipStream_t s, PairJob* d_table, const PairJob* h_table, size_t table_jobs, const PairJob* h_jobs_coarsest, const float* T_init,
                            int n, unsigned* clear, int clear_words)
{
  static_assert(sizeof(PairJob) % 8 == 0, "set_pose_upload_kernel moves 8-byte words");
  hipLaunchKernelGGL(set_pose_upload_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<unsigned long long*>(d_table),
                     reinterpret_cast<const unsigned long long*>(h_table), table_jobs * (sizeof(PairJob) / 8), h_jobs_coarsest, T_init, n, clear, clear_words);
}
}  // namespace bpvo_hip
