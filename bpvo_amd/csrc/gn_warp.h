// K6 warp + residual, device side: projection, validity, (cached) bilinear taps, residuals of one template point / one 256-point chunk.
#pragma once
#include "gn_common.h"

namespace bpvo_hip {

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

  // generic layout: records of `pitch` floats (= C, or the whole channel count when this job is one channel group of a wide descriptor)
  const int PT = (C == 8 || C == 1) ? C : j.pitch;
  if(valid) {
    const double wx = 1.0 - xf, wy = 1.0 - yf;
    const float* __restrict__ d0 = j.desc + ((size_t) yi * W + xi) * PT;
    const float* __restrict__ d1 = d0 + (size_t) W * PT;
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
        I00[c] = d0[c]; I01[c] = d0[PT + c]; I10[c] = d1[c]; I11[c] = d1[PT + c];
        I0[c] = j.pix[(size_t) i * PT + c];
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
        for(int c = 0; c < C; ++c) res[c] = 0.0f - j.pix[(size_t) i * PT + c];
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
// dense: the bracket step's dense form (bracket_chunk; chain launches only).
template <int C, bool FAST>
__device__ __forceinline__ void warp_chunk(const PairJob& j, int mode, unsigned chunk, BracketLds& s, bool dense = false)
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
    } else {      // generic C: point-major records [N][pitch]
#pragma unroll
      for(int c = 0; c < C; ++c) j.r[(size_t) i * j.pitch + c] = res[c];
    }
  }
  // bracket pass of the exact median (see bracket_chunk) while the residuals are in registers
  if(mode != 2 && (st->delta_scale > 1e-6f) && st->median_valid) {
    if(dense) bracket_chunk<C, true>(j, st->lo_key, st->hi_key, valid && in_block, hit && valid && in_block, res, chunk, (int) (threadIdx.x >> 6), s, true);
    else bracket_chunk<C, false>(j, st->lo_key, st->hi_key, valid && in_block, hit && valid && in_block, res, chunk, (int) (threadIdx.x >> 6), s, true);
  }
}

}  // namespace bpvo_hip
