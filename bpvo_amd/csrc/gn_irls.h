// K8 M-estimator weights + normal equations, device side: one tile of points by 256 threads (throughput and latency forms).
#pragma once
#include "gn_warp.h"

namespace bpvo_hip {

// ------------------------------------------------------------------------------------------------------------------
// K8 irls_reduce.  reference: MEstimator::ComputeWeights SIMD bodies (bpvo/mestimator.cc:242-282 Huber, :303-366 Tukey;
// `valid` ignored there, Q12/Q14) fused with LinearSystemBuilderReduction::rankUpdatePoint
// (bpvo/linear_system_builder.cc:140-205): w' = w * float(valid); H += (w' J_a) J_b (upper triangle); G += (w' r) J;
// e += (w' r) r.  Summation order differs from the reference's serial loop (Q15): per thread over channels and points,
// then a wavefront shuffle tree, then LDS across the 4 waves; per-block partials are combined in fixed order in f64 by
// gn_step, so the result is deterministic run to run.
constexpr int kNumAcc = 30;   // 21 H + 6 G + e + #valid points + tap-cache hits (fused path)
// The sums of the normal equations are accumulated with fused multiply-adds: H, G and the function value are tolerance-compared with
// the reference (its own summation order and SSE lanes differ from any one here; the rank-2 form above already regroups
// them), a fused multiply-add is the more accurate of the two, and the reduction is co-limited by its VALU work: ~100 of the ~330 flops
// per point go.  Residuals, weights and valid flags — the bit-exact quantities — do not pass through here.
#ifndef K8_FMA
#define K8_FMA 1
#endif
__device__ __forceinline__ float irls_mad(float a, float b, float c)
{
#if K8_FMA
  return __builtin_fmaf(a, b, c);
#else
  return a * b + c;
#endif
}

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
                                          IrlsPartLds& s_part, bool has, float* __restrict__ partials, bool agent_store = false)
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
    } else {      // generic C: point-major r[N][pitch], grad[N][2][pitch] (pitch = C, or the whole channel count for a channel group)
      const size_t PT = (size_t) j.pitch;
#pragma unroll
      for(int c = 0; c < C; ++c) {
        rr[c] = j.r[(size_t) i * PT + c];
        Ix[c] = j.grad[((size_t) i * 2 + 0) * PT + c];
        Iy[c] = j.grad[((size_t) i * 2 + 1) * PT + c];
      }
    }
    float Sxx = 0.0f, Sxy = 0.0f, Syy = 0.0f, Gx = 0.0f, Gy = 0.0f;
#pragma unroll
    for(int c = 0; c < C; ++c) {
      const float r = rr[c];
      const float w = mest_weight<LOSS>(r, sigma_inv) * v;
      const float wx = w * Ix[c], wy = w * Iy[c];
      Sxx = irls_mad(wx, Ix[c], Sxx);
      Sxy = irls_mad(wx, Iy[c], Sxy);
      Syy = irls_mad(wy, Iy[c], Syy);
      Gx = irls_mad(wx, r, Gx);
      Gy = irls_mad(wy, r, Gy);
      acc[27] = irls_mad(w * r, r, acc[27]);
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
        const float pa = irls_mad(Sxy, B[a], Sxx * A[a]);      // coefficient of A[b]
        const float qa = irls_mad(Syy, B[a], Sxy * A[a]);      // coefficient of B[b]
#pragma unroll
        for(int b = a; b < 6; ++b) { acc[idx] = irls_mad(qa, B[b], irls_mad(pa, A[b], acc[idx])); ++idx; }
      }
#pragma unroll
      for(int a = 0; a < 6; ++a) acc[21 + a] = irls_mad(Gy, B[a], irls_mad(Gx, A[a], acc[21 + a]));
    }
  }

  // wavefront tree (64 lanes, the pairing of a __shfl_down ladder without its LDS round trips: wave_tree_sums), then LDS across
  // the 4 waves
  const int lane = vtid & 63, wave = vtid >> 6;
  wave_tree_sums_to<kNumAcc>(acc, lane, s_part[wave]);
  __syncthreads();
  if(vtid < kNumAcc && has) {
    const float v = (s_part[0][vtid] + s_part[1][vtid]) + (s_part[2][vtid] + s_part[3][vtid]);
    // agent_store (step_in_reduce: another workgroup of the SAME launch reads the partial): written through to where every XCD sees
    // it, instead of a release fence that writes back the whole L2
    if(agent_store) __hip_atomic_store(partials + (size_t) tile * kPartialStride + vtid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else partials[(size_t) tile * kPartialStride + vtid] = v;
  }
}

// the form irls_reduce uses: one 256-thread workgroup = one tile
template <int C, int LOSS, bool FUSED>
__device__ __forceinline__ void irls_block(const PairJob& j, const GNState* __restrict__ st, int pts_per_block, bool agent_store = false)
{
  if((int) blockIdx.x * pts_per_block >= j.n) return;
  __shared__ IrlsPartLds s_part;
  irls_tile<C, LOSS, FUSED>(j, st, pts_per_block, blockIdx.x, threadIdx.x, s_part, true, j.partials, agent_store);
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
                                              IrlsPartLds& s_part, bool has, float* __restrict__ partials, bool agent_store = false)
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
      Sxx = irls_mad(wx, Ix[c], Sxx);
      Sxy = irls_mad(wx, Iy[c], Sxy);
      Syy = irls_mad(wy, Iy[c], Syy);
      Gx = irls_mad(wx, r, Gx);
      Gy = irls_mad(wy, r, Gy);
      acc[27] = irls_mad(w * r, r, acc[27]);
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
      const float pa = irls_mad(Sxy, B[a], Sxx * A[a]);
      const float qa = irls_mad(Syy, B[a], Sxy * A[a]);
#pragma unroll
      for(int b = a; b < 6; ++b) { acc[idx] = irls_mad(qa, B[b], irls_mad(pa, A[b], acc[idx])); ++idx; }
    }
#pragma unroll
    for(int a = 0; a < 6; ++a) acc[21 + a] = irls_mad(Gy, B[a], irls_mad(Gx, A[a], acc[21 + a]));
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

  const int lane = vtid & 63, wave = vtid >> 6;
  wave_tree_sums_to<kNumAcc>(acc, lane, s_part[wave]);
  __syncthreads();
  if(vtid < kNumAcc && has) {
    const float v = (s_part[0][vtid] + s_part[1][vtid]) + (s_part[2][vtid] + s_part[3][vtid]);
    if(agent_store) __hip_atomic_store(partials + (size_t) tile * kPartialStride + vtid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else partials[(size_t) tile * kPartialStride + vtid] = v;
  }
}

}  // namespace bpvo_hip
