// Validation mode "reference_reduction": the normal equations summed EXACTLY as the reference's default (serial) build sums them.
//
// reference: LinearSystemBuilderReduction::Run, serial branch (bpvo/linear_system_builder.cc:239-266): for i = 0 .. C*N-1 over the
// channel-major arrays [c*N + i], rankUpdatePoint (:140-205): w' = W[i] * float(valid[i]); the packed upper 2x2 blocks accumulate
// (w' * J[a]) * J[b] — one f32 multiply by w', one f32 multiply, one f32 add per slot and entry; G += (w' * R[i]) * J; res += (w' * R[i]) * R[i];
// toEigen (:207-221) keeps the upper triangle.  The product's own reduction (gn_irls.h) regroups these sums (rank-2 form, FMA, wave tree,
// f64 block combine: SURVEY.md Q15) and is tolerance-compared; THIS kernel is the checker that shows everything else on the path — residuals,
// valid flags, robust scale, weights, Jacobian rows, the 6x6 solve, the SE(3) update, every stop test — to be exact: with it H, G, f_norm, and
// hence every iterate, numIterations and status of PoseEstimatorBase::run, have the bits of the reference's index-order f32 sums.
//
// Shape: one 256-thread workgroup per workspace.  All threads form the entries of a chunk (weight, Jacobian row, residual) in LDS, then
// lane s < 28 of the first wave owns accumulator slot s and walks the chunk in index order.  the reduction itself ~100 x slower than irls_reduce, an estimate 30 - 50 x (scripts/reference_mode_cost.py) (one wave does the
// adds of a whole pair, one dependent f32 add per entry): a checker, chain path only — the host (estimate.hip) keeps the persistent / team /
// fused / step-in-reduce forms away from a context in this mode.
#include "kernels.h"

#include "gn_common.h"
#include "gn_irls.h"

namespace bpvo_hip {

constexpr int kRefChunk = 1024;      // entries per LDS chunk (32 KiB of 8-float records)

__device__ __forceinline__ float ref_weight(int loss, float r, float sigma_inv)
{
  switch(loss) {
    case BPVO_LOSS_HUBER: return mest_weight<BPVO_LOSS_HUBER>(r, sigma_inv);
    case BPVO_LOSS_TUKEY: return mest_weight<BPVO_LOSS_TUKEY>(r, sigma_inv);
    default: return 1.0f;
  }
}

// TILED8: the C = 8 layout (types.h tile_index, float4 pieces); otherwise point-major records of C floats (C = 1 included)
template <bool TILED8>
__global__ __launch_bounds__(256) void reference_reduce_kernel(const PairJob* __restrict__ jobs, ActiveSet act, int C, int loss)
{
  const PairJob& j = jobs[active_workspace(act, blockIdx.x)];
  const GNState* __restrict__ st = j.st;
  if(!st->active) return;
  __shared__ float rec[kRefChunk][8];      // J[0..5], r, w'
  __shared__ unsigned s_valid;
  const int tid = threadIdx.x;
  if(tid == 0) s_valid = 0u;

  const int N = j.n;
  const long long total = (long long) C * N;
  const float sigma_inv = 1.0f / st->scale;
  const float s_nrm[4] = {j.nrm[0], j.nrm[1], j.nrm[2], j.nrm[3]};
  const bool dspace = j.dspace != 0;
  const float ds_fx = j.K[0], ds_fy = j.K[4], ds_fx_i = 1.0f / j.K[0], ds_fy_i = 1.0f / j.K[4], ds_b_i = 1.0f / j.b;

  // accumulator slot of this lane: s < 21 the upper triangle (a, b) in toEigen's order of gn_logic's unpack; 21 .. 26 G[a]; 27 the squared norm
  int ia = 6, ib = 6;
  if(tid < 21) {
    int idx = 0;
    for(int a = 0; a < 6; ++a)
      for(int b = a; b < 6; ++b) {
        if(idx == tid) { ia = a; ib = b; }
        ++idx;
      }
  } else if(tid < 27) {
    ia = 6; ib = tid - 21;         // (w' * r) * J[a]
  }                                // tid == 27: (w' * r) * r
  float acc = 0.0f;
  unsigned nvalid = 0u;

  for(long long k0 = 0; k0 < total; k0 += kRefChunk) {
    const int m = (int) ((total - k0) < (long long) kRefChunk ? (total - k0) : (long long) kRefChunk);
    for(int e = tid; e < m; e += 256) {
      const long long k = k0 + e;
      const int c = (int) (k / N), i = (int) (k - (long long) c * N);
      float r, Ix, Iy;
      if constexpr(TILED8) {
        r = j.r[tile_index<2>(i, c >> 2) * 4 + (c & 3)];
        Ix = j.grad[tile_index<4>(i, c >> 2) * 4 + (c & 3)];
        Iy = j.grad[tile_index<4>(i, 2 + (c >> 2)) * 4 + (c & 3)];
      } else {
        r = j.r[(size_t) i * C + c];
        Ix = j.grad[((size_t) i * 2 + 0) * C + c];
        Iy = j.grad[((size_t) i * 2 + 1) * C + c];
      }
      const unsigned v = j.valid[i];
      const float4 Pt = j.pts[i];
      float J[6];
      if(dspace) dspace_jac_row(Pt.x, Pt.y, Pt.z, ds_fx, ds_fy, ds_fx_i, ds_fy_i, ds_b_i, Ix, Iy, J);
      else jac_row(jac_point(Pt.x, Pt.y, Pt.z, s_nrm), Ix, Iy, J);
      const float w = ref_weight(loss, r, sigma_inv) * (float) v;       // _W[i] * static_cast<float>(_valid[i])
#pragma unroll
      for(int q = 0; q < 6; ++q) rec[e][q] = J[q];
      rec[e][6] = r;
      rec[e][7] = w;
      if(c == 0) nvalid += v;
    }
    __syncthreads();
    if(tid < 28) {
      // (w' * J[a]) * J[b]   |   (w' * r) * J[a]   |   (w' * r) * r — the products of eight entries first (independent), then their adds in index order
      // (no contraction, no reassociation: -ffp-contract=off, no fast-math)
      int e = 0;
      for(; e + 8 <= m; e += 8) {
        float p[8];
#pragma unroll
        for(int q = 0; q < 8; ++q) p[q] = (rec[e + q][7] * rec[e + q][ia]) * rec[e + q][ib];
#pragma unroll
        for(int q = 0; q < 8; ++q) acc = acc + p[q];
      }
      for(; e < m; ++e) acc = acc + (rec[e][7] * rec[e][ia]) * rec[e][ib];
    }
    __syncthreads();
  }
  if(nvalid) atomicAdd(&s_valid, nvalid);
  __syncthreads();
  float* __restrict__ out = j.partials;
  if(tid < 28) out[tid] = acc;
  else if(tid == 28) out[28] = (float) s_valid;
  else if(tid == 29) out[29] = 0.0f;
}

void launch_reference_reduce(hipStream_t s, const GNLaunch& g)
{
  if(g.max_points <= 0) return;
  if(g.C == 8) hipLaunchKernelGGL(reference_reduce_kernel<true>, dim3(g.npairs), dim3(256), 0, s, g.jobs, g.active, g.C, g.loss);
  else hipLaunchKernelGGL(reference_reduce_kernel<false>, dim3(g.npairs), dim3(256), 0, s, g.jobs, g.active, g.C, g.loss);
}

}  // namespace bpvo_hip
