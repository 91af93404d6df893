// Per-frame kernels (gfx950) of the descriptors the reference builds as chains of whole-image operations — Laplacian, IntensityAndGradient,
// DescriptorFields (first / second order), CentralDifference, LATCH: every operation one launch over the level, work
// planes in FrameJob::scratch, results written into the pixel-interleaved descriptor records.  Not the benchmarked path (bit-planes and
// intensity live in kernels_frame.hip); built for parity with the reference's operator set, bit-exact against the oracle.
#include <algorithm>
#include <cstdlib>

#include "frame_common.h"

namespace bpvo_hip {

// ---- LaplacianDescriptor::compute (reference: bpvo/gradient_descriptor.cc:64-67): cv::Laplacian(u8 -> f32), kernel size 1
// ({0,1,0,1,-4,1,0,1,0}), 3 ({2,0,2,0,-8,0,2,0,2}), 5 or 7 (Sobel second derivatives), BORDER_REFLECT_101; integer-valued,
// hence exact in f32.
__global__ __launch_bounds__(256) void laplacian_kernel(const FrameJob* jobs, int ksize)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(x >= W || y >= R) return;
  const uint8_t* __restrict__ I = j.img;
  if(ksize > 3) {
    // ksize 5 / 7: d2/dx2 + d2/dy2 with the separable Sobel kernels of cv::getSobelKernels (second derivative x binomial
    // smoothing): [1 0 -2 0 1] x [1 4 6 4 1] and [1 2 -1 -4 -1 2 1] x [1 6 15 20 15 6 1].  Integer sums far below 2^24
    // (and below the 16-bit work type OpenCV uses for ksize 5): exact in any order.
    const int d5[5] = {1, 0, -2, 0, 1}, s5[5] = {1, 4, 6, 4, 1};
    const int d7[7] = {1, 2, -1, -4, -1, 2, 1}, s7[7] = {1, 6, 15, 20, 15, 6, 1};
    const int r = ksize >> 1;
    int acc = 0;
    for(int dy = 0; dy < ksize; ++dy) {
      const uint8_t* row = I + (size_t) reflect101_wide(y - r + dy, R) * W;
      const int dyk = ksize == 5 ? d5[dy] : d7[dy], syk = ksize == 5 ? s5[dy] : s7[dy];
      for(int dx = 0; dx < ksize; ++dx) {
        const int dxk = ksize == 5 ? d5[dx] : d7[dx], sxk = ksize == 5 ? s5[dx] : s7[dx];
        acc += (dxk * syk + sxk * dyk) * (int) row[reflect101_wide(x - r + dx, W)];
      }
    }
    j.desc[(size_t) y * W + x] = (float) acc;
    return;
  }
  const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W), ym = reflect101(y - 1, R), yp = reflect101(y + 1, R);
  const float k_edge = ksize == 3 ? 0.0f : 1.0f, k_diag = ksize == 3 ? 2.0f : 0.0f, k_ctr = ksize == 3 ? -8.0f : -4.0f;
  const uint8_t *rm = I + (size_t) ym * W, *r0 = I + (size_t) y * W, *rp = I + (size_t) yp * W;
  float v = k_diag * (float) rm[xm] + k_edge * (float) rm[x] + k_diag * (float) rm[xp];
  v += k_edge * (float) r0[xm] + k_ctr * (float) r0[x] + k_edge * (float) r0[xp];
  v += k_diag * (float) rp[xm] + k_edge * (float) rp[x] + k_diag * (float) rp[xp];
  j.desc[(size_t) y * W + x] = v;
}

// ---- GradientDescriptor::compute (reference: bpvo/gradient_descriptor.cc:42-63) with sigma <= 0: channels (I, Ix, Iy),
// Ix / Iy = xgradient / ygradient (bpvo/imgproc.h:214-265): 0.5 * central difference, one-sided 0.5 * (I1 - I0) at the borders.
__global__ __launch_bounds__(256) void gradient_descriptor_kernel(const FrameJob* jobs)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(x >= W || y >= R) return;
  const uint8_t* __restrict__ I = j.img;
  auto at = [&](int yy, int xx) { return (float) I[(size_t) yy * W + xx]; };
  const int xa = x == 0 ? 0 : (x == W - 1 ? W - 2 : x - 1), xb = x == 0 ? 1 : (x == W - 1 ? W - 1 : x + 1);
  const int ya = y == 0 ? 0 : (y == R - 1 ? R - 2 : y - 1), yb = y == 0 ? 1 : (y == R - 1 ? R - 1 : y + 1);
  float* d = j.desc + ((size_t) y * W + x) * 3;
  d[0] = at(y, x);
  d[1] = 0.5f * (at(y, xb) - at(y, xa));
  d[2] = 0.5f * (at(yb, x) - at(ya, x));
}

// ---- DescriptorFields / DescriptorFields2ndOrder (reference: bpvo/gradient_descriptor.cc:100-160): chains of plane operations
// -- convertTo, imsmooth (5 x 5 f32 Gaussian, bpvo/imgproc.cc:166-171), xgradient / ygradient (bpvo/imgproc.h:214-265),
// splitPosNeg (gradient_descriptor.cc:80-98) -- each one launch of this kernel.  A plane code >= 0 is a work plane of
// FrameJob::scratch, a code < 0 is descriptor channel -1-code of the interleaved [npix][C] records.
enum { DF_CONVERT = 0, DF_GAUSS_ROW, DF_GAUSS_COL, DF_GRAD_X, DF_GRAD_Y, DF_SPLIT, DF_U8_ROW, DF_U8_COL, DF_SHIFT_DIFF, DF_TO_CH0,
       DF_GAUSS_ROW_N, DF_GAUSS_COL_N, DF_U8_ROW_N, DF_U8_COL_N };
struct PlaneRef { float* p; int stride; };
__device__ __forceinline__ PlaneRef df_plane(const FrameJob& j, int code, int C)
{
  if(code >= 0) return PlaneRef{j.scratch + (size_t) code * j.rows * j.cols, 1};
  return PlaneRef{j.desc + (-1 - code), C};
}
// CentralDifferenceDescriptor (bpvo/central_difference_descriptor.cc:36-131) adds: the u8 5 x 5 fixed-point Gaussian of the
// image (DF_U8_ROW keeps the int row sums as bit patterns in a work plane, DF_U8_COL rounds them to the u8 value, held as
// float), the image minus its clamped shift by (i0, i1) (DF_SHIFT_DIFF), and the copy of channel 0 into the compact
// FrameJob::ch0 plane that the C = 8 kernels expect (DF_TO_CH0).
// Kernels wider than 5 taps (imsmooth with sigma >= 2.5, the automatic size of GradientDescriptor's pre-smoothing) take
// the generic forms of OpenCV 2.4's filter engine: DF_GAUSS_ROW_N s = k[0]*S[x-r]; s += k[j]*S[x-r+j] (RowFilter, left to
// right), DF_GAUSS_COL_N s = k[r]*S0; s += k[r+j]*(S[+j] + S[-j]) (SymmColumnFilter); DF_U8_*_N the 8-bit fixed-point pair.
__global__ __launch_bounds__(256) void df_plane_kernel(const FrameJob* jobs, int op, int src_code, int dst_code, int dst2_code, int C,
                                                       float k0, float k1, float k2, int i0, int i1, int i2, GaussTaps gt)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(x >= W || y >= R) return;
  const PlaneRef S = df_plane(j, src_code, C), D = df_plane(j, dst_code, C);
  auto at = [&](int yy, int xx) { return S.p[((size_t) yy * W + xx) * S.stride]; };
  const size_t q = (size_t) y * W + x;
  float v;
  switch(op) {
    case DF_CONVERT: v = (float) j.img[q]; break;
    case DF_GAUSS_ROW:   // s = S[0]*k0 + (S[-1]+S[1])*k1 + (S[-2]+S[2])*k2
      v = at(y, x) * k0 + (at(y, reflect101(x - 1, W)) + at(y, reflect101(x + 1, W))) * k1 +
          (at(y, reflect101(x - 2, W)) + at(y, reflect101(x + 2, W))) * k2;
      break;
    case DF_GAUSS_COL:   // s = k0*S0; s += k1*(S+1 + S-1); s += k2*(S+2 + S-2)
      v = k0 * at(y, x);
      v += k1 * (at(reflect101(y + 1, R), x) + at(reflect101(y - 1, R), x));
      v += k2 * (at(reflect101(y + 2, R), x) + at(reflect101(y - 2, R), x));
      break;
    case DF_GRAD_X:
      v = x == 0 ? 0.5f * (at(y, 1) - at(y, 0)) : (x == W - 1 ? 0.5f * (at(y, x) - at(y, x - 1)) : 0.5f * (at(y, x + 1) - at(y, x - 1)));
      break;
    case DF_GRAD_Y:
      v = y == 0 ? 0.5f * (at(1, x) - at(0, x)) : (y == R - 1 ? 0.5f * (at(y, x) - at(y - 1, x)) : 0.5f * (at(y + 1, x) - at(y - 1, x)));
      break;
    case DF_U8_ROW: {
      const uint8_t* row = j.img + (size_t) y * W;
      const int t = row[x] * i0 + (row[reflect101(x - 1, W)] + row[reflect101(x + 1, W)]) * i1 +
                    (row[reflect101(x - 2, W)] + row[reflect101(x + 2, W)]) * i2;
      v = __int_as_float(t);
      break;
    }
    case DF_U8_COL: {
      auto it = [&](int yy) { return __float_as_int(at(yy, x)); };
      const int t = (it(y) * i0 + (it(reflect101(y - 1, R)) + it(reflect101(y + 1, R))) * i1 +
                     (it(reflect101(y - 2, R)) + it(reflect101(y + 2, R))) * i2 + (1 << 15)) >> 16;
      v = (float) min(255, max(0, t));
      break;
    }
    case DF_SHIFT_DIFF:
      v = at(y, x) - at(min(max(y + i1, 0), R - 1), min(max(x + i0, 0), W - 1));
      break;
    case DF_GAUSS_ROW_N: {
      const int r = gt.n >> 1;
      v = gt.k[0] * at(y, reflect101_wide(x - r, W));
      for(int t = 1; t < gt.n; ++t) v += gt.k[t] * at(y, reflect101_wide(x - r + t, W));
      break;
    }
    case DF_GAUSS_COL_N: {
      const int r = gt.n >> 1;
      v = gt.k[r] * at(y, x);
      for(int t = 1; t <= r; ++t) v += gt.k[r + t] * (at(reflect101_wide(y + t, R), x) + at(reflect101_wide(y - t, R), x));
      break;
    }
    case DF_U8_ROW_N: {
      const uint8_t* row = j.img + (size_t) y * W;
      const int r = gt.n >> 1;
      int t = 0;
      for(int q = 0; q < gt.n; ++q) t += gt.ki[q] * row[reflect101_wide(x - r + q, W)];
      v = __int_as_float(t);
      break;
    }
    case DF_U8_COL_N: {
      const int r = gt.n >> 1;
      int t = 0;
      for(int q = 0; q < gt.n; ++q) t += gt.ki[q] * __float_as_int(at(reflect101_wide(y - r + q, R), x));
      t = (t + (1 << 15)) >> 16;
      v = (float) min(255, max(0, t));
      break;
    }
    case DF_TO_CH0:
      if(j.ch0) j.ch0[q] = at(y, x);
      return;
    default: {           // DF_SPLIT
      const float s = at(y, x);
      const PlaneRef N = df_plane(j, dst2_code, C);
      N.p[q * N.stride] = s < 0 ? s : 0.0f;
      v = s >= 0 ? s : 0.0f;
    }
  }
  D.p[q * D.stride] = v;
}

// ---- wide descriptors built channel by channel (CentralDifference: 48 ... 360 channels, LATCH: 8 ... 512): a column pass that wrote ONE channel of the
// pixel-interleaved records stored 4 bytes every 4 C — a partial sector per pixel and channel, 0.88 ms per 640x480 frame for 48 channels against 0.44 ms
// for 128 frames of bit-planes.  So the row passes of up to eight consecutive channels go to eight work planes (coalesced), and ONE column pass forms the
// eight channels of a pixel and stores them as one contiguous 32-byte piece of its record.  Same operations per channel, same order: same bits.
// (a) the row pass of the smoothing of (image - its clamped shift by (ox, oy)) without the difference plane in between: DF_SHIFT_DIFF + DF_GAUSS_ROW(_N)
// (blockIdx.z = frame * 8 + channel of the group: the row passes of a group's eight offsets are ONE launch)
struct CdOffsets { signed char ox[8], oy[8]; };
__global__ __launch_bounds__(256) void cd_diff_row_kernel(const FrameJob* jobs, int src_code, int dst_code0, int nch, CdOffsets o, float k0, float k1, float k2, GaussTaps gt)
{
  const int kch = blockIdx.z & 7;
  if(kch >= nch) return;
  const FrameJob& j = jobs[blockIdx.z >> 3];
  const int W = j.cols, R = j.rows;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(x >= W || y >= R) return;
  const int ox = o.ox[kch], oy = o.oy[kch];
  const float* __restrict__ S = j.scratch + (size_t) src_code * R * W;
  const float* __restrict__ r0 = S + (size_t) y * W;
  const float* __restrict__ r1 = S + (size_t) min(max(y + oy, 0), R - 1) * W;
  auto d = [&](int xx) { return r0[xx] - r1[min(max(xx + ox, 0), W - 1)]; };
  float v;
  if(gt.n == 5) {
    v = d(x) * k0 + (d(reflect101(x - 1, W)) + d(reflect101(x + 1, W))) * k1 + (d(reflect101(x - 2, W)) + d(reflect101(x + 2, W))) * k2;
  } else {
    const int r = gt.n >> 1;
    v = gt.k[0] * d(reflect101_wide(x - r, W));
    for(int t = 1; t < gt.n; ++t) v += gt.k[t] * d(reflect101_wide(x - r + t, W));
  }
  (j.scratch + (size_t) (dst_code0 + kch) * R * W)[(size_t) y * W + x] = v;
}
// (b) the column pass (DF_GAUSS_COL / DF_GAUSS_COL_N) of work planes src_code .. src_code + nch - 1 into channels c0 .. c0 + nch - 1
__global__ __launch_bounds__(256) void df_col8_kernel(const FrameJob* jobs, int src_code, int nch, int c0, int C, float k0, float k1, float k2, GaussTaps gt)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(x >= W || y >= R) return;
  const size_t plane = (size_t) R * W;
  const float* __restrict__ S = j.scratch + (size_t) src_code * plane;
  float v[8];
  if(gt.n == 5) {
    const size_t q0 = (size_t) y * W + x, qp1 = (size_t) reflect101(y + 1, R) * W + x, qm1 = (size_t) reflect101(y - 1, R) * W + x,
                 qp2 = (size_t) reflect101(y + 2, R) * W + x, qm2 = (size_t) reflect101(y - 2, R) * W + x;
#pragma unroll
    for(int k = 0; k < 8; ++k) {
      if(k >= nch) { v[k] = 0.0f; continue; }
      const float* __restrict__ P = S + (size_t) k * plane;
      float a = k0 * P[q0];
      a += k1 * (P[qp1] + P[qm1]);
      a += k2 * (P[qp2] + P[qm2]);
      v[k] = a;
    }
  } else {
    const int r = gt.n >> 1;
#pragma unroll
    for(int k = 0; k < 8; ++k) v[k] = k < nch ? gt.k[r] * S[(size_t) k * plane + (size_t) y * W + x] : 0.0f;
    for(int t = 1; t <= r; ++t) {
      const size_t qa = (size_t) reflect101_wide(y + t, R) * W + x, qb = (size_t) reflect101_wide(y - t, R) * W + x;
#pragma unroll
      for(int k = 0; k < 8; ++k)
        if(k < nch) v[k] += gt.k[r + t] * (S[(size_t) k * plane + qa] + S[(size_t) k * plane + qb]);
    }
  }
  float* __restrict__ dst = j.desc + ((size_t) y * W + x) * C + c0;
  if(nch == 8 && (C & 3) == 0 && (c0 & 3) == 0) {
    reinterpret_cast<float4*>(dst)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(dst)[1] = make_float4(v[4], v[5], v[6], v[7]);
  } else {
#pragma unroll
    for(int k = 0; k < 8; ++k) if(k < nch) dst[k] = v[k];
  }
}
// (c) CentralDifference without smoothing afterwards: the differences of up to eight offsets straight into their channels
__global__ __launch_bounds__(256) void cd_diff8_kernel(const FrameJob* jobs, int src_code, int nch, int c0, int C, CdOffsets o)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(x >= W || y >= R) return;
  const float* __restrict__ S = j.scratch + (size_t) src_code * R * W;
  const float ctr = S[(size_t) y * W + x];
  float* __restrict__ dst = j.desc + ((size_t) y * W + x) * C + c0;
#pragma unroll
  for(int k = 0; k < 8; ++k)
    if(k < nch) dst[k] = ctr - S[(size_t) min(max(y + (int) o.oy[k], 0), R - 1) * W + min(max(x + (int) o.ox[k], 0), W - 1)];
}

// ---- LatchDescriptor, evaluated densely (reference: bpvo/latch_descriptor.cc:124-167 compute, :170-262 CalcuateSums, :264-493 pixelTests,
// :1041-1086 LatchDescriptor::compute).  Key points are the pixels border <= y < R - border - 1, border <= x < W - border - 1 (row-major),
// border = 24 + K, K = latchHalfSsdSize.  Work planes of FrameJob::scratch: the smoothed u8 image, the [key points][bytes] descriptor
// bytes, one extracted channel, one smoothing temporary.  The descriptor bytes come LAST: they take latchNumBytes bytes per key point — one float
// plane up to 4 bytes, 16 planes for 64 (the context sizes FrameJob::scratch for them: 3 + ceil(bytes / 4) planes).
enum { LATCH_P_GRAY = 0, LATCH_P_CH = 1, LATCH_P_TMP = 2, LATCH_P_BYTES = 3 };
__device__ __forceinline__ uint8_t* latch_u8_plane(const FrameJob& j, int plane)
{
  return reinterpret_cast<uint8_t*>(j.scratch.get() + (size_t) plane * j.rows * j.cols);
}

// cv::GaussianBlur(image, gray, Size(3,3), 2, 2) on u8 (:147): OpenCV 2.4's 8-bit fixed-point separable filter — taps cvRound(k * 256),
// row pass u8 -> int, column pass (sum + 2^15) >> 16 saturated, BORDER_REFLECT_101 — as census_blur_kernel; integer arithmetic, any order.
__global__ __launch_bounds__(256) void latch_blur3_kernel(const FrameJob* jobs, int kc, int ks)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(x >= W || y >= R) return;
  const uint8_t* __restrict__ I = j.img;
  const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
  auto row = [&](int yy) { const uint8_t* r = I + (size_t) yy * W; return (int) r[x] * kc + ((int) r[xm] + (int) r[xp]) * ks; };
  const int v = (row(y) * kc + (row(reflect101(y - 1, R)) + row(reflect101(y + 1, R))) * ks + (1 << 15)) >> 16;
  latch_u8_plane(j, LATCH_P_GRAY)[(size_t) y * W + x] = (uint8_t) min(255, max(0, v));
}

// The descriptor bytes of a 64 x 8 tile of key points.  The smoothed image under the tile — the tile plus the 24 + K pixels any mini-patch
// can reach on every side: the 48 x 48 patch of every key point of the tile — is staged in LDS once; a thread then evaluates its key
// points' bits from LDS: bit j of byte ix = [suma < sumc] for triplet 8 ix + (7 - j), suma / sumc the integer sums of squared differences
// of the (2K+1)^2 mini-patches at (a, b) and (c, b) (:239-256: (int)(double(diff)^2) of u8 differences, i.e. exact integers).
// `offsets`: the 6 * 8 * BYTES triplet coordinates as CalcuateSums uses them (rotated by the constant key-point angle and clamped when
// latchRotationInvariance is on; prepared by the host, bpvo_hip.hip).
constexpr int LT_W = 64, LT_H = 8;
template <int BYTES>
__global__ __launch_bounds__(256) void latch_bits_kernel(const FrameJob* jobs, const signed char* __restrict__ offsets, int K)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
  __shared__ signed char s_off[48 * BYTES];
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int H = 24 + K;
  const int nx = W - 2 * H - 1, ny = R - 2 * H - 1;       // key points per row / rows of key points
  const int kx0 = blockIdx.x * LT_W, ky0 = blockIdx.y * LT_H;
  if(nx <= 0 || ny <= 0 || kx0 >= nx || ky0 >= ny) return;
  const int tid = threadIdx.x;
  const int TW = LT_W + 2 * H, TH = LT_H + 2 * H, pitch = (TW + 3) & ~3;
  const uint8_t* __restrict__ gray = latch_u8_plane(j, LATCH_P_GRAY);
  // key point (kx, ky) sits at pixel (H + kx, H + ky): the tile's first staged pixel is (kx0, ky0)
  for(int i = tid; i < TH * TW; i += 256) {
    const int ly = i / TW, lx = i - ly * TW;
    s_tile[ly * pitch + lx] = gray[(size_t) min(ky0 + ly, R - 1) * W + min(kx0 + lx, W - 1)];
  }
  for(int i = tid; i < 48 * BYTES; i += 256) s_off[i] = offsets[i];
  __syncthreads();
  uint8_t* __restrict__ out = latch_u8_plane(j, LATCH_P_BYTES);
  const int lx = tid & 63;
#pragma unroll
  for(int rr = 0; rr < LT_H / 4; ++rr) {
    const int ly = (tid >> 6) + 4 * rr;
    const int kx = kx0 + lx, ky = ky0 + ly;
    if(kx >= nx || ky >= ny) continue;
    const uint8_t* centre = s_tile + (ly + H) * pitch + lx + H;
    for(int ix = 0; ix < BYTES; ++ix) {
      unsigned byte = 0;
      for(int b = 7; b >= 0; --b) {
        const signed char* t = s_off + (ix * 8 + (7 - b)) * 6;
        const uint8_t* pa = centre + (int) t[1] * pitch + (int) t[0];
        const uint8_t* pb = centre + (int) t[3] * pitch + (int) t[2];
        const uint8_t* pc = centre + (int) t[5] * pitch + (int) t[4];
        int suma = 0, sumc = 0;
        for(int iy = -K; iy <= K; ++iy)
          for(int ixx = -K; ixx <= K; ++ixx) {
            const int o = iy * pitch + ixx;
            const int vb = (int) pb[o];
            const int da = (int) pa[o] - vb, dc = (int) pc[o] - vb;
            suma += da * da;
            sumc += dc * dc;
          }
        byte |= (unsigned) (suma < sumc) << b;
      }
      out[((size_t) ky * nx + kx) * BYTES + ix] = (uint8_t) byte;
    }
  }
}

// Channel 8 c + bit of the dense descriptor before its smoothing (:1056-1080): 255 * bit - 128 inside the key-point region, 0 outside.
// The byte a key point reads is byte c + k of the row-major [key points][bytes] buffer, k its row-major index — the reference advances the
// pointer of a COLUMN view by one byte per key point (`_buffer.col(c).ptr()`, `*src_ptr++`): for latchNumBytes = 1 that is byte c of key
// point k, for more bytes it is byte (c + k) % bytes of key point (c + k) / bytes; restated as the reference computes it.
__global__ __launch_bounds__(256) void latch_extract_kernel(const FrameJob* jobs, int c_byte, int bit, int K)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(x >= W || y >= R) return;
  const int H = 24 + K;
  const int nx = W - 2 * H - 1;
  float v = 0.0f;
  if(x >= H && x < W - H - 1 && y >= H && y < R - H - 1) {
    const unsigned val = latch_u8_plane(j, LATCH_P_BYTES)[(size_t) c_byte + (size_t) (y - H) * nx + (x - H)];
    v = 255.0f * (float) ((val >> bit) & 1u) + -128.0f;
  }
  (j.scratch.get() + (size_t) LATCH_P_CH * R * W)[(size_t) y * W + x] = v;
}

// ---- host-callable launchers ------------------------------------------------------------------------------------
void launch_laplacian(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int ksize)
{
  hipLaunchKernelGGL(laplacian_kernel, grid2d(W, R, nframes), dim3(256), 0, s, jobs, ksize);
}
// smoothing of a work plane / channel with either form of the kernel (5 taps: the small-kernel ops, wider: the generic ones)
template <class Op>
static void df_smooth(Op&& op, int src, int tmp, int dst, const GaussTaps& g)
{
  if(g.n == 5) {
    const float k[3] = {g.k[2], g.k[3], g.k[4]};
    op(DF_GAUSS_ROW, src, tmp, k, nullptr);
    op(DF_GAUSS_COL, tmp, dst, k, nullptr);
  } else {
    op(DF_GAUSS_ROW_N, src, tmp, nullptr, &g);
    op(DF_GAUSS_COL_N, tmp, dst, nullptr, &g);
  }
}
void launch_gradient_descriptor(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, const GaussTaps& pre)
{
  if(pre.n == 0) {
    hipLaunchKernelGGL(gradient_descriptor_kernel, grid2d(W, R, nframes), dim3(256), 0, s, jobs);
    return;
  }
  // GradientDescriptor::compute with sigma > 0 (bpvo/gradient_descriptor.cc:42-63): channel 0 keeps the unsmoothed
  // intensities, the gradients are taken of cv::GaussianBlur(I, Size(), sigma)
  const dim3 grid = grid2d(W, R, nframes);
  auto op = [&](int o, int src, int dst, const float* k, const GaussTaps* g) {
    hipLaunchKernelGGL(df_plane_kernel, grid, dim3(256), 0, s, jobs, o, src, dst, 0, 3, k ? k[0] : 0.0f, k ? k[1] : 0.0f, k ? k[2] : 0.0f, 0, 0, 0,
                       g ? *g : GaussTaps());
  };
  enum { P_S = 0, P_TMP = 1 };
  op(DF_CONVERT, 0, -1, nullptr, nullptr);
  df_smooth(op, -1, P_TMP, P_S, pre);
  op(DF_GRAD_X, P_S, -2, nullptr, nullptr);
  op(DF_GRAD_Y, P_S, -3, nullptr, nullptr);
}
// one level of DescriptorFields (second_order = 0: 5 channels) or DescriptorFields2ndOrder (10 channels); g1 / g2 are the
// imsmooth kernels of sigma1 / sigma2 (n = 0: sigma <= 0, no smoothing)
void launch_descriptor_fields(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int second_order, const GaussTaps& g1,
                              const GaussTaps& g2)
{
  const int C = second_order ? 10 : 5;
  const dim3 grid = grid2d(W, R, nframes);
  auto op5 = [&](int o, int src, int dst, const float* k, const GaussTaps* g, int dst2 = 0) {
    hipLaunchKernelGGL(df_plane_kernel, grid, dim3(256), 0, s, jobs, o, src, dst, dst2, C, k ? k[0] : 0.0f, k ? k[1] : 0.0f, k ? k[2] : 0.0f, 0, 0, 0,
                       g ? *g : GaussTaps());
  };
  auto op = [&](int o, int src, int dst, int dst2 = 0) { op5(o, src, dst, nullptr, nullptr, dst2); };
  auto ch = [](int c) { return -1 - c; };
  enum { P_I0 = 0, P_I = 1, P_B1 = 2, P_B2 = 3, P_POS = 4, P_NEG = 5, P_TMP = 6, P_ROW = 7 };
  auto smooth = [&](int src, int dst, const GaussTaps& g) { df_smooth(op5, src, P_TMP, dst, g); };
  // the smoothed channels come in ascending order, two per split: their row passes go to the work planes P_ROW .. P_ROW + 7 and one column pass
  // writes up to eight of them into the records at once (df_col8_kernel: contiguous stores instead of one strided float per pixel and channel)
  const float k5[3] = {g2.k[2], g2.k[3], g2.k[4]};
  int batch_c0 = -1, batch_n = 0;
  auto flush = [&]() {
    if(batch_n) hipLaunchKernelGGL(df_col8_kernel, grid, dim3(256), 0, s, jobs, (int) P_ROW, batch_n, batch_c0, C, k5[0], k5[1], k5[2], g2);
    batch_n = 0;
  };
  auto smooth_to_channel = [&](int src, int c) {
    if(batch_n && (c != batch_c0 + batch_n || batch_n == 8)) flush();
    if(batch_n == 0) batch_c0 = c;
    if(g2.n == 5) op5(DF_GAUSS_ROW, src, P_ROW + batch_n, k5, nullptr);
    else op5(DF_GAUSS_ROW_N, src, P_ROW + batch_n, nullptr, &g2);
    ++batch_n;
  };
  auto split = [&](int src, int cpos, int cneg) {
    if(g2.n > 0) {
      op(DF_SPLIT, src, P_POS, P_NEG);
      smooth_to_channel(P_POS, cpos);
      smooth_to_channel(P_NEG, cneg);
    } else {
      op(DF_SPLIT, src, ch(cpos), ch(cneg));
    }
  };
  const int I0 = second_order ? P_I0 : ch(0);     // first order keeps the unsmoothed intensities as channel 0
  op(DF_CONVERT, 0, I0);
  int I = I0;
  if(g1.n > 0) { smooth(I0, P_I, g1); I = P_I; }
  if(!second_order) {
    op(DF_GRAD_X, I, P_B1); split(P_B1, 1, 2);
    op(DF_GRAD_Y, I, P_B1); split(P_B1, 3, 4);
  } else {
    op(DF_GRAD_X, I, P_B1);    split(P_B1, 0, 1);   // Ix
    op(DF_GRAD_X, P_B1, P_B2); split(P_B2, 2, 3);   // Ixx
    split(P_B2, 4, 5);                              // "Ixy": the reference splits Ixx again (gradient_descriptor.cc:149-150)
    op(DF_GRAD_Y, I, P_B1);    split(P_B1, 6, 7);   // Iy
    op(DF_GRAD_Y, P_B1, P_B2); split(P_B2, 8, 9);   // Iyy
  }
  flush();
}
// one level of CentralDifferenceDescriptor: C = (2r+1)^2 - 1 channels; before = the u8 blur of sigma_before (fixed-point taps),
// after = the f32 kernel of sigma_after (n = 0: not applied)
void launch_central_difference(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int radius, const GaussTaps& before,
                               const GaussTaps& after)
{
  const int C = (2 * radius + 1) * (2 * radius + 1) - 1;
  const dim3 grid = grid2d(W, R, nframes);
  auto opi = [&](int o, int src, int dst, const float* k, const GaussTaps* g, int i0 = 0, int i1 = 0, int i2 = 0) {
    hipLaunchKernelGGL(df_plane_kernel, grid, dim3(256), 0, s, jobs, o, src, dst, 0, C, k ? k[0] : 0.0f, k ? k[1] : 0.0f, k ? k[2] : 0.0f, i0, i1, i2,
                       g ? *g : GaussTaps());
  };
  auto op5 = [&](int o, int src, int dst, const float* k, const GaussTaps* g) { opi(o, src, dst, k, g); };
  enum { P_IMG = 0, P_DIFF = 1, P_TMP = 2 };
  if(before.n == 5) {
    opi(DF_U8_ROW, 0, P_TMP, nullptr, nullptr, before.ki[2], before.ki[3], before.ki[4]);
    opi(DF_U8_COL, P_TMP, P_IMG, nullptr, nullptr, before.ki[2], before.ki[3], before.ki[4]);
  } else if(before.n > 5) {
    opi(DF_U8_ROW_N, 0, P_TMP, nullptr, &before);
    opi(DF_U8_COL_N, P_TMP, P_IMG, nullptr, &before);
  } else {
    opi(DF_CONVERT, 0, P_IMG, nullptr, nullptr);
  }
  // channels in groups of eight: their row passes into the work planes P_ROW .. P_ROW + 7, one column pass for the group (see cd_diff_row_kernel)
  enum { P_ROW = 3 };
  const float k5[3] = {after.k[2], after.k[3], after.k[4]};
  int c = 0, in_group = 0;
  CdOffsets offs = {};
  auto flush = [&]() {
    if(in_group == 0) return;
    if(after.n > 0) {
      for(int f0 = 0; f0 < nframes; f0 += 8000) {      // (grid.z <= 65535)
        const dim3 grid8(grid.x, grid.y, (unsigned) std::min(8000, nframes - f0) * 8u);
        hipLaunchKernelGGL(cd_diff_row_kernel, grid8, dim3(256), 0, s, jobs + f0, (int) P_IMG, (int) P_ROW, in_group, offs, k5[0], k5[1], k5[2], after);
      }
      hipLaunchKernelGGL(df_col8_kernel, grid, dim3(256), 0, s, jobs, (int) P_ROW, in_group, c - in_group, C, k5[0], k5[1], k5[2], after);
    } else {
      hipLaunchKernelGGL(cd_diff8_kernel, grid, dim3(256), 0, s, jobs, (int) P_IMG, in_group, c - in_group, C, offs);
    }
    in_group = 0;
  };
  for(int oy = -radius; oy <= radius; ++oy)
    for(int ox = -radius; ox <= radius; ++ox) {
      if(ox == 0 && oy == 0) continue;
      offs.ox[in_group] = (signed char) ox; offs.oy[in_group] = (signed char) oy;
      ++c;
      if(++in_group == 8) flush();
    }
  flush();
  (void) op5;
  if(C == 8) opi(DF_TO_CH0, -1, 0, nullptr, nullptr);
}

// one level of LatchDescriptor: 8 * bytes channels.  `offsets`: device table of 48 * bytes coordinates; (kc, ks): fixed-point taps of the
// 3 x 3 blur of sigma 2; after: the imsmooth kernel of sigma 1.75 every channel is smoothed with (:1082)
void launch_latch(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int bytes, int K, const signed char* offsets, int kc, int ks,
                  const GaussTaps& after)
{
  const int C = 8 * bytes, H = 24 + K;
  const dim3 grid = grid2d(W, R, nframes);
  hipLaunchKernelGGL(latch_blur3_kernel, grid, dim3(256), 0, s, jobs, kc, ks);
  const int nx = W - 2 * H - 1, ny = R - 2 * H - 1;
  if(nx > 0 && ny > 0) {
    const dim3 tiles((nx + LT_W - 1) / LT_W, (ny + LT_H - 1) / LT_H, nframes);
    const size_t lds = (size_t) (LT_H + 2 * H) * ((LT_W + 2 * H + 3) & ~3);
    switch(bytes) {
      case 1: hipLaunchKernelGGL(latch_bits_kernel<1>, tiles, dim3(256), lds, s, jobs, offsets, K); break;
      case 2: hipLaunchKernelGGL(latch_bits_kernel<2>, tiles, dim3(256), lds, s, jobs, offsets, K); break;
      case 4: hipLaunchKernelGGL(latch_bits_kernel<4>, tiles, dim3(256), lds, s, jobs, offsets, K); break;
      case 8: hipLaunchKernelGGL(latch_bits_kernel<8>, tiles, dim3(256), lds, s, jobs, offsets, K); break;
      case 16: hipLaunchKernelGGL(latch_bits_kernel<16>, tiles, dim3(256), lds, s, jobs, offsets, K); break;
      case 32: hipLaunchKernelGGL(latch_bits_kernel<32>, tiles, dim3(256), lds, s, jobs, offsets, K); break;
      default: hipLaunchKernelGGL(latch_bits_kernel<64>, tiles, dim3(256), lds, s, jobs, offsets, K); break;
    }
  }
  auto op5 = [&](int o, int src, int dst, const float* k, const GaussTaps* g) {
    hipLaunchKernelGGL(df_plane_kernel, grid, dim3(256), 0, s, jobs, o, src, dst, 0, C, k ? k[0] : 0.0f, k ? k[1] : 0.0f, k ? k[2] : 0.0f, 0, 0, 0,
                       g ? *g : GaussTaps());
  };
  // the eight channels of a descriptor byte: their row passes into eight work planes behind the descriptor bytes, one column pass for the byte
  // (df_col8_kernel: one contiguous 32-byte store per pixel instead of eight strided ones)
  const int p_row = LATCH_P_BYTES + (bytes + 3) / 4;
  const float k5[3] = {after.k[2], after.k[3], after.k[4]};
  for(int c = 0; c < bytes; ++c) {
    for(int bit = 0; bit < 8; ++bit) {
      hipLaunchKernelGGL(latch_extract_kernel, grid, dim3(256), 0, s, jobs, c, bit, K);
      if(after.n == 5) op5(DF_GAUSS_ROW, LATCH_P_CH, p_row + bit, k5, nullptr);
      else op5(DF_GAUSS_ROW_N, LATCH_P_CH, p_row + bit, nullptr, &after);
    }
    hipLaunchKernelGGL(df_col8_kernel, grid, dim3(256), 0, s, jobs, p_row, 8, 8 * c, C, k5[0], k5[1], k5[2], after);
  }
  if(C == 8) op5(DF_TO_CH0, -1, 0, nullptr, nullptr);
}

}  // namespace bpvo_hip
