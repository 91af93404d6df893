// Stereo front-end on the device: the reference's default StereoAlgorithm, OpenCV 2.4 block matching with
// the parameters of utils/stereo_algorithm.cc:63-82 (cvFindStereoCorrespondenceBM: x-Sobel pre-filter, SAD over a square
// window, winner takes all, texture and uniqueness tests, parabola sub-pixel step in 1/16 px) followed by the conversion
// disp16 * (1/16) -> f32 of utils/stereo_algorithm.cc:108.  The disparity map is written straight into device memory (a frame
// slot), so a frame needs two u8 uploads instead of one u8 and one f32.  The algorithm is OpenCV's (third party, absent from
// the reference tree and from this image): restated from the published 2.4 sources, parity UNPINNED — the tests compare with a
// CPU restatement of the same sources.  All arithmetic is integer: evaluation order is irrelevant.
//
// Layout of the matcher: one wavefront = 64 adjacent window columns x ST_ROWS output rows.  For one disparity the lanes hold the
// column sums of |L - R| over the window rows (sliding down the rows: + one row, - one row); the horizontal window sum is
// P(x + w) - P(x - w - 1) of a wavefront prefix sum (DPP adds), so nothing but the two image tiles goes through LDS.  The
// 64 - 2w centre lanes own an output column each and track, over the disparities, the minimum, its neighbours (sub-pixel
// step) and the four smallest sums (the uniqueness test needs the smallest sum outside mind +- 1: it is among the four
// smallest).  The right-image tile is read with the linear addressing of the original: the window of the last output columns
// runs up to SADWindowSize / 2 pixels past the end of its row — into the next row, and past the end of the image the index is
// clamped to the last pixel (the original reads whatever follows the buffer there).
#include <algorithm>

#include "kernels.h"

namespace bpvo_hip {

constexpr int ST_ROWS = 8;          // output rows per wavefront (the per-pixel state lives in registers: 14 values per row)
constexpr int ST_WAVES = 4;         // wavefronts (row groups) per workgroup
constexpr int ST_MAX_WSZ = 21;

// prefilterXSobel: [1 2 1]^T x [-1 0 1] with rows reflected (101), clipped to [-cap, cap] + cap; first / last column and an
// unpaired last row = cap (the original walks the rows in pairs)
__global__ __launch_bounds__(256) void stereo_prefilter_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int cols, int cap,
                                                              size_t frame_stride)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(x >= cols || y >= rows) return;
  src += frame_stride * blockIdx.z;
  dst += frame_stride * blockIdx.z;
  uint8_t out = (uint8_t) cap;
  const bool unpaired = (rows & 1) && y == rows - 1;
  if(!unpaired && x > 0 && x < cols - 1) {
    const int ym = y > 0 ? y - 1 : (rows > 1 ? y + 1 : y), yp = y < rows - 1 ? y + 1 : (rows > 1 ? y - 1 : y);
    const uint8_t* r0 = src + (size_t) ym * cols;
    const uint8_t* r1 = src + (size_t) y * cols;
    const uint8_t* r2 = src + (size_t) yp * cols;
    const int v = ((int) r0[x + 1] - (int) r0[x - 1]) + 2 * ((int) r1[x + 1] - (int) r1[x - 1]) + ((int) r2[x + 1] - (int) r2[x - 1]);
    out = (uint8_t) (min(max(v, -cap), cap) + cap);
  }
  dst[(size_t) y * cols + x] = out;
}

// inclusive scan over the 64 lanes on the VALU: DPP row shifts inside the rows of 16 (lanes without a source add 0), then the two row
// broadcasts — six ds_bpermute round trips as a __shfl_up ladder, once per (disparity, row) of the matcher's inner loop; integer sums
__device__ __forceinline__ int wave_incl_scan(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, true);
  return v;
}

template <int WSZ>
__global__ __launch_bounds__(64 * ST_WAVES) void stereo_bm_kernel(const uint8_t* __restrict__ Lp, const uint8_t* __restrict__ Rp, float* __restrict__ disp,
                                                                   int rows, int cols, int ndisp, int mindisp, int cap, int texture_threshold,
                                                                   int uniqueness_ratio, size_t frame_stride)
{
  constexpr int W2 = WSZ / 2;
  constexpr int OUTW = 64 - 2 * W2;                 // output columns per wavefront
  constexpr int TR = ST_ROWS + 2 * W2;              // tile rows per wavefront
  extern __shared__ uint8_t smem[];                 // [ST_WAVES] x { L tile TR x 64, R tile TR x (64 + ndisp - 1) }
  const int rw = 64 + ndisp - 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint8_t* Lt = smem + (size_t) wave * TR * (64 + rw);
  uint8_t* Rt = Lt + TR * 64;
  Lp += frame_stride * blockIdx.z;
  Rp += frame_stride * blockIdx.z;
  disp += frame_stride * blockIdx.z;

  const int lofs = max(ndisp - 1 + mindisp, 0);     // (rofs = 0: minDisparity >= 1 - ndisp is checked by the host)
  const int width1 = min(cols - ndisp + 1, cols - lofs);   // (the original overruns the row by minDisparity columns: cut at the last one)
  const int xseg0 = blockIdx.x * OUTW;              // first output column of the wavefront in x units (X = lofs + x)
  const int y0 = (blockIdx.y * ST_WAVES + wave) * ST_ROWS;
  if(y0 >= rows || xseg0 >= width1) return;         // whole wavefronts only: no workgroup barrier below
  const size_t npix = (size_t) rows * cols;

  // tiles: window column of lane l is xc = xseg0 - W2 + l; left column clamp(lofs + xc), right columns clamp(xc) + d (linear)
  const int o_r = min(max(xseg0 - W2, 0), cols - 1);
  for(int r = 0; r < TR; ++r) {
    const int yy = min(max(y0 - W2 + r, 0), rows - 1);
    Lt[r * 64 + lane] = Lp[(size_t) yy * cols + min(max(lofs + xseg0 - W2 + lane, 0), cols - 1)];
    for(int t = lane; t < rw; t += 64) Rt[r * rw + t] = Rp[min((size_t) yy * cols + (size_t) (o_r + t), npix - 1)];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int r_off = min(max(xseg0 - W2 + lane, 0), cols - 1) - o_r;     // this lane's column in the R tile at d = 0
  int lcol[TR];
#pragma unroll
  for(int r = 0; r < TR; ++r) lcol[r] = Lt[r * 64 + lane];

  // texture: sum over the window of |L - cap|
  int tsum[ST_ROWS];
  {
    int cs = 0;
#pragma unroll
    for(int r = 0; r < WSZ; ++r) cs += abs(lcol[r] - cap);
#pragma unroll
    for(int i = 0; i < ST_ROWS; ++i) {
      const int P = wave_incl_scan(cs);
      const int hi = __shfl(P, min(lane + W2, 63)), lo = __shfl(P, max(lane - W2 - 1, 0));
      tsum[i] = hi - (lane - W2 - 1 >= 0 ? lo : 0);
      if(i + 1 < ST_ROWS) cs += abs(lcol[i + WSZ] - cap) - abs(lcol[i] - cap);
    }
  }

  // per output pixel state over the disparities
  int minsad[ST_ROWS], mind[ST_ROWS], nb_n[ST_ROWS], nb_p[ST_ROWS], prev[ST_ROWS];
  int t4s[ST_ROWS][4], t4d[ST_ROWS][4];       // four smallest (sum, d), ascending
#pragma unroll
  for(int i = 0; i < ST_ROWS; ++i) {
    minsad[i] = 0x7fffffff; mind[i] = -1; nb_n[i] = 0; nb_p[i] = 0; prev[i] = 0;
#pragma unroll
    for(int k = 0; k < 4; ++k) { t4s[i][k] = 0x7fffffff; t4d[i][k] = -1; }
  }

  for(int d = 0; d < ndisp; ++d) {
    int diff[TR];
#pragma unroll
    for(int r = 0; r < TR; ++r) diff[r] = abs(lcol[r] - (int) Rt[r * rw + r_off + d]);
    int cs = 0;
#pragma unroll
    for(int r = 0; r < WSZ; ++r) cs += diff[r];
#pragma unroll
    for(int i = 0; i < ST_ROWS; ++i) {
      const int P = wave_incl_scan(cs);
      const int hi = __shfl(P, min(lane + W2, 63)), lo = __shfl(P, max(lane - W2 - 1, 0));
      const int s = hi - (lane - W2 - 1 >= 0 ? lo : 0);
      if(i + 1 < ST_ROWS) cs += diff[i + WSZ] - diff[i];
      // winner takes all (first minimum), its neighbours for the sub-pixel step
      if(d == mind[i] + 1 && mind[i] >= 0) nb_p[i] = s;
      if(s < minsad[i]) { minsad[i] = s; mind[i] = d; nb_n[i] = prev[i]; nb_p[i] = s; }   // nb_p is overwritten at d + 1 if it exists
      prev[i] = s;
      // four smallest sums (ties keep the earlier disparity first)
      if(s < t4s[i][3]) {
        int cs_ = s, cd_ = d;
#pragma unroll
        for(int k = 0; k < 4; ++k) {
          if(cs_ < t4s[i][k]) { const int ts = t4s[i][k], td = t4d[i][k]; t4s[i][k] = cs_; t4d[i][k] = cd_; cs_ = ts; cd_ = td; }
        }
      }
    }
  }

  if(lane < W2 || lane >= 64 - W2) return;
  const int x = xseg0 + lane - W2;
  if(x >= width1) return;
  const float filtered = (float) ((mindisp - 1) << 4) * (1.0f / 16.0f);
#pragma unroll
  for(int i = 0; i < ST_ROWS; ++i) {
    const int y = y0 + i;
    if(y >= rows) break;
    float out = filtered;
    bool ok = tsum[i] >= texture_threshold;
    const int md = mind[i];
    if(ok && uniqueness_ratio > 0) {
      const int thresh = minsad[i] + (minsad[i] * uniqueness_ratio / 100);
#pragma unroll
      for(int k = 0; k < 4; ++k)
        if(t4d[i][k] >= 0 && t4s[i][k] <= thresh && (t4d[i][k] < md - 1 || t4d[i][k] > md + 1)) ok = false;
    }
    if(ok) {
      // sad[-1] = sad[1] (= p when mind = 0), sad[ndisp] = sad[ndisp - 2] (= n when mind = ndisp - 1)
      const int n = md > 0 ? nb_n[i] : nb_p[i];
      const int p = md < ndisp - 1 ? nb_p[i] : nb_n[i];
      const int dd = p + n - 2 * minsad[i] + abs(p - n);
      const int d16 = ((ndisp - md - 1 + mindisp) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4;
      out = (float) (short) d16 * (1.0f / 16.0f);
    }
    disp[(size_t) y * cols + lofs + x] = out;
  }
}

__global__ __launch_bounds__(256) void stereo_fill_kernel(float* __restrict__ disp, size_t n, float v)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if(i < n) disp[i] = v;
}

void launch_stereo_prefilter(hipStream_t s, const uint8_t* src, uint8_t* dst, int rows, int cols, int cap, int nframes)
{
  hipLaunchKernelGGL(stereo_prefilter_kernel, dim3((cols + 63) / 64, (rows + 3) / 4, nframes), dim3(256), 0, s, src, dst, rows, cols, cap,
                     (size_t) rows * cols);
}

template <int WSZ>
static bool launch_bm_w(hipStream_t s, const StereoLaunch& g)
{
  constexpr int W2 = WSZ / 2, OUTW = 64 - 2 * W2, TR = ST_ROWS + 2 * W2;
  const int width1 = std::min(g.cols - g.ndisp + 1, g.cols - (g.ndisp - 1 + g.mindisp));
  const size_t lds = (size_t) ST_WAVES * TR * (64 + 64 + g.ndisp - 1);
  if(lds > 160 * 1024) return false;
  static bool attr_set = false;      // (one attribute per instantiation; harmless to repeat)
  if(!attr_set && lds > 64 * 1024) {
    (void) hipFuncSetAttribute((const void*) stereo_bm_kernel<WSZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid((width1 + OUTW - 1) / OUTW, (g.rows + ST_ROWS * ST_WAVES - 1) / (ST_ROWS * ST_WAVES), g.nframes);
  hipLaunchKernelGGL(stereo_bm_kernel<WSZ>, grid, dim3(64 * ST_WAVES), lds, s, g.left_pre, g.right_pre, g.disp, g.rows, g.cols, g.ndisp, g.mindisp, g.cap,
                     g.texture_threshold, g.uniqueness_ratio, (size_t) g.rows * g.cols);
  return true;
}

// the whole disparity map: invalid value everywhere, then the matcher over the columns that have all disparities
bool launch_stereo_bm(hipStream_t s, const StereoLaunch& g)
{
  const size_t n = (size_t) g.rows * g.cols * g.nframes;
  const float filtered = (float) ((g.mindisp - 1) << 4) * (1.0f / 16.0f);
  hipLaunchKernelGGL(stereo_fill_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, g.disp, n, filtered);
  const int lofs = g.ndisp - 1 + g.mindisp, width1 = std::min(g.cols - g.ndisp + 1, g.cols - lofs);
  if(lofs >= g.cols || width1 < 1) return true;     // findStereoCorrespondenceBM: nothing to match, all FILTERED
  switch(g.wsz) {
    case 5: return launch_bm_w<5>(s, g);
    case 7: return launch_bm_w<7>(s, g);
    case 9: return launch_bm_w<9>(s, g);
    case 11: return launch_bm_w<11>(s, g);
    case 13: return launch_bm_w<13>(s, g);
    case 15: return launch_bm_w<15>(s, g);
    case 17: return launch_bm_w<17>(s, g);
    case 19: return launch_bm_w<19>(s, g);
    case 21: return launch_bm_w<21>(s, g);
    default: return false;
  }
}

}  // namespace bpvo_hip
