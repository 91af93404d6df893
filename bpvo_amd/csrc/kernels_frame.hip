// Per-frame kernels (gfx950): image pyramid, census / bit-planes descriptor, saliency, pixel selection,
// 3-D points, Hartley normalisation, template (pixels + Jacobians).  All are batched over frames with blockIdx.z.
//
// HBM layout: the descriptor of a level is PIXEL-INTERLEAVED, desc[(y*W + x)*C + c], so that the C
// channels of one pixel are one 32-byte record (C = 8): a bilinear gather touches 2 x 64 contiguous bytes instead
// of 32 scattered dwords, and every kernel below writes/reads full records with 16-byte accesses.
#include <algorithm>
#include <cstdlib>

#include "frame_common.h"

namespace bpvo_hip {

// ---- input ingest: one launch copies the u8 image and f32 disparity of every frame of a batch from a packed device
// buffer ([frame][rows*cols]) into the frame slots (VisualOdometryFrame::setData's image.copyTo / disparity.copyTo,
// reference: bpvo/vo_frame.cc:50-51) instead of 2 memcpy calls per frame.
// skip_odd_disp: pair batches (A0, B0, A1, B1, ...) — the current frame B of a pair never becomes a template, its disparity is not copied
// Levels in one launch (few frames: every per-level launch is a 5 - 20 us latency floor, four levels of them in a row): the grid is sized for
// the FINEST level of the group, z = level x nframes + frame, and the workgroups that fall outside a coarser level's own grid leave at
// once.  `jobs` is the finest level's row of the table [level][job_pitch]; a per-level launch has z < nframes: its own row.
__device__ __forceinline__ const FrameJob& level_job(const FrameJob* jobs, unsigned z, int nframes, int job_pitch)
{
  const unsigned lvl = z / (unsigned) nframes;
  return jobs[(size_t) lvl * job_pitch + (z - lvl * (unsigned) nframes)];
}

__global__ __launch_bounds__(256) void ingest_kernel(const FrameJob* jobs, const uint8_t* images, const float* disps, size_t npix, int skip_odd_disp)
{
  const FrameJob& j = jobs[blockIdx.z];
  const uint8_t* __restrict__ si = images + (size_t) blockIdx.z * npix;
  // skip_odd_disp 2: the disparities are packed for the even frames only ([frame / 2][npix]: the upload pipeline of host batches)
  const float* __restrict__ sd = disps + (size_t) (skip_odd_disp == 2 ? blockIdx.z / 2 : blockIdx.z) * npix;
  uint8_t* __restrict__ di = const_cast<uint8_t*>(j.img.get());
  float* __restrict__ dd = const_cast<float*>(j.disp.get());
  const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x, stride = (size_t) gridDim.x * 256;
  const bool w4 = (npix % 4 == 0) && (((uintptr_t) si | (uintptr_t) di) % 4 == 0);
  if(w4) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(si);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(di);
    for(size_t k = t; k < npix / 4; k += stride) d4[k] = s4[k];
  } else {
    for(size_t k = t; k < npix; k += stride) di[k] = si[k];
  }
  if(skip_odd_disp && (blockIdx.z & 1)) return;
  const bool f4 = (npix % 4 == 0) && (((uintptr_t) sd | (uintptr_t) dd) % 16 == 0);
  if(f4) {
    const float4* s4 = reinterpret_cast<const float4*>(sd);
    float4* d4 = reinterpret_cast<float4*>(dd);
    for(size_t k = t; k < npix / 4; k += stride) d4[k] = load_stream(s4 + k);   // the packed input is read once
  } else {
    for(size_t k = t; k < npix; k += stride) dd[k] = sd[k];
  }
}

// ---- K0: cv::pyrDown u8 (reference call site: bpvo/image_pyramid.cc:49).  [1 4 6 4 1]^2 / 256 with (s + 128) >> 8,
// BORDER_REFLECT_101, dst = ((W+1)/2, (R+1)/2).  Integer arithmetic: evaluation order is irrelevant.
// Through LDS: a workgroup stages the (2*64+3) x (2*16+3) source pixels of a 64 x 16 output
// tile with row-contiguous loads (whole dwords where the tile allows: see below), forms the horizontal [1 4 6 4 1] sums once per (source row, output column) and the vertical ones from
// those.  Integer arithmetic, same values.  1024 pairs of 1241x376, the three levels of a step: 2.5 ms (one thread per output from HBM) -> 1.8 (LDS) ->
// 1.46 (dword staging) -> 0.81 ms (round 4: four sums per thread from two 8-byte LDS reads, v_alignbyte + v_dot4; dword stores; four tiles per
// workgroup with the next tile's loads in flight).
constexpr int PDT_W = 64, PDT_H = 16;
constexpr int PDS_W = 2 * PDT_W + 3, PDS_H = 2 * PDT_H + 3;
constexpr int PDS_DW = (PDS_W + 3 + 3) / 4;        // dwords that cover PDS_W bytes starting at any byte of a dword (34)
constexpr int PDS_PITCH = 4 * (PDS_DW + 2);        // bytes per staged row (144)
constexpr int PD_STACK = 4;                        // vertically adjacent tiles a workgroup walks: the next tile's loads fly under the current one's passes
constexpr int PD_NPRE = (PDS_H * PDS_DW + 255) / 256;
__global__ __launch_bounds__(256) void pyrdown_u8_lds_kernel(const FrameJob* src_jobs, const FrameJob* dst_jobs)
{
  __shared__ __attribute__((aligned(16))) uint8_t s_src[PDS_H][PDS_PITCH];
  __shared__ __attribute__((aligned(8))) uint16_t s_h[PDS_H][PDT_W];      // <= 16 * 255
  __shared__ int s_off[PDS_H];
  const FrameJob& sj = src_jobs[blockIdx.z];
  const FrameJob& dj = dst_jobs[blockIdx.z];
  const int sw = sj.cols, sh = sj.rows, dw = dj.cols, dh = dj.rows;
  const int dx0 = blockIdx.x * PDT_W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint8_t* __restrict__ s = sj.img;
  uint8_t* __restrict__ const d = const_cast<uint8_t*>(dj.img.get());
  const int c_first = 2 * dx0 - 2;
  // Tiles whose source columns (and the up to three bytes either side that whole-dword loads add) lie inside one image row: the window
  // is fetched as aligned dwords, 34 per row, and a row's bytes start `off` (0..3, its address modulo 4) into its LDS row.  The
  // others (the first and last tile of a row of tiles, narrow images) go byte by byte through REFLECT_101.
  const bool fast = c_first >= 3 && c_first + PDS_W + 3 <= sw;
  auto row_ptr = [&](int dy0, int r) { return s + (size_t) reflect101(min(2 * dy0 - 2 + r, sh + 1), sh) * sw + c_first; };
  uint32_t pre[PD_NPRE];
  auto prefetch = [&](int dy0) {      // (every lane loads: clamped index, no branch around the loads)
#pragma unroll
    for(int k = 0; k < PD_NPRE; ++k) {
      const int i = min((int) threadIdx.x + k * 256, PDS_H * PDS_DW - 1);
      const int r = i / PDS_DW, l = i - r * PDS_DW;
      const uintptr_t a = reinterpret_cast<uintptr_t>(row_ptr(dy0, r));
      // (dwords past the row's window are not staged; the address stays inside the frame's image: a row's window ends 3 bytes before the row does)
      const unsigned ll = min((unsigned) l, ((unsigned) (a & 3u) + PDS_W - 1) / 4u);
      pre[k] = *reinterpret_cast<const uint32_t*>(a - (a & 3u) + 4u * ll);
    }
  };
  const int dy_first = blockIdx.y * (PDT_H * PD_STACK);
  if(fast) prefetch(dy_first);
  for(int t = 0; t < PD_STACK; ++t) {
    const int dy0 = dy_first + t * PDT_H;
    if(dy0 >= dh) break;
    if(fast) {
#pragma unroll
      for(int k = 0; k < PD_NPRE; ++k) {
        const int i = (int) threadIdx.x + k * 256;
        if(i < PDS_H * PDS_DW) { const int r = i / PDS_DW, l = i - r * PDS_DW; reinterpret_cast<uint32_t*>(&s_src[r][0])[l] = pre[k]; }
      }
      if(t + 1 < PD_STACK && dy0 + PDT_H < dh) prefetch(dy0 + PDT_H);
    } else {
      for(int r = wave; r < PDS_H; r += 4) {
        const uint8_t* row = s + (size_t) reflect101(min(2 * dy0 - 2 + r, sh + 1), sh) * sw;
#pragma unroll
        for(int c0 = 0; c0 < PDS_W; c0 += 64) {
          const int c = c0 + lane;
          if(c < PDS_W) s_src[r][c] = row[reflect101(min(c_first + c, sw + 1), sw)];
        }
      }
    }
    // byte offset of every staged row inside its LDS row (its address modulo 4 in the dword path, 0 otherwise)
    if(threadIdx.x < PDS_H) s_off[threadIdx.x] = fast ? (int) (reinterpret_cast<uintptr_t>(row_ptr(dy0, threadIdx.x)) & 3u) : 0;
    __syncthreads();
    // horizontal [1 4 6 4 1]: a thread forms FOUR neighbouring sums of one staged row from the 11 bytes they cover — two 8-byte LDS reads,
    // the window shifted to the row's byte offset with v_alignbyte, every sum a v_dot4 over (1 4 6 4) plus its fifth byte (each staged
    // byte was read five times, one byte per LDS instruction, before)
    for(int idx = threadIdx.x; idx < PDS_H * (PDT_W / 4); idx += 256) {
      const int r = idx >> 4, xq = idx & 15;
      const uint2* q = reinterpret_cast<const uint2*>(&s_src[r][8 * xq]);
      const uint2 lo = q[0], hi = q[1];
      const unsigned off = (unsigned) s_off[r];
      const unsigned d0 = __builtin_amdgcn_alignbyte(lo.y, lo.x, off), d1 = __builtin_amdgcn_alignbyte(hi.x, lo.y, off),
                     d2 = __builtin_amdgcn_alignbyte(hi.y, hi.x, off);
      const unsigned kW = 0x04060401u;      // bytes (1, 4, 6, 4), lowest first
      const unsigned h0 = __builtin_amdgcn_udot4(d0, kW, d1 & 0xffu, false);
      const unsigned h1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), kW, (d1 >> 16) & 0xffu, false);
      const unsigned h2 = __builtin_amdgcn_udot4(d1, kW, d2 & 0xffu, false);
      const unsigned h3 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), kW, (d2 >> 16) & 0xffu, false);
      *reinterpret_cast<uint2*>(&s_h[r][4 * xq]) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
    }
    __syncthreads();
    // vertical pass: a thread finishes four neighbouring pixels of one output row and stores them as one (possibly unaligned) dword
    const int ty = threadIdx.x >> 4, xq = threadIdx.x & 15;
    const int y = dy0 + ty, x = dx0 + 4 * xq;
    if(y < dh && x < dw) {
      unsigned v[4];
      uint2 tt[5];
#pragma unroll
      for(int k = 0; k < 5; ++k) tt[k] = *reinterpret_cast<const uint2*>(&s_h[2 * ty + k][4 * xq]);
#pragma unroll
      for(int e = 0; e < 4; ++e) {
        unsigned a[5];
#pragma unroll
        for(int k = 0; k < 5; ++k) { const unsigned w = (e < 2) ? tt[k].x : tt[k].y; a[k] = (e & 1) ? (w >> 16) : (w & 0xffffu); }
        v[e] = (a[0] + 4u * a[1] + 6u * a[2] + 4u * a[3] + a[4] + 128u) >> 8;
      }
      uint8_t* o = d + (size_t) y * dw + x;
      if(x + 3 < dw) {
        typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
        *reinterpret_cast<u32_unaligned*>(o) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
      } else {
        for(int e = 0; e < 4 && x + e < dw; ++e) o[e] = (uint8_t) v[e];
      }
    }
    __syncthreads();      // (the next tile's staging overwrites s_src / s_h)
  }
}

// ---- K0 for a few frames: up to three pyrDown steps in ONE launch.  One level is a 15 - 20 us launch whatever the frame (a stage of one
// or two frames fills a fraction of the chip), and the levels depend on each other: three of them are 53 us of a single pair's 220 us
// frame stage.  Here a workgroup owns an 8 x 8 tile of the COARSEST level of the group and computes, in LDS, everything below it that
// the tile depends on — 19 x 19 pixels of the level before, 41 x 41 of the one before that, from 85 x 85 source pixels — and stores
// the part of every level that lies under its own tile (the halos are recomputed by the neighbours: integer arithmetic, same values).
// Same definition as pyrdown_u8_lds_kernel: [1 4 6 4 1]^2, (s + 128) >> 8, BORDER_REFLECT_101 on every level's own coordinates.
constexpr int PA_T = 8;
constexpr int pa_edge(int steps) { return steps == 0 ? PA_T : 2 * pa_edge(steps - 1) + 3; }      // 8, 19, 41, 85
template <int NL>
__global__ __launch_bounds__(256) void pyramid_levels_kernel(const FrameJob* jobs /* the SOURCE level's row of the table */, int job_pitch)
{
  constexpr int E0 = pa_edge(NL), E1 = pa_edge(NL - 1);
  __shared__ uint8_t s_a[E0 * E0];           // the level being read
  __shared__ uint8_t s_b[E1 * E1];           // the level being written
  __shared__ uint16_t s_h[E0 * E1];          // horizontal sums of the level being read (<= 16 * 255)
  int W[NL + 1], H[NL + 1], xlo[NL + 1], xhi[NL + 1], ylo[NL + 1], yhi[NL + 1];      // (hi inclusive)
#pragma unroll
  for(int k = 0; k <= NL; ++k) { const FrameJob& j = jobs[(size_t) k * job_pitch + blockIdx.z]; W[k] = j.cols; H[k] = j.rows; }
  xlo[NL] = blockIdx.x * PA_T; xhi[NL] = min(xlo[NL] + PA_T, W[NL]) - 1;
  ylo[NL] = blockIdx.y * PA_T; yhi[NL] = min(ylo[NL] + PA_T, H[NL]) - 1;
#pragma unroll
  for(int k = NL - 1; k >= 0; --k) {
    xlo[k] = max(0, 2 * xlo[k + 1] - 2); xhi[k] = min(W[k] - 1, 2 * xhi[k + 1] + 2);
    ylo[k] = max(0, 2 * ylo[k + 1] - 2); yhi[k] = min(H[k] - 1, 2 * yhi[k + 1] + 2);
  }
  const int tid = threadIdx.x;
  {   // the source region
    const uint8_t* __restrict__ src = jobs[blockIdx.z].img;
    const int nx = xhi[0] - xlo[0] + 1, ny = yhi[0] - ylo[0] + 1;
    for(int i = tid; i < nx * ny; i += 256) {
      const int r = i / nx, c = i - r * nx;
      s_a[r * E0 + c] = src[(size_t) (ylo[0] + r) * W[0] + xlo[0] + c];
    }
  }
  __syncthreads();
  uint8_t* cur = s_a;
  uint8_t* nxt = s_b;
  int pc = E0, pn = E1;      // row pitches of the two buffers
#pragma unroll
  for(int k = 0; k < NL; ++k) {
    const int snx = xhi[k] - xlo[k] + 1, sny = yhi[k] - ylo[k] + 1;          // region of level k in `cur`
    const int dnx = xhi[k + 1] - xlo[k + 1] + 1, dny = yhi[k + 1] - ylo[k + 1] + 1;
    (void) snx;
    for(int i = tid; i < sny * dnx; i += 256) {      // horizontal sums: (source row, destination column)
      const int r = i / dnx, c = i - r * dnx;
      const int x2 = 2 * (xlo[k + 1] + c);
      const uint8_t* row = cur + r * pc - xlo[k];
      const unsigned a0 = row[reflect101(x2 - 2, W[k])], a1 = row[reflect101(x2 - 1, W[k])], a2 = row[x2], a3 = row[reflect101(x2 + 1, W[k])],
                     a4 = row[reflect101(x2 + 2, W[k])];
      s_h[r * E1 + c] = (uint16_t) (a0 + 4u * a1 + 6u * a2 + 4u * a3 + a4);
    }
    __syncthreads();
    // the level's owned part: under this workgroup's tile of the coarsest level
    const int sh = NL - (k + 1);
    const int ox0 = (blockIdx.x * PA_T) << sh, ox1 = min(((blockIdx.x + 1) * PA_T) << sh, W[k + 1]);
    const int oy0 = (blockIdx.y * PA_T) << sh, oy1 = min(((blockIdx.y + 1) * PA_T) << sh, H[k + 1]);
    uint8_t* __restrict__ dst = const_cast<uint8_t*>(jobs[(size_t) (k + 1) * job_pitch + blockIdx.z].img.get());
    for(int i = tid; i < dny * dnx; i += 256) {
      const int r = i / dnx, c = i - r * dnx;
      const int y = ylo[k + 1] + r, x = xlo[k + 1] + c, y2 = 2 * y;
      const uint16_t* col = s_h + c - ylo[k] * E1;
      const unsigned b0 = col[reflect101(y2 - 2, H[k]) * E1], b1 = col[reflect101(y2 - 1, H[k]) * E1], b2 = col[y2 * E1],
                     b3 = col[reflect101(y2 + 1, H[k]) * E1], b4 = col[reflect101(y2 + 2, H[k]) * E1];
      const uint8_t v = (uint8_t) ((b0 + 4u * b1 + 6u * b2 + 4u * b3 + b4 + 128u) >> 8);
      nxt[r * pn + c] = v;
      if(x >= ox0 && x < ox1 && y >= oy0 && y < oy1) dst[(size_t) y * W[k + 1] + x] = v;
    }
    __syncthreads();
    { uint8_t* t = cur; cur = nxt; nxt = t; }
    { const int t = pc; pc = pn; pn = t; }
  }
}
// `steps` (1 .. 3) levels below src_row in one launch; dW x dR = the coarsest of them.  The caller (frames.hip) takes this form only when every
// level is at least 8 pixels either way: the reflections then stay inside a workgroup's regions.
void launch_pyramid_levels(hipStream_t s, const FrameJob* src_row, int job_pitch, int steps, int dW, int dR, int nframes)
{
  const dim3 grid((dW + PA_T - 1) / PA_T, (dR + PA_T - 1) / PA_T, nframes);
  if(steps == 3) hipLaunchKernelGGL(pyramid_levels_kernel<3>, grid, dim3(256), 0, s, src_row, job_pitch);
  else if(steps == 2) hipLaunchKernelGGL(pyramid_levels_kernel<2>, grid, dim3(256), 0, s, src_row, job_pitch);
  else hipLaunchKernelGGL(pyramid_levels_kernel<1>, grid, dim3(256), 0, s, src_row, job_pitch);
}

// ---- IntensityDescriptor::compute: u8 -> f32 (reference: bpvo/intensity_descriptor.cc:31-43)
__global__ __launch_bounds__(256) void intensity_kernel(const FrameJob* jobs, int nframes, int job_pitch)
{
  const FrameJob& j = level_job(jobs, blockIdx.z, nframes, job_pitch);
  const int n = j.rows * j.cols;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if(i >= n) return;
  if(i + 3 < n && ((uintptr_t) (j.img + i) & 3) == 0) {
    const uchar4 v = *reinterpret_cast<const uchar4*>(j.img + i);
    *reinterpret_cast<float4*>(j.desc + i) = make_float4((float) v.x, (float) v.y, (float) v.z, (float) v.w);
  } else {
    for(int k = i; k < n && k < i + 4; ++k) j.desc[k] = (float) j.img[k];
  }
}

// ---- K1a: census transform (reference: bpvo/census.cc:42-91, bpvo/v128.h:102-105).
// bit k = [neighbour_k >= centre], neighbours (-1,-1),(-1,0),(-1,+1),(0,-1),(0,+1),(+1,-1),(+1,0),(+1,+1); 1-px border = 0.
// Workgroups of 64 x 16 pixels: every thread walks 4 rows of one column with a sliding 3 x 3 window (6 rows x 3 bytes loaded
// for 4 outputs).  Tiny one-pixel-per-thread workgroups were bound by the workgroup dispatch rate (~1.15 WG/ns), not by HBM.
constexpr int ROWS_PER_THREAD = 4;
constexpr int CENSUS_ROWS = 8;   // rows per thread (4: 166, 8: 149, 16: 178 us per launch at 256 pairs)
__global__ __launch_bounds__(256) void census_kernel(const FrameJob* jobs)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * CENSUS_ROWS;
  const int W = j.cols, R = j.rows;
  if(x >= W || y0 >= R) return;
  const bool xin = x > 0 && x < W - 1;
  const int xm = max(x - 1, 0), xp = min(x + 1, W - 1);
  // all (CENSUS_ROWS + 2) x 3 bytes are loaded before the first use: a rolling window issued one row per iteration
  // and every iteration waited for its own loads (PMC: 81 % of the wave cycles waiting, 1.6 waves per SIMD in flight)
  uint8_t w[CENSUS_ROWS + 2][3];
#pragma unroll
  for(int k = 0; k < CENSUS_ROWS + 2; ++k) {
    const uint8_t* p = j.img + (size_t) min(max(y0 - 1 + k, 0), R - 1) * W;
    w[k][0] = p[xm]; w[k][1] = p[x]; w[k][2] = p[xp];
  }
#pragma unroll
  for(int k = 0; k < CENSUS_ROWS; ++k) {
    const int y = y0 + k;
    if(y < R) {
      uint8_t out = 0;
      if(xin && y > 0 && y < R - 1) {
        const uint8_t ctr = w[k + 1][1];
        out = (uint8_t) (((w[k][0] >= ctr) << 0) | ((w[k][1] >= ctr) << 1) | ((w[k][2] >= ctr) << 2) | ((w[k + 1][0] >= ctr) << 3) |
                         ((w[k + 1][2] >= ctr) << 4) | ((w[k + 2][0] >= ctr) << 5) | ((w[k + 2][1] >= ctr) << 6) | ((w[k + 2][2] >= ctr) << 7));
      }
      j.cen[(size_t) y * W + x] = out;
    }
  }
}

// ---- K1a': census of the 3x3-smoothed image, sigmaPriorToCensusTransform > 0 (reference: bpvo/census.cc:63-66 ->
// cv::GaussianBlur(u8, Size(3,3), s, s) [ext: OpenCV 2.4 8-bit separable filter in fixed point: taps round(k * 256),
// row pass u8 -> int, column pass (sum + 2^15) >> 16, saturated; BORDER_REFLECT_101]).  The smoothed image is never
// written out: a 64 x 4 output tile stages its source pixels + 2-px halo (reflected coordinates) in LDS, the row and
// column passes run in LDS, and the census reads the 66 x 6 smoothed tile.
constexpr int CB_TW = 64, CB_TH = 4;
__global__ __launch_bounds__(256) void census_blur_kernel(const FrameJob* jobs, int kc, int ks, int nframes, int job_pitch)
{
  __shared__ uint8_t s_src[(CB_TH + 4) * (CB_TW + 4)];
  __shared__ int s_tmp[(CB_TH + 4) * (CB_TW + 2)];
  __shared__ uint8_t s_b[(CB_TH + 2) * (CB_TW + 2)];
  const FrameJob& j = level_job(jobs, blockIdx.z, nframes, job_pitch);
  const int W = j.cols, R = j.rows;
  const int x0 = blockIdx.x * CB_TW, y0 = blockIdx.y * CB_TH;
  if(x0 >= W || y0 >= R) return;      // (a coarser level inside a launch sized for the finest)
  const int tid = threadIdx.x;
  // source tile: rows y0-2 .. y0+CB_TH+1, columns x0-2 .. x0+CB_TW+1
  for(int i = tid; i < (CB_TH + 4) * (CB_TW + 4); i += 256) {
    const int ly = i / (CB_TW + 4), lx = i - ly * (CB_TW + 4);
    const int gy = reflect101(min(y0 + ly - 2, R + 1), R), gx = reflect101(min(x0 + lx - 2, W + 1), W);
    s_src[i] = j.img[(size_t) gy * W + gx];
  }
  __syncthreads();
  // row pass at columns x0-1 .. x0+CB_TW for every staged row
  for(int i = tid; i < (CB_TH + 4) * (CB_TW + 2); i += 256) {
    const int ly = i / (CB_TW + 2), lx = i - ly * (CB_TW + 2);
    const uint8_t* p = s_src + ly * (CB_TW + 4) + lx + 1;    // centre = column x0-1+lx
    s_tmp[i] = (int) p[0] * kc + ((int) p[-1] + (int) p[1]) * ks;
  }
  __syncthreads();
  // column pass at rows y0-1 .. y0+CB_TH
  for(int i = tid; i < (CB_TH + 2) * (CB_TW + 2); i += 256) {
    const int ly = i / (CB_TW + 2), lx = i - ly * (CB_TW + 2);
    const int* t = s_tmp + (ly + 1) * (CB_TW + 2) + lx;
    const int v = (t[0] * kc + (t[-(CB_TW + 2)] + t[CB_TW + 2]) * ks + (1 << 15)) >> 16;
    s_b[i] = (uint8_t) min(255, max(0, v));
  }
  __syncthreads();
  const int lx = tid & 63, ly = tid >> 6;
  const int x = x0 + lx, y = y0 + ly;
  if(x >= W || y >= R) return;
  uint8_t out = 0;
  if(x > 0 && x < W - 1 && y > 0 && y < R - 1) {
    constexpr int P = CB_TW + 2;
    const uint8_t* p = s_b + (ly + 1) * P + lx + 1;
    const uint8_t c = p[0];
    out = (uint8_t) (((p[-P - 1] >= c) << 0) | ((p[-P] >= c) << 1) | ((p[-P + 1] >= c) << 2) | ((p[-1] >= c) << 3) |
                     ((p[1] >= c) << 4) | ((p[P - 1] >= c) << 5) | ((p[P] >= c) << 6) | ((p[P + 1] >= c) << 7));
  }
  j.cen[(size_t) y * W + x] = out;
}

// ---- K1b: 8 bit-planes + cv::GaussianBlur 5x5 (reference: bpvo/bitplanes_descriptor.cc:37-57).
// One 256-thread workgroup produces a 64 x 8 tile of 32-byte pixel records.  The census bytes of the tile + 2-px halo
// (REFLECT_101 on the coordinates) are staged in LDS, the horizontal pass is written to LDS for all 8 planes
// (12 rows x 64 cols x 8 f32 = 24 KB; with the spread census tile 32 KB: 5 workgroups per CU), the vertical pass streams full records to HBM with two 16-byte
// stores per pixel.
// Arithmetic per plane, f32, no fusing, exactly OpenCV's symmetric 5-tap filters:
//   row: t = S0*k0 + (S-1 + S+1)*k1 + (S-2 + S+2)*k2        column: s = k0*T0; s += k1*(T+1 + T-1); s += k2*(T+2 + T-2)
#ifndef BP_TW_VALUE       // wider tiles (longer store runs) measured slower: 128 x 4 +15 %, 256 x 2 +75 % (row-pass redundancy)
#define BP_TW_VALUE 64
#define BP_TH_VALUE 8
#endif
constexpr int BP_TW = BP_TW_VALUE, BP_TH = BP_TH_VALUE, BP_HALO = 2;
// vertically adjacent tiles a workgroup walks (kernel argument `stack`): 8 (64 rows) for batches — the prologue of a workgroup, a memory
// round trip nothing covers, is paid half as often: 11.35 -> 11.03 ms for the four levels of 2048 frames — and 4 for single frames, whose
// 640x480 level 0 would otherwise be 80 workgroups on 256 CUs
// row-pass results in LDS: one float4 plane per half (channels 0-3, 4-7) of (BP_TH + 4) rows x BP_TW columns; the second plane starts
// 128 bytes out of phase with the first, so that the two lanes of a pixel (same position, different plane) hit different banks
constexpr int BP_PLANE = (BP_TH + 2 * BP_HALO) * BP_TW + 8;
// The workgroup walks `stack` vertically adjacent tiles; the census bytes of the next tile are fetched into registers
// before the current tile's passes run, so the global-load latency is hidden behind the LDS/VALU work instead of being
// exposed once per (short-lived) workgroup.
// FROM_IMAGE (sigma_ct <= 0, the default): the census transform is fused in — the workgroup stages the u8 IMAGE tile with a
// 3-px halo and computes the census bytes of its tile + 2-px halo in LDS (reference: bpvo/census.cc:42-91, see census_kernel:
// same bits, 0 on the 1-px image border), so the census image never exists in HBM: one launch and ~2.3 B/px of traffic less
// per level.  The REFLECT_101 of the blur acts on CENSUS coordinates, so a staged census position outside the image maps to
// an interior one first and reads the image around that.
// The row pass through a table: a bit-plane sample is 0 or 1, so t = S0*k0 + (S-1 + S+1)*k1 + (S-2 + S+2)*k2 takes one of 2 x 3 x 3
// values, selected by (S0, S-1 + S+1, S-2 + S+2).  The census byte of a staged position is spread to one BYTE per plane (two words:
// planes 0-3, 4-7), the five positions of a row window are combined bytewise into the index a + 3 b + 9 S0 of all eight planes at
// once, and the eighteen values — evaluated by the kernel itself with the very expression above — are looked up in LDS (different
// entries lie in different banks, equal ones broadcast: no conflicts).  ~30 instead of ~136 VALU operations per position, same bits.
__device__ __forceinline__ uint2 spread_planes(unsigned c)
{
  unsigned lo = c & 0xFu, hi = (c >> 4) & 0xFu;
  lo = (lo | (lo << 14)) & 0x00030003u; lo = (lo | (lo << 7)) & 0x01010101u;
  hi = (hi | (hi << 14)) & 0x00030003u; hi = (hi | (hi << 7)) & 0x01010101u;
  return make_uint2(lo, hi);
}

template <bool FROM_IMAGE>
__global__ __launch_bounds__(256) void bitplanes_blur_kernel(const FrameJob* jobs, float k0, float k1, float k2, int stack, int nframes, int job_pitch)
{
  constexpr int CR = BP_TH + 2 * BP_HALO, CC = BP_TW + 2 * BP_HALO;   // staged census rows / columns
  constexpr int CW = CC + 4;                                             // padded LDS row pitch of the census tile
  constexpr int IR = CR + 2, IC = CC + 2, IW = IC + 2;                   // image window rows / columns / LDS pitch (FROM_IMAGE)
  constexpr int NSRC = FROM_IMAGE ? IR * IC : CR * CC;                   // bytes staged per tile
  constexpr int NPRE = (NSRC + 255) / 256;                               // ... per thread
  __shared__ uint2 s_cen[CR * CW];          // census bytes spread to one byte per plane (spread_planes)
  __shared__ uint8_t s_img[FROM_IMAGE ? IR * IW : 4];
  __shared__ float s_row[(BP_PLANE + CR * BP_TW) * 4];
  __shared__ float s_lut[18];
  const FrameJob& j = level_job(jobs, blockIdx.z, nframes, job_pitch);
  const int W = j.cols, R = j.rows;
  const int x0 = blockIdx.x * BP_TW;
  if(x0 >= W || (int) blockIdx.y * BP_TH * stack >= R) return;      // (a coarser level inside a launch sized for the finest)
  const int tid = threadIdx.x;
  const uint8_t* __restrict__ cen = FROM_IMAGE ? j.img : j.cen;
  // the job's pointers in registers: read through `j` inside the loop they are re-loaded after every store (the stores might alias the
  // job table), and on gfx9 the wait for such a load also waits for every store before it (vmcnt counts both, in order)
  float* __restrict__ const desc = j.desc;
  float* __restrict__ const ch0 = j.ch0;
  // A LAZY level (FrameJob::lazy; template frames of a pair batch at the NMS levels): the records of such a level are read at the five
  // stencil positions of one pixel in ~18 only — 72 % of them never — so the level keeps its census bytes and channel 0 (what the saliency
  // map needs) and template_build forms the stencils' records from the census bytes: the kernel then runs the census stage, one plane of
  // eight in the two passes, and stores 5 bytes per pixel instead of 36.
  // The tile columns at the right edge stay dense: the saliency of the columns the reference's SIMD body treats apart — x < 4 (formed from
  // columns n - 4 + x) and x >= n = W & ~3 (Q7) — reads whole records of columns n - 5 .. W - 1 (saliency_generic).  And when W is a
  // multiple of 4, column x = 3 is formed at xs = W - 1, whose right neighbour is the reference's read past the end of the row: the
  // record of column 0 of the NEXT row — so the first tile column stays dense too.
  const bool lazy_level = FROM_IMAGE && j.lazy != 0;
  const bool lazy = lazy_level && x0 + BP_TW + 8 <= (W & ~3) && !(x0 == 0 && (W & 3) == 0);
  uint8_t* __restrict__ const cen_out = j.cen;
  if(tid < 18) {      // entry a + 3 b + 9 S0 (visible after the first barrier below)
    const float S0 = (float) (tid / 9), A = (float) (tid % 3), B = (float) ((tid / 3) % 3);
    s_lut[tid] = S0 * k0 + A * k1 + B * k2;
  }

  uint8_t pre[NPRE];
  // (every lane loads, the lanes past the end of the window a byte they do not stage: no branch and no zero-initialisation around the
  // loads — with them the compiler waits for the whole memory queue, the previous tile's stores included, before it reuses `pre`)
  auto prefetch = [&](int y0) {
#pragma unroll
    for(int k = 0; k < NPRE; ++k) {
      const int i = min(tid + k * 256, NSRC - 1);
      if constexpr(FROM_IMAGE) {
        // image window rows y0-3 .. y0+BP_TH+2, columns x0-3 .. x0+BP_TW+2, clamped (clamped positions are never used: a census
        // position on the image border is 0 without looking at its neighbours)
        const int ly = i / IC, lx = i - ly * IC;
        const int gy = min(max(y0 + ly - BP_HALO - 1, 0), R - 1), gx = min(max(x0 + lx - BP_HALO - 1, 0), W - 1);
        pre[k] = cen[(size_t) gy * W + gx];
      } else {
        const int ly = i / CC, lx = i - ly * CC;
        const int gy = reflect101(min(y0 + ly - BP_HALO, R + 1), R), gx = reflect101(min(x0 + lx - BP_HALO, W + 1), W);
        pre[k] = cen[(size_t) gy * W + gx];
      }
    }
  };
  // registers -> LDS: the image window (FROM_IMAGE) or the spread census bytes of the tile
  auto stage = [&]() {
#pragma unroll
    for(int k = 0; k < NPRE; ++k) {
      const int i = tid + k * 256;
      if constexpr(FROM_IMAGE) {
        if(i < IR * IC) { const int ly = i / IC, lx = i - ly * IC; s_img[ly * IW + lx] = pre[k]; }
      } else {
        if(i < CR * CC) { const int ly = i / CC, lx = i - ly * CC; s_cen[ly * CW + lx] = spread_planes(pre[k]); }
      }
    }
  };
  // Order of a tile: census (from s_img) | barrier | next tile's loads issued, row pass | barrier | next tile staged, column pass with
  // its stores | barrier.  The staging sits BEFORE the stores on purpose: its wait for the loads is a wait for everything older in the
  // memory queue (one in-order counter for loads and stores on gfx9) — placed at the top of the next tile it waited for the stores the
  // column pass had just issued, a full memory round trip per tile; here the youngest stores in the queue are a census and a row pass old.
  const int ybase = blockIdx.y * BP_TH * stack;
  prefetch(ybase);
  stage();
  __syncthreads();
  for(int t = 0; t < stack; ++t) {
    const int y0 = ybase + t * BP_TH;
    if(y0 >= R) break;
    if constexpr(FROM_IMAGE) {
      // census of the staged positions: (gy, gx) = REFLECT_101 of the census coordinate, read from the image window
      for(int i = tid; i < CR * CC; i += 256) {
        const int ly = i / CC, lx = i - ly * CC;
        const int gy = reflect101(min(y0 + ly - BP_HALO, R + 1), R), gx = reflect101(min(x0 + lx - BP_HALO, W + 1), W);
        uint8_t out = 0;
        if(gx > 0 && gx < W - 1 && gy > 0 && gy < R - 1) {
          // window origin: image row y0 - 3, column x0 - 3 (the reflected position lies inside the window: |reflection| <= 2 px)
          const uint8_t* p = s_img + (gy - (y0 - BP_HALO - 1)) * IW + (gx - (x0 - BP_HALO - 1));
          const uint8_t c = p[0];
          out = (uint8_t) (((p[-IW - 1] >= c) << 0) | ((p[-IW] >= c) << 1) | ((p[-IW + 1] >= c) << 2) | ((p[-1] >= c) << 3) |
                           ((p[1] >= c) << 4) | ((p[IW - 1] >= c) << 5) | ((p[IW] >= c) << 6) | ((p[IW + 1] >= c) << 7));
        }
        s_cen[ly * CW + lx] = spread_planes(out);
        // lazy level: the census byte image is what template_build works from (the position's own pixel: no reflection inside the tile)
        if(lazy_level && ly >= BP_HALO && ly < BP_HALO + BP_TH && lx >= BP_HALO && lx < BP_HALO + BP_TW && y0 + ly - BP_HALO < R && x0 + lx - BP_HALO < W)
          cen_out[(size_t) (y0 + ly - BP_HALO) * W + (x0 + lx - BP_HALO)] = out;
      }
      __syncthreads();
    }
    const bool more = t + 1 < stack && y0 + BP_TH < R;
    if(more) prefetch(y0 + BP_TH);

    if(lazy) {
      // plane 0 only: the same table entry, the same column expression as the full form below
      for(int i = tid; i < CR * BP_TW; i += 256) {
        const int ly = i / BP_TW, lx = i - ly * BP_TW;
        const uint2* c = s_cen + ly * CW + lx;
        const unsigned ilo = (c[1].x + c[3].x) + 3u * (c[0].x + c[4].x) + 9u * c[2].x;
        s_row[i] = s_lut[ilo & 0xffu];
      }
      __syncthreads();
      if(more) stage();
      for(int i = tid; i < BP_TH * BP_TW; i += 256) {
        const int ly = i / BP_TW, lx = i - ly * BP_TW;
        const int gx = x0 + lx, gy = y0 + ly;
        if(gx >= W || gy >= R) continue;
        const float* Th = s_row + ly * BP_TW + lx;
        constexpr int pitch = BP_TW;
        float o = k0 * Th[2 * pitch]; o += k1 * (Th[3 * pitch] + Th[pitch]); o += k2 * (Th[4 * pitch] + Th[0]);
        ch0[(size_t) gy * W + gx] = o;
      }
      __syncthreads();
      continue;
    }
    // horizontal pass: (BP_TH + 4) rows x BP_TW columns, 5 positions per thread
    for(int i = tid; i < CR * BP_TW; i += 256) {
      const int ly = i / BP_TW, lx = i - ly * BP_TW;
      const uint2* c = s_cen + ly * CW + lx;   // c[0..4] = columns x-2..x+2
      const uint2 cm2 = c[0], cm1 = c[1], c0 = c[2], cp1 = c[3], cp2 = c[4];
      const unsigned ilo = (cm1.x + cp1.x) + 3u * (cm2.x + cp2.x) + 9u * c0.x;     // bytewise, <= 17: no carries
      const unsigned ihi = (cm1.y + cp1.y) + 3u * (cm2.y + cp2.y) + 9u * c0.y;
      float tt[8];
#pragma unroll
      for(int b = 0; b < 4; ++b) {
        tt[b] = s_lut[(ilo >> (8 * b)) & 0xffu];
        tt[4 + b] = s_lut[(ihi >> (8 * b)) & 0xffu];
      }
      // two planes of 4 channels each: consecutive lanes are 16 bytes apart in either plane, so the 16-byte LDS accesses of
      // both passes are bank-conflict free (one [8]-float record per pixel would put lanes 32 bytes apart: 2-way conflicts)
      float4* o = reinterpret_cast<float4*>(s_row);
      o[i] = make_float4(tt[0], tt[1], tt[2], tt[3]);
      o[BP_PLANE + i] = make_float4(tt[4], tt[5], tt[6], tt[7]);
    }
    __syncthreads();
    if(more) stage();

    // vertical pass.  A LANE PAIR per pixel: the even lane forms channels 0-3, the odd lane channels 4-7, so that every store instruction
    // of a wavefront covers 1 KB of contiguous records (32 pixels) instead of every other 16 bytes of 2 KB: the kernel runs at the rate of
    // its stores plus what of its arithmetic they do not cover (profiles/r03_bitplanes_stores.txt): 8.86 -> 8.68 ms per level-0 launch of 2048 frames
    for(int u = tid; u < 2 * BP_TH * BP_TW; u += 256) {
      const int i = u >> 1, h = u & 1;
      const int ly = i / BP_TW, lx = i - ly * BP_TW;
      const int gx = x0 + lx, gy = y0 + ly;
      if(gx >= W || gy >= R) continue;
      const float4* Th = reinterpret_cast<const float4*>(s_row) + h * BP_PLANE + ly * BP_TW + lx;     // row ly of s_row is image row gy-2
      constexpr int pitch = BP_TW;
      const float4 Tm2 = Th[0], Tm1 = Th[pitch], T0 = Th[2 * pitch], Tp1 = Th[3 * pitch], Tp2 = Th[4 * pitch];
      float4 o4;
      o4.x = k0 * T0.x; o4.x += k1 * (Tp1.x + Tm1.x); o4.x += k2 * (Tp2.x + Tm2.x);
      o4.y = k0 * T0.y; o4.y += k1 * (Tp1.y + Tm1.y); o4.y += k2 * (Tp2.y + Tm2.y);
      o4.z = k0 * T0.z; o4.z += k1 * (Tp1.z + Tm1.z); o4.z += k2 * (Tp2.z + Tm2.z);
      o4.w = k0 * T0.w; o4.w += k1 * (Tp1.w + Tm1.w); o4.w += k2 * (Tp2.w + Tm2.w);
      store_stream(reinterpret_cast<float4*>(desc + ((size_t) gy * W + gx) * 8) + h, o4);
      if(h == 0 && ch0) ch0[(size_t) gy * W + gx] = o4.x;      // (not for frames that only ever serve as the current frame of a pair batch)
    }
    __syncthreads();   // s_cen / s_row are rewritten by the next tile
  }
}

// sigma_bp <= 0: planes without smoothing (reference: bpvo/bitplanes_descriptor.cc:52-56 skipped)
__global__ __launch_bounds__(256) void bitplanes_noblur_kernel(const FrameJob* jobs)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int n = j.rows * j.cols;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if(i >= n) return;
  const unsigned c = j.cen[i];
  float4* d = reinterpret_cast<float4*>(j.desc + (size_t) i * 8);
  if(j.ch0) j.ch0[i] = (float) (c & 1u);
  d[0] = make_float4((float) (c & 1u), (float) ((c >> 1) & 1u), (float) ((c >> 2) & 1u), (float) ((c >> 3) & 1u));
  d[1] = make_float4((float) ((c >> 4) & 1u), (float) ((c >> 5) & 1u), (float) ((c >> 6) & 1u), (float) ((c >> 7) & 1u));
}

// ---- K3: saliency map (reference: bpvo/dense_descriptor.cc:92-100 -> bpvo/imgproc.cc:45-74 and :104-127).
// Closed form of the reference's memory effects, store bug of gradientAbsoluteMagnitudeAcc included (Q7, Q7b); p is the
// linear pixel index, reads at p-1 / p+1 deliberately run across row ends like the reference's pointer arithmetic:
//   g_c(p)    = |I_c[p-1] - I_c[p+1]| + |I_c[p-W] - I_c[p+W]|             (4-wide SSE body)
//   tail_c(p) = |I_c[p+1] - I_c[p-1]| + |I_c[p+W] + I_c[p-W]|             (scalar tail, `+` in the y term)
//   n = W & ~3.  rows 0, R-1 -> 0;  x = W-1 -> 0;
//   C == 1: x < n -> g_0, else tail_0.
//   C  > 1: 4 <= x < n -> g_0 (channel 0 only);  x < 4 -> S0(n-4+x) + g_{C-1}(n-4+x) with S0(W-1) = 0;
//           n <= x < W-1 -> ((tail_0 + tail_1) + ...) + tail_{C-1}.
template <int C>
__device__ __forceinline__ float grad_abs(const float* __restrict__ I, size_t p, int W, int c)
{
  const float Ix = fabsf(I[(p - 1) * C + c] - I[(p + 1) * C + c]);
  const float Iy = fabsf(I[(p - W) * C + c] - I[(p + W) * C + c]);
  return Ix + Iy;
}
template <int C>
__device__ __forceinline__ float grad_tail(const float* __restrict__ I, size_t p, int W, int c)
{
  return fabsf(I[(p + 1) * C + c] - I[(p - 1) * C + c]) + fabsf(I[(p + W) * C + c] + I[(p - W) * C + c]);
}

template <int C>
__global__ __launch_bounds__(256) void saliency_kernel(const FrameJob* jobs)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * ROWS_PER_THREAD;
  if(x >= W || y0 >= R) return;
  const float* __restrict__ I = j.desc;
  const int n = W & ~3;
  // (no early exit from the unrolled loop: with a `break` the compiler keeps the rows' loads in program order and every
  // row waits for its own loads)
#pragma unroll
  for(int k = 0; k < ROWS_PER_THREAD; ++k) {
    const int y = min(y0 + k, R - 1);
    float S = 0.0f;
    if(y >= 1 && y <= R - 2 && x != W - 1) {
      const size_t row = (size_t) y * W;
      if(x >= n) {
        S = grad_tail<C>(I, row + x, W, 0);
        for(int c = 1; c < C; ++c) S += grad_tail<C>(I, row + x, W, c);
      } else if(C == 1) {
        S = grad_abs<C>(I, row + x, W, 0);
      } else if(x >= 4) {
        if(C == 8 && j.ch0) S = grad_abs<1>(j.ch0, row + x, W, 0);     // channel 0 from its compact plane
        else S = grad_abs<C>(I, row + x, W, 0);
      } else {
        const int xs = n - 4 + x;
        const float S0 = (xs == W - 1) ? 0.0f : grad_abs<C>(I, row + xs, W, 0);
        S = S0 + grad_abs<C>(I, row + xs, W, C - 1);
      }
    }
    j.sal[(size_t) y * W + x] = S;
  }
}

// ---- K4: pixel selection (reference: bpvo/template_data.cc:43-89, bpvo/imgproc.h:117-160).
// Pass 1 (select_flag_kernel): candidate flag per pixel = saliency >= minSaliency && IsLocalMax && disparity gate,
// count per 256-pixel chunk of the row-major scan.  Pass 2 (select_scan_kernel): exclusive scan of the chunk counts,
// N = total & ~15 (the reference drops the LAST N mod 16 points).  Pass 3 (select_write_kernel): order-preserving
// compaction = chunk offset + in-chunk rank, keeps ranks < N, writes (y*W+x) and makePoint (rigid_body_warp.h:47-60).
__device__ __forceinline__ bool is_local_max(const float* __restrict__ S, int W, int radius, int y, int x)
{
  if(radius <= 0) return true;
  const float* p = S + (size_t) y * W + x;
  const float v = p[0];
  if(radius == 1) {   // WITH_SIMD form: 3 rows x 4 cols (cols -1..+2), strict > (Q8)
    // all 11 neighbours are loaded before any comparison (no short-circuit: that would serialise 11 memory latencies)
    float a[4], b[4], c[4];
#pragma unroll
    for(int k = 0; k < 4; ++k) { a[k] = p[k - 1 - W]; b[k] = p[k - 1]; c[k] = p[k - 1 + W]; }
    bool ok = true;
#pragma unroll
    for(int k = 0; k < 4; ++k) {
      if(k != 1) ok = ok & (v > b[k]);
      ok = ok & (v > a[k]) & (v > c[k]);
    }
    return ok;
  }
  for(int r = -radius; r <= radius; ++r)
    for(int c = -radius; c <= radius; ++c)
      if(!(r == 0 && c == 0) && p[r * W + c] >= v) return false;
  return true;
}

// A thread owns 4 consecutive pixels of the row-major scan, a wavefront therefore exactly one 256-pixel chunk (the unit
// of the order-preserving scan): chunk counts and in-chunk ranks are wavefront operations, no LDS, no barrier.  The 3 x 4
// NMS windows of the 4 pixels overlap (3 rows x 7 columns instead of 4 x 11 loads) when they lie in one row.
#ifndef SEL_PX_VALUE
#define SEL_PX_VALUE 4
#endif
constexpr int SEL_PX = SEL_PX_VALUE;            // pixels per thread (4, 8 or 16; measured: 8 is 3-7 % slower, 16 12-24 %)
constexpr int SEL_BLOCK_PX = 256 * SEL_PX;      // pixels per workgroup
constexpr int SEL_GROUP = 256 / SEL_PX;         // lanes that share one 256-pixel chunk (64, 32 or 16)
__global__ __launch_bounds__(256) void select_flag_kernel(const FrameJob* jobs, float min_saliency, float min_disp,
                                                          float max_disp, int border)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols, R = j.rows;
  const int npix = W * R;
  const int p0 = blockIdx.x * SEL_BLOCK_PX + threadIdx.x * SEL_PX;
  const float* __restrict__ S = j.sal;
  bool f[SEL_PX];
#pragma unroll
  for(int e = 0; e < SEL_PX; ++e) f[e] = false;
  if(p0 < npix) {
    const int y0 = p0 / W, x0 = p0 - y0 * W;
    const bool one_row = x0 + SEL_PX <= W && p0 + SEL_PX <= npix;
    const bool inner = one_row && j.nms_radius == 1 && y0 >= border && y0 < R - border - 1 && x0 >= border && x0 + SEL_PX - 1 < W - border - 1;
    if(inner) {
      // shared window: rows y0-1..y0+1, columns x0-1..x0+SEL_PX+1, all loaded before the first comparison
      float a[SEL_PX + 3], b[SEL_PX + 3], c[SEL_PX + 3];
      const float* q = S + (size_t) y0 * W + x0 - 1;
#pragma unroll
      for(int k = 0; k < SEL_PX + 3; ++k) { a[k] = q[k - W]; b[k] = q[k]; c[k] = q[k + W]; }
#pragma unroll
      for(int e = 0; e < SEL_PX; ++e) {
        const float v = b[e + 1];
        bool ok = v >= min_saliency;
#pragma unroll
        for(int k = 0; k < 4; ++k) {      // columns x-1 .. x+2 of pixel e = window entries e .. e+3, strict > (Q8)
          if(k != 1) ok = ok & (v > b[e + k]);
          ok = ok & (v > a[e + k]) & (v > c[e + k]);
        }
        f[e] = ok;
      }
    } else {
#pragma unroll
      for(int e = 0; e < SEL_PX; ++e) {
        const int p = p0 + e;
        if(p < npix) {
          const int y = p / W, x = p - y * W;
          if(y >= border && y < R - border - 1 && x >= border && x < W - border - 1)
            f[e] = S[p] >= min_saliency && is_local_max(S, W, j.nms_radius, y, x);
        }
      }
    }
    // disparity gate for the survivors (full-resolution map, template_data.cc:73-83)
#pragma unroll
    for(int e = 0; e < SEL_PX; ++e) {
      if(!f[e]) continue;
      const int p = p0 + e;
      const int y = p / W, x = p - y * W;
      const float d = j.disp[(size_t) (1 << j.level) * ((size_t) y * j.disp_cols + x)];
      f[e] = (d >= min_disp && d <= max_disp);
    }
    if(p0 + SEL_PX <= npix) {
#pragma unroll
      for(int e = 0; e < SEL_PX; e += 4)
        *reinterpret_cast<uchar4*>(j.flag + p0 + e) = make_uchar4(f[e], f[e + 1], f[e + 2], f[e + 3]);
    } else {
      for(int e = 0; e < SEL_PX && p0 + e < npix; ++e) j.flag[p0 + e] = f[e] ? 1 : 0;
    }
  }
  // chunk count = sum over the SEL_GROUP lanes of the chunk
  int cnt = 0;
#pragma unroll
  for(int e = 0; e < SEL_PX; ++e) cnt += (int) f[e];
#pragma unroll
  for(int o = SEL_GROUP / 2; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
  const int chunk = p0 / 256;
  if((threadIdx.x & (SEL_GROUP - 1)) == 0 && chunk * 256 < npix) j.blk_count[chunk] = cnt;
}

__global__ __launch_bounds__(1024) void select_scan_kernel(const FrameJob* jobs)
{
  const FrameJob& j = jobs[blockIdx.x];
  const int nblk = (j.rows * j.cols + 255) / 256;
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  if(threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for(int base = 0; base < nblk; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < nblk) ? j.blk_count[i] : 0;
    int incl = v;   // inclusive scan in the wave
#pragma unroll
    for(int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if(lane >= o) incl += t;
    }
    if(lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int woff = 0;
    for(int w = 0; w < wave; ++w) woff += s_wave[w];
    const int carry = s_carry;
    if(i < nblk) j.blk_count[i] = carry + woff + incl - v;   // exclusive offset
    __syncthreads();
    if(threadIdx.x == 1023) s_carry = carry + woff + incl;
    __syncthreads();
  }
  if(threadIdx.x == 0) {
    int total = s_carry;
    if(total > j.cap) total = j.cap;
    *j.n_out = total & ~15;
  }
}

__global__ __launch_bounds__(256) void select_write_kernel(const FrameJob* jobs)
{
  const FrameJob& j = jobs[blockIdx.z];
  const int W = j.cols;
  const int npix = W * j.rows;
  const int p0 = blockIdx.x * SEL_BLOCK_PX + threadIdx.x * SEL_PX;
  const int wave_p0 = blockIdx.x * SEL_BLOCK_PX + (threadIdx.x & ~63) * SEL_PX;
  if(wave_p0 >= npix) return;                 // whole wavefront past the image
  const int chunk = p0 / 256;
  unsigned char fe[SEL_PX];
#pragma unroll
  for(int e = 0; e < SEL_PX; ++e) fe[e] = 0;
  if(p0 + SEL_PX <= npix) {
#pragma unroll
    for(int e = 0; e < SEL_PX; e += 4) {
      const uchar4 v = *reinterpret_cast<const uchar4*>(j.flag + p0 + e);
      fe[e] = v.x; fe[e + 1] = v.y; fe[e + 2] = v.z; fe[e + 3] = v.w;
    }
  } else {
    for(int e = 0; e < SEL_PX && p0 + e < npix; ++e) fe[e] = j.flag[p0 + e];
  }
  int mine = 0;
#pragma unroll
  for(int e = 0; e < SEL_PX; ++e) mine += (int) fe[e];
  // exclusive prefix over the SEL_GROUP lanes of the chunk = rank inside the chunk
  const int gl = threadIdx.x & (SEL_GROUP - 1);
  int incl = mine;
#pragma unroll
  for(int o = 1; o < SEL_GROUP; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if(gl >= o) incl += t;
  }
  if(!mine) return;
  int rank = j.blk_count[chunk] + incl - mine;
  const int N = *j.n_out;
  const float fx = j.K[0], fy = j.K[4], cx = j.K[2], cy = j.K[5];
  const float Bf = j.b * fx;
#pragma unroll
  for(int e = 0; e < SEL_PX; ++e) {
    if(!fe[e]) continue;
    const int r = rank++;
    if(r >= N) break;
    const int p = p0 + e;
    const int y = p / W, x = p - y * W;
    const float d = j.disp[(size_t) (1 << j.level) * ((size_t) y * j.disp_cols + x)];
    const float Z = (float) ((double) Bf * (1.0 / (double) d));
    const float X = ((float) x - cx) * Z * (1.0f / fx);
    const float Y = ((float) y - cy) * Z * (1.0f / fy);
    // DisparitySpaceWarp::makePoint (bpvo/disparity_space_warp.h:31-34): (x - cx, y - cy, d, 1)
    j.pts[r] = j.dspace ? make_float4((float) x - cx, (float) y - cy, d, 1.0f) : make_float4(X, Y, Z, 1.0f);
    j.inds[r] = p;
  }
}

// ---- K3 + K4 in tiles (the form launch_saliency_select uses when the NMS radius is <= 1): saliency, the 3 x 4 NMS test and the disparity gate
// in ONE pass over 64 x 32 pixel tiles.  Channel 0 of the descriptor (its compact plane when C = 8) is staged in LDS with row-contiguous
// loads, the saliency of the tile and its halo is formed from there (the closed form above; the few columns the reference's SIMD body
// treats differently, x < 4 and x >= W & ~3, go through the per-pixel functions), written out once (bpvo_hip_get_saliency) and
// kept in LDS for the NMS windows.  A candidate flag is one BIT: a wavefront ballots the 64 pixels of a row segment into one 64-bit
// word, words[(y * WPR + x / 64)], WPR = ceil(W / 64) — words in (y, x / 64) order are pixels in row-major order, the order of the
// reference's scan.  select_words_scan_kernel turns the words' popcounts into exclusive offsets, select_words_write_kernel hands the
// set bits of a group of words to the lanes of a wavefront 64 at a time: rank = offset of the group + index of the bit in the group.
// Same values and order as the three-pass form above (which stays for radii > 1).
constexpr int ST_W = 64, ST_H = 32, ST_ROWS_PER_WAVE = ST_H / 4;
constexpr int ST_CH_ROWS = ST_H + 4, ST_CH_COLS = ST_W + 5, ST_CH_PITCH = 72;     // channel 0: rows y0-2 .. y0+H+1, columns x0-2 .. x0+W+2
constexpr int ST_S_ROWS = ST_H + 2, ST_S_COLS = ST_W + 3, ST_S_PITCH = 68;        // saliency:  rows y0-1 .. y0+H,   columns x0-1 .. x0+W+1

template <int C>
__device__ __forceinline__ float saliency_generic(const FrameJob& j, int x, int y)
{
  // the per-pixel closed form (see saliency_kernel), for the columns the tile does not cover
  const int W = j.cols, R = j.rows, n = W & ~3;
  if(!(y >= 1 && y <= R - 2 && x != W - 1)) return 0.0f;
  const float* __restrict__ I = j.desc;
  const size_t row = (size_t) y * W;
  if(x >= n) {
    float S = grad_tail<C>(I, row + x, W, 0);
    for(int c = 1; c < C; ++c) S += grad_tail<C>(I, row + x, W, c);
    return S;
  }
  if(C == 1 || x >= 4) return grad_abs<C>(I, row + x, W, 0);
  const int xs = n - 4 + x;
  const float S0 = (xs == W - 1) ? 0.0f : grad_abs<C>(I, row + xs, W, 0);
  return S0 + grad_abs<C>(I, row + xs, W, C - 1);
}

template <int C>
__global__ __launch_bounds__(256) void saliency_select_tile_kernel(const FrameJob* jobs, float min_saliency, float min_disp, float max_disp, int border, int nframes, int job_pitch)
{
  __shared__ float s_ch[ST_CH_ROWS][ST_CH_PITCH];
  __shared__ float s_sal[ST_S_ROWS][ST_S_PITCH];
  const FrameJob& j = level_job(jobs, blockIdx.z, nframes, job_pitch);
  const int W = j.cols, R = j.rows, n = W & ~3;
  const int x0 = blockIdx.x * ST_W, y0 = blockIdx.y * ST_H;
  if(x0 >= W || y0 >= R) return;      // (a coarser level inside a launch sized for the finest)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // channel 0 with its halo (coordinates clamped into the image: such entries only feed saliencies that are defined as 0 or replaced)
  // (a frame whose descriptor was computed without the compact plane — the current frame of a pair batch made a template after all — reads
  // channel 0 out of the records)
  const bool compact = C == 8 && j.ch0;
  const float* __restrict__ ch = compact ? j.ch0.get() : j.desc.get();
  const int CS = compact ? 1 : C;
  // (flat index over rows x pitch: every pass of the 256 threads is 3.5 full rows instead of one row plus a 5-lane remainder)
  for(int i = threadIdx.x; i < ST_CH_ROWS * ST_CH_PITCH; i += 256) {
    const int r = i / ST_CH_PITCH, cc = i - r * ST_CH_PITCH;
    if(cc < ST_CH_COLS) {
      const int y = min(max(y0 - 2 + r, 0), R - 1), xx = min(max(x0 - 2 + cc, 0), W - 1);
      s_ch[r][cc] = ch[((size_t) y * W + xx) * CS];
    }
  }
  __syncthreads();
  for(int i = threadIdx.x; i < ST_S_ROWS * ST_S_PITCH; i += 256) {
    const int r = i / ST_S_PITCH, cc = i - r * ST_S_PITCH;
    if(cc >= ST_S_COLS) continue;
    const int y = y0 - 1 + r, xx = x0 - 1 + cc;
    // the SIMD body's form from the tile, unconditionally; the exceptions (image border, the columns the reference treats apart) after it
    float S = fabsf(s_ch[r + 1][cc] - s_ch[r + 1][cc + 2]) + fabsf(s_ch[r][cc + 1] - s_ch[r + 2][cc + 1]);
    const bool inside = xx >= 0 && xx < W && y >= 0 && y < R;
    if(!(xx >= 4 && xx < n && xx != W - 1 && y >= 1 && y <= R - 2)) S = (inside && (xx < 4 || xx >= n)) ? saliency_generic<C>(j, xx, y) : 0.0f;
    s_sal[r][cc] = S;
  }
  __syncthreads();
  // candidates: one column per lane, ST_ROWS_PER_WAVE rows per wave, the 3 x 4 window slides down the column
  const int x = x0 + lane, c = lane + 1;
  const int radius = j.nms_radius;
  const bool col_ok = x >= border && x < W - border - 1;
  const int ry0 = wave * ST_ROWS_PER_WAVE;        // first row of this wave inside the tile
  // strict > against the 11 neighbours (Q8: 3 rows, columns x-1 .. x+2) = v > their maximum (saliencies are finite sums of |.|);
  // per row: side = max of columns x-1, x+1, x+2, full = max(side, column x)
  float vrow[ST_ROWS_PER_WAVE + 2], side[ST_ROWS_PER_WAVE + 2], full[ST_ROWS_PER_WAVE + 2];
#pragma unroll
  for(int q = 0; q < ST_ROWS_PER_WAVE + 2; ++q) {
    const float* t = &s_sal[ry0 + q][c - 1];
    vrow[q] = t[1];
    side[q] = fmaxf(fmaxf(t[0], t[2]), t[3]);
    full[q] = fmaxf(side[q], t[1]);
  }
  bool ok[ST_ROWS_PER_WAVE];
#pragma unroll
  for(int q = 0; q < ST_ROWS_PER_WAVE; ++q) {
    const int y = y0 + ry0 + q;
    const float v = vrow[q + 1];
    bool t = col_ok && y >= border && y < R - border - 1 && v >= min_saliency;
    if(radius == 1) t = t & (v > fmaxf(fmaxf(full[q], side[q + 1]), full[q + 2]));
    ok[q] = t;
  }
  // disparity gate for the survivors (full-resolution map, template_data.cc:73-83): all rows' loads in flight together
  float dv[ST_ROWS_PER_WAVE];
#pragma unroll
  for(int q = 0; q < ST_ROWS_PER_WAVE; ++q)
    dv[q] = ok[q] ? j.disp[(size_t) (1 << j.level) * ((size_t) (y0 + ry0 + q) * j.disp_cols + x)] : 0.0f;
  unsigned long long words = 0;      // lane q keeps the word of row q
#pragma unroll
  for(int q = 0; q < ST_ROWS_PER_WAVE; ++q) {
    const unsigned long long m = __ballot(ok[q] && dv[q] >= min_disp && dv[q] <= max_disp);
    if(lane == q) words = m;
  }
  const int y = y0 + ry0 + lane;
  if(lane < ST_ROWS_PER_WAVE && y < R) j.words[(size_t) y * ((W + 63) / 64) + blockIdx.x] = words;
  // the saliency map itself (bpvo_hip_get_saliency), LAST: stores issued before the gate's loads would be waited for with them (one in-order
  // counter for loads and stores), a memory round trip per tile
  if(x < W) {
    float* __restrict__ sal_out = j.sal;
#pragma unroll
    for(int q = 0; q < ST_ROWS_PER_WAVE; ++q) {
      const int yy = y0 + ry0 + q;
      if(yy < R) sal_out[(size_t) yy * W + x] = vrow[q + 1];
    }
  }
}

// exclusive scan of the words' popcounts in (y, x / 64) order; N = total & ~15 (the reference drops the LAST N mod 16 points)
__global__ __launch_bounds__(1024) void select_words_scan_kernel(const FrameJob* jobs, int nframes, int job_pitch)
{
  const FrameJob& j = level_job(jobs, blockIdx.x, nframes, job_pitch);
  const int nw = j.rows * ((j.cols + 63) / 64);
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  if(threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for(int base = 0; base < nw; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < nw) ? __popcll(j.words[i]) : 0;
    int incl = v;
#pragma unroll
    for(int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if(lane >= o) incl += t;
    }
    if(lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int woff = 0;
    for(int w = 0; w < wave; ++w) woff += s_wave[w];
    const int carry = s_carry;
    if(i < nw) j.blk_count[i] = carry + woff + incl - v;
    __syncthreads();
    if(threadIdx.x == 1023) s_carry = carry + woff + incl;
    __syncthreads();
  }
  if(threadIdx.x == 0) {
    int total = s_carry;
    if(total > j.cap) total = j.cap;
    *j.n_out = total & ~15;
  }
}

// order-preserving compaction: a wavefront takes SW_WORDS consecutive words and hands their set bits to its lanes 64 at a time (the t-th
// set bit of the group has rank offset + t: the arithmetic below runs on full wavefronts at the sparse levels too); writes (y*W+x)
// and makePoint (rigid_body_warp.h:47-60) of the bits whose rank is below N
constexpr int SW_WORDS = 8;
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n)      // position of the n-th (0-based) set bit of m
{
  int pos = 0;
#pragma unroll
  for(int w = 32; w >= 1; w >>= 1) {
    const int cnt = __popcll(m & ((1ull << w) - 1ull));
    if(n >= cnt) { n -= cnt; m >>= w; pos += w; }
  }
  return pos;
}
__global__ __launch_bounds__(256) void select_words_write_kernel(const FrameJob* jobs, int nframes, int job_pitch)
{
  const FrameJob& j = level_job(jobs, blockIdx.z, nframes, job_pitch);
  const int W = j.cols, WPR = (W + 63) / 64;
  const int nw = j.rows * WPR;
  const int lane = threadIdx.x & 63;
  const int w0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * SW_WORDS;
  if(w0 >= nw) return;
  unsigned long long m[SW_WORDS];
  int before[SW_WORDS + 1];        // set bits of the group in front of word k
  before[0] = 0;
#pragma unroll
  for(int k = 0; k < SW_WORDS; ++k) {
    m[k] = (w0 + k < nw) ? j.words[w0 + k] : 0ull;
    before[k + 1] = before[k] + __popcll(m[k]);
  }
  const int total = before[SW_WORDS];
  if(total == 0) return;
  const int off = j.blk_count[w0];
  const int N = *j.n_out;
  const float fx = j.K[0], fy = j.K[4], cx = j.K[2], cy = j.K[5];
  const float Bf = j.b * fx;
  // all disparities of the group first (at most SW_WORDS rounds of 64 set bits), then the arithmetic and the stores: a store issued
  // before the next round's load would be waited for together with it
  int px[SW_WORDS], py[SW_WORDS];
  float dd[SW_WORDS];
#pragma unroll
  for(int it = 0; it < SW_WORDS; ++it) {
    const int t = lane + 64 * it;
    px[it] = -1; py[it] = 0; dd[it] = 1.0f;
    if(t < total && off + t < N) {
      int k = 0;
      unsigned long long mk = m[0];
      int first = 0;
#pragma unroll
      for(int q = 1; q < SW_WORDS; ++q)
        if(t >= before[q]) { k = q; mk = m[q]; first = before[q]; }
      const int bit = nth_set_bit(mk, t - first);
      const int w = w0 + k;
      py[it] = w / WPR; px[it] = (w - py[it] * WPR) * 64 + bit;
      dd[it] = j.disp[(size_t) (1 << j.level) * ((size_t) py[it] * j.disp_cols + px[it])];
    }
  }
#pragma unroll
  for(int it = 0; it < SW_WORDS; ++it) {
    if(px[it] < 0) continue;
    const int r = off + lane + 64 * it, x = px[it], y = py[it];
    const float d = dd[it];
    const float Z = (float) ((double) Bf * (1.0 / (double) d));
    const float X = ((float) x - cx) * Z * (1.0f / fx);
    const float Y = ((float) y - cy) * Z * (1.0f / fy);
    // DisparitySpaceWarp::makePoint (bpvo/disparity_space_warp.h:31-34): (x - cx, y - cy, d, 1)
    j.pts[r] = j.dspace ? make_float4((float) x - cx, (float) y - cy, d, 1.0f) : make_float4(X, Y, Z, 1.0f);
    j.inds[r] = y * W + x;
  }
}

// ---- Hartley normalisation (reference: bpvo/warps.cc:27-48, bpvo/rigid_body_warp.h:62-71).
// The reference sums N points sequentially in f32; to reproduce its rounding the sums here are sequential too: chunks staged in LDS, one
// wave adding in point order.  It runs once per keyframe and level, all levels and frames side by side.  The dependent adds are the whole
// cost (26 k points at level 0 of a 1241x376 frame, two passes), and a wave issues one instruction every four cycles whatever it is: so
// the adds take their operand from ANOTHER LANE of the row (DPP row_shl: no instruction of its own) — lane i of a row loads four
// consecutive elements, lane 0 of the row adds the 64 of them in order, one `v_add_f32_dpp` per element and one LDS read per sixteen.
// In the first pass the four rows of the wave keep the chains of x, y, z (and w) side by side.
// (History: one `ds_read_b32` + one add per element, unrolled by 16 / 64: 387 / 336 us per setTemplate of one 1241x376 frame.)
// Written as one asm block per batch: the compiler folds `update_dpp` + add into the same `v_add_f32_dpp`, but then separates every two
// of them by `s_nop 1` — its hazard table asks for two wait states between a VALU write and a DPP instruction that reads the register,
// whichever operand reads it; the hardware needs them for the operand that goes through the lane crossbar (src0: loaded from LDS here),
// not for the accumulator on the ordinary port (the sums are compared bit for bit with the sequential sums of the CPU restatement at
// every size the suite runs: tests/test_gpu_parity.py).  Both forms are built — the kernel takes the choice as a template parameter, the
// context's option "normalization_form" selects it, and tests/test_gpu_parity.py runs one against the other (a toolchain or hardware
// change that breaks the assumption shows there).  The hand-scheduled form is the default ONLY for the architecture it was measured on:
// NRM_DPP_ASM = 0 (any other target, or a build that says so) makes the compiler's form the only one.
#ifndef NRM_DPP_ASM
#if defined(__gfx950__) || !defined(__HIP_DEVICE_COMPILE__)
#define NRM_DPP_ASM 1
#else
#define NRM_DPP_ASM 0
#endif
#endif
#define NRM_ROW_(I) \
  "v_add_f32_dpp %0, %1, %0 row_shl:" #I " row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
  "v_add_f32_dpp %0, %2, %0 row_shl:" #I " row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
  "v_add_f32_dpp %0, %3, %0 row_shl:" #I " row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
  "v_add_f32_dpp %0, %4, %0 row_shl:" #I " row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
// value of lane (l + I) of l's row of 16 (0 where there is none: only lane 0 of a row is ever used)
template <int I>
__device__ __forceinline__ float nrm_row_lane(float v)
{
  if constexpr(I == 0) return v;
  else return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + I /* row_shl:I */, 0xf, 0xf, true));
}
template <int I>
__device__ __forceinline__ void nrm_add_row_c(float& acc, const float (&v)[4])
{
  if constexpr(I < 16) {
#pragma unroll
    for(int q = 0; q < 4; ++q) acc += nrm_row_lane<I>(v[q]);      // elements 4 I .. 4 I + 3 of the batch, in order
    nrm_add_row_c<I + 1>(acc, v);
  }
}
// acc (lane 0 of every row) += the 64 elements of the batch its row holds, in order
template <bool ASM>
__device__ __forceinline__ void nrm_add_batch(float& acc, const float (&v)[4])
{
#if NRM_DPP_ASM
  if constexpr(ASM) {
  asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4\n"
               NRM_ROW_(1) NRM_ROW_(2) NRM_ROW_(3) NRM_ROW_(4) NRM_ROW_(5) NRM_ROW_(6) NRM_ROW_(7) NRM_ROW_(8)
               NRM_ROW_(9) NRM_ROW_(10) NRM_ROW_(11) NRM_ROW_(12) NRM_ROW_(13) NRM_ROW_(14) NRM_ROW_(15)
               : "+v"(acc) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
  return;
  }
#endif
  nrm_add_row_c<0>(acc, v);
}
constexpr int NRM_THREADS = 256, NRM_CHUNK = 512;   // 20 KB of LDS: seven workgroups per CU (1024-point chunks: three; 0.93 -> 0.59 ms per 1024-pair step)
// FORM 2: no cross-lane traffic at all — the chunk is staged in LDS component by component, every lane of a row reads the SAME four consecutive
// elements of its row's component with one ds_read_b128 (a broadcast), and the chain is four plain dependent v_add_f32 per read.  Plain C++: no
// hazard assumption, nothing a toolchain can break — the portable form, and the third voice of the bit-identity test — but measured SLOWER than the
// DPP chains (310 against 170 us for a 1241x376 template, 4.0 against 2.2 ms for a dense 640x480 one: scripts/nrm_forms.py), so FORM 1 (asm DPP
// chains, gfx950) stays the default and FORM 0 (the compiler's DPP form) the fallback for other targets.
template <int FORM>
__device__ __forceinline__ void nrm_chain_b128(float& acc, const float4* __restrict__ sp, int n4)
{
  // n4 float4 (a multiple of 4: chunks hold multiples of 16 points) added in element order; the next four reads are in flight under these adds
  float4 a[4], b[4];
#pragma unroll
  for(int q = 0; q < 4; ++q) { a[q] = sp[q]; b[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
  for(int k = 0; k < n4; k += 4) {
    if(k + 4 < n4) {
#pragma unroll
      for(int q = 0; q < 4; ++q) b[q] = sp[k + 4 + q];
    }
#pragma unroll
    for(int q = 0; q < 4; ++q) { acc += a[q].x; acc += a[q].y; acc += a[q].z; acc += a[q].w; }
#pragma unroll
    for(int q = 0; q < 4; ++q) a[q] = b[q];
  }
}
// FORM 3: FORM 2's layout and reads, the adds as blocks of sixteen back-to-back plain `v_add_f32` (a dependent plain add issues every 6 cycles
// on gfx950, one with a DPP operand every 7: scripts/micro/addchain.hip), 32 elements per buffer and two buffers, so that a buffer's eight
// ds_read_b128 are in flight under the other's 192 cycles of adds; wave 0 does nothing else (the other three waves stage the chunks and form
// the second pass's distances).  Elements beyond `cnt` up to the next multiple of 32 are the zeros the staging wrote (x + 0 = x).
#define NRM_ADD16_(A, B, C, D) \
  asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4\n" \
               "v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %6\n v_add_f32 %0, %0, %7\n v_add_f32 %0, %0, %8\n" \
               "v_add_f32 %0, %0, %9\n v_add_f32 %0, %0, %10\n v_add_f32 %0, %0, %11\n v_add_f32 %0, %0, %12\n" \
               "v_add_f32 %0, %0, %13\n v_add_f32 %0, %0, %14\n v_add_f32 %0, %0, %15\n v_add_f32 %0, %0, %16\n" \
               : "+v"(acc) : "v"(A.x), "v"(A.y), "v"(A.z), "v"(A.w), "v"(B.x), "v"(B.y), "v"(B.z), "v"(B.w), \
                             "v"(C.x), "v"(C.y), "v"(C.z), "v"(C.w), "v"(D.x), "v"(D.y), "v"(D.z), "v"(D.w))
__device__ __forceinline__ void nrm_chain_asm(float& acc, const float4* __restrict__ sp, int cnt)
{
  const int blocks = (cnt + 31) / 32;
  float4 x[8], y[8];
#pragma unroll
  for(int q = 0; q < 8; ++q) x[q] = sp[q];
  // (the reads are unconditional — the block index clamped to the last one, whose values then go unused — so that the compiler can count
  // them: it waits for the older buffer only, and the newer one's reads stay in flight under the adds)
  for(int b = 0; b < blocks; b += 2) {
    const int by = min(b + 1, blocks - 1), bx = min(b + 2, blocks - 1);
#pragma unroll
    for(int q = 0; q < 8; ++q) y[q] = sp[by * 8 + q];
    NRM_ADD16_(x[0], x[1], x[2], x[3]);
    NRM_ADD16_(x[4], x[5], x[6], x[7]);
#pragma unroll
    for(int q = 0; q < 8; ++q) x[q] = sp[bx * 8 + q];
    if(b + 1 < blocks) {
      NRM_ADD16_(y[0], y[1], y[2], y[3]);
      NRM_ADD16_(y[4], y[5], y[6], y[7]);
    }
  }
}
template <int FORM>
__global__ __launch_bounds__(NRM_THREADS) void normalization_kernel(const FrameJob* jobs, int job_pitch, int first_level,
                                                                    int with_normalization)
{
  const FrameJob& j = jobs[(size_t) (first_level + blockIdx.y) * job_pitch + blockIdx.x];
  const int N = *j.n_out;
  const int tid = threadIdx.x;
  if(!with_normalization || N == 0) {
    if(tid == 0) { j.nrm[0] = 1.0f; j.nrm[1] = 0.0f; j.nrm[2] = 0.0f; j.nrm[3] = 0.0f; }
    return;
  }
  // Chunks of 512 points are staged in LDS with coalesced loads (zeros beyond N: x + 0 = x for every partial sum, which is never -0);
  // wave 0 then adds them strictly in point order, in batches of 64.  (a) Row r of wave 0 keeps the chain of component r, (b) the
  // global loads of chunk i + 1 are in flight while chunk i is being added (registers -> the other LDS buffer afterwards), and (c) in
  // the second pass every thread forms the distances of chunk i + 1 while wave 0 accumulates those of chunk i.  Same order, same roundings.
  constexpr bool ASM = FORM == 1;
  constexpr bool COMP = FORM >= 2;            // component-major staging (FORM 2, 3)
  // FORM 3: wave 0 only adds; threads 64 .. 255 stage the chunks and form the distances (three points each per chunk)
  constexpr int W0 = FORM == 3 ? 64 : 0, NW = NRM_THREADS - W0;
  __shared__ float4 s_pts[2][NRM_CHUNK];      // FORM 2, 3: the same 16 KB as s_comp[2][4][NRM_CHUNK], component by component
  __shared__ __align__(16) float s_dist[2][NRM_CHUNK];
  float (*s_comp)[4][NRM_CHUNK] = reinterpret_cast<float (*)[4][NRM_CHUNK]>(&s_pts[0][0]);
  auto stage = [&](int buf, int k, const float4& p) {
    if constexpr(COMP) { s_comp[buf][0][k] = p.x; s_comp[buf][1][k] = p.y; s_comp[buf][2][k] = p.z; s_comp[buf][3][k] = p.w; }
    else s_pts[buf][k] = p;
  };
  __shared__ float s_c[4];
  constexpr int PER = (NRM_CHUNK + NW - 1) / NW;     // points per (staging) thread and chunk
  const int wt = tid - W0;                            // index among the staging threads (< 0: wave 0 of FORM 3)
  auto mine = [&](int q) { return wt >= 0 && q * NW + wt < NRM_CHUNK; };
  const int nchunks = (N + NRM_CHUNK - 1) / NRM_CHUNK;
  const int row = (tid >> 4) & 3, li = tid & 15;
  float4 pre[PER];
  auto fetch = [&](int chunk) {
    const int base = chunk * NRM_CHUNK;
#pragma unroll
    for(int q = 0; q < PER; ++q) {
      const int k = base + q * NW + wt;
      pre[q] = (mine(q) && k < N) ? j.pts[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  };
  fetch(0);
#pragma unroll
  for(int q = 0; q < PER; ++q) if(mine(q)) stage(0, q * NW + wt, pre[q]);
  __syncthreads();
  float c = 0.0f;
  // FORM 3: the points of chunk i + 2 are requested while chunk i is being added and chunk i + 1, requested an iteration earlier, is staged —
  // a chunk's adds take 1.3 us, a trip to HBM up to two: with one chunk of lead the barrier waited for the loads (0.4 us per chunk)
  float4 pa[PER], pb[PER];
  auto fetch_to = [&](int chunk, float4 (&dst)[PER]) {
    const int base = chunk * NRM_CHUNK;
#pragma unroll
    for(int q = 0; q < PER; ++q) {
      const int k = base + q * NW + wt;
      dst[q] = (mine(q) && k < N) ? j.pts[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  };
  auto chunk_cnt = [&](int chunk) { return min(NRM_CHUNK, N - chunk * NRM_CHUNK); };
  if constexpr(FORM == 3) {
    auto stage_from = [&](int buf, const float4 (&src)[PER]) {
#pragma unroll
      for(int q = 0; q < PER; ++q) if(mine(q)) stage(buf, q * NW + wt, src[q]);
    };
    fetch_to(1, pa);
    for(int ch = 0; ch < nchunks; ch += 2) {
      fetch_to(ch + 2, pb);
      if(tid < 64) nrm_chain_asm(c, reinterpret_cast<const float4*>(s_comp[0][row]), chunk_cnt(ch));
      if(ch + 1 < nchunks) stage_from(1, pa);
      __syncthreads();
      if(ch + 1 >= nchunks) break;
      fetch_to(ch + 3, pa);
      if(tid < 64) nrm_chain_asm(c, reinterpret_cast<const float4*>(s_comp[1][row]), chunk_cnt(ch + 1));
      if(ch + 2 < nchunks) stage_from(0, pb);
      __syncthreads();
    }
  }
  for(int ch = 0; ch < nchunks && FORM != 3; ++ch) {      // (forms 0 - 2: one chunk of lead)
    const int cur = ch & 1;
    const int cnt = min(NRM_CHUNK, N - ch * NRM_CHUNK);
    if(ch + 1 < nchunks) fetch(ch + 1);
    if constexpr(FORM == 2) {
      if(tid < 64) nrm_chain_b128<FORM>(c, reinterpret_cast<const float4*>(s_comp[cur][row]), cnt / 4);
    } else if(tid < 64) {
      const float* sp = reinterpret_cast<const float*>(s_pts[cur]) + row + 16 * li;      // component `row` of point 4 li of a batch
      float v[4], nx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for(int q = 0; q < 4; ++q) v[q] = sp[4 * q];
      __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0) here, so that the loop does not wait for the NEXT batch's reads before every batch
      for(int b = 0; b < cnt; b += 64) {
        if(b + 64 < cnt) {      // the next batch's LDS reads are in flight under this batch's adds
#pragma unroll
          for(int q = 0; q < 4; ++q) nx[q] = sp[4 * (b + 64 + q)];
        }
        nrm_add_batch<ASM>(c, v);
#pragma unroll
        for(int q = 0; q < 4; ++q) v[q] = nx[q];
      }
    }
    if(ch + 1 < nchunks) {
#pragma unroll
      for(int q = 0; q < PER; ++q) if(mine(q)) stage(cur ^ 1, q * NW + wt, pre[q]);
    }
    __syncthreads();
  }
  const float fN = (float) N;
  if(tid < 64 && li == 0) s_c[row] = c / fN;
  __syncthreads();
  const float c0 = s_c[0], c1 = s_c[1], c2 = s_c[2], c3 = s_c[3];
  float dpre[PER];
  auto dists = [&](int chunk) {
    const int base = chunk * NRM_CHUNK;
#pragma unroll
    for(int q = 0; q < PER; ++q) {
      const int k = base + q * NW + wt;
      const bool in = mine(q) && k < N;
      const float4 p = in ? j.pts[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      const float d0 = p.x - c0, d1 = p.y - c1, d2 = p.z - c2, d3 = p.w - c3;
      dpre[q] = in ? sqrtf((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) : 0.0f;
    }
  };
  dists(0);
#pragma unroll
  for(int q = 0; q < PER; ++q) if(mine(q)) s_dist[0][q * NW + wt] = dpre[q];
  __syncthreads();
  float m = 0.0f;
  if constexpr(FORM == 3) {      // the same two chunks of lead: points requested two chunks ahead, their distances formed and stored one chunk ahead
    auto dist_store = [&](int buf, const float4 (&src)[PER], int chunk) {
      const int base = chunk * NRM_CHUNK;
#pragma unroll
      for(int q = 0; q < PER; ++q) {
        if(!mine(q)) continue;
        const float4 p = src[q];
        const float d0 = p.x - c0, d1 = p.y - c1, d2 = p.z - c2, d3 = p.w - c3;
        s_dist[buf][q * NW + wt] = (base + q * NW + wt < N) ? sqrtf((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) : 0.0f;
      }
    };
    fetch_to(1, pa);
    for(int ch = 0; ch < nchunks; ch += 2) {
      fetch_to(ch + 2, pb);
      if(tid < 64) nrm_chain_asm(m, reinterpret_cast<const float4*>(s_dist[0]), chunk_cnt(ch));
      if(ch + 1 < nchunks) dist_store(1, pa, ch + 1);
      __syncthreads();
      if(ch + 1 >= nchunks) break;
      fetch_to(ch + 3, pa);
      if(tid < 64) nrm_chain_asm(m, reinterpret_cast<const float4*>(s_dist[1]), chunk_cnt(ch + 1));
      if(ch + 2 < nchunks) dist_store(0, pb, ch + 2);
      __syncthreads();
    }
  }
  for(int ch = 0; ch < nchunks && FORM != 3; ++ch) {
    const int cur = ch & 1;
    const int cnt = min(NRM_CHUNK, N - ch * NRM_CHUNK);
    if(ch + 1 < nchunks) dists(ch + 1);
    if constexpr(FORM == 2) {
      if(tid < 64) nrm_chain_b128<FORM>(m, reinterpret_cast<const float4*>(s_dist[cur]), cnt / 4);
    } else if(tid < 64) {
      float4 d4 = *reinterpret_cast<const float4*>(&s_dist[cur][4 * li]), n4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      for(int b = 0; b < cnt; b += 64) {
        if(b + 64 < cnt) n4 = *reinterpret_cast<const float4*>(&s_dist[cur][b + 64 + 4 * li]);
        const float v[4] = {d4.x, d4.y, d4.z, d4.w};
        nrm_add_batch<ASM>(m, v);
        d4 = n4;
      }
    }
    if(ch + 1 < nchunks) {
#pragma unroll
      for(int q = 0; q < PER; ++q) if(mine(q)) s_dist[cur ^ 1][q * NW + wt] = dpre[q];
    }
    __syncthreads();
  }
  if(tid == 0) {
    m /= fN;
    const float s = (float) (sqrt(3.0) / (double) fmaxf(m, 1e-6f));
    j.nrm[0] = s; j.nrm[1] = c0; j.nrm[2] = c1; j.nrm[3] = c2;
  }
}

// ---- K5: template pixels, central-difference gradients and 1x6 Jacobians
// (reference: bpvo/template_data.cc:102-137; Jacobian = bpvo/rigid_body_warp.cc:60-315 in the SSE code's operation
// gradients pre-multiplied by fx, fy).  The 1x6 Jacobian rows themselves are recomputed on the fly by irls_reduce (types.h
// jac_row), so only pix and (Ix, Iy) are stored, in the tiled layout of types.h.
// One row-pass value vector of the bit-planes blur at census position (px, r) from the census byte image: the five bytes of the row
// window (REFLECT_101 on the census coordinates, as bitplanes_blur_kernel stages them), spread to a byte per plane, combined into the
// table index of all eight planes, looked up — the operations of that kernel's row pass, value for value.
__device__ __forceinline__ void bp_row_from_census(const uint8_t* __restrict__ cen, int W, int R, int px, int r, const float* __restrict__ lut, float (&T)[8])
{
  const uint8_t* row = cen + (size_t) reflect101(r, R) * W;
  const uint2 cm2 = spread_planes(row[reflect101(px - 2, W)]), cm1 = spread_planes(row[reflect101(px - 1, W)]), c0 = spread_planes(row[reflect101(px, W)]),
              cp1 = spread_planes(row[reflect101(px + 1, W)]), cp2 = spread_planes(row[reflect101(px + 2, W)]);
  const unsigned ilo = (cm1.x + cp1.x) + 3u * (cm2.x + cp2.x) + 9u * c0.x;
  const unsigned ihi = (cm1.y + cp1.y) + 3u * (cm2.y + cp2.y) + 9u * c0.y;
#pragma unroll
  for(int b = 0; b < 4; ++b) {
    T[b] = lut[(ilo >> (8 * b)) & 0xffu];
    T[4 + b] = lut[(ihi >> (8 * b)) & 0xffu];
  }
}
// ... and the column pass over five of them (rows y-2 .. y+2 of one column): bitplanes_blur_kernel's expression
__device__ __forceinline__ void bp_col_pass(const float (&Tm2)[8], const float (&Tm1)[8], const float (&T0)[8], const float (&Tp1)[8], const float (&Tp2)[8],
                                            float k0, float k1, float k2, float (&out)[8])
{
#pragma unroll
  for(int c = 0; c < 8; ++c) {
    float o = k0 * T0[c];
    o += k1 * (Tp1[c] + Tm1[c]);
    o += k2 * (Tp2[c] + Tm2[c]);
    out[c] = o;
  }
}

template <int C>
__global__ __launch_bounds__(256) void template_build_kernel(const FrameJob* jobs, int grad_cd5, float k0, float k1, float k2, int nframes, int job_pitch)
{
  const FrameJob& j = level_job(jobs, blockIdx.z, nframes, job_pitch);
  __shared__ float s_lut[18];
  if(C == 8 && j.lazy) {      // (uniform over the workgroup: a frame's level is lazy or it is not)
    if(threadIdx.x < 18) {    // the row-pass table of bitplanes_blur_kernel: entry a + 3 b + 9 S0
      const float S0 = (float) (threadIdx.x / 9), A = (float) (threadIdx.x % 3), B = (float) ((threadIdx.x / 3) % 3);
      s_lut[threadIdx.x] = S0 * k0 + A * k1 + B * k2;
    }
    __syncthreads();
  }
  const int N = *j.n_out;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if(i >= N) return;
  const int W = j.cols;
  const int ii = j.inds[i];
  // DisparitySpaceWarp::jacobian takes the raw gradients (multiplying by 1.0f is exact)
  const float fx = j.dspace ? 1.0f : j.K[0], fy = j.dspace ? 1.0f : j.K[4];
  const float* __restrict__ D = j.desc;
  float pixv[C], Ix[C], Iy[C];
  const float NN = 1.0f / 18.0f;
  if constexpr(C == 8) {
    // a pixel's 8 channels are one 32-byte record: every neighbour is fetched with two 16-byte loads (one request per
    // 128-byte line and record instead of eight dword requests that thrash the 16 KB L1 between them)
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(D) + (size_t) ii * 2;
    const ptrdiff_t rs = (ptrdiff_t) W * 2;   // row stride in float4
    auto load8 = [&](ptrdiff_t off, float (&v)[8]) {
      const float4 a = rec[off], b = rec[off + 1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    };
    float xp[8], xm[8], yp[8], ym[8];
    if(j.lazy) {
      // the five records of the CD3 stencil, formed from the census bytes exactly as bitplanes_blur_kernel forms them (lazy levels are
      // CD3 only: the host keeps CD5 templates dense): column x over rows y-3 .. y+3 serves (x, y-1), (x, y), (x, y+1); columns x -+ 1
      // over rows y-2 .. y+2 serve (x-1, y), (x+1, y)
      const int R = j.rows, x = ii % W, y = ii / W;
      const uint8_t* __restrict__ cen = j.cen;
      {
        float Tc[7][8];
#pragma unroll
        for(int r = 0; r < 7; ++r) bp_row_from_census(cen, W, R, x, y - 3 + r, s_lut, Tc[r]);
        bp_col_pass(Tc[0], Tc[1], Tc[2], Tc[3], Tc[4], k0, k1, k2, ym);
        bp_col_pass(Tc[1], Tc[2], Tc[3], Tc[4], Tc[5], k0, k1, k2, pixv);
        bp_col_pass(Tc[2], Tc[3], Tc[4], Tc[5], Tc[6], k0, k1, k2, yp);
      }
#pragma unroll
      for(int side = 0; side < 2; ++side) {
        float Ts[5][8];
#pragma unroll
        for(int r = 0; r < 5; ++r) bp_row_from_census(cen, W, R, x + (side ? 1 : -1), y - 2 + r, s_lut, Ts[r]);
        if(side) bp_col_pass(Ts[0], Ts[1], Ts[2], Ts[3], Ts[4], k0, k1, k2, xp);
        else bp_col_pass(Ts[0], Ts[1], Ts[2], Ts[3], Ts[4], k0, k1, k2, xm);
      }
    } else {
      load8(0, pixv);
      load8(2, xp); load8(-2, xm); load8(rs, yp); load8(-rs, ym);
    }
    if(!grad_cd5) {
#pragma unroll
      for(int c = 0; c < 8; ++c) {
        Ix[c] = fx * (0.5f * (xp[c] - xm[c]));
        Iy[c] = fy * (0.5f * (yp[c] - ym[c]));
      }
    } else {
      float xp2[8], xm2[8], yp2[8], ym2[8];
      load8(4, xp2); load8(-4, xm2); load8(2 * rs, yp2); load8(-2 * rs, ym2);
#pragma unroll
      for(int c = 0; c < 8; ++c) {
        Ix[c] = fx * (NN * (1.0f * xm2[c] - 8.0f * xm[c] + 8.0f * xp[c] - 1.0f * xp2[c]));
        Iy[c] = fy * (NN * (1.0f * ym2[c] - 8.0f * ym[c] + 8.0f * yp[c] - 1.0f * yp2[c]));
      }
    }
  } else {
#pragma unroll
    for(int c = 0; c < C; ++c) {
      const float* cc = D + (size_t) ii * C + c;
      float gx, gy;
      if(!grad_cd5) {
        gx = 0.5f * (cc[C] - cc[-C]);
        gy = 0.5f * (cc[(size_t) W * C] - cc[-(ptrdiff_t) W * C]);
      } else {
        gx = NN * (1.0f * cc[-2 * C] - 8.0f * cc[-C] + 8.0f * cc[C] - 1.0f * cc[2 * C]);
        gy = NN * (1.0f * cc[-2 * (ptrdiff_t) W * C] - 8.0f * cc[-(ptrdiff_t) W * C] + 8.0f * cc[(ptrdiff_t) W * C] - 1.0f * cc[2 * (ptrdiff_t) W * C]);
      }
      pixv[c] = cc[0];
      Ix[c] = fx * gx;      // Ix = _mm_mul_ps(FX, Ix) (rigid_body_warp.cc:103-104)
      Iy[c] = fy * gy;
    }
  }
  // tiled stores (types.h tile_index): consecutive lanes write consecutive vectors
  if constexpr(C == 8) {
    float4* pv = reinterpret_cast<float4*>(j.pix.get());
    store_stream(pv + tile_index<2>(i, 0), make_float4(pixv[0], pixv[1], pixv[2], pixv[3]));
    store_stream(pv + tile_index<2>(i, 1), make_float4(pixv[4], pixv[5], pixv[6], pixv[7]));
    float4* gv = reinterpret_cast<float4*>(j.grad.get());
    store_stream(gv + tile_index<4>(i, 0), make_float4(Ix[0], Ix[1], Ix[2], Ix[3]));
    store_stream(gv + tile_index<4>(i, 1), make_float4(Ix[4], Ix[5], Ix[6], Ix[7]));
    store_stream(gv + tile_index<4>(i, 2), make_float4(Iy[0], Iy[1], Iy[2], Iy[3]));
    store_stream(gv + tile_index<4>(i, 3), make_float4(Iy[4], Iy[5], Iy[6], Iy[7]));
  } else {      // generic C: point-major pix[N][C], grad[N][2][C] (C = 1: pix[N], grad[N] as (Ix, Iy) pairs)
#pragma unroll
    for(int c = 0; c < C; ++c) {
      j.pix[(size_t) i * C + c] = pixv[c];
      j.grad[((size_t) i * 2 + 0) * C + c] = Ix[c];
      j.grad[((size_t) i * 2 + 1) * C + c] = Iy[c];
    }
  }
}

// template_build for descriptors of more than 48 channels: the generic branch above with a run-time channel loop (no per-channel arrays); same
// arithmetic per channel, point-major pix[N][C], grad[N][2][C]
__global__ __launch_bounds__(256) void template_build_wide_kernel(const FrameJob* jobs, int C, int grad_cd5, int nframes, int job_pitch)
{
  const FrameJob& j = level_job(jobs, blockIdx.z, nframes, job_pitch);
  const int N = *j.n_out;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if(i >= N) return;
  const int W = j.cols;
  const int ii = j.inds[i];
  const float fx = j.dspace ? 1.0f : j.K[0], fy = j.dspace ? 1.0f : j.K[4];
  const float* __restrict__ D = j.desc;
  const float NN = 1.0f / 18.0f;
  const ptrdiff_t sx = C, sy = (ptrdiff_t) W * C;
#pragma unroll 4
  for(int c = 0; c < C; ++c) {
    const float* cc = D + (size_t) ii * C + c;
    float gx, gy;
    if(!grad_cd5) {
      gx = 0.5f * (cc[sx] - cc[-sx]);
      gy = 0.5f * (cc[sy] - cc[-sy]);
    } else {
      gx = NN * (1.0f * cc[-2 * sx] - 8.0f * cc[-sx] + 8.0f * cc[sx] - 1.0f * cc[2 * sx]);
      gy = NN * (1.0f * cc[-2 * sy] - 8.0f * cc[-sy] + 8.0f * cc[sy] - 1.0f * cc[2 * sy]);
    }
    j.pix[(size_t) i * C + c] = cc[0];
    j.grad[((size_t) i * 2 + 0) * C + c] = fx * gx;
    j.grad[((size_t) i * 2 + 1) * C + c] = fy * gy;
  }
}

// point counts of every (frame, level) of a batch into one contiguous array, so that the host reads them back with a
// single copy instead of one per frame
__global__ void gather_counts_kernel(const FrameJob* jobs, int job_pitch, int nframes, int first_level, int num_levels, int* out /*[nframes][kMaxLevels]*/)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if(t >= nframes * kMaxLevels) return;
  const int i = t / kMaxLevels, l = t - i * kMaxLevels;
  out[t] = (l >= first_level && l < num_levels) ? *jobs[(size_t) l * job_pitch + i].n_out : 0;
}

// Jacobians in the reference layout for the C ABI accessor (bpvo_hip_get_jacobians): J[(c*N + i)*6 + k]
template <int C>
__global__ __launch_bounds__(256) void export_jacobians_kernel(const FrameJob* job, float* out)
{
  const FrameJob& j = *job;
  const int N = *j.n_out;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if(i >= N) return;
  const float4 P = j.pts[i];
  const JacPoint jp = jac_point(P.x, P.y, P.z, j.nrm);
  float Ix[C], Iy[C];
  if constexpr(C == 8) {
    const float4* gv = reinterpret_cast<const float4*>(j.grad.get());
    const float4 a = gv[tile_index<4>(i, 0)], b = gv[tile_index<4>(i, 1)], c = gv[tile_index<4>(i, 2)], d = gv[tile_index<4>(i, 3)];
    Ix[0] = a.x; Ix[1] = a.y; Ix[2] = a.z; Ix[3] = a.w; Ix[4] = b.x; Ix[5] = b.y; Ix[6] = b.z; Ix[7] = b.w;
    Iy[0] = c.x; Iy[1] = c.y; Iy[2] = c.z; Iy[3] = c.w; Iy[4] = d.x; Iy[5] = d.y; Iy[6] = d.z; Iy[7] = d.w;
  } else {
#pragma unroll
    for(int c = 0; c < C; ++c) { Ix[c] = j.grad[((size_t) i * 2 + 0) * C + c]; Iy[c] = j.grad[((size_t) i * 2 + 1) * C + c]; }
  }
#pragma unroll
  for(int c = 0; c < C; ++c) {
    float J[6];
    if(j.dspace) dspace_jac_row(P.x, P.y, P.z, j.K[0], j.K[4], 1.0f / j.K[0], 1.0f / j.K[4], 1.0f / j.b, Ix[c], Iy[c], J);
    else jac_row(jp, Ix[c], Iy[c], J);
    float* o = out + ((size_t) c * N + i) * 6;
#pragma unroll
    for(int k = 0; k < 6; ++k) o[k] = J[k];
  }
}

// ... for descriptors of more than 48 channels: one thread per (point, channel)
__global__ __launch_bounds__(256) void export_jacobians_wide_kernel(const FrameJob* job, int C, float* out)
{
  const FrameJob& j = *job;
  const int N = *j.n_out;
  const size_t k = (size_t) blockIdx.x * 256 + threadIdx.x;
  if(k >= (size_t) N * C) return;
  const int c = (int) (k / N), i = (int) (k - (size_t) c * N);
  const float4 P = j.pts[i];
  const float Ix = j.grad[((size_t) i * 2 + 0) * C + c], Iy = j.grad[((size_t) i * 2 + 1) * C + c];
  float J[6];
  if(j.dspace) dspace_jac_row(P.x, P.y, P.z, j.K[0], j.K[4], 1.0f / j.K[0], 1.0f / j.K[4], 1.0f / j.b, Ix, Iy, J);
  else jac_row(jac_point(P.x, P.y, P.z, j.nrm), Ix, Iy, J);
  float* o = out + k * 6;
#pragma unroll
  for(int q = 0; q < 6; ++q) o[q] = J[q];
}

// ---- host-callable launchers ------------------------------------------------------------------------------------
static inline dim3 grid2d_rows(int W, int R, int nz) { return dim3((W + 63) / 64, (R + 4 * ROWS_PER_THREAD - 1) / (4 * ROWS_PER_THREAD), nz); }

void launch_ingest(hipStream_t s, const FrameJob* jobs_level0, const uint8_t* d_images, const float* d_disps, size_t npix, int nframes, int skip_odd_disp)
{
  const int blocks = (int) std::min<size_t>(256, (npix / 4 + 255) / 256);
  hipLaunchKernelGGL(ingest_kernel, dim3(blocks, 1, nframes), dim3(256), 0, s, jobs_level0, d_images, d_disps, npix, skip_odd_disp);
}
void launch_pyrdown(hipStream_t s, const FrameJob* src, const FrameJob* dst, int dW, int dR, int nframes)
{
  hipLaunchKernelGGL(pyrdown_u8_lds_kernel, dim3((dW + PDT_W - 1) / PDT_W, (dR + PDT_H * PD_STACK - 1) / (PDT_H * PD_STACK), nframes), dim3(256), 0, s, src, dst);
}
void launch_intensity(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, int nlevels, int job_pitch)
{
  hipLaunchKernelGGL(intensity_kernel, dim3((W * R + 1023) / 1024, 1, nframes * nlevels), dim3(256), 0, s, jobs, nframes, job_pitch);
}
// (nlevels > 1: the smoothed-census form only)
void launch_census(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, const int* blur_taps, int nlevels, int job_pitch)
{
  if(blur_taps)
    hipLaunchKernelGGL(census_blur_kernel, dim3((W + CB_TW - 1) / CB_TW, (R + CB_TH - 1) / CB_TH, nframes * nlevels), dim3(256), 0, s,
                       jobs, blur_taps[0], blur_taps[1], nframes, job_pitch);
  else
    hipLaunchKernelGGL(census_kernel, dim3((W + 63) / 64, (R + 4 * CENSUS_ROWS - 1) / (4 * CENSUS_ROWS), nframes), dim3(256), 0, s, jobs);
}
void launch_bitplanes(hipStream_t s, const FrameJob* jobs, int W, int R, int nframes, float sigma, const float k[3], int from_image, int nlevels,
                      int job_pitch)
{
  // tiles a workgroup walks: few frames do not fill the chip with stacks of four (1241x376 x 2 frames: 480 workgroups of 32 rows)
  const int stack = nframes >= 16 ? 8 : (nframes > 4 ? 4 : 1);
  const dim3 grid((W + BP_TW - 1) / BP_TW, (R + BP_TH * stack - 1) / (BP_TH * stack), nframes * nlevels);
  if(sigma > 0.0f && from_image)
    hipLaunchKernelGGL(bitplanes_blur_kernel<true>, grid, dim3(256), 0, s, jobs, k[0], k[1], k[2], stack, nframes, job_pitch);
  else if(sigma > 0.0f)
    hipLaunchKernelGGL(bitplanes_blur_kernel<false>, grid, dim3(256), 0, s, jobs, k[0], k[1], k[2], stack, nframes, job_pitch);
  else
    hipLaunchKernelGGL(bitplanes_noblur_kernel, dim3((W * R + 255) / 256, 1, nframes), dim3(256), 0, s, jobs);
}
void launch_saliency_select(hipStream_t s, const FrameJob* jobs, int C, int W, int R, int nframes, int nms_radius, float min_saliency,
                            float min_disp, float max_disp, int border, int nlevels, int job_pitch)
{
  if(nms_radius <= 1) {
    // tiles: saliency + NMS + gate in one pass, candidate bits, word scan, lane-per-pixel compaction
    const int WPR = (W + 63) / 64, nw = R * WPR;
    auto tile = [&](auto c) {
      hipLaunchKernelGGL(saliency_select_tile_kernel<decltype(c)::value>, dim3(WPR, (R + ST_H - 1) / ST_H, nframes * nlevels), dim3(256), 0, s, jobs,
                         min_saliency, min_disp, max_disp, border, nframes, job_pitch);
    };
    if(C <= 48 || !dispatch_wide_channels(C, tile)) dispatch_channels(C, tile);
    hipLaunchKernelGGL(select_words_scan_kernel, dim3(nframes * nlevels), dim3(1024), 0, s, jobs, nframes, job_pitch);
    hipLaunchKernelGGL(select_words_write_kernel, dim3((nw + 4 * SW_WORDS - 1) / (4 * SW_WORDS), 1, nframes * nlevels), dim3(256), 0, s, jobs, nframes,
                       job_pitch);
    return;
  }
  // larger NMS windows: the saliency map first, then flag bytes / chunk scan / compaction straight from it
  auto plain = [&](auto c) { hipLaunchKernelGGL(saliency_kernel<decltype(c)::value>, grid2d_rows(W, R, nframes), dim3(256), 0, s, jobs); };
  if(C <= 48 || !dispatch_wide_channels(C, plain)) dispatch_channels(C, plain);
  const int nblk = (W * R + SEL_BLOCK_PX - 1) / SEL_BLOCK_PX;
  hipLaunchKernelGGL(select_flag_kernel, dim3(nblk, 1, nframes), dim3(256), 0, s, jobs, min_saliency, min_disp, max_disp, border);
  hipLaunchKernelGGL(select_scan_kernel, dim3(nframes), dim3(1024), 0, s, jobs);
  hipLaunchKernelGGL(select_write_kernel, dim3(nblk, 1, nframes), dim3(256), 0, s, jobs);
}
// Small control tables (job rows, initial poses) copied by a KERNEL that reads the pinned host rows directly: while the upload pipeline
// keeps the DMA engines busy with 45 MB chunks a hipMemcpyAsync of a few KB on a lane's stream waits its turn behind them
__global__ __launch_bounds__(256) void copy_rows_kernel(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, size_t pitch8,
                                                        size_t width8, int rows)
{
  const size_t n = width8 * (size_t) rows;
  for(size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
    const size_t r = i / width8, k = i - r * width8;
    dst[r * pitch8 + k] = src[r * pitch8 + k];
  }
}
void launch_copy_rows(hipStream_t s, void* dst, const void* src_host_pinned, size_t pitch_bytes, size_t width_bytes, int rows)
{
  const size_t n = width_bytes / 8 * (size_t) rows;
  if(n == 0) return;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned) std::min<size_t>(64, (n + 255) / 256)), dim3(256), 0, s, (unsigned long long*) dst,
                     (const unsigned long long*) src_host_pinned, pitch_bytes / 8, width_bytes / 8, rows);
}
void launch_normalization(hipStream_t s, const FrameJob* jobs, int job_pitch, int nframes, int first_level, int num_levels,
                          int with_normalization, int form)
{
  const dim3 grid(nframes, num_levels - first_level);
  // 4 (the default): FORM 3 — plain adds on a wave of its own, 10 % faster on a dense template, at 118 registers instead of 28 — where every
  // workgroup of the launch is resident anyway; FORM 1 for the launches of a large batch, whose throughput is the number of resident chains
  if(form == 4) form = (nframes * (num_levels - first_level) <= 1024) ? 3 : 1;
  if(form == 3) hipLaunchKernelGGL(normalization_kernel<3>, grid, dim3(NRM_THREADS), 0, s, jobs, job_pitch, first_level, with_normalization);
  else if(form == 2) hipLaunchKernelGGL(normalization_kernel<2>, grid, dim3(NRM_THREADS), 0, s, jobs, job_pitch, first_level, with_normalization);
  else if(form == 1) hipLaunchKernelGGL(normalization_kernel<1>, grid, dim3(NRM_THREADS), 0, s, jobs, job_pitch, first_level, with_normalization);
  else hipLaunchKernelGGL(normalization_kernel<0>, grid, dim3(NRM_THREADS), 0, s, jobs, job_pitch, first_level, with_normalization);
}
void launch_gather_counts(hipStream_t s, const FrameJob* jobs, int job_pitch, int nframes, int first_level, int num_levels, int* out)
{
  hipLaunchKernelGGL(gather_counts_kernel, dim3((nframes * kMaxLevels + 255) / 256), dim3(256), 0, s, jobs, job_pitch, nframes, first_level,
                     num_levels, out);
}
void launch_template_build(hipStream_t s, const FrameJob* jobs, int C, int max_points, int nframes, int grad_cd5, const float gauss_k[3], int nlevels,
                           int job_pitch)
{
  if(max_points <= 0) return;
  const dim3 g((max_points + 255) / 256, 1, nframes * nlevels);
  if(C > 48) { hipLaunchKernelGGL(template_build_wide_kernel, g, dim3(256), 0, s, jobs, C, grad_cd5, nframes, job_pitch); return; }
  dispatch_channels(C, [&](auto c) {
    hipLaunchKernelGGL(template_build_kernel<decltype(c)::value>, g, dim3(256), 0, s, jobs, grad_cd5, gauss_k[0], gauss_k[1], gauss_k[2], nframes, job_pitch);
  });
}

void launch_export_jacobians(hipStream_t s, const FrameJob* job, int C, int n, float* out)
{
  if(n <= 0) return;
  const dim3 g((n + 255) / 256);
  if(C > 48) { hipLaunchKernelGGL(export_jacobians_wide_kernel, dim3((unsigned) (((size_t) n * C + 255) / 256)), dim3(256), 0, s, job, C, out); return; }
  dispatch_channels(C, [&](auto c) { hipLaunchKernelGGL(export_jacobians_kernel<decltype(c)::value>, g, dim3(256), 0, s, job, out); });
}

}  // namespace bpvo_hip
