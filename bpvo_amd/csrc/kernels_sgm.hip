// Stereo front-end, second matcher: the reference's in-tree semi-global matcher SgmStereo (utils/sgm.{h,cc} — the
// SPS-stereo SGM; selected by `StereoAlgorithm = SGM`, utils/stereo_algorithm.cc:42-59,127-133; conf/kitti_eval.cfg:27 and
// conf/kitti_stereo.cfg:5 run the KITTI evaluation with it).  Integer arithmetic throughout (int16 saturating path costs; two double
// expressions: the census weight and the sub-pixel step), so the bar is bit-equality with the oracle's restatement, which cites the
// source line by line (parity unpinned: utils/sgm.cc includes OpenCV and cannot be built here).
//
// The CPU code is one sequential sweep per pass that carries two paths at once; on the device the same sums are computed from
// their definitions, the sequential part reduced to what is sequential in the algorithm:
//   sgm_census_sobel   capped x-Sobel (right image mirrored, as the original stores it) and the 5x5 / 3x3 census code       per pixel
//   sgm_pixel_cost     Birchfield-Tomasi style sampling-insensitive |dSobel| + weighted census Hamming distance            per (y, x, d)
//   sgm_box_rows/cols  (2r+1)^2 box sum with clamped coordinates = the sliding row / column sums of the original, incl. the rows and
//                      the column the original never writes (S1 in the oracle); separable, four disparities per thread      per (y, x, d)
//   sgm_right_cost     the right image's cost volume: right(x, d) = left(x + d, d), clamped                                  per (y, x, d)
//   sgm_path_packed    one scanline (a row for the horizontal paths, a column for the vertical ones) per wavefront, lanes = pairs of
//                      disparities, sequential along the line: L(p, d) = min(L'(d), L'(d +- 1) + P1, min L' + P2) - (min L' + P2) +
//                      C(p, d) in packed int16 saturating arithmetic, neighbours and the wave minimum by DPP                   per line
//   sgm_wta            winner takes all + the original's sub-pixel expression in double                                      per pixel
//   speckle filter     connected components (|difference| <= 2 * factor between 4-neighbours) by lock-free union-find over run labels, sizes by
//                      atomics, regions of <= 100 pixels zeroed — the flood fill of the original finds the same components
//   sgm_lr_check       left-right consistency of the left map, conversion to float
// Bounds (profiles/r04_stereo_pmc.txt): the scanline kernel is a chain of `width` (or `height`) dependent steps per wavefront with
// 2 (2 H + 2 W) wavefronts in flight; everything else streams the 1- and 2-byte volumes once or twice.
#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace bpvo_hip {

namespace {

__device__ __forceinline__ int sat16(int v) { return min(32767, max(-32768, v)); }

__global__ __launch_bounds__(256) void sgm_census_sobel_kernel(const uint8_t* __restrict__ img, uint8_t* __restrict__ sobel, int* __restrict__ census,
                                                              int rows, int cols, int pitch, int cap, int crad, int flip)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(y >= rows) return;
  // the row pitch is wider than the image: the padding (and the image border) holds `cap` (memset in the original)
  if(x >= cols) {
    if(x < pitch) sobel[(size_t) pitch * y + x] = (uint8_t) cap;
    return;
  }
  const int c = img[(size_t) cols * y + x];
  int code = 0;
  for(int oy = -crad; oy <= crad; ++oy)
    for(int ox = -crad; ox <= crad; ++ox) {
      code <<= 1;
      const int yy = y + oy, xx = x + ox;
      if(yy >= 0 && yy < rows && xx >= 0 && xx < cols && (int) img[(size_t) cols * yy + xx] >= c) code += 1;
    }
  census[(size_t) cols * y + x] = code;
  int v = cap;
  const bool interior = y >= 1 && y < rows - 1 && x >= 1 && x < cols - 1;
  if(interior) {
    const uint8_t* r0 = img + (size_t) cols * (y - 1);
    const uint8_t* r1 = r0 + cols;
    const uint8_t* r2 = r1 + cols;
    v = ((int) r0[x + 1] + 2 * (int) r1[x + 1] + (int) r2[x + 1]) - ((int) r0[x - 1] + 2 * (int) r1[x - 1] + (int) r2[x - 1]);
    v = v > cap ? 2 * cap : (v < -cap ? 0 : v + cap);
  }
  // interior pixels land mirrored for the right image; the border columns keep `cap` at their own (unmirrored) place — which the
  // mirrored interior never touches (columns 0 and W - 1 map onto each other)
  const int xo = (flip && interior) ? cols - x - 1 : x;
  sobel[(size_t) pitch * y + xo] = (uint8_t) v;
}

__device__ __forceinline__ void half_minmax(const uint8_t* __restrict__ row, int x, int W, int& mn, int& mx)
{
  const int c = row[x];
  const int l = x > 0 ? (c + (int) row[x - 1]) / 2 : c;
  const int r = x < W - 1 ? (c + (int) row[x + 1]) / 2 : c;
  mn = min(min(l, r), c);
  mx = max(max(l, r), c);
}

// pixel-wise cost u8 [rows][cols][D] per ROW TILE: a workgroup handles SPC_TX pixels of one row; what a cost needs of the left pixel (centre,
// half-pixel minimum and maximum, census word) and of the SPC_TX + D - 1 right positions the tile's disparities reach is formed ONCE in LDS (one
// workgroup per pixel recomputing it per disparity from six loads: 200 us per 1241 x 376 x 128 frame, now 47), the weighted Hamming term
// (uint8) ((double) ham * cweight) comes from a 33-entry table of that very expression, and a thread writes four consecutive disparities as
// one dword.  D a multiple of 4, D <= 256.
constexpr int SPC_TX = 32, SPC_MAXD = 256;
__global__ __launch_bounds__(256) void sgm_pixel_cost_tile_kernel(const uint8_t* __restrict__ sl, const uint8_t* __restrict__ sr, const int* __restrict__ cl,
                                                                   const int* __restrict__ cr, uint8_t* __restrict__ pc, int rows, int cols, int pitch, int D,
                                                                   double cweight)
{
  __shared__ uint8_t s_lc[SPC_TX], s_lmin[SPC_TX], s_lmax[SPC_TX];
  __shared__ int s_cl[SPC_TX];
  __shared__ uint8_t s_rc[SPC_TX + SPC_MAXD], s_rmin[SPC_TX + SPC_MAXD], s_rmax[SPC_TX + SPC_MAXD];
  __shared__ int s_cr[SPC_TX + SPC_MAXD];
  __shared__ uint8_t s_ham[33];
  const int y = blockIdx.y, x0 = blockIdx.x * SPC_TX, tid = threadIdx.x;
  const uint8_t* lrow = sl + (size_t) pitch * y;
  const uint8_t* rrow = sr + (size_t) pitch * y;               // mirrored: original column x - d sits at cols - 1 - x + d
  const int nr = SPC_TX + D - 1;
  // right positions ri0 + k, k = 0 .. nr - 1, with ri0 = cols - 1 - (x0 + SPC_TX - 1) (those outside the row are never indexed: d <= x)
  const int ri0 = cols - SPC_TX - x0;
  // census words of the right image at columns cx0 + k, cx0 = x0 - (D - 1)
  const int cx0 = x0 - D + 1;
  for(int k = tid; k < nr; k += 256) {
    const int ri = ri0 + k;
    if(ri >= 0 && ri < cols) {
      int mn, mx;
      half_minmax(rrow, ri, cols, mn, mx);
      s_rc[k] = rrow[ri]; s_rmin[k] = (uint8_t) mn; s_rmax[k] = (uint8_t) mx;
    }
    const int cx = cx0 + k;
    if(cx >= 0 && cx < cols) s_cr[k] = cr[(size_t) cols * y + cx];
  }
  if(tid < SPC_TX && x0 + tid < cols) {
    const int x = x0 + tid;
    int mn, mx;
    half_minmax(lrow, x, cols, mn, mx);
    s_lc[tid] = lrow[x]; s_lmin[tid] = (uint8_t) mn; s_lmax[tid] = (uint8_t) mx;
    s_cl[tid] = cl[(size_t) cols * y + x];
  }
  if(tid >= 64 && tid < 64 + 33) s_ham[tid - 64] = (uint8_t) ((double) (tid - 64) * cweight);
  __syncthreads();
  const int dq = D >> 2;
  for(int it = tid; it < SPC_TX * dq; it += 256) {
    const int xi = it / dq, q = it - xi * dq;
    const int x = x0 + xi;
    if(x >= cols) break;                                       // (items are ordered by pixel)
    const int lc = s_lc[xi], lmin = s_lmin[xi], lmax = s_lmax[xi];
    const unsigned lcen = (unsigned) s_cl[xi];
    unsigned out = 0;
#pragma unroll
    for(int e = 0; e < 4; ++e) {
      const int d = min(4 * q + e, x);                          // costs beyond d = x repeat the one at d = x (S2)
      const int kr = (SPC_TX - 1 - xi) + d;                     // = cols - 1 - x + d - ri0
      const int kc = (xi + D - 1) - d;                          // = x - d - cx0
      const int rc = s_rc[kr], rmin = s_rmin[kr], rmax = s_rmax[kr];
      int l2r = max(0, lc - rmax);
      l2r = max(l2r, rmin - lc);
      int r2l = max(0, rc - lmax);
      r2l = max(r2l, lmin - rc);
      const int sad = min(l2r, r2l);
      const int ham = __popc(lcen ^ (unsigned) s_cr[kc]);
      out |= (unsigned) (uint8_t) (sad + (int) s_ham[ham]) << (8 * e);
    }
    *reinterpret_cast<unsigned*>(pc + ((size_t) cols * y + x) * D + 4 * q) = out;
  }
}

// (2r+1)^2 box sums with clamped coordinates; rows y + r >= rows and (for y >= 1) column 0 stay 0 (S1).  Separable: the sum over the
// window's columns first (u16, at most (2r+1) * 255), then over its rows — 2 (2r+1) reads per output instead of (2r+1)^2; integer sums,
// any order (round 3 evaluated the 25 clamped taps per output directly: 0.6 ms per 1241 x 376 x 128 frame, profiles/r03_stereo.txt).
// A thread owns four consecutive disparities of one pixel: 4-byte / 8-byte accesses, coalesced along d.
__global__ __launch_bounds__(256) void sgm_box_rows_kernel(const uint8_t* __restrict__ pc, uint16_t* __restrict__ rowsum, int rows, int cols, int D, int wrad)
{
  const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x;          // (pixel, group of 4 disparities)
  const int dq = D >> 2;
  const size_t p = t / dq;
  if(p >= (size_t) rows * cols) return;
  const int d = (int) (t - p * dq) * 4;
  const int x = (int) (p % cols);
  const size_t row0 = p - x;
  int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for(int ox = -wrad; ox <= wrad; ++ox) {
    const int xx = min(max(x + ox, 0), cols - 1);
    const uchar4 v = *reinterpret_cast<const uchar4*>(pc + (row0 + xx) * D + d);
    s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
  }
  *reinterpret_cast<ushort4*>(rowsum + p * D + d) = make_ushort4((uint16_t) s0, (uint16_t) s1, (uint16_t) s2, (uint16_t) s3);
}
__global__ __launch_bounds__(256) void sgm_box_cols_kernel(const uint16_t* __restrict__ rowsum, uint16_t* __restrict__ cost, int rows, int cols, int D, int wrad)
{
  const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x;
  const int dq = D >> 2;
  const size_t p = t / dq;
  if(p >= (size_t) rows * cols) return;
  const int d = (int) (t - p * dq) * 4;
  const int x = (int) (p % cols), y = (int) (p / cols);
  int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if(y + wrad < rows && (y == 0 || x >= 1)) {
    for(int oy = -wrad; oy <= wrad; ++oy) {
      const int yy = min(max(y + oy, 0), rows - 1);
      const ushort4 v = *reinterpret_cast<const ushort4*>(rowsum + ((size_t) cols * yy + x) * D + d);
      s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
  }
  *reinterpret_cast<ushort4*>(cost + p * D + d) = make_ushort4((uint16_t) s0, (uint16_t) s1, (uint16_t) s2, (uint16_t) s3);
}

__global__ void sgm_right_cost_kernel(const uint16_t* __restrict__ lcost, uint16_t* __restrict__ rcost, int rows, int cols, int D)
{
  const int x = blockIdx.x, y = blockIdx.y, d = threadIdx.x;
  if(d >= D) return;
  const int dd = min(d, cols - 1 - x);                         // past the image the last valid disparity's cost repeats
  rcost[((size_t) cols * y + x) * D + d] = lcost[((size_t) cols * y + x + dd) * D + dd];
}

// The same through LDS (D <= 128): the gather above reads one 2-byte value per lane, every lane from another 256-byte row (eight times
// the volume in requests: profiles/r04_stereo_pmc.txt).  Here a workgroup fetches the SRC_TX + D - 1 rows its SRC_TX pixels reach as
// whole rows (16-byte loads; neighbouring tiles share most of them in the L2) and reads the diagonal from LDS.
constexpr int SRC_TX = 64, SRC_MAXD = 128, SRC_PITCH = SRC_MAXD + 8;      // halfwords per staged row: 272 bytes, 16-byte aligned, banks spread along the diagonal
__global__ __launch_bounds__(256) void sgm_right_cost_tile_kernel(const uint16_t* __restrict__ lcost, uint16_t* __restrict__ rcost, int rows, int cols, int D)
{
  __shared__ __attribute__((aligned(16))) uint16_t s_rows[(SRC_TX + SRC_MAXD - 1) * SRC_PITCH];
  // Workgroups are handed to the XCDs round robin, and every XCD has its own L2: consecutive tiles of a row, which share all but SRC_TX of
  // their rows, go to ONE XCD (each takes a contiguous eighth of the tiles): 563 -> 119.5 MB read per 1241 x 376 x 128 frame, the volume itself.
  // (a 1-D grid of 8 * per workgroups, per = ceil(tiles / 8))
  const int gx = (cols + SRC_TX - 1) / SRC_TX, ntile = gx * rows, per = (int) gridDim.x >> 3;
  const int tile = ((int) blockIdx.x & 7) * per + ((int) blockIdx.x >> 3);
  if(tile >= ntile) return;
  const int y = tile / gx, x0 = (tile - y * gx) * SRC_TX, tid = threadIdx.x;
  const int nrow = min(SRC_TX + D - 1, cols - x0);             // rows of pixels x0 .. x0 + nrow - 1 (past the image nothing is read: dd <= cols - 1 - x)
  const int v8 = D >> 3;                                       // 16-byte vectors per row
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(lcost + ((size_t) cols * y + x0) * D);
  for(int i = tid; i < nrow * v8; i += 256) {
    const int r = i / v8, c = i - r * v8;
    *reinterpret_cast<uint4*>(&s_rows[r * SRC_PITCH + 8 * c]) = src[(size_t) r * v8 + c];
  }
  __syncthreads();
  const int dq = D >> 2;
  for(int it = tid; it < SRC_TX * dq; it += 256) {
    const int xl = it / dq, q = it - xl * dq;
    const int x = x0 + xl;
    if(x >= cols) break;
    const int last = cols - 1 - x;                             // past the image the last valid disparity's cost repeats
    uint16_t v[4];
#pragma unroll
    for(int e = 0; e < 4; ++e) {
      const int dd = min(4 * q + e, last);
      v[e] = s_rows[(xl + dd) * SRC_PITCH + dd];
    }
    *reinterpret_cast<ushort4*>(rcost + ((size_t) cols * y + x) * D + 4 * q) = make_ushort4(v[0], v[1], v[2], v[3]);
  }
}

// One scanline per wavefront.  The four directions of a cost volume — along the rows left to right and back, along the columns down and
// up — depend on the cost volume alone, and so do the two volumes (left, right image): ALL EIGHT sets of scanlines run in ONE launch,
// 2 * (2 rows + 2 cols) wavefronts, every direction writing its path costs L into a volume of its own (the original adds them into one
// sum volume as it goes; the winner-takes-all kernel adds the four in the original's order).
// L(p, d) = min(L'(d), L'(d +- 1) + P1, min L' + P2) - (min L' + P2) + C(p, d) in int16 saturating arithmetic.  A lane holds 2 NP
// consecutive disparities as PAIRS of 16-bit values in 32-bit registers: v_pk_add_i16 / v_pk_sub_i16 with clamp and v_pk_min_i16 do two
// disparities per instruction and the saturation for free (the costs are below 2^15: stereo_check rejects window radii whose box sums
// are not).  What is sequential is a step's dependent chain — neighbours across lanes, the recurrence, the wave minimum that the next
// step needs — so the chain is kept on the VALU: neighbours by DPP wave shifts, the minimum by the DPP row ladder + v_readlane (__shfl
// compiles to ds_bpermute, an LDS round trip each, eight in a row per step).  Memory: batches of SGM_PF steps, see the loop.
#ifndef SGM_PRIO_VALUE
#define SGM_PRIO_VALUE 1
#endif
#ifndef SGM_PF_VALUE
#define SGM_PF_VALUE 16
#endif
constexpr int SGM_PF = SGM_PF_VALUE;
typedef short sgm_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sgm_s16x2 sgm_pk(unsigned u) { return __builtin_bit_cast(sgm_s16x2, u); }
__device__ __forceinline__ unsigned sgm_bits(sgm_s16x2 v) { return __builtin_bit_cast(unsigned, v); }
template <int NP>
__global__ __launch_bounds__(64) void sgm_path_packed_kernel(const uint16_t* __restrict__ cost_l, const uint16_t* __restrict__ cost_r, int16_t* __restrict__ Lvol,
                                                             int rows, int cols, int D, int P1, int P2)
{
  constexpr int PF = SGM_PF, V = 2 * NP;
  // dispatch order: the rows of both cost volumes first (with cols > rows the longest chains of the launch), then the columns
  int side, path, line;
  {
    int q = blockIdx.x;
    if(q < 4 * rows) { side = q / (2 * rows); q -= side * 2 * rows; path = q < rows ? 0 : 2; line = q < rows ? q : q - rows; }
    else { q -= 4 * rows; side = q / (2 * cols); q -= side * 2 * cols; path = q < cols ? 1 : 3; line = q < cols ? q : q - cols; }
  }
  const int vertical = path & 1, dir = path < 2 ? 1 : -1;
#if SGM_PRIO_VALUE
  if(!vertical && cols > rows) __builtin_amdgcn_s_setprio(3);      // the longest chains of the launch first (kernels_sgbm.hip, sgbm_path_kernel)
#endif
  const uint16_t* __restrict__ cost = side ? cost_r : cost_l;
  const size_t vol = (size_t) rows * cols * D;
  int16_t* __restrict__ L = Lvol + (size_t) (side * 4 + path) * vol;

  const int lane = threadIdx.x;
  const int nsteps = vertical ? rows : cols;
  const size_t step_stride = (vertical ? (size_t) cols * D : (size_t) D);
  const size_t base = vertical ? (size_t) line * D : (size_t) line * cols * D;
  const int d0 = lane * V;
  const bool live = d0 < D;                                    // (D is a multiple of 16 and V divides it: a lane is all in or all out)
  unsigned prev[NP];
#pragma unroll
  for(int j = 0; j < NP; ++j) prev[j] = 0u;
  int prev_min = 0;
  const sgm_s16x2 P1v = {(short) P1, (short) P1};
  auto offset = [&](int st) { return base + (size_t) (dir > 0 ? st : nsteps - 1 - st) * step_stride + d0; };
  // The cost loads are UNCONDITIONAL — the step clamped to the line's last one, the lanes beyond D reading disparity 0: a load inside a
  // branch makes the compiler wait for it at the end of the branch (s_waitcnt vmcnt(0) right behind every load: the first version of this
  // kernel spent a full memory round trip per step, 1241 of them per row — that, not arithmetic or shuffles, was its 1.1 ms).
  const int d0_load = live ? d0 : 0;
  auto load_offset = [&](int st) { st = min(st, nsteps - 1); return base + (size_t) (dir > 0 ? st : nsteps - 1 - st) * step_stride + d0_load; };
  using word_t = typename std::conditional<NP == 1, uint32_t, uint64_t>::type;
  // A scanline is walked in BATCHES of PF steps: the PF cost words of a batch are requested back to back, the PF steps run on registers,
  // the PF words of path costs are stored back to back.  Loads and stores share one counter on this hardware and may complete out of
  // order with respect to each other, so the compiler can only wait for "everything" (vmcnt(0)) once both kinds are in flight: a load per
  // step next to a store per step — the software pipeline of the first versions — degenerates into one full memory round trip per STEP
  // (1241 of them per row).  In batches the round trip is paid once per PF steps.
  for(int s0 = 0; s0 < nsteps; s0 += PF) {
    word_t cw[PF], ow[PF];
#pragma unroll
    for(int i = 0; i < PF; ++i) cw[i] = *reinterpret_cast<const word_t*>(cost + load_offset(s0 + i));
#pragma unroll
    for(int i = 0; i < PF; ++i) {
      const word_t cword = cw[i];
      const short pm = (short) (prev_min + P2);
      const sgm_s16x2 pmv = {pm, pm};
      // neighbours by DPP wave shifts (a VALU move: ~10 cycles; ds_bpermute, what __shfl compiles to, is an LDS round trip of ~120): lane
      // 0 / lane 63 keep the `old` operand — the sentinel 32767 below disparity 0 / (for D = 64 V) behind the last one
      const unsigned left = (unsigned) __builtin_amdgcn_update_dpp((int) 0x7fff0000u, (int) prev[NP - 1], 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
      const unsigned right = (unsigned) __builtin_amdgcn_update_dpp((int) 0x00007fffu, (int) prev[0], 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
      unsigned cur[NP];
      int mn = 32767;
#pragma unroll
      for(int j = 0; j < NP; ++j) {
        const unsigned below = j > 0 ? prev[j - 1] : left;
        unsigned above = j < NP - 1 ? prev[j + 1] : right;
        if(d0 + 2 * j + 2 >= D) above = 0x7fffu;              // the sentinel behind the last disparity
        const sgm_s16x2 lm = sgm_pk((below >> 16) | (prev[j] << 16));      // (d - 1) of both halves
        const sgm_s16x2 lp = sgm_pk((prev[j] >> 16) | (above << 16));      // (d + 1) of both halves
        const sgm_s16x2 c = sgm_pk((unsigned) (cword >> (32 * j)));
        sgm_s16x2 a = __builtin_elementwise_min(sgm_pk(prev[j]), __builtin_elementwise_add_sat(lm, P1v));
        a = __builtin_elementwise_min(a, __builtin_elementwise_add_sat(lp, P1v));
        a = __builtin_elementwise_min(a, pmv);
        a = __builtin_elementwise_add_sat(__builtin_elementwise_sub_sat(a, pmv), c);
        cur[j] = sgm_bits(a);
        if(live) mn = min(mn, min((int) a.x, (int) a.y));
      }
      // wave minimum by the DPP ladder (row shifts inside the rows of 16 lanes, then the two row broadcasts): lane 63 ends up with it
      mn = min(mn, __builtin_amdgcn_update_dpp(mn, mn, 0x111 /*row_shr:1*/, 0xf, 0xf, false));
      mn = min(mn, __builtin_amdgcn_update_dpp(mn, mn, 0x112 /*row_shr:2*/, 0xf, 0xf, false));
      mn = min(mn, __builtin_amdgcn_update_dpp(mn, mn, 0x114 /*row_shr:4*/, 0xf, 0xf, false));
      mn = min(mn, __builtin_amdgcn_update_dpp(mn, mn, 0x118 /*row_shr:8*/, 0xf, 0xf, false));
      mn = min(mn, __builtin_amdgcn_update_dpp(mn, mn, 0x142 /*row_bcast:15*/, 0xa, 0xf, false));
      mn = min(mn, __builtin_amdgcn_update_dpp(mn, mn, 0x143 /*row_bcast:31*/, 0xc, 0xf, false));
      word_t out = 0;
#pragma unroll
      for(int j = 0; j < NP; ++j) out |= (word_t) cur[j] << (32 * j);
      ow[i] = out;
      // (steps past the end of the line — the last batch of a ragged line — leave the state alone: nothing follows them)
      if(s0 + i < nsteps) {
        prev_min = __builtin_amdgcn_readlane(mn, 63);
#pragma unroll
        for(int j = 0; j < NP; ++j) prev[j] = cur[j];
      }
    }
    if(live) {
#pragma unroll
      for(int i = 0; i < PF; ++i)
        if(s0 + i < nsteps) *reinterpret_cast<word_t*>(L + offset(s0 + i)) = ow[i];
    }
  }
}

// winner takes all (first minimum) + the sub-pixel expression of the original in double.  A wavefront per pixel: the D sums are read
// coalesced (V per lane), the first minimum is the wave minimum of (sum, d) packed into one integer.
// The sum of the four path costs in the original's order and saturating arithmetic: ((((0 + rows forward) + columns forward) + rows
// backward) + columns backward) (utils/sgm.cc:831-836, pass after pass).
__device__ __forceinline__ int sgm_sum4(const int16_t* __restrict__ L, size_t vol, size_t i)
{
  int s = sat16(0 + (int) L[i]);
  s = sat16(s + (int) L[vol + i]);
  s = sat16(s + (int) L[2 * vol + i]);
  return sat16(s + (int) L[3 * vol + i]);
}
template <int V>
__global__ __launch_bounds__(256) void sgm_wta_kernel(const int16_t* __restrict__ L4, uint16_t* __restrict__ disp, size_t npix, int D, double factor)
{
  const size_t p = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if(p >= npix) return;
  const size_t vol = npix * D;
  int v[V];
  int key = 0x7fffffff;
#pragma unroll
  for(int k = 0; k < V; ++k) {
    const int d = lane * V + k;
    v[k] = d < D ? sgm_sum4(L4, vol, p * D + d) : 32767;
    if(d < D) key = min(key, ((v[k] + 32768) << 9) | d);      // (d < 512)
  }
  // minimum over the wave on the VALU (DPP row ladder + v_readlane: six ds_bpermute round trips as __shfl_xor), and the three sums the
  // sub-pixel step needs out of the lanes' registers instead of twelve more dependent loads by lane 0
  key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x111 /*row_shr:1*/, 0xf, 0xf, false));
  key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x112 /*row_shr:2*/, 0xf, 0xf, false));
  key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x114 /*row_shr:4*/, 0xf, 0xf, false));
  key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x118 /*row_shr:8*/, 0xf, 0xf, false));
  key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x142 /*row_bcast:15*/, 0xa, 0xf, false));
  key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x143 /*row_bcast:31*/, 0xc, 0xf, false));
  const int bd = __builtin_amdgcn_readlane(key, 63) & 511;
  auto sum_at = [&](int dsel) {      // dsel uniform over the wave
    const int li = __builtin_amdgcn_readfirstlane(dsel / V), rem = dsel % V;
    int r = 0;
#pragma unroll
    for(int k = 0; k < V; ++k) {
      const int t = __builtin_amdgcn_readlane(v[k], li);
      r = (k == rem) ? t : r;
    }
    return r;
  };
  int out;
  if(bd > 0 && bd < D - 1) {
    const int c = sum_at(bd), l = sum_at(bd - 1), r = sum_at(bd + 1);
    double w;
    if(r < l) w = (double) bd * factor + (double) (r - l) / (double) (c - l) / 2.0 * factor + 0.5;
    else w = (double) bd * factor + (double) (r - l) / (double) (c - r) / 2.0 * factor + 0.5;
    // static_cast<int> on x86: truncation, INT_MIN for NaN / infinities / out of range (a zero denominator: S5)
    const bool ok = (w == w) && w > -2147483649.0 && w < 2147483648.0;
    out = ok ? (int) w : (int) 0x80000000;
  } else {
    out = (int) ((double) bd * factor);
  }
  if(lane == 0) disp[p] = (uint16_t) out;
}

// ---- speckle filter: connected components by lock-free union-find (roots = smallest pixel index of the component)
__device__ __forceinline__ int uf_find(int* __restrict__ lab, int a)
{
  int p = lab[a];
  while(p != a) { a = p; p = lab[a]; }
  return a;
}
__device__ __forceinline__ void uf_union(int* __restrict__ lab, int a, int b)
{
  for(;;) {
    a = uf_find(lab, a);
    b = uf_find(lab, b);
    if(a == b) return;
    if(a < b) { const int t = a; a = b; b = t; }       // link the larger root under the smaller
    const int old = atomicMin(&lab[a], b);
    if(old == a) return;
    a = old;
  }
}
// Two pixels are connected when both are non-zero and differ by at most max_diff.  Labels START as the first pixel of the horizontal run
// a pixel belongs to, found inside the wavefront by a ballot of the run starts (no atomics; a run that enters the wavefront from the left
// starts, for now, at the wavefront's first pixel): one union per wavefront ties such a run to its left part, and a vertical pair
// (p, p + cols) needs a union only where its left neighbour pair does not already imply it — (p - 1, p), (p - 1 + cols, p + cols) and
// (p - 1, p - 1 + cols) all connected — i.e. about once per pair of overlapping runs instead of once per pixel.  Same components as
// linking every neighbour pair (round 3: 0.27 ms per launch on a 1241 x 376 plane, every pixel of which is one component).
__device__ __forceinline__ bool sgm_connected(int a, int b, int max_diff) { return a != 0 && b != 0 && abs(a - b) <= max_diff; }
__global__ __launch_bounds__(256) void sgm_cc_init_kernel(const uint16_t* __restrict__ img, int* __restrict__ lab, int* __restrict__ size, int rows, int cols,
                                                         int max_diff)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int npix = rows * cols;
  const int lane = threadIdx.x & 63;
  const int v = p < npix ? (int) img[p] : 0;
  const int x = p < npix ? p % cols : 0;
  const bool start = p >= npix || v == 0 || x == 0 || !sgm_connected(v, (int) img[p - 1], max_diff);     // (a zero pixel "starts" nothing: it breaks runs)
  const unsigned long long starts = __ballot(start) | 1ull;          // lane 0 stands in for a run that comes in from the left
  if(p >= npix) return;
  const unsigned long long upto = starts & (~0ull >> (63 - lane));   // run starts at lanes <= this one
  const int first = 63 - __clzll((long long) upto);
  lab[p] = v != 0 ? p - (lane - first) : -1;
  size[p] = 0;
}
__global__ __launch_bounds__(256) void sgm_cc_merge_kernel(const uint16_t* __restrict__ img, int* __restrict__ lab, int rows, int cols, int max_diff)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  if(p >= rows * cols) return;
  const int v = img[p];
  if(v == 0) return;
  const int x = p % cols, y = p / cols;
  // a run cut by the wavefront boundary: its first pixel of this wavefront and the pixel before it
  if((threadIdx.x & 63) == 0 && x > 0 && sgm_connected(v, (int) img[p - 1], max_diff)) uf_union(lab, p, p - 1);
  if(y < rows - 1) {
    const int q = img[p + cols];
    if(sgm_connected(v, q, max_diff)) {
      bool implied = false;
      if(x > 0) {
        const int vl = img[p - 1], ql = img[p + cols - 1];
        implied = sgm_connected(vl, v, max_diff) && sgm_connected(ql, q, max_diff) && sgm_connected(vl, ql, max_diff);
      }
      if(!implied) uf_union(lab, p, p + cols);
    }
  }
}
__global__ __launch_bounds__(256) void sgm_cc_count_kernel(int* __restrict__ lab, int* __restrict__ size, int npix)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  int r = -1;
  if(p < npix && lab[p] >= 0) {
    r = uf_find(lab, p);
    lab[p] = r;                                            // (path compression; roots never change after the merge kernel)
  }
  // the pixels of a wavefront mostly share one root (a plane is ONE component of 400 k pixels: one atomic per pixel on one address took
  // 4.3 ms): the lanes with the leader's root are counted with one atomic, then the next root, ... — and the root the workgroup's first
  // lanes have is counted in LDS first, one global atomic per workgroup (7 300 waves on one address were 90 us of queueing)
  __shared__ int s_root, s_cnt;
  if(threadIdx.x == 0) { s_root = r; s_cnt = 0; }
  __syncthreads();
  const int wg_root = s_root;
  unsigned long long todo = __ballot(r >= 0);
  while(todo) {
    const int leader = __ffsll((long long) todo) - 1;
    const int lr = __builtin_amdgcn_readlane(r, leader);
    const unsigned long long same = __ballot(r == lr) & todo;
    if((threadIdx.x & 63) == leader) {
      if(lr == wg_root) atomicAdd(&s_cnt, (int) __popcll(same));
      else atomicAdd(&size[lr], (int) __popcll(same));
    }
    todo &= ~same;
  }
  __syncthreads();
  if(threadIdx.x == 0 && s_cnt > 0) atomicAdd(&size[wg_root], s_cnt);
}
__global__ __launch_bounds__(256) void sgm_cc_apply_kernel(uint16_t* __restrict__ img, const int* __restrict__ lab, const int* __restrict__ size, int npix,
                                                          int max_size)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  if(p >= npix || lab[p] < 0) return;
  if(size[lab[p]] <= max_size) img[p] = 0;
}

// enforceLeftRightConsistency (left half) + disparity / factor -> float
__global__ __launch_bounds__(256) void sgm_lr_check_kernel(const uint16_t* __restrict__ dl, const uint16_t* __restrict__ dr, float* __restrict__ out, int rows,
                                                          int cols, double factor, int thresh)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  if(p >= rows * cols) return;
  const int x = p % cols;
  int v = dl[p];
  if(v != 0) {
    const int ld = (int) ((double) v / factor + 0.5);
    if(x - ld < 0) v = 0;
    else {
      const int rd = (int) ((double) dr[p - ld] / factor + 0.5);
      if(rd == 0 || abs(ld - rd) > thresh) v = 0;
    }
  }
  out[p] = (float) ((double) v / factor);
}

}  // namespace

void launch_speckle_filter_u16(hipStream_t s, uint16_t* img, int* lab, int* size, int rows, int cols, int max_diff, int max_size)
{
  const size_t npix = (size_t) rows * cols;
  const unsigned nb = (unsigned) ((npix + 255) / 256);
  hipLaunchKernelGGL(sgm_cc_init_kernel, dim3(nb), dim3(256), 0, s, img, lab, size, rows, cols, max_diff);
  hipLaunchKernelGGL(sgm_cc_merge_kernel, dim3(nb), dim3(256), 0, s, img, lab, rows, cols, max_diff);
  hipLaunchKernelGGL(sgm_cc_count_kernel, dim3(nb), dim3(256), 0, s, lab, size, (int) npix);
  hipLaunchKernelGGL(sgm_cc_apply_kernel, dim3(nb), dim3(256), 0, s, img, lab, size, (int) npix, max_size);
}

size_t sgm_scratch_bytes(int rows, int cols, int D)
{
  const size_t npix = (size_t) rows * cols, pitch = (size_t) cols + 15 - (cols - 1) % 16;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  return 2 * up(pitch * rows) + 2 * up(npix * 4) + up(npix * D) + 2 * up(npix * D * 2) + up(8 * npix * D * 2) + 2 * up(npix * 2) + 2 * up(npix * 4);
}

// SGMStereo::compute (utils/sgm.cc:250-285) for `nframes` rectified pairs, one after the other on the stream (the cost volumes of a
// 1241 x 376 x 128 frame are 120 MB each: the scratch is per context, not per frame)
bool launch_stereo_sgm(hipStream_t s, const SgmLaunch& g)
{
  const int rows = g.rows, cols = g.cols, D = g.ndisp;
  if(D <= 0 || D % 16 || D > 256) return false;
  const size_t npix = (size_t) rows * cols;
  const int pitch = cols + 15 - (cols - 1) % 16;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  unsigned char* w = (unsigned char*) g.scratch;
  uint8_t* sob_l = w; w += up((size_t) pitch * rows);
  uint8_t* sob_r = w; w += up((size_t) pitch * rows);
  int* cen_l = (int*) w; w += up(npix * 4);
  int* cen_r = (int*) w; w += up(npix * 4);
  uint8_t* pc = w; w += up(npix * D);
  uint16_t* cost_l = (uint16_t*) w; w += up(npix * D * 2);
  uint16_t* cost_r = (uint16_t*) w; w += up(npix * D * 2);
  int16_t* Lvol = (int16_t*) w; w += up(8 * npix * D * 2);      // path costs: [side][direction][rows][cols][D]
  uint16_t* disp_l = (uint16_t*) w; w += up(npix * 2);
  uint16_t* disp_r = (uint16_t*) w; w += up(npix * 2);
  int* lab = (int*) w; w += up(npix * 4);
  int* size = (int*) w; w += up(npix * 4);
  const int cap = (std::min(std::max(g.sobel_cap, 15), 127)) | 1;
  const int dthreads = (D + 63) / 64 * 64;
  const dim3 gpix((pitch + 63) / 64, (rows + 3) / 4), gxy(cols, rows);
  const unsigned nb = (unsigned) ((npix + 255) / 256);
  for(int f = 0; f < g.nframes; ++f) {
    const uint8_t* L = g.left + npix * f;
    const uint8_t* R = g.right + npix * f;
    hipLaunchKernelGGL(sgm_census_sobel_kernel, gpix, dim3(256), 0, s, L, sob_l, cen_l, rows, cols, pitch, cap, g.census_radius, 0);
    hipLaunchKernelGGL(sgm_census_sobel_kernel, gpix, dim3(256), 0, s, R, sob_r, cen_r, rows, cols, pitch, cap, g.census_radius, 1);
    hipLaunchKernelGGL(sgm_pixel_cost_tile_kernel, dim3((cols + SPC_TX - 1) / SPC_TX, rows), dim3(256), 0, s, sob_l, sob_r, cen_l, cen_r, pc, rows, cols, pitch, D,
                       g.census_weight);
    {
      // (the row sums borrow the first path-cost volume: the scanline kernel overwrites it later on the same stream)
      uint16_t* rowsum = reinterpret_cast<uint16_t*>(Lvol);
      const unsigned nbq = (unsigned) ((npix * (size_t) (D / 4) + 255) / 256);
      hipLaunchKernelGGL(sgm_box_rows_kernel, dim3(nbq), dim3(256), 0, s, pc, rowsum, rows, cols, D, g.window_radius);
      hipLaunchKernelGGL(sgm_box_cols_kernel, dim3(nbq), dim3(256), 0, s, rowsum, cost_l, rows, cols, D, g.window_radius);
    }
    if(D <= SRC_MAXD) hipLaunchKernelGGL(sgm_right_cost_tile_kernel, dim3(8u * (unsigned) (((cols + SRC_TX - 1) / SRC_TX * rows + 7) / 8)), dim3(256), 0, s, cost_l, cost_r, rows, cols, D);
    else hipLaunchKernelGGL(sgm_right_cost_kernel, gxy, dim3(dthreads), 0, s, cost_l, cost_r, rows, cols, D);
    {
      const dim3 gp((unsigned) (2 * (2 * rows + 2 * cols)));
      if(D <= 128) hipLaunchKernelGGL(sgm_path_packed_kernel<1>, gp, dim3(64), 0, s, cost_l, cost_r, Lvol, rows, cols, D, g.P1, g.P2);
      else hipLaunchKernelGGL(sgm_path_packed_kernel<2>, gp, dim3(64), 0, s, cost_l, cost_r, Lvol, rows, cols, D, g.P1, g.P2);
    }
    for(int side = 0; side < 2; ++side) {
      uint16_t* disp = side == 0 ? disp_l : disp_r;
      const int16_t* L4 = Lvol + (size_t) side * 4 * npix * D;
      {
        const dim3 gw((unsigned) ((npix + 3) / 4));
        if(D <= 64) hipLaunchKernelGGL(sgm_wta_kernel<1>, gw, dim3(256), 0, s, L4, disp, npix, D, g.disparity_factor);
        else if(D <= 128) hipLaunchKernelGGL(sgm_wta_kernel<2>, gw, dim3(256), 0, s, L4, disp, npix, D, g.disparity_factor);
        else hipLaunchKernelGGL(sgm_wta_kernel<4>, gw, dim3(256), 0, s, L4, disp, npix, D, g.disparity_factor);
      }
      // speckleFilter(100, 2 * factor) (utils/sgm.cc:898)
      launch_speckle_filter_u16(s, disp, lab, size, rows, cols, (int) (2 * g.disparity_factor), 100);
    }
    hipLaunchKernelGGL(sgm_lr_check_kernel, dim3(nb), dim3(256), 0, s, disp_l, disp_r, g.disp + npix * f, rows, cols, g.disparity_factor, g.consistency_threshold);
  }
  return true;
}

}  // namespace bpvo_hip
