// Stereo front-end, third matcher: `StereoAlgorithm = SGBM` — cv::StereoSGBM of OpenCV 2.4 as the reference constructs
// it (utils/stereo_algorithm.cc:25-40; run :113-121; conf/kitti_seq_0.cfg:6), single-pass mode (five path directions; the two-pass mode
// cannot be reached through the reference's constructor call: nine positional arguments into a constructor of eleven), then medianBlur(3), filterSpeckles and / 16.
// OpenCV's source is absent from the reference tree: the oracle restates the published algorithm (parity unpinned), these kernels equal
// the oracle bit for bit.  Integer arithmetic throughout.
//
// The CPU code is one sequential sweep over the rows that carries four paths at once and a fifth on the way back; here every stage is
// computed from its definition, the sequential part reduced to what is sequential in the algorithm:
//   sgbm_planes        clipped x-Sobel and raw plane (first / last column = tab[0]) of both images                       per pixel
//   sgbm_pixel_cost    Birchfield-Tomasi cost on both planes (raw plane >> 2), u8, cost columns x in [maxD, width)      per (y, x, d)
//   sgbm_box_rows/cols SAD window: clamped sums over the cost columns, then over the rows, with the rows and the column the original
//                      never refreshes (rows >= height - SW/2 repeat the last computed row, cost column 0 keeps row 0)   per (y, x, d)
//   sgbm_path          one scanline per wavefront — rows both ways, columns down, both down-going diagonals — lanes = pairs of
//                      disparities in packed int16, neighbours and the wave minimum by DPP; a path starts from L = 0, min L = 0
//                      (the zeroed borders of the original's buffers)                                                    per line
//   sgbm_wta / sgbm_lr a wavefront per cost pixel: S = sat(sat(L0 + L1 + L2 + L3) + L4), first minimum, uniqueness, sub-pixel parabola
//                      (C division), the right view's votes (atomics on (cost, column) keys: smallest cost, then the column the
//                      original meets first); then the left-right check per pixel                                        per pixel
//   sgbm_median3       3 x 3 median of the int16 map, replicated border; filterSpeckles by the union-find kernels of kernels_sgm.hip
// Bounds: the scanline kernel is a chain of `width1` (or `height`) dependent steps per wavefront; everything else streams the volumes.
#include <algorithm>

#include "kernels.h"

namespace bpvo_hip {

namespace {

constexpr int kInvalidCost = 32767;

// ---- planes: [0] clip(x-Sobel) + ftzero, [1] raw; columns 0 and cols - 1 of BOTH hold tab[0] = ftzero (calcPixelCostBT's prologue)
__global__ __launch_bounds__(256) void sgbm_planes_kernel(const uint8_t* __restrict__ img, uint8_t* __restrict__ planes /*[2][rows][cols]*/, int rows, int cols, int ftzero)
{
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if(x >= cols) return;
  const size_t npix = (size_t) rows * cols;
  int sob = ftzero, raw = ftzero;
  if(x > 0 && x < cols - 1) {
    const uint8_t* r1 = img + (size_t) y * cols;
    const uint8_t* r0 = y > 0 ? r1 - cols : r1;
    const uint8_t* r2 = y < rows - 1 ? r1 + cols : r1;
    const int g = ((int) r1[x + 1] - (int) r1[x - 1]) * 2 + (int) r0[x + 1] - (int) r0[x - 1] + (int) r2[x + 1] - (int) r2[x - 1];
    sob = min(max(g, -ftzero), ftzero) + ftzero;
    raw = r1[x];
  }
  planes[(size_t) y * cols + x] = (uint8_t) sob;
  planes[npix + (size_t) y * cols + x] = (uint8_t) raw;
}

// min / max over the half-sample neighbourhood of column x of a plane row (image edge: the value itself)
__device__ __forceinline__ void bt_interval(const uint8_t* __restrict__ row, int x, int cols, int& v, int& lo, int& hi)
{
  v = row[x];
  const int l = x > 0 ? (v + (int) row[x - 1]) / 2 : v;
  const int r = x < cols - 1 ? (v + (int) row[x + 1]) / 2 : v;
  lo = min(min(l, r), v);
  hi = max(max(l, r), v);
}

// pix[y][x1][d] (u8: at most 2 ftzero + 63) for the cost columns x = minX1 + x1.  A workgroup handles PC_TX cost columns of one row: the
// (value, lo, hi) triples of the left columns and of the PC_TX + D - 1 right columns they reach are formed once in LDS.
constexpr int PC_TX = 64;
__global__ __launch_bounds__(256) void sgbm_pixel_cost_kernel(const uint8_t* __restrict__ pl, const uint8_t* __restrict__ pr, uint8_t* __restrict__ pix, int rows,
                                                             int cols, int minD, int D, int minX1, int width1)
{
  extern __shared__ unsigned s_bt[];      // [2 planes][left PC_TX | right PC_TX + D - 1] packed v | lo << 8 | hi << 16
  const int y = blockIdx.y, x1_0 = blockIdx.x * PC_TX;
  const int nr = PC_TX + D - 1;
  const size_t npix = (size_t) rows * cols;
  unsigned* sl = s_bt;                    // [2][PC_TX]
  unsigned* sr = s_bt + 2 * PC_TX;        // [2][nr]
  for(int i = threadIdx.x; i < 2 * PC_TX; i += 256) {
    const int pln = i / PC_TX, k = i - pln * PC_TX;
    const int x = min(minX1 + x1_0 + k, cols - 1);
    int v, lo, hi;
    bt_interval(pl + pln * npix + (size_t) y * cols, x, cols, v, lo, hi);
    sl[i] = (unsigned) v | ((unsigned) lo << 8) | ((unsigned) hi << 16);
  }
  // right columns x - d for x in [x0, x0 + PC_TX), d in [minD, minD + D): from x0 - (minD + D - 1) to x0 + PC_TX - 1 - minD
  const int xr0 = minX1 + x1_0 - (minD + D - 1);
  for(int i = threadIdx.x; i < 2 * nr; i += 256) {
    const int pln = i / nr, k = i - pln * nr;
    const int x = min(max(xr0 + k, 0), cols - 1);
    int v, lo, hi;
    bt_interval(pr + pln * npix + (size_t) y * cols, x, cols, v, lo, hi);
    sr[i] = (unsigned) v | ((unsigned) lo << 8) | ((unsigned) hi << 16);
  }
  __syncthreads();
  const int quads = D / 4;
  for(int i = threadIdx.x; i < PC_TX * quads; i += 256) {
    const int k = i / quads, q = i - k * quads;
    const int x1 = x1_0 + k;
    if(x1 >= width1) continue;
    unsigned out = 0;
#pragma unroll
    for(int e = 0; e < 4; ++e) {
      const int d = 4 * q + e;                       // disparity index (real disparity minD + d)
      const int rk = k + (D - 1) - d;                // position of column x - (minD + d) in the right strip
      int cost = 0;
#pragma unroll
      for(int pln = 0; pln < 2; ++pln) {
        const unsigned a = sl[pln * PC_TX + k], b = sr[pln * nr + rk];
        const int u = a & 255, u0 = (a >> 8) & 255, u1 = (a >> 16) & 255;
        const int v = b & 255, v0 = (b >> 8) & 255, v1 = (b >> 16) & 255;
        const int c0 = max(max(0, u - v1), v0 - u);
        const int c1 = max(max(0, v - u1), u0 - v);
        cost += min(c0, c1) >> (pln == 0 ? 0 : 2);
      }
      out |= (unsigned) cost << (8 * e);
    }
    *reinterpret_cast<unsigned*>(pix + ((size_t) y * width1 + x1) * D + 4 * q) = out;
  }
}

// horizontal window sums, clamped to the cost columns: hs[y][x1][d] = sum_{dx = -SW2 .. SW2} pix[y][clamp(x1 + dx, 0, width1 - 1)][d]   (u16)
__global__ __launch_bounds__(256) void sgbm_box_rows_kernel(const uint8_t* __restrict__ pix, uint16_t* __restrict__ hs, int rows, int width1, int D, int SW2)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;      // (pixel, quad of disparities)
  const int quads = D / 4;
  if(i >= (size_t) rows * width1 * quads) return;
  const int q = (int) (i % quads);
  const size_t p = i / quads;
  const int x1 = (int) (p % width1);
  const size_t rowbase = (p - x1) * (size_t) D;
  unsigned s[4] = {0, 0, 0, 0};
  for(int dx = -SW2; dx <= SW2; ++dx) {
    const int xx = min(max(x1 + dx, 0), width1 - 1);
    const unsigned w = *reinterpret_cast<const unsigned*>(pix + rowbase + (size_t) xx * D + 4 * q);
    s[0] += w & 255u; s[1] += (w >> 8) & 255u; s[2] += (w >> 16) & 255u; s[3] += w >> 24;
  }
  *reinterpret_cast<ushort4*>(hs + p * D + 4 * q) = make_ushort4((unsigned short) s[0], (unsigned short) s[1], (unsigned short) s[2], (unsigned short) s[3]);
}
// vertical sums, clamped to the image rows, with what the original never refreshes: rows y >= rows - SH2 repeat row rows - SH2 - 1 (at
// least row 0), and cost column 0 keeps row 0's value; stored as int16 like the original's buffer (the caller keeps the sums below 2^15)
__global__ __launch_bounds__(256) void sgbm_box_cols_kernel(const uint16_t* __restrict__ hs, int16_t* __restrict__ cost, int rows, int width1, int D, int SH2)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  const int quads = D / 4;
  if(i >= (size_t) rows * width1 * quads) return;
  const int q = (int) (i % quads);
  const size_t p = i / quads;
  const int x1 = (int) (p % width1);
  int y = (int) (p / width1);
  if(y >= rows - SH2) y = max(rows - SH2 - 1, 0);
  if(x1 == 0) y = 0;
  unsigned s[4] = {0, 0, 0, 0};
  for(int dy = -SH2; dy <= SH2; ++dy) {
    const int yy = min(max(y + dy, 0), rows - 1);
    const ushort4 w = *reinterpret_cast<const ushort4*>(hs + ((size_t) yy * width1 + x1) * D + 4 * q);
    s[0] += w.x; s[1] += w.y; s[2] += w.z; s[3] += w.w;
  }
  *reinterpret_cast<short4*>(cost + p * D + 4 * q) = make_short4((short) s[0], (short) s[1], (short) s[2], (short) s[3]);
}

// ---- the five path families.  Family f, line l -> start (x0, y0), step (dx, dy), number of steps.
//   0: -> along row l          1: down-right diagonal   2: down column l    3: down-left diagonal   4: <- along row l
// The diagonals start on the top row (l < width1: (l, 0)) or on the side they enter through (l >= width1: row l - width1 + 1).
struct SgbmLine { int x0, y0, dx, dy, n; };
__device__ __forceinline__ SgbmLine sgbm_line(int fam, int l, int rows, int width1)
{
  SgbmLine s;
  switch(fam) {
    case 0: s = {0, l, 1, 0, width1}; break;
    case 4: s = {width1 - 1, l, -1, 0, width1}; break;
    case 2: s = {l, 0, 0, 1, rows}; break;
    case 1:
      if(l < width1) s = {l, 0, 1, 1, min(width1 - l, rows)};
      else { const int y = l - width1 + 1; s = {0, y, 1, 1, min(width1, rows - y)}; }
      break;
    default:
      if(l < width1) s = {l, 0, -1, 1, min(l + 1, rows)};
      else { const int y = l - width1 + 1; s = {width1 - 1, y, -1, 1, min(width1, rows - y)}; }
      break;
  }
  return s;
}
__host__ __device__ inline int sgbm_family_lines(int fam, int rows, int width1)
{
  return (fam == 0 || fam == 4) ? rows : fam == 2 ? width1 : width1 + rows - 1;
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk(unsigned u) { return __builtin_bit_cast(s16x2, u); }
__device__ __forceinline__ unsigned bits(s16x2 v) { return __builtin_bit_cast(unsigned, v); }
#ifndef SGBM_PRIO_VALUE
#define SGBM_PRIO_VALUE 1
#endif
#ifndef SGBM_PF_VALUE
#define SGBM_PF_VALUE 16
#endif
// L(p, d) = C(p, d) + min(L'(d), L'(d - 1) + P1, L'(d + 1) + P1, min L' + P2) - (min L' + P2); L' = 0, min L' = 0 ahead of the first step.
// A lane holds 2 NP consecutive disparities as pairs of int16 (v_pk_* with clamp: the values stay inside int16 — sgbm limits — so the
// saturating forms equal the original's int arithmetic); neighbours by DPP wave shifts, the wave minimum by the DPP ladder.  The cost
// words of SGBM_PF steps are requested back to back, the steps run on registers, the path costs are stored back to back.
template <int NP>
__global__ __launch_bounds__(64) void sgbm_path_kernel(const int16_t* __restrict__ cost, int16_t* __restrict__ Lvol, int rows, int width1, int D, int P1, int P2)
{
  constexpr int PF = SGBM_PF_VALUE, V = 2 * NP;
  // dispatch order: the long chains (the rows, both ways) first, then the columns, then the diagonals
  int fam = 0, l = blockIdx.x;
  {
    const int order[5] = {0, 4, 2, 1, 3};
    int k = 0;
    for(; k < 5; ++k) {
      const int nl = sgbm_family_lines(order[k], rows, width1);
      if(l < nl) break;
      l -= nl;
    }
    fam = order[min(k, 4)];
  }
  const SgbmLine ln = sgbm_line(fam, l, rows, width1);
#if SGBM_PRIO_VALUE
  // The launch lasts as long as its longest chains — the rows, `width1` dependent steps against at most `rows` for everything else — while
  // every SIMD time-slices four or five lines: the rows get the issue priority, the short lines fill the gaps.
  if(fam == 0 || fam == 4) __builtin_amdgcn_s_setprio(3);
#endif
  const size_t vol = (size_t) rows * width1 * D;
  int16_t* __restrict__ L = Lvol + (size_t) fam * vol;
  const int lane = threadIdx.x;
  const int d0 = lane * V;
  const bool live = d0 < D;
  const int d0_load = live ? d0 : 0;
  const long long step_stride = ((long long) ln.dy * width1 + ln.dx) * D;
  const long long base = ((long long) ln.y0 * width1 + ln.x0) * D;
  unsigned prev[NP];
#pragma unroll
  for(int j = 0; j < NP; ++j) prev[j] = 0u;
  int prev_min = 0;
  const s16x2 P1v = {(short) P1, (short) P1};
  using word_t = typename std::conditional<NP == 1, uint32_t, uint64_t>::type;
#ifndef SGBM_PIPE_VALUE
#define SGBM_PIPE_VALUE 0
#endif
  // SGBM_PIPE_VALUE 1: the cost words of the NEXT batch requested before the steps of this one run.  Measured: 0.719 against 0.715 ms per
  // 1241 x 376 x 128 frame — no gain (the batch's own 16 loads are consumed with partial waits as they arrive; the chain of dependent steps
  // is what a line costs), so the plain form is the default.
  word_t cnext[PF];
  if(SGBM_PIPE_VALUE) {
#pragma unroll
    for(int i = 0; i < PF; ++i) cnext[i] = *reinterpret_cast<const word_t*>(cost + base + (long long) min(i, ln.n - 1) * step_stride + d0_load);
  }
  for(int s0 = 0; s0 < ln.n; s0 += PF) {
    word_t cw[PF], ow[PF];
    if(SGBM_PIPE_VALUE) {
#pragma unroll
      for(int i = 0; i < PF; ++i) cw[i] = cnext[i];
      if(s0 + PF < ln.n) {
#pragma unroll
        for(int i = 0; i < PF; ++i) cnext[i] = *reinterpret_cast<const word_t*>(cost + base + (long long) min(s0 + PF + i, ln.n - 1) * step_stride + d0_load);
      }
    } else {
#pragma unroll
      for(int i = 0; i < PF; ++i) cw[i] = *reinterpret_cast<const word_t*>(cost + base + (long long) min(s0 + i, ln.n - 1) * step_stride + d0_load);
    }
#pragma unroll
    for(int i = 0; i < PF; ++i) {
      const word_t cword = cw[i];
      const short pm = (short) (prev_min + P2);
      const s16x2 pmv = {pm, pm};
      const unsigned left = (unsigned) __builtin_amdgcn_update_dpp((int) 0x7fff0000u, (int) prev[NP - 1], 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
      const unsigned right = (unsigned) __builtin_amdgcn_update_dpp((int) 0x00007fffu, (int) prev[0], 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
      unsigned cur[NP];
      int mn = 32767;
#pragma unroll
      for(int j = 0; j < NP; ++j) {
        const unsigned below = j > 0 ? prev[j - 1] : left;
        unsigned above = j < NP - 1 ? prev[j + 1] : right;
        if(d0 + 2 * j + 2 >= D) above = 0x7fffu;              // the sentinel behind the last disparity
        const s16x2 lm = pk((below >> 16) | (prev[j] << 16));
        const s16x2 lp = pk((prev[j] >> 16) | (above << 16));
        const s16x2 c = pk((unsigned) (cword >> (32 * j)));
        s16x2 a = __builtin_elementwise_min(pk(prev[j]), __builtin_elementwise_add_sat(lm, P1v));
        a = __builtin_elementwise_min(a, __builtin_elementwise_add_sat(lp, P1v));
        a = __builtin_elementwise_min(a, pmv);
        a = __builtin_elementwise_add_sat(__builtin_elementwise_sub_sat(a, pmv), c);
        cur[j] = bits(a);
        if(live) mn = min(mn, min((int) a.x, (int) a.y));
      }
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
      if(s0 + i < ln.n) {
        prev_min = __builtin_amdgcn_readlane(mn, 63);
#pragma unroll
        for(int j = 0; j < NP; ++j) prev[j] = cur[j];
      }
    }
    if(live) {
#pragma unroll
      for(int i = 0; i < PF; ++i)
        if(s0 + i < ln.n) *reinterpret_cast<word_t*>(L + base + (long long) (s0 + i) * step_stride + d0) = ow[i];
    }
  }
}

// ---- selection in two launches that fill the chip.
// sgbm_wta: a WAVEFRONT per cost pixel — S = sat(sat(L0 + L1 + L2 + L3) + L4) (the original's order), first minimum, uniqueness, sub-pixel
// parabola (C division) -> the candidate map (int, scaled by 16), and the pixel's vote for the right view: a global atomicMin on a
// (cost, column) key per right-image column — smallest cost wins, among equals the column the original meets first (it walks the row from
// the right with a strict `>`).  (Round 5, first form: one workgroup per image row with the votes in LDS — 376 workgroups, 252 us per
// 1241 x 376 x 128 frame at 2 TB/s; profiles/r05_stereo_pmc_first.txt.)
__device__ __forceinline__ int sat16i(int v) { return min(max(v, -32768), 32767); }
template <int V>      // disparities per lane: D <= 64 V
__global__ __launch_bounds__(256) void sgbm_wta_kernel(const int16_t* __restrict__ Lvol, int* __restrict__ cand, unsigned* __restrict__ votes, int rows, int cols,
                                                      int width1, int D, int minD, int minX1, int uniqueness)
{
  const int lane = threadIdx.x & 63;
  const size_t p = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);      // cost pixel
  if(p >= (size_t) rows * width1) return;
  const int y = (int) (p / width1), x = (int) (p % width1);
  const size_t vol = (size_t) rows * width1 * D;
  const size_t at = p * D;
  int S[V];
  unsigned best = 0xffffffffu;       // (S + 32768) << 16 | d: the first minimum
#pragma unroll
  for(int k = 0; k < V; ++k) {
    const int d = lane + 64 * k;
    int s = kInvalidCost;
    if(d < D) {
      const int l0 = Lvol[at + d], l1 = Lvol[vol + at + d], l2 = Lvol[2 * vol + at + d], l3 = Lvol[3 * vol + at + d], l4 = Lvol[4 * vol + at + d];
      s = sat16i(sat16i(l0 + l1 + l2 + l3) + l4);
      best = min(best, ((unsigned) (s + 32768) << 16) | (unsigned) d);
    }
    S[k] = s;
  }
#pragma unroll
  for(int o = 32; o >= 1; o >>= 1) best = min(best, (unsigned) __shfl_xor((int) best, o));
  const int minS = (int) (best >> 16) - 32768, bestDisp = (int) (best & 0xffffu);
  if(minS >= kInvalidCost) return;                        // (every sum saturated: the original's strict `<` finds no minimum)
  bool clash = false;
#pragma unroll
  for(int k = 0; k < V; ++k) {
    const int d = lane + 64 * k;
    if(d < D && S[k] * (100 - uniqueness) < minS * 100 && abs(bestDisp - d) > 1) clash = true;
  }
  if(__any(clash)) return;
  int sm = 0, sp = 0;      // neighbours of the minimum for the parabola
  {
    const int dm = bestDisp - 1, dp = bestDisp + 1;
#pragma unroll
    for(int k = 0; k < V; ++k) {
      const int vm = __shfl(S[k], dm & 63), vp = __shfl(S[k], dp & 63);
      if((dm >> 6) == k) sm = vm;
      if((dp >> 6) == k) sp = vp;
    }
  }
  if(lane == 0) {
    int d = bestDisp;
    const int x2 = x + minX1 - d - minD;
    atomicMin(&votes[(size_t) y * cols + x2], ((unsigned) (minS + 32768) << 16) | (unsigned) (65535 - x));
    if(0 < d && d < D - 1) {
      const int denom2 = max(sm + sp - 2 * minS, 1);
      d = d * 16 + ((sm - sp) * 16 + denom2) / (denom2 * 2);
    } else {
      d *= 16;
    }
    cand[(size_t) y * cols + x + minX1] = d + minD * 16;
  }
}
// left-right check of the candidates against the right view's votes (both roundings of the sub-pixel disparity get a chance) -> int16 map
__global__ __launch_bounds__(256) void sgbm_lr_kernel(const int* __restrict__ cand, const unsigned* __restrict__ votes, int16_t* __restrict__ disp, int rows, int cols,
                                                     int width1, int minD, int minX1, int disp12MaxDiff)
{
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if(x >= cols) return;
  const int INVALID_SCALED = (minD - 1) * 16;
  const size_t row = (size_t) y * cols;
  int d1 = cand[row + x];
  if(d1 != INVALID_SCALED && x >= minX1 && x < minX1 + width1) {
    const int dlo = d1 >> 4, dhi = (d1 + 15) >> 4;
    const int xa = x - dlo, xb = x - dhi;
    auto vote = [&](int xx) -> int {      // disp2ptr[xx]: d + minD of the winning column, or the invalid value (scaled: the original's own mix)
      const unsigned v = votes[row + xx];
      if(v == 0xffffffffu) return INVALID_SCALED;
      const int xw = 65535 - (int) (v & 0xffffu);
      return xw + minX1 - xx;             // x2 = x + minX1 - d - minD  =>  d + minD = x + minX1 - x2
    };
    if(0 <= xa && xa < cols && 0 <= xb && xb < cols) {
      const int va = vote(xa), vb = vote(xb);
      if(va >= minD && abs(va - dlo) > disp12MaxDiff && vb >= minD && abs(vb - dhi) > disp12MaxDiff) d1 = INVALID_SCALED;
    }
  }
  disp[row + x] = (int16_t) d1;
}
__global__ __launch_bounds__(256) void sgbm_init_select_kernel(int* __restrict__ cand, unsigned* __restrict__ votes, size_t npix, int invalid_scaled)
{
  const size_t p = (size_t) blockIdx.x * 256 + threadIdx.x;
  if(p < npix) { cand[p] = invalid_scaled; votes[p] = 0xffffffffu; }
}

// medianBlur(disp, disp, 3) on int16, replicated border; also writes the speckle filter's view (u16, 0 = invalid) when asked
__device__ __forceinline__ void sort2(int& a, int& b) { const int lo = min(a, b), hi = max(a, b); a = lo; b = hi; }
__global__ __launch_bounds__(256) void sgbm_median3_kernel(const int16_t* __restrict__ src, int16_t* __restrict__ dst, uint16_t* __restrict__ shifted, int rows, int cols,
                                                          int invalid_scaled)
{
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if(x >= cols) return;
  int v[9];
  int k = 0;
  if(cols == 1 || rows == 1) {      // medianBlur's one-pixel-wide case: median of three along the long side
    const int len = cols == 1 ? rows : cols, i = cols == 1 ? y : x;
    int a = src[max(i - 1, 0)], b = src[i], c = src[min(i + 1, len - 1)];
    sort2(a, b); sort2(b, c); sort2(a, b);
    dst[i] = (int16_t) b;
    if(shifted) shifted[i] = (uint16_t) (b - invalid_scaled);
    return;
  }
#pragma unroll
  for(int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for(int dx = -1; dx <= 1; ++dx) v[k++] = src[(size_t) min(max(y + dy, 0), rows - 1) * cols + min(max(x + dx, 0), cols - 1)];
  // median of nine by the classic 19-exchange network
  sort2(v[1], v[2]); sort2(v[4], v[5]); sort2(v[7], v[8]); sort2(v[0], v[1]); sort2(v[3], v[4]); sort2(v[6], v[7]);
  sort2(v[1], v[2]); sort2(v[4], v[5]); sort2(v[7], v[8]); sort2(v[0], v[3]); sort2(v[5], v[8]); sort2(v[4], v[7]);
  sort2(v[3], v[6]); sort2(v[1], v[4]); sort2(v[2], v[5]); sort2(v[4], v[7]); sort2(v[4], v[2]); sort2(v[6], v[4]);
  sort2(v[4], v[2]);
  const size_t p = (size_t) y * cols + x;
  dst[p] = (int16_t) v[4];
  if(shifted) shifted[p] = (uint16_t) (v[4] - invalid_scaled);
}
// int16 (or the speckle filter's shifted u16 view) -> float / 16  (disp16.convertTo(CV_32FC1, 1.0 / 16.0), utils/stereo_algorithm.cc:120)
__global__ __launch_bounds__(256) void sgbm_to_float_kernel(const int16_t* __restrict__ d16, const uint16_t* __restrict__ shifted, float* __restrict__ out, size_t npix,
                                                           int invalid_scaled)
{
  const size_t p = (size_t) blockIdx.x * 256 + threadIdx.x;
  if(p >= npix) return;
  const int v = shifted ? (int) shifted[p] + invalid_scaled : (int) d16[p];
  out[p] = (float) v * (1.0f / 16.0f);
}

__global__ __launch_bounds__(256) void sgbm_fill_kernel(int16_t* __restrict__ d, size_t n, int16_t v)
{
  const size_t p = (size_t) blockIdx.x * 256 + threadIdx.x;
  if(p < n) d[p] = v;
}

}  // namespace

// what the kernels serve, and why: false + reason
bool sgbm_serves(const SgbmLaunch& g, const char** why)
{
  auto no = [&](const char* w) { if(why) *why = w; return false; };
  if(g.full_dp) return no("SGBM: fullDP cannot be reached through the reference's StereoSGBM constructor call (utils/stereo_algorithm.cc:30-39) and is not built");
  if(g.min_disp < 0) return no("SGBM: minDisparity >= 0 on the device path");
  if(g.ndisp <= 0 || g.ndisp % 16 || g.ndisp > 256) return no("SGBM: numberOfDisparities must be a positive multiple of 16, <= 256 on the device path");
  const int SW = g.sad_window > 0 ? g.sad_window : 5, SW2 = SW / 2;
  const int width1 = g.cols - (g.min_disp + g.ndisp);
  if(width1 > 0 && width1 <= SW2) return no("SGBM: fewer cost columns than half a SAD window");
  const int ftzero = std::max(g.pre_filter_cap, 15) | 1;
  if(ftzero > 127) return no("SGBM: preFilterCap <= 127 (8-bit planes)");
  const int max_cost = (2 * SW2 + 1) * (2 * SW2 + 1) * (2 * ftzero + 63);
  const int P1 = g.P1 > 0 ? g.P1 : 2, P2 = std::max(g.P2 > 0 ? g.P2 : 5, P1 + 1);
  // int16 buffers: the original wraps (costs) or relies on the values fitting (path costs); inside these limits nothing does
  if(max_cost > 32767) return no("SGBM: SADWindowSize^2 * (2 * max(preFilterCap, 15) + 63) must stay below 2^15 on the device path");
  if(max_cost + P2 >= 32767 || P2 >= 32767) return no("SGBM: P2 + the largest window cost must stay below 2^15 on the device path");
  return true;
}

size_t sgbm_scratch_bytes(int rows, int cols, int min_disp, int D)
{
  const size_t npix = (size_t) rows * cols;
  const int width1 = std::max(cols - (min_disp + D), 1);
  const size_t nvol = (size_t) rows * width1 * D;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  return 2 * up(2 * npix) + up(nvol) + up(nvol * 2) + up(5 * nvol * 2) + 2 * up(npix * 2) + up(npix * 2) + 2 * up(npix * 4);
}

// StereoSGBM::operator() + the conversion of StereoAlgorithm::run, for `nframes` rectified pairs one after the other on the stream
bool launch_stereo_sgbm(hipStream_t s, const SgbmLaunch& g)
{
  if(!sgbm_serves(g, nullptr)) return false;
  const int rows = g.rows, cols = g.cols, D = g.ndisp, minD = g.min_disp, maxD = minD + D;
  const size_t npix = (size_t) rows * cols;
  const int minX1 = maxD, width1 = cols - maxD;
  const int SW = g.sad_window > 0 ? g.sad_window : 5, SW2 = SW / 2;
  const int ftzero = std::max(g.pre_filter_cap, 15) | 1;
  const int uniq = g.uniqueness_ratio >= 0 ? g.uniqueness_ratio : 10;
  const int d12 = g.disp12_max_diff > 0 ? g.disp12_max_diff : 1;
  const int P1 = g.P1 > 0 ? g.P1 : 2, P2 = std::max(g.P2 > 0 ? g.P2 : 5, P1 + 1);
  const int invalid_scaled = (minD - 1) * 16;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t nvol = (size_t) rows * std::max(width1, 1) * D;
  unsigned char* w = (unsigned char*) g.scratch;
  uint8_t* planes_l = w; w += up(2 * npix);
  uint8_t* planes_r = w; w += up(2 * npix);
  uint8_t* pix = w; w += up(nvol);
  int16_t* cost = (int16_t*) w; w += up(nvol * 2);
  int16_t* Lvol = (int16_t*) w; w += up(5 * nvol * 2);
  int16_t* d16a = (int16_t*) w; w += up(npix * 2);
  int16_t* d16b = (int16_t*) w; w += up(npix * 2);
  uint16_t* shifted = (uint16_t*) w; w += up(npix * 2);
  int* lab = (int*) w; w += up(npix * 4);
  int* size = (int*) w; w += up(npix * 4);
  const unsigned nb = (unsigned) ((npix + 255) / 256);
  for(int f = 0; f < g.nframes; ++f) {
    const uint8_t* L = g.left + npix * f;
    const uint8_t* R = g.right + npix * f;
    if(width1 > 0) {
      const dim3 gpl((cols + 255) / 256, rows);
      hipLaunchKernelGGL(sgbm_planes_kernel, gpl, dim3(256), 0, s, L, planes_l, rows, cols, ftzero);
      hipLaunchKernelGGL(sgbm_planes_kernel, gpl, dim3(256), 0, s, R, planes_r, rows, cols, ftzero);
      const size_t lds = sizeof(unsigned) * (size_t) (2 * PC_TX + 2 * (PC_TX + D - 1));
      hipLaunchKernelGGL(sgbm_pixel_cost_kernel, dim3((width1 + PC_TX - 1) / PC_TX, rows), dim3(256), lds, s, planes_l, planes_r, pix, rows, cols, minD, D, minX1, width1);
      {
        uint16_t* hs = reinterpret_cast<uint16_t*>(Lvol);      // (the row sums borrow the first path volume: the scanline kernel overwrites it later)
        const unsigned nbq = (unsigned) (((size_t) rows * width1 * (D / 4) + 255) / 256);
        hipLaunchKernelGGL(sgbm_box_rows_kernel, dim3(nbq), dim3(256), 0, s, pix, hs, rows, width1, D, SW2);
        hipLaunchKernelGGL(sgbm_box_cols_kernel, dim3(nbq), dim3(256), 0, s, hs, cost, rows, width1, D, SW2);
      }
      int lines = 0;
      for(int fam = 0; fam < 5; ++fam) lines += sgbm_family_lines(fam, rows, width1);
      if(D <= 128) hipLaunchKernelGGL(sgbm_path_kernel<1>, dim3((unsigned) lines), dim3(64), 0, s, cost, Lvol, rows, width1, D, P1, P2);
      else hipLaunchKernelGGL(sgbm_path_kernel<2>, dim3((unsigned) lines), dim3(64), 0, s, cost, Lvol, rows, width1, D, P1, P2);
      // (candidates and votes borrow the speckle filter's label / size planes: they are consumed before that filter runs)
      int* cand = lab;
      unsigned* votes = reinterpret_cast<unsigned*>(size);
      hipLaunchKernelGGL(sgbm_init_select_kernel, dim3(nb), dim3(256), 0, s, cand, votes, npix, invalid_scaled);
      const dim3 gw((unsigned) (((size_t) rows * width1 + 3) / 4));
      if(D <= 64) hipLaunchKernelGGL(sgbm_wta_kernel<1>, gw, dim3(256), 0, s, Lvol, cand, votes, rows, cols, width1, D, minD, minX1, uniq);
      else if(D <= 128) hipLaunchKernelGGL(sgbm_wta_kernel<2>, gw, dim3(256), 0, s, Lvol, cand, votes, rows, cols, width1, D, minD, minX1, uniq);
      else hipLaunchKernelGGL(sgbm_wta_kernel<4>, gw, dim3(256), 0, s, Lvol, cand, votes, rows, cols, width1, D, minD, minX1, uniq);
      hipLaunchKernelGGL(sgbm_lr_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, s, cand, votes, d16a, rows, cols, width1, minD, minX1, d12);
    } else {
      // no cost column at all (the image is narrower than the disparity range): every pixel invalid
      hipLaunchKernelGGL(sgbm_fill_kernel, dim3(nb), dim3(256), 0, s, d16a, npix, (int16_t) invalid_scaled);
    }
    const bool speckle = g.speckle_window > 0;
    hipLaunchKernelGGL(sgbm_median3_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, s, d16a, d16b, speckle ? shifted : (uint16_t*) nullptr, rows, cols, invalid_scaled);
    if(speckle) {
      // filterSpeckles(disp, (minDisparity - 1) * 16, speckleWindowSize, 16 * speckleRange): on the map shifted so that invalid = 0
      launch_speckle_filter_u16(s, shifted, lab, size, rows, cols, 16 * g.speckle_range, g.speckle_window);
      hipLaunchKernelGGL(sgbm_to_float_kernel, dim3(nb), dim3(256), 0, s, d16b, shifted, g.disp + npix * f, npix, invalid_scaled);
    } else {
      hipLaunchKernelGGL(sgbm_to_float_kernel, dim3(nb), dim3(256), 0, s, d16b, (const uint16_t*) nullptr, g.disp + npix * f, npix, invalid_scaled);
    }
  }
  return true;
}

}  // namespace bpvo_hip
