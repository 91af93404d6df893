// ORACLE — TEST INFRASTRUCTURE ONLY (see orc.h).
// Stereo front-end, third matcher: `StereoAlgorithm = SGBM | SemiGlobalBlockMatching` (utils/stereo_algorithm.cc:25-40 construction,
// :113-121 run; selected by conf/kitti_seq_0.cfg:6): cv::StereoSGBM of OpenCV 2.4, then medianBlur(3) (inside its operator()), the
// optional filterSpeckles, and disp16.convertTo(CV_32F, 1/16).
// The matcher lives in OpenCV 2.4 (modules/calib3d/src/stereosgbm.cpp: calcPixelCostBT, computeDisparitySGBM, filterSpeckles;
// modules/imgproc/src/smooth.cpp: medianBlur on CV_16S), a third-party dependency ABSENT from /root/reference and from this image:
// PARITY UNPINNED.  This file restates the published algorithm of OpenCV 2.4.x [ext], the scalar branches (the SSE2 branches compute the
// same numbers while nothing leaves the int16 range); tests/test_stereo.py checks it against a numpy evaluation of the same definition.
//
// What the reference contributes is the constructor call, which passes NINE positional arguments to a constructor of eleven
//     StereoSGBM(minDisparity, numDisparities, SADWindowSize, P1 = 0, P2 = 0, disp12MaxDiff = 0, preFilterCap = 0, uniquenessRatio = 0,
//                speckleWindowSize = 0, speckleRange = 0, fullDP = false)
// so that the config keys land one slot off (Q22): `uniquenessRatio` -> disp12MaxDiff, `speckleWindowSize` -> preFilterCap,
// `speckleRange` -> uniquenessRatio, `(bool) fullDP` -> speckleWindowSize (0 or 1); speckleRange and fullDP keep their defaults: the
// two-pass mode cannot be reached through the reference, and it is not restated here (SgbmParams::fullDP must be 0).  The mapping is the
// caller's (include/bpvo_hip/vo.hpp StereoParameters::fromReferenceConfigSGBM); this file takes the cv::StereoSGBM fields.
//
// Facts of the 2.4 code that a reader of the paper would not guess, all restated:
//   * matching costs exist for the columns x in [maxD, width) only (width1 = width - maxD of them, minDisparity >= 0); the window sums clamp
//     to that range and to the image rows;
//   * for y > 0 the cost of column minX1 (x = 0 of the cost buffer) is never updated: it keeps the value of row 0;
//   * for the last SADWindowSize / 2 rows the cost buffer is not updated at all: they reuse the costs of row height - 1 - SADWindowSize / 2;
//   * a path that enters from outside the cost buffer starts from L = 0, min L = 0: its first cost is C - P2;
//   * both planes of calcPixelCostBT (x-Sobel through the clip table, raw intensity) have their first and last column set to tab[0];
//   * disp12MaxDiff <= 0 means 1, uniquenessRatio < 0 means 10, P1 <= 0 means 2, P2 <= 0 means 5, then P2 = max(P2, P1 + 1).
#include "orc.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace orc {

typedef int16_t CostType;
static const int kDispShift = 4, kDispScale = 16;
static const int kMaxCost = SHRT_MAX;

static inline CostType sat16(int v) { return (CostType) std::min(std::max(v, (int) SHRT_MIN), (int) SHRT_MAX); }

// calcPixelCostBT for image row y: cost[(x - minX1) * D + (d - minD)] += BT(sobel plane) + (BT(raw plane) >> 2)
static void pixelCostBT(const uint8_t* img1, const uint8_t* img2, int rows, int width, int y, int minD, int maxD, int ftzero, CostType* cost)
{
  const int D = maxD - minD, minX1 = std::max(maxD, 0), maxX1 = width + std::min(minD, 0), width1 = maxX1 - minX1;
  auto tab = [&](int v) { return std::min(std::max(v, -ftzero), ftzero) + ftzero; };      // clipTab[v + TAB_OFS]
  std::vector<int> p1(2 * (size_t) width), p2(2 * (size_t) width);                        // [plane][real column] (the original mirrors p2)
  const uint8_t* r1 = img1 + (size_t) y * width;
  const uint8_t* r2 = img2 + (size_t) y * width;
  const int n = y > 0 ? -width : 0, s = y < rows - 1 ? width : 0;
  for(int c = 0; c < 2; ++c) p1[c * width] = p1[c * width + width - 1] = p2[c * width] = p2[c * width + width - 1] = tab(0);
  for(int x = 1; x < width - 1; ++x) {
    p1[x] = tab((r1[x + 1] - r1[x - 1]) * 2 + r1[x + n + 1] - r1[x + n - 1] + r1[x + s + 1] - r1[x + s - 1]);
    p2[x] = tab((r2[x + 1] - r2[x - 1]) * 2 + r2[x + n + 1] - r2[x + n - 1] + r2[x + s + 1] - r2[x + s - 1]);
    p1[width + x] = r1[x];
    p2[width + x] = r2[x];
  }
  std::memset(cost, 0, sizeof(CostType) * (size_t) width1 * D);
  std::vector<int> v0(width), v1(width);
  for(int c = 0; c < 2; ++c) {
    const int* q1 = p1.data() + c * width;
    const int* q2 = p2.data() + c * width;
    const int diff_scale = c == 0 ? 0 : 2;
    for(int x = 0; x < width; ++x) {          // half-sample interval of the right image around column x
      const int v = q2[x];
      const int vl = x < width - 1 ? (v + q2[x + 1]) / 2 : v;      // (mirrored storage: the original's x - 1 is the real x + 1; min / max are symmetric)
      const int vr = x > 0 ? (v + q2[x - 1]) / 2 : v;
      v0[x] = std::min(std::min(vl, vr), v);
      v1[x] = std::max(std::max(vl, vr), v);
    }
    for(int x = minX1; x < maxX1; ++x) {
      const int u = q1[x];
      const int ul = x > 0 ? (u + q1[x - 1]) / 2 : u;
      const int ur = x < width - 1 ? (u + q1[x + 1]) / 2 : u;
      const int u0 = std::min(std::min(ul, ur), u), u1 = std::max(std::max(ul, ur), u);
      for(int d = minD; d < maxD; ++d) {
        const int xr = x - d;
        const int v = q2[xr];
        const int c0 = std::max(std::max(0, u - v1[xr]), v0[xr] - u);
        const int c1 = std::max(std::max(0, v - u1), u0 - v);
        CostType& dst = cost[(size_t) (x - minX1) * D + (d - minD)];
        dst = (CostType) (dst + (std::min(c0, c1) >> diff_scale));
      }
    }
  }
}

// medianBlur(disp, disp, 3) on CV_16S: 3 x 3 median, replicated border (smooth.cpp medianBlur_SortNet)
static void medianBlur3(std::vector<int16_t>& img, int rows, int cols)
{
  if(cols == 1 || rows == 1) {
    // the one-pixel-wide special case of medianBlur_SortNet: median of (prev, cur, next) along the long side, ends replicated
    const int len = cols == 1 ? rows : cols;
    std::vector<int16_t> src(img);
    for(int i = 0; i < len; ++i) {
      int a = src[std::max(i - 1, 0)], b = src[i], c = src[std::min(i + 1, len - 1)];
      if(a > b) std::swap(a, b);
      if(b > c) std::swap(b, c);
      if(a > b) std::swap(a, b);
      img[i] = (int16_t) b;
    }
    return;
  }
  std::vector<int16_t> src(img);
  for(int y = 0; y < rows; ++y)
    for(int x = 0; x < cols; ++x) {
      int16_t v[9];
      int k = 0;
      for(int dy = -1; dy <= 1; ++dy)
        for(int dx = -1; dx <= 1; ++dx)
          v[k++] = src[(size_t) std::min(std::max(y + dy, 0), rows - 1) * cols + std::min(std::max(x + dx, 0), cols - 1)];
      std::nth_element(v, v + 4, v + 9);
      img[(size_t) y * cols + x] = v[4];
    }
}

// cv::filterSpeckles(img, newVal, maxSpeckleSize, maxDiff): 4-connected regions (neighbours both != newVal, |difference| <= maxDiff) of
// at most maxSpeckleSize pixels are set to newVal
static void filterSpeckles16(std::vector<int16_t>& img, int rows, int cols, int newVal, int maxSpeckleSize, int maxDiff)
{
  const size_t npix = (size_t) rows * cols;
  std::vector<int> labels(npix, 0);
  std::vector<int> stack(npix);
  std::vector<uint8_t> rtype(npix + 1, 0);
  int curlabel = 0;
  for(int i = 0; i < rows; ++i)
    for(int j = 0; j < cols; ++j) {
      const size_t at = (size_t) i * cols + j;
      if(img[at] == newVal) continue;
      if(labels[at]) {
        if(rtype[labels[at]]) img[at] = (int16_t) newVal;
        continue;
      }
      ++curlabel;
      labels[at] = curlabel;
      size_t top = 0;
      size_t p = at;
      int count = 0;
      for(;;) {
        ++count;
        const int px = (int) (p % cols), py = (int) (p / cols);
        const int dp = img[p];
        auto visit = [&](size_t q) {
          if(!labels[q] && img[q] != newVal && std::abs(dp - img[q]) <= maxDiff) { labels[q] = curlabel; stack[top++] = q; }
        };
        if(px < cols - 1) visit(p + 1);
        if(px > 0) visit(p - 1);
        if(py < rows - 1) visit(p + cols);
        if(py > 0) visit(p - cols);
        if(top == 0) break;
        p = stack[--top];
      }
      if(count <= maxSpeckleSize) { rtype[curlabel] = 1; img[at] = (int16_t) newVal; }
      else rtype[curlabel] = 0;
    }
}

// computeDisparitySGBM (single pass: 5 directions) -> disp (CV_16S, 4 fractional bits)
static bool computeDisparitySGBM(const uint8_t* img1, const uint8_t* img2, int rows, int width, const SgbmParams& sp, std::vector<int16_t>& disp)
{
  const int minD = sp.minDisparity, maxD = minD + sp.numberOfDisparities;
  const int SW = sp.SADWindowSize > 0 ? sp.SADWindowSize : 5;
  const int ftzero = std::max(sp.preFilterCap, 15) | 1;
  const int uniquenessRatio = sp.uniquenessRatio >= 0 ? sp.uniquenessRatio : 10;
  const int disp12MaxDiff = sp.disp12MaxDiff > 0 ? sp.disp12MaxDiff : 1;
  const int P1 = sp.P1 > 0 ? sp.P1 : 2, P2 = std::max(sp.P2 > 0 ? sp.P2 : 5, P1 + 1);
  const int height = rows;
  const int minX1 = std::max(maxD, 0), maxX1 = width + std::min(minD, 0);
  const int D = maxD - minD, width1 = maxX1 - minX1;
  const int INVALID_DISP = minD - 1, INVALID_DISP_SCALED = INVALID_DISP * kDispScale;
  const int SW2 = SW / 2, SH2 = SW / 2;
  disp.assign((size_t) rows * width, (int16_t) INVALID_DISP_SCALED);
  if(minX1 >= maxX1) return true;
  if(D % 16 != 0 || D <= 0) return false;
  if(width1 <= SW2) return false;       // (the original indexes pixDiff[SW2 * D] without a check)

  const size_t rowsz = (size_t) width1 * D;
  // hsum[k]: horizontal window sums of image row k (clamped to the cost columns), int16 like the original's buffers
  std::vector<CostType> pix(rowsz);
  std::vector<std::vector<CostType>> hsum((size_t) height);
  auto hsum_row = [&](int k) -> const std::vector<CostType>& {
    std::vector<CostType>& h = hsum[(size_t) k];
    if(!h.empty()) return h;
    h.assign(rowsz, 0);
    pixelCostBT(img1, img2, rows, width, k, minD, maxD, ftzero, pix.data());
    for(int d = 0; d < D; ++d) {
      int acc = 0;
      for(int x = 0; x <= SW2; ++x) acc = (CostType) (acc + pix[(size_t) x * D + d] * (x == 0 ? SW2 + 1 : 1));
      h[d] = (CostType) acc;
      for(int x = 1; x < width1; ++x) {
        const int add = pix[(size_t) std::min(x + SW2, width1 - 1) * D + d], sub = pix[(size_t) std::max(x - SW2 - 1, 0) * D + d];
        h[(size_t) x * D + d] = (CostType) (h[(size_t) (x - 1) * D + d] + add - sub);
      }
    }
    return h;
  };

  std::vector<CostType> C(rowsz, 0), S(rowsz);
  // L_r of the current and the previous row, directions 0..3, one cell of border left and right, D + 2 disparities (the sentinels)
  const int D2 = D + 2;
  const size_t cell = (size_t) 4 * D2;
  std::vector<CostType> LrA((size_t) (width1 + 2) * cell, 0), LrB((size_t) (width1 + 2) * cell, 0);
  std::vector<CostType> mnA((size_t) (width1 + 2) * 4, 0), mnB((size_t) (width1 + 2) * 4, 0);
  CostType* Lr[2] = {LrA.data(), LrB.data()};
  CostType* minLr[2] = {mnA.data(), mnB.data()};
  auto L = [&](int k, int x, int r) { return Lr[k] + (size_t) (x + 1) * cell + (size_t) r * D2 + 1; };       // [-1 .. D]
  auto M = [&](int k, int x, int r) -> CostType& { return minLr[k][(size_t) (x + 1) * 4 + r]; };

  std::vector<int16_t> disp2(width);
  std::vector<CostType> disp2cost(width);
  for(int y = 0; y < height; ++y) {
    // ---- matching cost of the row (SAD window over the Birchfield-Tomasi pixel costs)
    if(y == 0) {
      for(int k = 0; k <= SH2; ++k) {
        const std::vector<CostType>& h = hsum_row(std::min(k, height - 1));
        const int scale = k == 0 ? SH2 + 1 : 1;
        for(size_t i = 0; i < rowsz; ++i) C[i] = (CostType) (C[i] + h[i] * scale);
      }
    } else {
      const int k = y + SH2;
      if(k < height) {
        const std::vector<CostType>& add = hsum_row(k);
        const std::vector<CostType>& sub = hsum_row(std::max(y - SH2 - 1, 0));
        for(size_t i = (size_t) D; i < rowsz; ++i) C[i] = (CostType) (C[i] + add[i] - sub[i]);      // (column 0 of the buffer keeps row 0's cost)
      }                                                                                             // (k >= height: the buffer is left as it is)
    }
    if(y - SH2 - 2 >= 0) std::vector<CostType>().swap(hsum[(size_t) (y - SH2 - 2)]);               // (no longer needed)
    std::fill(S.begin(), S.end(), (CostType) 0);

    // ---- borders of the current row's L_r: zero (a path that enters from outside starts at L = 0, min L = 0)
    for(int r = 0; r < 4; ++r) {
      std::fill(L(0, -1, r) - 1, L(0, -1, r) - 1 + D2, (CostType) 0);
      std::fill(L(0, width1, r) - 1, L(0, width1, r) - 1 + D2, (CostType) 0);
      M(0, -1, r) = 0; M(0, width1, r) = 0;
    }
    // ---- directions 0 (from x - 1), 1 (x - 1, y - 1), 2 (x, y - 1), 3 (x + 1, y - 1)
    for(int x = 0; x < width1; ++x) {
      const int delta[4] = {M(0, x - 1, 0) + P2, M(1, x - 1, 1) + P2, M(1, x, 2) + P2, M(1, x + 1, 3) + P2};
      CostType* Lp[4] = {L(0, x - 1, 0), L(1, x - 1, 1), L(1, x, 2), L(1, x + 1, 3)};
      for(int r = 0; r < 4; ++r) Lp[r][-1] = Lp[r][D] = (CostType) kMaxCost;
      int minL[4] = {kMaxCost, kMaxCost, kMaxCost, kMaxCost};
      const CostType* Cp = C.data() + (size_t) x * D;
      CostType* Sp = S.data() + (size_t) x * D;
      for(int d = 0; d < D; ++d) {
        const int Cpd = Cp[d];
        int sum = Sp[d];
        for(int r = 0; r < 4; ++r) {
          const int Lv = Cpd + std::min((int) Lp[r][d], std::min(Lp[r][d - 1] + P1, std::min(Lp[r][d + 1] + P1, delta[r]))) - delta[r];
          L(0, x, r)[d] = (CostType) Lv;
          minL[r] = std::min(minL[r], Lv);
          sum += Lv;
        }
        Sp[d] = sat16(sum);
      }
      for(int r = 0; r < 4; ++r) M(0, x, r) = (CostType) minL[r];
    }

    // ---- direction 4 (from x + 1), winner takes all, uniqueness, sub-pixel, the right image's view
    int16_t* d1 = disp.data() + (size_t) y * width;
    for(int x = 0; x < width; ++x) { d1[x] = (int16_t) INVALID_DISP_SCALED; disp2[x] = (int16_t) INVALID_DISP_SCALED; disp2cost[x] = (CostType) kMaxCost; }
    for(int x = width1 - 1; x >= 0; --x) {
      CostType* Sp = S.data() + (size_t) x * D;
      int minS = kMaxCost, bestDisp = -1;
      {
        const int delta0 = M(0, x + 1, 0) + P2;
        CostType* Lp0 = L(0, x + 1, 0);
        Lp0[-1] = Lp0[D] = (CostType) kMaxCost;
        CostType* Lx = L(0, x, 0);
        const CostType* Cp = C.data() + (size_t) x * D;
        int minL0 = kMaxCost;
        for(int d = 0; d < D; ++d) {
          const int L0 = Cp[d] + std::min((int) Lp0[d], std::min(Lp0[d - 1] + P1, std::min(Lp0[d + 1] + P1, delta0))) - delta0;
          Lx[d] = (CostType) L0;
          minL0 = std::min(minL0, L0);
          const int Sval = Sp[d] = sat16(Sp[d] + L0);
          if(Sval < minS) { minS = Sval; bestDisp = d; }
        }
        M(0, x, 0) = (CostType) minL0;
      }
      int d;
      for(d = 0; d < D; ++d)
        if(Sp[d] * (100 - uniquenessRatio) < minS * 100 && std::abs(bestDisp - d) > 1) break;
      if(d < D) continue;
      d = bestDisp;
      const int x2 = x + minX1 - d - minD;
      if(disp2cost[x2] > minS) { disp2cost[x2] = (CostType) minS; disp2[x2] = (int16_t) (d + minD); }
      if(0 < d && d < D - 1) {
        const int denom2 = std::max(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
        d = d * kDispScale + ((Sp[d - 1] - Sp[d + 1]) * kDispScale + denom2) / (denom2 * 2);
      } else {
        d *= kDispScale;
      }
      d1[x + minX1] = (int16_t) (d + minD * kDispScale);
    }
    // ---- left-right check against the right image's view (both roundings of the sub-pixel disparity get a chance)
    for(int x = minX1; x < maxX1; ++x) {
      const int dd = d1[x];
      if(dd == INVALID_DISP_SCALED) continue;
      const int _d = dd >> kDispShift, d_ = (dd + kDispScale - 1) >> kDispShift;
      const int _x = x - _d, x_ = x - d_;
      if(0 <= _x && _x < width && disp2[_x] >= minD && std::abs(disp2[_x] - _d) > disp12MaxDiff &&
         0 <= x_ && x_ < width && disp2[x_] >= minD && std::abs(disp2[x_] - d_) > disp12MaxDiff)
        d1[x] = (int16_t) INVALID_DISP_SCALED;
    }
    std::swap(Lr[0], Lr[1]);
    std::swap(minLr[0], minLr[1]);
  }
  return true;
}

// StereoAlgorithm::run, SemiGlobalBlockMatching branch (utils/stereo_algorithm.cc:113-121): StereoSGBM::operator() = computeDisparitySGBM +
// medianBlur(3) + filterSpeckles if speckleWindowSize > 0; then convertTo(CV_32FC1, 1/16)
bool stereoSGBM(const uint8_t* left, const uint8_t* right, int rows, int cols, const SgbmParams& sp, float* dmap)
{
  if(sp.fullDP) return false;                       // unreachable through the reference (header)
  if(sp.minDisparity < 0) return false;             // (the cost-column bookkeeping above is written for minDisparity >= 0)
  if(sp.numberOfDisparities <= 0 || sp.numberOfDisparities % 16 != 0) return false;
  std::vector<int16_t> d16;
  if(!computeDisparitySGBM(left, right, rows, cols, sp, d16)) return false;
  medianBlur3(d16, rows, cols);
  if(sp.speckleWindowSize > 0)
    filterSpeckles16(d16, rows, cols, (sp.minDisparity - 1) * kDispScale, sp.speckleWindowSize, kDispScale * sp.speckleRange);
  for(size_t i = 0; i < d16.size(); ++i) dmap[i] = (float) d16[i] * (1.0f / 16.0f);
  return true;
}

}  // namespace orc
