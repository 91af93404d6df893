// ORACLE — TEST INFRASTRUCTURE ONLY (see orc.h).
// Stereo front-end, second matcher: the reference's in-tree semi-global matcher `SgmStereo` (utils/sgm.{h,cc}; selected by
// `StereoAlgorithm = SGM`, utils/stereo_algorithm.cc:42-59,127-133, conf/kitti_eval.cfg:27, conf/kitti_stereo.cfg:5).  Its
// source IS in the reference tree (GPL code of the SPS-stereo library, compiled only WITH_GPL_CODE), but it includes OpenCV and
// bpvo/types.h (Eigen), so it cannot be built here: PARITY UNPINNED, like the rest of the oracle.  Everything in it is integer
// arithmetic (two double expressions: the census weight and the sub-pixel step, both restated operation for operation), so this
// restatement — scalar, written from the definitions rather than from the SSE code — has one right answer per input; every block
// cites the lines it follows, including the places where the original's behaviour is an accident of its loops (S1 - S6 below).
//
//   S1  the last `windowRadius` rows of the cost image are never written and stay 0 (calcRowCosts only produces row y when
//       y + windowRadius < height, utils/sgm.cc:517-526,579), and for y >= 1 column x = 0 is never written either (:533)
//   S2  costs at disparities d > x repeat the cost at d = x (:607-609,659-661), the Hamming part the weighted distance at d = x
//       (:709-712)
//   S3  the right Sobel image is stored mirrored (:398-409) and read at width - 1 - x + d (:601,677)
//   S4  path costs subtract (previous minimum + P2), not the previous minimum (:813-816): they go negative; all int16
//       saturating arithmetic
//   S5  sub-pixel step (:859-869): a zero denominator (right neighbour equal to the centre) makes the double expression -inf or
//       NaN, static_cast<int> of which is INT_MIN on x86 and the stored unsigned short 0, i.e. "invalid"
//   S6  only the LEFT disparity image is returned; the right one serves the left-right check (:269-283,963-980)
#include "orc.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace orc {

namespace {

inline int16_t adds16(int a, int b) { return (int16_t) std::min(32767, std::max(-32768, a + b)); }     // _mm_adds_epi16
inline int16_t subs16(int a, int b) { return (int16_t) std::min(32767, std::max(-32768, a - b)); }     // _mm_subs_epi16

struct Sgm {
  int W, H, D, cap, crad, wrad, P1, P2, lr_thresh;
  double factor, cweight;
  int pitch;     // widthStep_ (:256)

  // computeCappedSobelImage (:377-413): interior pixels only, everything else = cap; the right image mirrored (S3)
  void sobel(const uint8_t* img, bool flip, std::vector<uint8_t>& out) const
  {
    out.assign((size_t) pitch * H, (uint8_t) cap);
    for(int y = 1; y < H - 1; ++y)
      for(int x = 1; x < W - 1; ++x) {
        int v = (img[W * (y - 1) + x + 1] + 2 * img[W * y + x + 1] + img[W * (y + 1) + x + 1]) -
                (img[W * (y - 1) + x - 1] + 2 * img[W * y + x - 1] + img[W * (y + 1) + x - 1]);
        v = v > cap ? 2 * cap : (v < -cap ? 0 : v + cap);
        out[(size_t) pitch * y + (flip ? W - x - 1 : x)] = (uint8_t) v;
      }
  }

  // computeCensusImage (:415-434): (2r+1)^2 bits, row-major, neighbour >= centre, outside the image 0
  void census(const uint8_t* img, std::vector<int>& out) const
  {
    out.resize((size_t) W * H);
    for(int y = 0; y < H; ++y)
      for(int x = 0; x < W; ++x) {
        const uint8_t c = img[W * y + x];
        int code = 0;
        for(int oy = -crad; oy <= crad; ++oy)
          for(int ox = -crad; ox <= crad; ++ox) {
            code <<= 1;
            if(y + oy >= 0 && y + oy < H && x + ox >= 0 && x + ox < W && img[W * (y + oy) + x + ox] >= c) code += 1;
          }
        out[(size_t) W * y + x] = code;
      }
  }

  // calcHalfPixelRight / the left half-pixel values of calcPixelwiseSAD (:581-594,684-699): min / max of the centre and the two
  // (truncating) half-way values to the neighbours
  static void half_minmax(const uint8_t* row, int x, int W, int& mn, int& mx)
  {
    const int c = row[x];
    const int l = x > 0 ? (c + row[x - 1]) / 2 : c;
    const int r = x < W - 1 ? (c + row[x + 1]) / 2 : c;
    mn = std::min(std::min(l, r), c);
    mx = std::max(std::max(l, r), c);
  }

  // calcPixelwiseSAD + addPixelwiseHamming (:581-733) for one image row: u8 cost [W][D]
  void pixelwise(const uint8_t* lrow, const uint8_t* rrow_flipped, const int* lcen, const int* rcen, std::vector<uint8_t>& cost) const
  {
    cost.resize((size_t) W * D);
    std::vector<uint8_t> rmin(W), rmax(W);
    for(int x = 0; x < W; ++x) { int a, b; half_minmax(rrow_flipped, x, W, a, b); rmin[x] = (uint8_t) a; rmax[x] = (uint8_t) b; }
    for(int x = 0; x < W; ++x) {
      const int lc = lrow[x];
      int lmin, lmax;
      half_minmax(lrow, x, W, lmin, lmax);
      uint8_t* c = cost.data() + (size_t) D * x;
      const int dmax = std::min(x, D - 1);
      for(int d = 0; d <= dmax; ++d) {
        const int ri = W - 1 - x + d;                                    // the mirrored right row: original column x - d
        const int rc = rrow_flipped[ri];
        int l2r = std::max(0, lc - (int) rmax[ri]);
        l2r = std::max(l2r, (int) rmin[ri] - lc);
        int r2l = std::max(0, rc - lmax);
        r2l = std::max(r2l, lmin - rc);
        c[d] = (uint8_t) std::min(l2r, r2l);
      }
      for(int d = dmax + 1; d < D; ++d) c[d] = c[d - 1];                 // S2
      // Hamming distance of the census codes, weighted in double and truncated, added in u8 (:701-733)
      int ham = 0;
      for(int d = 0; d <= dmax; ++d) {
        ham = __builtin_popcount((unsigned) (lcen[x] ^ rcen[x - d]));
        c[d] = (uint8_t) (c[d] + (uint8_t) (ham * cweight));
      }
      const int last = (uint8_t) (ham * cweight);
      for(int d = dmax + 1; d < D; ++d) c[d] = (uint8_t) (c[d] + last);  // S2
    }
  }

  // one row of the horizontal box sums (:444-465,527-563): window [x - r, x + r], columns clamped to the image, u16
  void row_box(const std::vector<uint8_t>& pix, std::vector<uint16_t>& agg) const
  {
    agg.assign((size_t) W * D, 0);
    for(int x = 0; x <= wrad; ++x) {
      const int scale = x == 0 ? wrad + 1 : 1;
      for(int d = 0; d < D; ++d) agg[d] = (uint16_t) (agg[d] + (uint16_t) (pix[(size_t) D * std::min(x, W - 1) + d] * scale));
    }
    for(int x = 1; x < W; ++x) {
      const uint8_t* add = pix.data() + (size_t) std::min(x + wrad, W - 1) * D;
      const uint8_t* sub = pix.data() + (size_t) std::max(x - wrad - 1, 0) * D;
      for(int d = 0; d < D; ++d) agg[(size_t) D * x + d] = (uint16_t) adds16(subs16((int16_t) agg[(size_t) D * (x - 1) + d], sub[d]), add[d]);
    }
  }

  // computeLeftCostImage (:345-375) = calcTopRowCost (:436-484) + calcRowCosts (:486-579): (2r+1)^2 box sums of the pixel-wise cost,
  // rows and columns clamped; S1
  void left_cost(const uint8_t* L, const uint8_t* R, std::vector<uint16_t>& cost) const
  {
    std::vector<uint8_t> ls, rs;
    std::vector<int> lc, rc;
    sobel(L, false, ls);
    sobel(R, true, rs);
    census(L, lc);
    census(R, rc);
    cost.assign((size_t) W * H * D, 0);
    const size_t RS = (size_t) W * D;
    std::vector<std::vector<uint16_t>> agg(H);        // (the original keeps a ring of 2r + 2 rows)
    std::vector<uint8_t> pix;
    auto row_agg = [&](int y) {
      if(!agg[y].empty()) return;
      pixelwise(ls.data() + (size_t) pitch * y, rs.data() + (size_t) pitch * y, lc.data() + (size_t) W * y, rc.data() + (size_t) W * y, pix);
      row_box(pix, agg[y]);
    };
    // top row: (r + 1) x row 0 + rows 1 .. r (each clamped to the last row)
    for(int i = 0; i <= wrad; ++i) {
      const int y = std::min(i, H - 1);
      if(i > 0 && y != i) {          // min(rowIndex, height - 1): a clamped index recomputes into the same ring slot; its row pointer has
        // moved on, though — images of fewer than r + 1 rows are not a case the reference can see (KITTI, Tsukuba); refuse below
      }
      row_agg(y);
      const int scale = i == 0 ? wrad + 1 : 1;
      for(size_t k = 0; k < RS; ++k) cost[k] = (uint16_t) (cost[k] + agg[y][k] * scale);
    }
    // rows 1 ..: previous row - box row (y - r - 1, clamped) + box row (y + r), only while y + r < H (S1); column 0 untouched (S1)
    for(int y = 1; y < H; ++y) {
      if(y + wrad >= H) continue;
      row_agg(y + wrad);
      const std::vector<uint16_t>& add = agg[y + wrad];
      const std::vector<uint16_t>& sub = agg[std::max(y - wrad - 1, 0)];
      uint16_t* cur = cost.data() + RS * y;
      const uint16_t* prev = cur - RS;
      for(int x = 1; x < W; ++x)
        for(int d = 0; d < D; ++d) {
          const size_t k = (size_t) D * x + d;
          cur[k] = (uint16_t) adds16(subs16((int16_t) prev[k], (int16_t) sub[k]), (int16_t) add[k]);
        }
    }
  }

  // computeRightCostImage (:735-775): right(x, d) = left(x + d, d); past the image the last valid disparity's cost repeats
  void right_cost(const std::vector<uint16_t>& lcost, std::vector<uint16_t>& rcost) const
  {
    rcost.assign((size_t) W * H * D, 0);
    for(int y = 0; y < H; ++y) {
      const uint16_t* lrow = lcost.data() + (size_t) W * D * y;
      uint16_t* rrow = rcost.data() + (size_t) W * D * y;
      for(int x = 0; x < W; ++x)
        for(int d = 0; d <= std::min(x, D - 1); ++d) rrow[(size_t) D * (x - d) + d] = lrow[(size_t) D * x + d];
      for(int x = std::max(W - D + 1, 0); x < W; ++x) {
        const int maxd = W - x;
        const uint16_t last = rrow[(size_t) D * x + maxd - 1];
        for(int d = maxd; d < D; ++d) rrow[(size_t) D * x + d] = last;
      }
    }
  }

  // performSGM (:747-899): two sweeps (top-left to bottom-right, then back), in each one path along the row (from the previous x) and
  // one along the column (from the previous row): four directions in all.  L(p, d) = min(L'(d), L'(d-1) + P1, L'(d+1) + P1,
  // min L' + P2) - (min L' + P2) + C(p, d) in saturating int16 (S4); the sum of the four goes through winner-takes-all with a
  // parabola-like sub-pixel step (:841-872), then speckleFilter(100, 2 * factor) (:898).
  void aggregate(const std::vector<uint16_t>& cost, std::vector<uint16_t>& disp) const
  {
    std::vector<int16_t> sum((size_t) W * H * D, 0);
    disp.assign((size_t) W * H, 0);
    const int16_t kMax = SHRT_MAX;
    for(int pass = 0; pass < 2; ++pass) {
      const int step = pass == 0 ? 1 : -1;
      // path costs of the previous pixel of the row (path 0) and of the previous row (path 2, per column); before the first pixel /
      // row they are all zero, and so are their minima (:772-779,787-790)
      std::vector<int16_t> up((size_t) W * D, 0), up_new((size_t) W * D);
      std::vector<int16_t> up_min(W, 0), up_min_new(W);
      std::vector<int16_t> left(D), cur0(D), cur2(D);
      for(int yi = 0; yi < H; ++yi) {
        const int y = pass == 0 ? yi : H - 1 - yi;
        std::fill(left.begin(), left.end(), (int16_t) 0);
        int left_min = 0;
        for(int xi = 0; xi < W; ++xi) {
          const int x = pass == 0 ? xi : W - 1 - xi;
          const uint16_t* C = cost.data() + ((size_t) W * y + x) * D;
          int16_t* S = sum.data() + ((size_t) W * y + x) * D;
          const int16_t pm0 = (int16_t) (left_min + P2), pm2 = (int16_t) (up_min[x] + P2);
          const int16_t* u = up.data() + (size_t) D * x;
          int16_t n0 = kMax, n2 = kMax;
          for(int d = 0; d < D; ++d) {
            const int16_t l_m = d > 0 ? left[d - 1] : kMax, l_p = d < D - 1 ? left[d + 1] : kMax;
            const int16_t u_m = d > 0 ? u[d - 1] : kMax, u_p = d < D - 1 ? u[d + 1] : kMax;
            int16_t a = std::min(left[d], adds16(l_m, P1));
            a = std::min(a, adds16(l_p, P1));
            a = std::min(a, pm0);
            a = adds16(subs16(a, pm0), (int16_t) C[d]);
            int16_t b = std::min(u[d], adds16(u_m, P1));
            b = std::min(b, adds16(u_p, P1));
            b = std::min(b, pm2);
            b = adds16(subs16(b, pm2), (int16_t) C[d]);
            cur0[d] = a; cur2[d] = b;
            n0 = std::min(n0, a); n2 = std::min(n2, b);
            S[d] = adds16(adds16(S[d], a), b);
          }
          std::copy(cur0.begin(), cur0.end(), left.begin());
          left_min = n0;
          std::copy(cur2.begin(), cur2.end(), up_new.begin() + (size_t) D * x);
          up_min_new[x] = n2;
        }
        up.swap(up_new);
        up_min.swap(up_min_new);
        (void) step;
        if(pass == 1) {
          for(int x = 0; x < W; ++x) {
            const int16_t* S = sum.data() + ((size_t) W * y + x) * D;
            int best = S[0], bd = 0;
            for(int d = 1; d < D; ++d)
              if(S[d] < best) { best = S[d]; bd = d; }
            int out;
            if(bd > 0 && bd < D - 1) {
              const int c = S[bd], l = S[bd - 1], r = S[bd + 1];
              double v;
              if(r < l) v = bd * factor + static_cast<double>(r - l) / (c - l) / 2.0 * factor + 0.5;
              else v = bd * factor + static_cast<double>(r - l) / (c - r) / 2.0 * factor + 0.5;
              // static_cast<int>(double): truncation; NaN / infinities / out of range -> INT_MIN on x86 (cvttsd2si), S5
              out = (std::isfinite(v) && v > -2147483649.0 && v < 2147483648.0) ? (int) v : INT_MIN;
            } else {
              out = (int) (bd * factor);
            }
            disp[(size_t) W * y + x] = (uint16_t) out;
          }
        }
      }
    }
    speckle(100, (int) (2 * factor), disp);
  }

  // speckleFilter (:901-961): 4-connected regions of non-zero pixels whose neighbouring values differ by at most maxDifference;
  // regions of at most maxSpeckleSize pixels are set to 0.  (A flood fill in the original; the regions are the connected components
  // of that neighbour relation whatever the order.)
  void speckle(int max_size, int max_diff, std::vector<uint16_t>& img) const
  {
    std::vector<int> label((size_t) W * H, 0);
    std::vector<int> stack;
    std::vector<int> members;
    int next = 0;
    for(int p0 = 0; p0 < W * H; ++p0) {
      if(img[p0] == 0 || label[p0] != 0) continue;
      ++next;
      stack.assign(1, p0);
      members.clear();
      label[p0] = next;
      while(!stack.empty()) {
        const int p = stack.back();
        stack.pop_back();
        members.push_back(p);
        const int x = p % W, y = p / W, v = img[p];
        auto visit = [&](int q) {
          if(label[q] == 0 && img[q] != 0 && std::abs(v - (int) img[q]) <= max_diff) { label[q] = next; stack.push_back(q); }
        };
        if(x < W - 1) visit(p + 1);
        if(x > 0) visit(p - 1);
        if(y < H - 1) visit(p + W);
        if(y > 0) visit(p - W);
      }
      if((int) members.size() <= max_size)
        for(int p : members) label[p] = -1;       // zeroed below (the original zeroes them as the scan reaches them: same result)
    }
    for(int p = 0; p < W * H; ++p)
      if(label[p] < 0) img[p] = 0;
  }

  // enforceLeftRightConsistency (:963-1011), the left half (S6)
  void lr_check(std::vector<uint16_t>& dl, const std::vector<uint16_t>& dr) const
  {
    for(int y = 0; y < H; ++y)
      for(int x = 0; x < W; ++x) {
        uint16_t& v = dl[(size_t) W * y + x];
        if(v == 0) continue;
        const int ld = (int) ((double) v / factor + 0.5);
        if(x - ld < 0) { v = 0; continue; }
        const int rd = (int) ((double) dr[(size_t) W * y + x - ld] / factor + 0.5);
        if(rd == 0 || std::abs(ld - rd) > lr_thresh) v = 0;
      }
  }
};

}  // namespace

// SgmStereo::compute (utils/sgm.cc:166-184 -> SGMStereo::compute :250-285) with SgmStereo::Config (:47-56; the values the reference's
// StereoAlgorithm reads from its config file, utils/stereo_algorithm.cc:46-56).  Returns false for arguments the original throws on.
bool stereoSGM(const uint8_t* left, const uint8_t* right, int rows, int cols, const SgmParams& sp, float* dmap)
{
  if(sp.numberOfDisparities <= 0 || sp.numberOfDisparities % 16) return false;                     // :168-171, :208-210
  if(sp.censusRadius < 1 || sp.censusRadius > 2 || sp.censusWeightFactor < 0) return false;           // :228-233
  if(sp.smoothnessPenaltySmall < 0 || sp.smoothnessPenaltyLarge < 0 || sp.smoothnessPenaltySmall >= sp.smoothnessPenaltyLarge) return false;   // :239-246
  if(sp.consistencyThreshold < 0 || !(sp.disparityFactor > 0)) return false;                        // :213-215,:252-254
  if(rows <= sp.windowRadius || sp.windowRadius < 0) return false;      // fewer rows than the aggregation window: outside what the restatement covers
  Sgm s;
  s.W = cols; s.H = rows; s.D = sp.numberOfDisparities;
  s.cap = std::min(std::max(sp.sobelCapValue, 15), 127) | 1;                                          // :225-226
  s.crad = sp.censusRadius; s.wrad = sp.windowRadius;
  s.P1 = sp.smoothnessPenaltySmall; s.P2 = sp.smoothnessPenaltyLarge; s.lr_thresh = sp.consistencyThreshold;
  s.factor = sp.disparityFactor; s.cweight = sp.censusWeightFactor;
  s.pitch = cols + 15 - (cols - 1) % 16;                                                              // :256
  std::vector<uint16_t> lcost, rcost, dl, dr;
  s.left_cost(left, right, lcost);
  s.right_cost(lcost, rcost);
  s.aggregate(lcost, dl);
  s.aggregate(rcost, dr);
  s.lr_check(dl, dr);
  for(size_t i = 0; i < (size_t) rows * cols; ++i) dmap[i] = static_cast<float>(dl[i] / s.factor);    // :277-281
  return true;
}

}  // namespace orc
