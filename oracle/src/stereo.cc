// ORACLE — TEST INFRASTRUCTURE ONLY (see orc.h).
// Stereo front-end: block matching, the reference's default StereoAlgorithm (utils/stereo_algorithm.cc:63-82,98-111):
//   cvFindStereoCorrespondenceBM(left, right, disp16, state)  with  preFilterType = CV_STEREO_BM_XSOBEL, preFilterCap = 31,
//   SADWindowSize = 15, minDisparity = 0, numberOfDisparities from the config, textureThreshold = 10, uniquenessRatio = 15,
//   speckle filter off, disp12MaxDiff = -1;  then disp16.convertTo(CV_32F, 1/16).
// The algorithm itself lives in OpenCV 2.4 (modules/calib3d/src/stereobm.cpp: prefilterXSobel, findStereoCorrespondenceBM —
// the scalar branch), which is a third-party dependency ABSENT from /root/reference and from this image: PARITY UNPINNED.
// This file restates the published algorithm of OpenCV 2.4.x [ext]; the reference's own contribution is the parameter set
// and the conversion above.  Documented choices where the original relies on memory layout:
//   * the SAD window of the last output columns reads the right image up to SADWindowSize/2 pixels past the end of its row
//     (rptr[d] with the window column clamped, not the sum): linear addressing as in the original (the next row's pixels);
//     past the end of the image the index is clamped to the last pixel (the original reads whatever follows the buffer).
//   * minDisparity > 0: the original's column loop runs minDisparity columns past the end of the row (left image and
//     output): the loop is cut at the last column.
#include "orc.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace orc {

// prefilterXSobel (stereobm.cpp): x-Sobel clipped to [-cap, cap] + cap, two rows at a time; an odd last row and the first /
// last column are set to tab[0 + OFS] = cap
void stereoPrefilterXSobel(const uint8_t* src, int rows, int cols, int ftzero, uint8_t* dst)
{
  const int OFS = 256 * 4, TABSZ = OFS * 2 + 256;
  std::vector<uint8_t> tab(TABSZ);
  for(int x = 0; x < TABSZ; ++x) tab[x] = (uint8_t) (x - OFS < -ftzero ? 0 : x - OFS > ftzero ? ftzero * 2 : x - OFS + ftzero);
  const uint8_t val0 = tab[0 + OFS];
  int y;
  for(y = 0; y < rows - 1; y += 2) {
    const uint8_t* srow1 = src + (size_t) y * cols;
    const uint8_t* srow0 = y > 0 ? srow1 - cols : rows > 1 ? srow1 + cols : srow1;
    const uint8_t* srow2 = y < rows - 1 ? srow1 + cols : rows > 1 ? srow1 - cols : srow1;
    const uint8_t* srow3 = y < rows - 2 ? srow1 + cols * 2 : srow1;
    uint8_t* dptr0 = dst + (size_t) y * cols;
    uint8_t* dptr1 = dptr0 + cols;
    dptr0[0] = dptr0[cols - 1] = dptr1[0] = dptr1[cols - 1] = val0;
    for(int x = 1; x < cols - 1; ++x) {
      const int d0 = srow0[x + 1] - srow0[x - 1], d1 = srow1[x + 1] - srow1[x - 1], d2 = srow2[x + 1] - srow2[x - 1],
                d3 = srow3[x + 1] - srow3[x - 1];
      dptr0[x] = tab[d0 + d1 * 2 + d2 + OFS];
      dptr1[x] = tab[d1 + d2 * 2 + d3 + OFS];
    }
  }
  for(; y < rows; ++y) std::memset(dst + (size_t) y * cols, val0, (size_t) cols);
}

// findStereoCorrespondenceBM (stereobm.cpp, scalar branch) on the whole image (dy0 = dy1 = 0: rows replicate at the top and
// bottom), evaluated per output pixel instead of with the sliding sums — the sums are integers, so the order is irrelevant.
void stereoBlockMatching(const uint8_t* left, const uint8_t* right, int rows, int cols, const StereoParams& sp, int16_t* disp)
{
  const int wsz = sp.SADWindowSize, wsz2 = wsz / 2;
  const int ndisp = sp.numberOfDisparities, mindisp = sp.minDisparity;
  const int lofs = std::max(ndisp - 1 + mindisp, 0), rofs = -std::min(ndisp - 1 + mindisp, 0);
  // width1 = width - rofs - ndisp + 1 in the original; for minDisparity > 0 that range overruns the row by minDisparity columns
  // (it writes the first columns of the next row): cut at the last column here (documented choice, header)
  const int width1 = std::min(cols - rofs - ndisp + 1, cols - lofs);
  const int ftzero = sp.preFilterCap;
  const int16_t FILTERED = (int16_t) ((mindisp - 1) << 4);
  if(lofs >= cols || rofs >= cols || width1 < 1) {
    for(size_t i = 0; i < (size_t) rows * cols; ++i) disp[i] = FILTERED;
    return;
  }
  const size_t npix = (size_t) rows * cols;
  auto R = [&](int y, int col) -> int {          // right image, linear addressing past the row end (see the header)
    const size_t i = std::min((size_t) y * cols + (size_t) col, npix - 1);
    return right[i];
  };
  std::vector<int> sad(ndisp + 2);
  for(int y = 0; y < rows; ++y) {
    int16_t* drow = disp + (size_t) y * cols;
    for(int x = 0; x < lofs; ++x) drow[x] = FILTERED;
    for(int x = lofs + width1; x < cols; ++x) drow[x] = FILTERED;
    for(int x = 0; x < width1; ++x) {
      int* s = sad.data() + 1;
      for(int d = 0; d < ndisp; ++d) s[d] = 0;
      int tsum = 0;
      for(int dy = -wsz2; dy <= wsz2; ++dy) {
        // hsad rows are clamped to [0, rows - 1] (hsad_sub = max(y - wsz2 - 1, -dy0), hsad = min(y + wsz2, height + dy1 - 1); the
        // initial sums replicate row 0, htext likewise)
        const int yy = std::min(std::max(y + dy, 0), rows - 1);
        for(int dx = -wsz2; dx <= wsz2; ++dx) {
          const int xc = x + dx;
          const int lcol = lofs + std::min(std::max(xc, -lofs), cols - lofs - 1);
          const int rcol = rofs + std::min(std::max(xc, -rofs), cols - rofs - 1);
          const int lval = left[(size_t) yy * cols + lcol];
          for(int d = 0; d < ndisp; ++d) s[d] += std::abs(lval - R(yy, rcol + d));
          tsum += std::abs(lval - ftzero);
        }
      }
      int minsad = INT_MAX, mind = -1;
      for(int d = 0; d < ndisp; ++d)
        if(s[d] < minsad) { minsad = s[d]; mind = d; }
      int16_t out = FILTERED;
      bool ok = tsum >= sp.textureThreshold;
      if(ok && sp.uniquenessRatio > 0) {
        const int thresh = minsad + (minsad * sp.uniquenessRatio / 100);
        for(int d = 0; d < ndisp; ++d)
          if(s[d] <= thresh && (d < mind - 1 || d > mind + 1)) { ok = false; break; }
      }
      if(ok) {
        s[-1] = s[1];
        s[ndisp] = s[ndisp - 2];
        const int p = s[mind + 1], n = s[mind - 1];
        const int dd = p + n - 2 * s[mind] + std::abs(p - n);
        out = (int16_t) (((ndisp - mind - 1 + mindisp) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4);
      }
      drow[lofs + x] = out;
    }
  }
}

// StereoAlgorithm::run, BlockMatching branch (utils/stereo_algorithm.cc:98-111)
void stereoBM(const uint8_t* left, const uint8_t* right, int rows, int cols, const StereoParams& sp, float* dmap)
{
  std::vector<uint8_t> lp((size_t) rows * cols), rp((size_t) rows * cols);
  stereoPrefilterXSobel(left, rows, cols, sp.preFilterCap, lp.data());
  stereoPrefilterXSobel(right, rows, cols, sp.preFilterCap, rp.data());
  std::vector<int16_t> d16((size_t) rows * cols);
  stereoBlockMatching(lp.data(), rp.data(), rows, cols, sp, d16.data());
  for(size_t i = 0; i < d16.size(); ++i) dmap[i] = (float) d16[i] * (1.0f / 16.0f);   // convertTo(CV_32FC1, 1.0 / 16.0)
}

}  // namespace orc
