// ORACLE — test infrastructure only (see orc.h).  Image-level operators on the hot path.
#include "orc.h"

#include <cassert>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <vector>

namespace orc {

// cv::borderInterpolate(p, len, BORDER_REFLECT_101) [ext: OpenCV 2.4 imgproc/filter.cpp]
static inline int reflect101(int p, int len)
{
  if(len == 1) return 0;
  while(p < 0 || p >= len) {
    if(p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}

// cv::pyrDown, u8, default border (reference call site: bpvo/image_pyramid.cc:49).
// [ext: OpenCV 2.4 imgproc/pyramids.cpp pyrDown_<FixPtCast<uchar,8>>]: separable [1 4 6 4 1], horizontal pass
// into int rows, vertical pass, dst = (sum + 128) >> 8, BORDER_REFLECT_101, dst = ((W+1)/2, (R+1)/2).
void pyrDownU8(const uint8_t* src, int rows, int cols, std::vector<uint8_t>& dst, int& drows, int& dcols)
{
  drows = (rows + 1) / 2;
  dcols = (cols + 1) / 2;
  dst.assign((size_t) drows * dcols, 0);
  std::vector<int> hrow((size_t) 5 * dcols);
  for(int y = 0; y < drows; ++y) {
    for(int k = 0; k < 5; ++k) {
      const int sy = reflect101(2 * y - 2 + k, rows);
      const uint8_t* s = src + (size_t) sy * cols;
      int* h = hrow.data() + (size_t) k * dcols;
      for(int x = 0; x < dcols; ++x) {
        const int x0 = reflect101(2 * x - 2, cols), x1 = reflect101(2 * x - 1, cols), x2 = reflect101(2 * x, cols),
                  x3 = reflect101(2 * x + 1, cols), x4 = reflect101(2 * x + 2, cols);
        h[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
      }
    }
    const int *r0 = hrow.data(), *r1 = r0 + dcols, *r2 = r1 + dcols, *r3 = r2 + dcols, *r4 = r3 + dcols;
    uint8_t* d = dst.data() + (size_t) y * dcols;
    for(int x = 0; x < dcols; ++x)
      d[x] = (uint8_t) ((r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x] + 128) >> 8);
  }
}

int imsmoothTaps(float sigma) { return std::max(5, 2 * (int) std::round((double) sigma) + 1); }
// cv::GaussianBlur with ksize = Size() on a CV_32F image [ext: OpenCV 2.4 smooth.cpp createGaussianFilter]:
// cvRound(sigma * 4 * 2 + 1) | 1 (cvRound = round half to even)
int autoGaussTapsF32(float sigma) { return ((int) std::nearbyint((double) sigma * 4.0 * 2.0 + 1.0)) | 1; }

// cv::getGaussianKernel(n, sigma, CV_32F) for sigma > 0 [ext: OpenCV 2.4 imgproc/smooth.cpp]:
// t = exp(-0.5/sigma^2 * x^2) in double, stored as float, sum of the floats in double, scaled by 1/sum.
static void gaussianKernelF32(int n, double sigma, float* k)
{
  const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  const double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for(int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    const double t = std::exp(scale2X * x * x);
    k[i] = (float) t;
    sum += k[i];
  }
  sum = 1. / sum;
  for(int i = 0; i < n; ++i) k[i] = (float) (k[i] * sum);
}

// cv::GaussianBlur(f32, Size(5,5), sigma, sigma) (reference call site: bpvo/bitplanes_descriptor.cc:56).
// [ext: OpenCV 2.4 filter.cpp SymmRowSmallFilter<float,float> ksize 5, SymmColumnFilter<Cast<float,float>>]:
// row pass s = S[0]*k0 + (S[-1]+S[1])*k1 + (S[-2]+S[2])*k2, column pass s = k0*S0; s += k1*(S1+S-1); s += k2*(S2+S-2),
// all f32, BORDER_REFLECT_101.
void gaussianBlurF32_5x5(const float* src, int rows, int cols, float sigma, float* dst)
{
  float kern[5];
  gaussianKernelF32(5, sigma, kern);
  const float k0 = kern[2], k1 = kern[3], k2 = kern[4];
  std::vector<float> tmp((size_t) rows * cols);
  for(int y = 0; y < rows; ++y) {
    const float* S = src + (size_t) y * cols;
    float* t = tmp.data() + (size_t) y * cols;
    for(int x = 0; x < cols; ++x) {
      const float sm1 = S[reflect101(x - 1, cols)], sp1 = S[reflect101(x + 1, cols)];
      const float sm2 = S[reflect101(x - 2, cols)], sp2 = S[reflect101(x + 2, cols)];
      t[x] = S[x] * k0 + (sm1 + sp1) * k1 + (sm2 + sp2) * k2;
    }
  }
  for(int y = 0; y < rows; ++y) {
    const float* S0 = tmp.data() + (size_t) y * cols;
    const float* Sm1 = tmp.data() + (size_t) reflect101(y - 1, rows) * cols;
    const float* Sp1 = tmp.data() + (size_t) reflect101(y + 1, rows) * cols;
    const float* Sm2 = tmp.data() + (size_t) reflect101(y - 2, rows) * cols;
    const float* Sp2 = tmp.data() + (size_t) reflect101(y + 2, rows) * cols;
    float* d = dst + (size_t) y * cols;
    for(int x = 0; x < cols; ++x) {
      float s0 = k0 * S0[x];
      s0 += k1 * (Sp1[x] + Sm1[x]);
      s0 += k2 * (Sp2[x] + Sm2[x]);
      d[x] = s0;
    }
  }
}

// cv::GaussianBlur(f32, Size(k,k), sigma, sigma) for k >= 7 (imsmooth with sigma >= 2.5, bpvo/imgproc.cc:166-171; the
// automatic kernel size of GradientDescriptor, bpvo/gradient_descriptor.cc:53).  [ext: OpenCV 2.4 filter.cpp — kernels
// wider than 5 leave the "small" row filter: RowFilter<float,float> s = k[0]*S[0]; s += k[j]*S[j] for j = 1..k-1 over the
// window's leftmost to rightmost tap, and SymmColumnFilter<Cast<float,float>> s = k0*S0; s += kj*(S+j + S-j)], f32,
// BORDER_REFLECT_101.  UNPINNED like every other OpenCV restatement here.
void gaussianBlurF32(const float* src, int rows, int cols, int ksize, float sigma, float* dst)
{
  if(ksize == 5) { gaussianBlurF32_5x5(src, rows, cols, sigma, dst); return; }
  if(ksize < 7 || ksize > kMaxGaussTaps || !(ksize & 1)) throw std::runtime_error("oracle: Gaussian kernel size not restated");
  float kern[kMaxGaussTaps];
  gaussianKernelF32(ksize, sigma, kern);
  const int r = ksize / 2;
  std::vector<float> tmp((size_t) rows * cols);
  for(int y = 0; y < rows; ++y) {
    const float* S = src + (size_t) y * cols;
    float* t = tmp.data() + (size_t) y * cols;
    for(int x = 0; x < cols; ++x) {
      float s0 = kern[0] * S[reflect101(x - r, cols)];
      for(int j = 1; j < ksize; ++j) s0 += kern[j] * S[reflect101(x - r + j, cols)];
      t[x] = s0;
    }
  }
  for(int y = 0; y < rows; ++y) {
    float* d = dst + (size_t) y * cols;
    for(int x = 0; x < cols; ++x) {
      float s0 = kern[r] * tmp[(size_t) y * cols + x];
      for(int j = 1; j <= r; ++j)
        s0 += kern[r + j] * (tmp[(size_t) reflect101(y + j, rows) * cols + x] + tmp[(size_t) reflect101(y - j, rows) * cols + x]);
      d[x] = s0;
    }
  }
}

// cv::GaussianBlur(u8, Size(k,k), sigma, sigma) for any odd k >= 5 in OpenCV 2.4's 8-bit fixed point (taps cvRound(k * 256),
// row pass u8 -> int, column pass (sum + 2^15) >> 16 saturated): integer sums, so the order of the taps does not matter.
void gaussianBlurU8(const uint8_t* src, int rows, int cols, int ksize, float sigma, uint8_t* dst)
{
  if(ksize == 5) { gaussianBlurU8_5x5(src, rows, cols, sigma, dst); return; }
  if(ksize < 7 || ksize > kMaxGaussTaps || !(ksize & 1)) throw std::runtime_error("oracle: Gaussian kernel size not restated");
  float kf[kMaxGaussTaps];
  gaussianKernelF32(ksize, sigma, kf);
  int ki[kMaxGaussTaps];
  for(int i = 0; i < ksize; ++i) ki[i] = (int) std::nearbyint((double) kf[i] * 256.0);
  const int r = ksize / 2;
  std::vector<int> tmp((size_t) rows * cols);
  for(int y = 0; y < rows; ++y) {
    const uint8_t* S = src + (size_t) y * cols;
    for(int x = 0; x < cols; ++x) {
      int s0 = 0;
      for(int j = 0; j < ksize; ++j) s0 += ki[j] * S[reflect101(x - r + j, cols)];
      tmp[(size_t) y * cols + x] = s0;
    }
  }
  for(int y = 0; y < rows; ++y)
    for(int x = 0; x < cols; ++x) {
      int s0 = 0;
      for(int j = 0; j < ksize; ++j) s0 += ki[j] * tmp[(size_t) reflect101(y - r + j, rows) * cols + x];
      const int v = (s0 + (1 << 15)) >> 16;
      dst[(size_t) y * cols + x] = (uint8_t) std::min(255, std::max(0, v));
    }
}

// cv::GaussianBlur(u8, Size(3,3), s, s) (reference call site: bpvo/census.cc:65, only when sigma_ct > 0).
// [ext: OpenCV 2.4 createSeparableLinearFilter 8-bit fixed point]: kernels round(k*256) (cvRound), row pass u8->int,
// column pass (sum + 2^15) >> 16 saturated.  Version dependent in OpenCV: UNPINNED, the
// headline configurations keep sigma_ct = -1 and never reach this.
void gaussianBlurU8_3x3(const uint8_t* src, int rows, int cols, float sigma, uint8_t* dst)
{
  float kf[3];
  gaussianKernelF32(3, sigma, kf);
  int ki[3];
  for(int i = 0; i < 3; ++i) ki[i] = (int) std::nearbyint((double) kf[i] * 256.0);
  std::vector<int> tmp((size_t) rows * cols);
  for(int y = 0; y < rows; ++y) {
    const uint8_t* S = src + (size_t) y * cols;
    for(int x = 0; x < cols; ++x)
      tmp[(size_t) y * cols + x] = S[x] * ki[1] + (S[reflect101(x - 1, cols)] + S[reflect101(x + 1, cols)]) * ki[2];
  }
  for(int y = 0; y < rows; ++y) {
    const int* S0 = tmp.data() + (size_t) y * cols;
    const int* Sm = tmp.data() + (size_t) reflect101(y - 1, rows) * cols;
    const int* Sp = tmp.data() + (size_t) reflect101(y + 1, rows) * cols;
    for(int x = 0; x < cols; ++x) {
      int v = (S0[x] * ki[1] + (Sm[x] + Sp[x]) * ki[2] + (1 << 15)) >> 16;
      dst[(size_t) y * cols + x] = (uint8_t) std::min(255, std::max(0, v));
    }
  }
}

// cv::GaussianBlur(u8, Size(5,5), s, s) (reference call site: imsmooth of the u8 image, bpvo/central_difference_descriptor.cc:117
// via bpvo/imgproc.cc:166-171).  Same fixed-point separable filter as the 3 x 3 case above with five taps
// [ext: OpenCV 2.4 SymmRowSmallFilter<uchar,int>, SymmColumnFilter<FixedPtCastEx<int,uchar>>]: integer arithmetic, so the
// evaluation order does not matter.  UNPINNED like the 3 x 3 form.
void gaussianBlurU8_5x5(const uint8_t* src, int rows, int cols, float sigma, uint8_t* dst)
{
  float kf[5];
  gaussianKernelF32(5, sigma, kf);
  int ki[5];
  for(int i = 0; i < 5; ++i) ki[i] = (int) std::nearbyint((double) kf[i] * 256.0);
  std::vector<int> tmp((size_t) rows * cols);
  for(int y = 0; y < rows; ++y) {
    const uint8_t* S = src + (size_t) y * cols;
    for(int x = 0; x < cols; ++x)
      tmp[(size_t) y * cols + x] = S[x] * ki[2] + (S[reflect101(x - 1, cols)] + S[reflect101(x + 1, cols)]) * ki[3] +
                                   (S[reflect101(x - 2, cols)] + S[reflect101(x + 2, cols)]) * ki[4];
  }
  for(int y = 0; y < rows; ++y) {
    const int* S0 = tmp.data() + (size_t) y * cols;
    const int* Sm1 = tmp.data() + (size_t) reflect101(y - 1, rows) * cols;
    const int* Sp1 = tmp.data() + (size_t) reflect101(y + 1, rows) * cols;
    const int* Sm2 = tmp.data() + (size_t) reflect101(y - 2, rows) * cols;
    const int* Sp2 = tmp.data() + (size_t) reflect101(y + 2, rows) * cols;
    for(int x = 0; x < cols; ++x) {
      int v = (S0[x] * ki[2] + (Sm1[x] + Sp1[x]) * ki[3] + (Sm2[x] + Sp2[x]) * ki[4] + (1 << 15)) >> 16;
      dst[(size_t) y * cols + x] = (uint8_t) std::min(255, std::max(0, v));
    }
  }
}

// bpvo/census.cc:42-91 with v128 `>=` (bpvo/v128.h:102-105): bit k of dst(y,x) = [ neighbour_k >= centre ], neighbour
// order (-1,-1),(-1,0),(-1,+1),(0,-1),(0,+1),(+1,-1),(+1,0),(+1,+1); rows 0 and R-1, cols 0 and W-1 are 0.
// The 16-wide SSE ops + one overlapping op at the right edge (census.cc:78-83, Q19) cover every interior pixel
// exactly once in value, which is what is restated; the reference needs cols >= 18 for that op to stay in the row.
void census(const uint8_t* src, int rows, int cols, float sigma_ct, uint8_t* dst)
{
  assert(cols >= 18 && rows >= 3);
  std::vector<uint8_t> blurred;
  const uint8_t* I = src;
  if(sigma_ct > 0.0f) {
    blurred.resize((size_t) rows * cols);
    gaussianBlurU8_3x3(src, rows, cols, sigma_ct, blurred.data());
    I = blurred.data();
  }
  std::memset(dst, 0, (size_t) rows * cols);
  for(int y = 1; y < rows - 1; ++y) {
    const uint8_t* p = I + (size_t) y * cols;
    uint8_t* d = dst + (size_t) y * cols;
    for(int x = 1; x < cols - 1; ++x) {
      const uint8_t c = p[x];
      d[x] = (uint8_t) (((p[x - cols - 1] >= c) << 0) | ((p[x - cols] >= c) << 1) | ((p[x - cols + 1] >= c) << 2) |
                        ((p[x - 1] >= c) << 3) | ((p[x + 1] >= c) << 4) | ((p[x + cols - 1] >= c) << 5) |
                        ((p[x + cols] >= c) << 6) | ((p[x + cols + 1] >= c) << 7));
    }
  }
}

// bpvo/imgproc.cc:33-43 gradientAbsMag for one element (4 lanes at once in the reference)
static inline float gradAbsMag1(const float* src, int stride)
{
  const float Ix = std::fabs(src[-1] - src[1]);
  const float Iy = std::fabs(src[-stride] - src[stride]);
  return Ix + Iy;
}

// bpvo/imgproc.cc:45-74 (Q7b).  Literal restatement, linear-memory reads/writes included:
// the x = 0 block reads src[-1] (= last pixel of the previous row), the scalar tail uses `+` for the y term and
// reads src[x+1] past the row end, `dst[x] = 0` with x == cols zeroes column 0 of the next row (overwritten when
// that row is processed), then dst[cols-1] = 0.  First and last rows are 0.
void gradientAbsoluteMagnitude(const float* src_ptr, int rows, int cols, float* dst_ptr)
{
  std::fill_n(dst_ptr, cols, 0.0f);
  const float* src = src_ptr + cols;
  float* dst = dst_ptr + cols;
  const int n = cols & ~3;
  for(int r = 2; r < rows; ++r) {
    int x = 0;
    for(; x < n; x += 4)
      for(int k = 0; k < 4; ++k) dst[x + k] = gradAbsMag1(src + x + k, cols);
    for(; x < cols; ++x)
      dst[x] = std::fabs(src[x + 1] - src[x - 1]) + std::fabs(src[x + cols] + src[x - cols]);
    dst[x] = 0.0f;
    dst[cols - 1] = 0.0f;
    dst += cols;
    src += cols;
  }
  std::fill_n(dst, cols, 0.0f);
}

// bpvo/imgproc.cc:104-127 (Q7).  Literal restatement of the store bug: every 4-wide result is stored to `dst`
// (row start), not `dst + x` (imgproc.cc:117), so for C > 1 columns 4..n-1 keep the channel-0 value and columns 0..3
// end up holding (block n-4..n-1 of the accumulator) + (this channel's gradient of that block).
void gradientAbsoluteMagnitudeAcc(const float* src, int rows, int cols, float* dst)
{
  const int n = cols & ~3;
  src += cols;
  dst += cols;
  for(int r = 2; r < rows; ++r, src += cols, dst += cols) {
    int x = 0;
    for(; x < n; x += 4) {
      float g[4];
      for(int k = 0; k < 4; ++k) g[k] = dst[x + k] + gradAbsMag1(src + x + k, cols);
      for(int k = 0; k < 4; ++k) dst[k] = g[k];
    }
    for(; x < cols; ++x)
      dst[x] += std::fabs(src[x - 1] - src[x + 1]) + std::fabs(src[x - cols] + src[x + cols]);
    dst[x] = 0.0f;
    dst[cols - 1] = 0.0f;
  }
}

// bpvo/imgproc.h:117-160 with WITH_SIMD (the build default, Q8): radius 1 tests a 3 rows x 4 cols window
// (cols -1..+2) with strict `>`; the centre row's mask 13 = lanes {0,2,3} skips only the centre.  Generic radius
// rejects on `>=`.
bool isLocalMax(const float* ptr, int stride, int radius, int row, int col)
{
  if(radius <= 0) return true;
  if(radius == 1) {
    const float* p = ptr + (size_t) row * stride + col;
    const float v = *p;
    for(int k = -1; k <= 2; ++k) {
      if(k != 0 && !(v > p[k])) return false;
      if(!(v > p[k - stride])) return false;
      if(!(v > p[k + stride])) return false;
    }
    return true;
  }
  const float v = ptr[(size_t) row * stride + col];
  for(int r = -radius; r <= radius; ++r)
    for(int c = -radius; c <= radius; ++c)
      if(!(!r && !c) && ptr[(size_t) (row + r) * stride + col + c] >= v) return false;
  return true;
}

}  // namespace orc
