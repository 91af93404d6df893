// ORACLE — test infrastructure only (see orc.h).  Dense descriptors.
#include "orc.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <utility>

namespace orc {

// bpvo/types.cc:31-66 (AlgorithmParameters() constructor defaults)
void defaultParams(Params& p)
{
  p.numPyramidLevels = -1;
  p.minImageDimensionForPyramid = 40;
  p.sigmaPriorToCensusTransform = -1.0f;
  p.sigmaBitPlanes = 0.5f;
  p.dfSigma1 = 0.75f;
  p.dfSigma2 = 1.75f;
  p.latchNumBytes = 1;
  p.latchRotationInvariance = 0;
  p.latchHalfSsdSize = 1;
  p.centralDifferenceRadius = 3;
  p.centralDifferenceSigmaBefore = 0.75f;
  p.centralDifferenceSigmaAfter = 1.75f;
  p.laplacianKernelSize = 1;
  p.maxIterations = 50;
  p.parameterTolerance = 1e-7f;
  p.functionTolerance = 1e-6f;
  p.gradientTolerance = 1e-8f;
  p.relaxTolerancesForCoarseLevels = 1;
  p.gradientEstimation = kCD3;
  p.interp = kLinear;
  p.lossFunction = kTukey;
  p.descriptor = kIntensity;
  p.verbosity = 0x20;  // kIteration
  p.minTranslationMagToKeyFrame = 0.15f;
  p.minRotationMagToKeyFrame = 5.0f;
  p.maxFractionOfGoodPointsToKeyFrame = 0.6f;
  p.goodPointThreshold = 0.85f;
  p.minNumPixelsForNonMaximaSuppression = 320 * 240;
  p.nonMaxSuppRadius = 1;
  p.minNumPixelsToWork = 256;
  p.minSaliency = 0.1f;
  p.minValidDisparity = 0.001f;
  p.maxValidDisparity = 512.0f;
  p.maxTestLevel = 0;
  p.withNormalization = 1;
}

// DenseDescriptor::Create + compute (bpvo/dense_descriptor.cc:38-90):
//   kIntensity -> IntensityDescriptor::compute (bpvo/intensity_descriptor.cc:31-43): u8 -> f32, exact (Mat::convertTo).
//   kLaplacian -> LaplacianDescriptor::compute (bpvo/gradient_descriptor.cc:64-67): one f32 channel, cv::Laplacian.
//   kIntensityAndGradient -> GradientDescriptor::compute (bpvo/gradient_descriptor.cc:42-63): (I, Ix, Iy).
//   kDescriptorFieldsFirstOrder / SecondOrder -> DescriptorFields[2ndOrder]::compute (bpvo/gradient_descriptor.cc:100-160): 5 / 10 channels.
//   kCentralDifference -> CentralDifferenceDescriptor::compute (bpvo/central_difference_descriptor.cc:113-131): (2r+1)^2 - 1 channels.
//   kLatch -> LatchDescriptor::compute (bpvo/latch_descriptor.cc:1041-1086): 8 * latchNumBytes channels, see latchDescriptor below.
//   kBitPlanes -> BitPlanesDescriptor::compute (bpvo/bitplanes_descriptor.cc:84-91): census(I, sigma_ct) then for each
//                 bit b: ExtractChannel (:37-57) dst = 1.0f * ((c & (1<<b)) >> b) - 0.0f, GaussianBlur 5x5 sigma_bp if > 0.
//                 The 8 channels are the reference's parallel_for range (:89-90) -> OpenMP here.
// ---- LATCH, evaluated densely (bpvo/latch_descriptor.cc).
// The sampling table: 512 triplets of patch-centre offsets, `sampling_points_arr` (:507-1019), numbers only (tests/tools/make_latch_table.py).
static const int kLatchPoints[3072] = {
#include "latch_table.inc"
};

// The triplet offsets as CalcuateSums uses them (:180-236): as they stand, or — rotationInvariance — rotated by the key point's angle and
// clamped to [-24, 24].  The dense evaluation creates its key points with cv::KeyPoint() (:135-141), whose angle is -1, so the rotation
// is the same for every pixel: angle = -1 * (float)(CV_PI / 180.f), cos / sin of the float (:259-262; the float overloads of <cmath> —
// with ::cos(double) instead the cosine could differ in its last bit; every one of the 3072 products is further than 1e-4 from an integer,
// so the truncated offsets are the same either way: tests/test_oracle_cpu.py::test_latch_rotated_offsets_are_insensitive_to_the_last_bit_of_the_cosine).
void latchOffsets(bool rotationInvariance, int n_ints, int* out)
{
  const float angle = -1.0f * (float) (3.1415926535897932384626433832795 / 180.f);
  const float cos_theta = std::cos(angle), sin_theta = std::sin(angle);
  for(int t = 0; t < n_ints; t += 2) {
    const int ax = kLatchPoints[t], ay = kLatchPoints[t + 1];
    int ax2 = ax, ay2 = ay;
    if(rotationInvariance) {
      ax2 = (int) (((float) ax) * cos_theta - ((float) ay) * sin_theta);
      ay2 = (int) (((float) ax) * sin_theta + ((float) ay) * cos_theta);
      ax2 = std::min(24, std::max(-24, ax2));
      ay2 = std::min(24, std::max(-24, ay2));
    }
    out[t] = ax2; out[t + 1] = ay2;
  }
}

// LatchDescriptor::compute (:1041-1086) over LATCHDescriptorExtractorImpl::compute (:124-167) and pixelTests<bytes> (:264-493):
//  1. key points: every pixel with border <= y < rows - border - 1, border <= x < cols - border - 1, row-major; border = 24 + half_ssd_size;
//  2. the image smoothed with cv::GaussianBlur(Size(3,3), 2, 2) (8-bit fixed point, gaussianBlurU8_3x3);
//  3. per key point and descriptor byte ix: bit j = 7 .. 0 is [suma < sumc] of triplet 8 * ix + (7 - j), suma / sumc the sums of squared
//     differences of the (2K+1)^2 mini-patches around a and b / around c and b (integers: (int)(double(difference)^2));
//  4. channel 8 c + i of the dense descriptor = 255 * bit i of a byte - 128 inside the key-point region, 0 outside, then imsmooth(1.75).
//     WHICH byte: the reference walks `_buffer.col(c).ptr()` with ++ (:1066-1078) — the pointer of a column view advanced by one BYTE per
//     key point, i.e. through the row-major [key points][bytes] buffer from offset c — so key point k of channel byte c reads byte
//     (c + k) % bytes of key point (c + k) / bytes.  For latchNumBytes = 1 (the default and the only value in conf/) that IS byte c of
//     key point k; for more bytes it is what the reference computes, restated as it is.
static void latchDescriptor(const Params& p, const uint8_t* img, int rows, int cols, Descriptor& d, int nthreads)
{
  const int bytes = p.latchNumBytes, K = p.latchHalfSsdSize;
  if(bytes != 1 && bytes != 2 && bytes != 4 && bytes != 8 && bytes != 16 && bytes != 32 && bytes != 64)
    throw std::runtime_error("descriptorSize must be 1, 2, 4, 8, 16, 32, or 64");      // :104
  const size_t n = (size_t) rows * cols;
  const int border = 24 + K;                                       // getBorder(): PATCH_SIZE / 2 + half_ssd_size
  const int y0 = border, y1 = rows - border - 1, x0 = border, x1 = cols - border - 1;
  const int ny = std::max(0, y1 - y0), nx = std::max(0, x1 - x0);
  const size_t nkp = (size_t) ny * nx;
  std::vector<uint8_t> gray(n);
  gaussianBlurU8_3x3(img, rows, cols, 2.0f, gray.data());
  std::vector<int> off(6 * 8 * bytes);
  latchOffsets(p.latchRotationInvariance != 0, (int) off.size(), off.data());
  std::vector<uint8_t> buffer(nkp * bytes + 1, 0);
  (void) nthreads;
#pragma omp parallel for num_threads(nthreads) if(nthreads > 1)
  for(int ky = 0; ky < ny; ++ky)
    for(int kx = 0; kx < nx; ++kx) {
      const int px = (int) ((float) (x0 + kx) + 0.5), py = (int) ((float) (y0 + ky) + 0.5);      // (int)(pt.pt.x + 0.5), pt a Point2f
      uint8_t* desc = buffer.data() + ((size_t) ky * nx + kx) * bytes;
      int count = 0;
      for(int ix = 0; ix < bytes; ++ix) {
        desc[ix] = 0;
        for(int j = 7; j >= 0; --j) {
          int suma = 0, sumc = 0;
          const int ax2 = off[count] + px, ay2 = off[count + 1] + py, bx2 = off[count + 2] + px, by2 = off[count + 3] + py,
                    cx2 = off[count + 4] + px, cy2 = off[count + 5] + py;
          for(int iy = -K; iy <= K; ++iy) {
            const uint8_t* Mi_a = gray.data() + (size_t) (ay2 + iy) * cols;
            const uint8_t* Mi_b = gray.data() + (size_t) (by2 + iy) * cols;
            const uint8_t* Mi_c = gray.data() + (size_t) (cy2 + iy) * cols;
            for(int ixx = -K; ixx <= K; ++ixx) {
              const double difa = Mi_a[ax2 + ixx] - Mi_b[bx2 + ixx];
              suma += (int) (difa * difa);
              const double difc = Mi_c[cx2 + ixx] - Mi_b[bx2 + ixx];
              sumc += (int) (difc * difc);
            }
          }
          desc[ix] += (uint8_t) ((suma < sumc) << j);
          count += 6;
        }
      }
    }
  d.ch.assign(8 * bytes, std::vector<float>());
#pragma omp parallel for num_threads(nthreads) if(nthreads > 1)
  for(int j = 0; j < 8 * bytes; ++j) {
    const int c = j / 8, bit = j % 8;
    std::vector<float> tmp(n, 0.0f);
    const uint8_t* src_ptr = buffer.data() + c;                    // _buffer.col(c).ptr<const uint8_t>(), then *src_ptr++
    for(int y = y0; y < y1; ++y)
      for(int x = x0; x < x1; ++x) {
        const uint8_t val = *src_ptr++;
        tmp[(size_t) y * cols + x] = 255.0f * (float) ((val & (1 << bit)) >> bit) + -128.0f;
      }
    d.ch[j].resize(n);
    gaussianBlurF32_5x5(tmp.data(), rows, cols, 1.75f, d.ch[j].data());       // imsmooth(ch, ch, 1.75): 5 x 5 (bpvo/imgproc.cc:166-171)
  }
}

void computeDescriptor(const Params& p, const uint8_t* img, int rows, int cols, Descriptor& d, int nthreads)
{
  d.rows = rows;
  d.cols = cols;
  const size_t n = (size_t) rows * cols;
  if(p.descriptor == kLatch) {
    latchDescriptor(p, img, rows, cols, d, nthreads);
    return;
  }
  if(p.descriptor == kIntensity) {
    d.ch.resize(1);
    d.ch[0].resize(n);
    for(size_t i = 0; i < n; ++i) d.ch[0][i] = (float) img[i];
    return;
  }
  if(p.descriptor == kIntensityAndGradient) {
    // GradientDescriptor::compute (bpvo/gradient_descriptor.cc:42-63): channel 0 = intensities (convertTo), channels 1 / 2 =
    // xgradient / ygradient (bpvo/imgproc.h:214-265) of the intensities — smoothed first with cv::GaussianBlur(Size(), s, s)
    // when s = sigmaPriorToCensusTransform > 0 (the factory hands that parameter over, bpvo/dense_descriptor.cc:47-50); the
    // kernel size is OpenCV's automatic one for CV_32F, restated for 5 taps and more (s >= 0.44).  Channel 0 stays unsmoothed.
    d.ch.assign(3, std::vector<float>(n));
    for(size_t i = 0; i < n; ++i) d.ch[0][i] = (float) img[i];
    const float S = 0.5f;                                         // imgradient_scale<float>() (bpvo/imgproc.h:205-209)
    std::vector<float> smoothed;
    if(p.sigmaPriorToCensusTransform > 0.0f) {
      const int k = autoGaussTapsF32(p.sigmaPriorToCensusTransform);
      if(k < 5 || k > kMaxGaussTaps) throw std::runtime_error("oracle: GradientDescriptor sigma outside the restated kernel sizes");
      smoothed.resize(n);
      gaussianBlurF32(d.ch[0].data(), rows, cols, k, p.sigmaPriorToCensusTransform, smoothed.data());
    }
    const float* I = smoothed.empty() ? d.ch[0].data() : smoothed.data();
    for(int y = 0; y < rows; ++y)
      for(int x = 0; x < cols; ++x) {
        const size_t q = (size_t) y * cols + x;
        // Ix.col(0) = S*(I.col(1) - I.col(0)); interior S*(I(x+1) - I(x-1)); Ix.col(W-1) = S*(I.col(W-1) - I.col(W-2))
        d.ch[1][q] = x == 0 ? S * (I[q + 1] - I[q]) : (x == cols - 1 ? S * (I[q] - I[q - 1]) : S * (I[q + 1] - I[q - 1]));
        d.ch[2][q] = y == 0 ? S * (I[q + cols] - I[q]) : (y == rows - 1 ? S * (I[q] - I[q - cols]) : S * (I[q + cols] - I[q - cols]));
      }
    return;
  }
  if(p.descriptor == kLaplacian) {
    // LaplacianDescriptor::compute (bpvo/gradient_descriptor.cc:64-67): cv::Laplacian(image, CV_32F, ksize).
    // [ext: OpenCV 2.4 deriv.cpp] ksize 1 / 3 -> filter2D with the 3x3 kernels {0,1,0,1,-4,1,0,1,0} / {2,0,2,0,-8,0,2,0,2},
    // scale 1, delta 0, BORDER_DEFAULT (= REFLECT_101).  Small integers in f32: exact in any evaluation order.
    // ksize 5 / 7 -> Sobel second derivatives d2/dx2 + d2/dy2 with the separable kernels of cv::getSobelKernels
    // ([1 0 -2 0 1] x [1 4 6 4 1], [1 2 -1 -4 -1 2 1] x [1 6 15 20 15 6 1]; work type 16S for ksize 5 on u8, never saturated:
    // |d2x + d2y| <= 16320), integer-valued as well.
    if(p.laplacianKernelSize == 5 || p.laplacianKernelSize == 7) {
      const int K = p.laplacianKernelSize, r = K / 2;
      static const int d5[5] = {1, 0, -2, 0, 1}, s5[5] = {1, 4, 6, 4, 1};
      static const int d7[7] = {1, 2, -1, -4, -1, 2, 1}, s7[7] = {1, 6, 15, 20, 15, 6, 1};
      const int* dk = K == 5 ? d5 : d7;
      const int* sk = K == 5 ? s5 : s7;
      auto reflw = [](int q, int len) { if(len == 1) return 0; while(q < 0 || q >= len) q = q < 0 ? -q : 2 * len - 2 - q; return q; };
      d.ch.resize(1);
      d.ch[0].resize(n);
      std::vector<int> dxx(n), sxx(n);                        // row passes: second derivative / smoothing along x
      for(int y = 0; y < rows; ++y)
        for(int x = 0; x < cols; ++x) {
          int a = 0, b = 0;
          for(int t = 0; t < K; ++t) {
            const int v = img[(size_t) y * cols + reflw(x - r + t, cols)];
            a += dk[t] * v;
            b += sk[t] * v;
          }
          dxx[(size_t) y * cols + x] = a;
          sxx[(size_t) y * cols + x] = b;
        }
      for(int y = 0; y < rows; ++y)
        for(int x = 0; x < cols; ++x) {
          int acc = 0;
          for(int t = 0; t < K; ++t) {
            const size_t q = (size_t) reflw(y - r + t, rows) * cols + x;
            acc += sk[t] * dxx[q] + dk[t] * sxx[q];
          }
          d.ch[0][(size_t) y * cols + x] = (float) acc;
        }
      return;
    }
    if(p.laplacianKernelSize != 1 && p.laplacianKernelSize != 3) throw std::runtime_error("oracle: Laplacian kernel sizes 1, 3, 5 and 7 only");
    const float k_edge = p.laplacianKernelSize == 3 ? 0.0f : 1.0f, k_diag = p.laplacianKernelSize == 3 ? 2.0f : 0.0f;
    const float k_ctr = p.laplacianKernelSize == 3 ? -8.0f : -4.0f;
    auto refl = [](int q, int len) { if(q < 0) q = -q; if(q >= len) q = 2 * len - 2 - q; return q < 0 ? 0 : q; };
    d.ch.resize(1);
    d.ch[0].resize(n);
    for(int y = 0; y < rows; ++y) {
      const int ym = refl(y - 1, rows), yp = refl(y + 1, rows);
      for(int x = 0; x < cols; ++x) {
        const int xm = refl(x - 1, cols), xp = refl(x + 1, cols);
        auto I = [&](int yy, int xx) { return (float) img[(size_t) yy * cols + xx]; };
        float v = k_diag * I(ym, xm) + k_edge * I(ym, x) + k_diag * I(ym, xp);
        v += k_edge * I(y, xm) + k_ctr * I(y, x) + k_edge * I(y, xp);
        v += k_diag * I(yp, xm) + k_edge * I(yp, x) + k_diag * I(yp, xp);
        d.ch[0][(size_t) y * cols + x] = v;
      }
    }
    return;
  }
  if(p.descriptor == kDescriptorFieldsFirstOrder || p.descriptor == kDescriptorFieldsSecondOrder) {
    // DescriptorFields::compute / DescriptorFields2ndOrder::compute (bpvo/gradient_descriptor.cc:100-160).
    // imsmooth (bpvo/imgproc.cc:166-171): cv::GaussianBlur with k = max(5, 2*round(sigma)+1) taps -> 5 x 5 for sigma < 2.5,
    // the f32 5 x 5 blur restated in imgproc.cc (larger kernels are not restated).
    if((p.dfSigma1 > 0.0f && imsmoothTaps(p.dfSigma1) > kMaxGaussTaps) || (p.dfSigma2 > 0.0f && imsmoothTaps(p.dfSigma2) > kMaxGaussTaps))
      throw std::runtime_error("oracle: imsmooth kernels wider than 31 taps are not restated");
    typedef std::vector<float> Plane;
    auto smooth = [&](const Plane& src, float sigma) { Plane o(n); gaussianBlurF32(src.data(), rows, cols, imsmoothTaps(sigma), sigma, o.data()); return o; };
    const float S = 0.5f;                                         // imgradient_scale<float>() (bpvo/imgproc.h:205-209)
    auto xgrad = [&](const Plane& I, Plane& o) {                  // bpvo/imgproc.h:214-238
      o.resize(n);
      for(int y = 0; y < rows; ++y)
        for(int x = 0; x < cols; ++x) {
          const size_t q = (size_t) y * cols + x;
          o[q] = x == 0 ? S * (I[q + 1] - I[q]) : (x == cols - 1 ? S * (I[q] - I[q - 1]) : S * (I[q + 1] - I[q - 1]));
        }
    };
    auto ygrad = [&](const Plane& I, Plane& o) {                  // bpvo/imgproc.h:240-265
      o.resize(n);
      for(int y = 0; y < rows; ++y)
        for(int x = 0; x < cols; ++x) {
          const size_t q = (size_t) y * cols + x;
          o[q] = y == 0 ? S * (I[q + cols] - I[q]) : (y == rows - 1 ? S * (I[q] - I[q - cols]) : S * (I[q + cols] - I[q - cols]));
        }
    };
    auto split = [&](const Plane& src, Plane& pos, Plane& neg) {   // splitPosNeg (gradient_descriptor.cc:80-98)
      pos.resize(n); neg.resize(n);
      for(size_t i = 0; i < n; ++i) {
        pos[i] = src[i] >= 0 ? src[i] : 0.0f;
        neg[i] = src[i] < 0 ? src[i] : 0.0f;
      }
      if(p.dfSigma2 > 0.0f) { pos = smooth(pos, p.dfSigma2); neg = smooth(neg, p.dfSigma2); }
    };
    Plane I0(n);
    for(size_t i = 0; i < n; ++i) I0[i] = (float) img[i];
    const Plane I = p.dfSigma1 > 0.0f ? smooth(I0, p.dfSigma1) : I0;
    if(p.descriptor == kDescriptorFieldsFirstOrder) {
      d.ch.assign(5, Plane());
      d.ch[0] = I0;                                               // channel 0 keeps the unsmoothed intensities (:105)
      Plane buffer;
      xgrad(I, buffer); split(buffer, d.ch[1], d.ch[2]);
      ygrad(I, buffer); split(buffer, d.ch[3], d.ch[4]);
    } else {
      d.ch.assign(10, Plane());
      Plane buffer1, buffer2;
      xgrad(I, buffer1);       split(buffer1, d.ch[0], d.ch[1]);   // Ix
      xgrad(buffer1, buffer2); split(buffer2, d.ch[2], d.ch[3]);   // Ixx
      ygrad(buffer2, buffer1); split(buffer2, d.ch[4], d.ch[5]);   // "Ixy": the reference splits buffer2 (Ixx) again (:149-150)
      ygrad(I, buffer1);       split(buffer1, d.ch[6], d.ch[7]);   // Iy
      ygrad(buffer1, buffer2); split(buffer2, d.ch[8], d.ch[9]);   // Iyy
    }
    return;
  }
  if(p.descriptor == kCentralDifference) {
    // CentralDifferenceDescriptor::compute (bpvo/central_difference_descriptor.cc:113-131): the u8 image, imsmooth'ed when
    // sigma_before > 0, minus its copy shifted by every offset of the (2r+1)^2 window except (0,0) (rows outer, columns
    // inner, clamped at the borders, :52-69), each channel imsmooth'ed when sigma_after > 0 (:71-72).
    const int R = p.centralDifferenceRadius;
    if(R <= 0) throw std::runtime_error("invalid radius");       // :19
    if((p.centralDifferenceSigmaBefore > 0.0f && imsmoothTaps(p.centralDifferenceSigmaBefore) > kMaxGaussTaps) ||
       (p.centralDifferenceSigmaAfter > 0.0f && imsmoothTaps(p.centralDifferenceSigmaAfter) > kMaxGaussTaps))
      throw std::runtime_error("oracle: imsmooth kernels wider than 31 taps are not restated");
    std::vector<uint8_t> I(img, img + n);
    if(p.centralDifferenceSigmaBefore > 0.0f)
      gaussianBlurU8(img, rows, cols, imsmoothTaps(p.centralDifferenceSigmaBefore), p.centralDifferenceSigmaBefore, I.data());
    const int C = (2 * R + 1) * (2 * R + 1) - 1;
    d.ch.assign(C, std::vector<float>());
    std::vector<std::pair<int, int>> offsets;                    // (x, y)
    for(int r = -R; r <= R; ++r)
      for(int c = -R; c <= R; ++c)
        if(!(r == 0 && c == 0)) offsets.push_back(std::make_pair(c, r));
    (void) nthreads;
#pragma omp parallel for num_threads(nthreads) if(nthreads > 1)
    for(int i = 0; i < C; ++i) {
      std::vector<float> tmp(n);
      const int x_off = offsets[i].first, y_off = offsets[i].second;
      for(int y = 0; y < rows; ++y) {
        const int y_i = std::min(std::max(y + y_off, 0), rows - 1);
        for(int x = 0; x < cols; ++x) {
          const int x_i = std::min(std::max(x + x_off, 0), cols - 1);
          tmp[(size_t) y * cols + x] = (float) I[(size_t) y * cols + x] - (float) I[(size_t) y_i * cols + x_i];
        }
      }
      if(p.centralDifferenceSigmaAfter > 0.0f) {
        d.ch[i].resize(n);
        gaussianBlurF32(tmp.data(), rows, cols, imsmoothTaps(p.centralDifferenceSigmaAfter), p.centralDifferenceSigmaAfter, d.ch[i].data());
      } else {
        d.ch[i].swap(tmp);
      }
    }
    return;
  }
  if(p.descriptor != kBitPlanes) throw std::runtime_error("oracle: unsupported descriptor");

  std::vector<uint8_t> C(n);
  census(img, rows, cols, p.sigmaPriorToCensusTransform, C.data());
  d.ch.resize(8);
  (void) nthreads;
#pragma omp parallel for num_threads(nthreads) if(nthreads > 1)
  for(int b = 0; b < 8; ++b) {
    std::vector<float> tmp(n);
    for(size_t i = 0; i < n; ++i) tmp[i] = 1.0f * (float) ((C[i] & (1 << b)) >> b) - 0.0f;
    d.ch[b].resize(n);
    if(p.sigmaBitPlanes > 0.0f)
      gaussianBlurF32_5x5(tmp.data(), rows, cols, p.sigmaBitPlanes, d.ch[b].data());
    else
      d.ch[b] = tmp;
  }
}

// DenseDescriptor::computeSaliencyMap (bpvo/dense_descriptor.cc:92-100); IntensityDescriptor's override
// (bpvo/intensity_descriptor.cc:45-53) is the C == 1 case of the same code.
void computeSaliencyMap(const Descriptor& d, std::vector<float>& S)
{
  S.assign((size_t) d.rows * d.cols, 0.0f);
  gradientAbsoluteMagnitude(d.ch[0].data(), d.rows, d.cols, S.data());
  for(int i = 1; i < d.numChannels(); ++i) gradientAbsoluteMagnitudeAcc(d.ch[i].data(), d.rows, d.cols, S.data());
}

}  // namespace orc
