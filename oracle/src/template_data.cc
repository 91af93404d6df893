// ORACLE — test infrastructure only (see orc.h).  TemplateData::setData / computeResiduals and PhotoError.
#include "orc.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace orc {

// TemplateData::setData (bpvo/template_data.cc:37-142).
void TemplateData::setData(const Descriptor& desc, const float* D_ptr, int Dcols)
{
  computeSaliencyMap(desc, saliency);                                    // :40

  const int rows = desc.rows, cols = desc.cols;
  int nms_radius = -1;                                                   // :43-49
  if(rows * cols >= params.minNumPixelsForNonMaximaSuppression) nms_radius = params.nonMaxSuppRadius;

  const int border = std::max(params.nonMaxSuppRadius, 3);               // :51

  std::vector<uint16_t> cand;                                            // :53-66 (coordinates stored as uint16_t, Q9)
  for(int y = border; y < rows - border - 1; ++y) {
    const float* srow = saliency.data() + (size_t) y * cols;
    for(int x = border; x < cols - border - 1; ++x) {
      if(srow[x] >= params.minSaliency && isLocalMax(saliency.data(), cols, nms_radius, y, x)) {
        cand.push_back((uint16_t) y);
        cand.push_back((uint16_t) x);
      }
    }
  }

  points.clear();
  inds.clear();
  for(size_t i = 0; i < cand.size(); i += 2) {                           // :73-83
    const int y = cand[i + 0], x = cand[i + 1];
    const float d = D_ptr[(1 << level) * (y * Dcols + x)];               // full-res map, not rescaled (Q20)
    if(d >= params.minValidDisparity && d <= params.maxValidDisparity) {
      float pt[4];
      warp.makePoint((float) x, (float) y, d, pt);
      points.insert(points.end(), pt, pt + 4);
      inds.push_back(y * cols + x);
    }
  }

  const int extra = (int) (inds.size() % 16);                            // :85-89 (Q10)
  if(extra) {
    points.resize(points.size() - 4 * (size_t) extra);
    inds.resize(inds.size() - extra);
  }

  if(params.withNormalization && !inds.empty()) warp.setNormalization(points);   // :91-92

  const int N = (int) inds.size();
  const int C = desc.numChannels();
  numChannels = C;
  pixels.assign((size_t) C * N, 0.0f);
  jacobians.assign((size_t) C * N * 6, 0.0f);

  const float NN = 1.0f / 18.0f;                                         // :102
  std::vector<float> IxIy(2 * (size_t) N);
  for(int c = 0; c < C; ++c) {                                           // :105-137
    const float* c_ptr = desc.ch[c].data();
    float* P_ptr = pixels.data() + (size_t) c * N;
    for(int i = 0; i < N; ++i) {
      const int ii = inds[i];
      P_ptr[i] = c_ptr[ii];
      const float* cc = c_ptr + ii;
      if(params.gradientEstimation == kCD3) {
        IxIy[2 * i + 0] = 0.5f * (cc[1] - cc[-1]);
        IxIy[2 * i + 1] = 0.5f * (cc[cols] - cc[-cols]);
      } else {
        IxIy[2 * i + 0] = NN * (1.0f * cc[-2] - 8.0f * cc[-1] + 8.0f * cc[1] - 1.0f * cc[2]);
        IxIy[2 * i + 1] = NN * (1.0f * cc[-2 * cols] - 8.0f * cc[-1 * cols] + 8.0f * cc[+1 * cols] - 1.0f * cc[2 * cols]);
      }
    }
    warp.computeJacobian(points.data(), N, IxIy.data(), jacobians.data() + (size_t) c * N * 6);
  }
}

// Floor (bpvo/photo_error.cc:255-265): trunc then -(i > v).  static_cast<int> of a double that does not fit is UB
// in C++; x86 cvttsd2si returns INT_MIN ("integer indefinite") for NaN/inf/out-of-range, which is what the
// reference build does and what is restated explicitly here.
static inline int FloorD(double v)
{
  if(!(v > -2147483648.0 && v < 2147483648.0)) return INT32_MIN;   // never a valid pixel either way
  const int i = (int) v;
  return i - (i > v);
}

// TemplateData::computeResiduals (bpvo/template_data.cc:174-189):
//   warp.setPose(pose); PhotoError::init (bpvo/photo_error.cc:344-363) in double; per channel PhotoError::run
//   (bpvo/photo_error.cc:365-389,446-449, kLinear) in the reference's parallel_for over channels (:188) -> OpenMP.
void TemplateData::computeResiduals(const Descriptor& desc, const M44& pose, std::vector<float>& residuals,
                                    std::vector<uint16_t>& valid, int nthreads)
{
  const int N = numPoints();
  if(N == 0) throw std::logic_error("you should call setData before calling computeResiduals");   // :177
  if(params.interp != kLinear) throw std::runtime_error("oracle: only kLinear interpolation is restated");
  warp.setPose(pose);

  valid.resize(N);
  residuals.resize(pixels.size());

  if(fast_warp) {
    // The reference's inactive all-float branch (PHOTO_ERROR_OPT): PhotoError::Impl::init -> projectPoints scalar form
    // (bpvo/project_points.cc:180-214), run / operator() (bpvo/photo_error.cc:118-214), dot = _mm_dp_ps(.., 0xff)
    // = (a0*b0 + a1*b1) + (a2*b2 + a3*b3); load order (p[0], p[1], p[stride], p[stride+1]).
    const int rows = desc.rows, cols = desc.cols;
    const int max_rows = rows - 1, max_cols = cols - 1;
    std::vector<int> inds_w(N);
    std::vector<float> Cc(4 * (size_t) N);
    for(int i = 0; i < N; ++i) {
      const float* X = points.data() + 4 * (size_t) i;
      float x[3];
      for(int r = 0; r < 3; ++r) {
        float s = warp.P[r * 4 + 0] * X[0];
        s += warp.P[r * 4 + 1] * X[1];
        s += warp.P[r * 4 + 2] * X[2];
        s += warp.P[r * 4 + 3] * X[3];
        x[r] = s;
      }
      const float w_i = 1.0f / x[2];
      float xf = w_i * x[0], yf = w_i * x[1];
      // (int) of a float that does not fit is UB in C++; cvttss2si returns INT_MIN, restated explicitly
      const bool in_range = (xf > -2147483648.0f) && (xf < 2147483648.0f) && (yf > -2147483648.0f) && (yf < 2147483648.0f);
      const int xi = in_range ? (int) xf : INT32_MIN, yi = in_range ? (int) yf : INT32_MIN;
      valid[i] = (uint16_t) (xi >= 0 && xi < max_cols && yi >= 0 && yi < max_rows);
      inds_w[i] = valid[i] ? yi * cols + xi : 0;
      xf -= (float) xi;
      yf -= (float) yi;
      const float xfyf = xf * yf;
      Cc[4 * i + 0] = xfyf - yf - xf + 1.0f;
      Cc[4 * i + 1] = xf - xfyf;
      Cc[4 * i + 2] = yf - xfyf;
      Cc[4 * i + 3] = xfyf;
    }
    const int C = desc.numChannels();
    for(int c = 0; c < C; ++c) {
      const float* I0_ptr = pixels.data() + (size_t) c * N;
      const float* I1_ptr = desc.ch[c].data();
      float* r_ptr = residuals.data() + (size_t) c * N;
      for(int i = 0; i < N; ++i) {
        float Iw = 0.0f;
        if(valid[i]) {
          const float* p = I1_ptr + inds_w[i];
          const float* k = Cc.data() + 4 * (size_t) i;
          Iw = (k[0] * p[0] + k[1] * p[1]) + (k[2] * p[cols] + k[3] * p[cols + 1]);
        }
        r_ptr[i] = Iw - I0_ptr[i];
      }
    }
    return;
  }

  const int rows = desc.rows, cols = desc.cols;
  const int border_lo = 0, border_hi = 1;                                // kLinear (photo_error.cc:347-348)
  double P[12];
  for(int k = 0; k < 12; ++k) P[k] = (double) warp.P[k];
  std::vector<double> xy(2 * (size_t) N);
  for(int i = 0; i < N; ++i) {
    const double X0 = points[4 * i + 0], X1 = points[4 * i + 1], X2 = points[4 * i + 2], X3 = points[4 * i + 3];
    // P * X, Eigen fixed 3x4 * 4x1 in double, index-order sums
    double u[3];
    for(int r = 0; r < 3; ++r) {
      double s = P[r * 4 + 0] * X0;
      s += P[r * 4 + 1] * X1;
      s += P[r * 4 + 2] * X2;
      s += P[r * 4 + 3] * X3;
      u[r] = s;
    }
    const double zi = 1.0 / u[2];                                        // normHomog (bpvo/eigen.h:10-23)
    const double x = zi * u[0], y = zi * u[1];
    xy[2 * i + 0] = x;
    xy[2 * i + 1] = y;
    const int xi = FloorD(x), yi = FloorD(y);
    valid[i] = (uint16_t) (xi >= border_lo && xi < cols - border_hi && yi >= border_lo && yi < rows - 1);   // Q11
  }

  const int C = desc.numChannels();
  (void) nthreads;
#pragma omp parallel for num_threads(nthreads) if(nthreads > 1)
  for(int c = 0; c < C; ++c) {
    const float* I0_ptr = pixels.data() + (size_t) c * N;
    const float* I1_ptr = desc.ch[c].data();
    float* r_ptr = residuals.data() + (size_t) c * N;
    const int stride = cols;
    for(int i = 0; i < N; ++i) {
      if(valid[i]) {
        double xf = xy[2 * i + 0], yf = xy[2 * i + 1];
        const int xi = FloorD(xf), yi = FloorD(yf);
        xf -= (double) xi;
        yf -= (double) yi;
        const int ii = yi * stride + xi;
        const double wx = (1.0 - xf);
        const double Iw = (1.0 - yf) * (I1_ptr[ii] * wx + I1_ptr[ii + 1] * xf) +
                          yf * (I1_ptr[ii + stride] * wx + I1_ptr[ii + stride + 1] * xf);
        r_ptr[i] = (float) (Iw - (double) I0_ptr[i]);
      } else {
        r_ptr[i] = 0.0f;
      }
    }
  }
}

}  // namespace orc
