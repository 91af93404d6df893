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

// Interpolation kernels of the standard PhotoError branch (bpvo/photo_error.cc:267-334), T = float.
// interpolateCosine: `auto m = (T(1) - std::cos(x*M_PI)) / 2.0` is a double (x*M_PI promotes); stored back as float.
static inline void interpolateCosine(float x, float* coeffs)
{
  const double m = (1.0 - std::cos((double) x * M_PI)) / 2.0;            // T(1) - double -> double
  coeffs[0] = (float) (1.0 - m);                                         // T(1) - m
  coeffs[1] = (float) m;
}

// interpolateCubic (:267-279): all float, A = -0.5; integer literals convert to float.
static inline void interpolateCubic(float x, float* coeffs)
{
  const float A = -0.5f;
  coeffs[0] = ((A * (x + 1.0f) - 5.0f * A) * (x + 1.0f) + 8.0f * A) * (x + 1.0f) - 4.0f * A;
  coeffs[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
  coeffs[2] = ((A + 2.0f) * (1.0f - x) - (A + 3.0f)) * (1.0f - x) * (1.0f - x) + 1.0f;
  coeffs[3] = 1.0f - coeffs[0] - coeffs[1] - coeffs[2];
}

// interpolateCubicHermite value form (:311-334), bias = tension = 0: the tangent terms divide by the double literal 2.0,
// so each is evaluated in double and their sum is rounded to float on assignment to `T m0, m1`.
static inline float interpolateCubicHermite(const float* y, float mu)
{
  const float bias = 0.0f, tension = 0.0f;
  const float mu2 = mu * mu;
  const float mu3 = mu * mu2;
  const float m0 = (float) (((double) ((y[1] - y[0]) * (1 + bias) * (1 - tension)) / 2.0) +
                            ((double) ((y[2] - y[1]) * (1 - bias) * (1 - tension)) / 2.0));
  const float m1 = (float) (((double) ((y[2] - y[1]) * (1 + bias) * (1 - tension)) / 2.0) +
                            ((double) ((y[3] - y[2]) * (1 - bias) * (1 - tension)) / 2.0));
  const float a0 = 2 * mu3 - 3 * mu2 + 1;
  const float a1 = mu3 - 2 * mu2 + mu;
  const float a2 = mu3 - mu2;
  const float a3 = -2 * mu3 + 3 * mu2;
  return a0 * y[1] + a1 * m0 + a2 * m1 + a3 * y[2];
}

// Eigen 3.2 fixed-size 4-float dot product [ext]: cwiseProduct().sum() is vectorised (one Packet4f) and reduced with
// predux, which under SSE3 (the reference builds with -msse4.1 -mavx) is two haddps: (a0 + a1) + (a2 + a3).
static inline float dot4(const float* a, const float* b)
{
  return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
}

// TemplateData::computeResiduals (bpvo/template_data.cc:174-189):
//   warp.setPose(pose); PhotoError::init (bpvo/photo_error.cc:344-363) in double; per channel PhotoError::run
//   (bpvo/photo_error.cc:365-389,446-449, kLinear) in the reference's parallel_for over channels (:188) -> OpenMP.
void TemplateData::computeResiduals(const Descriptor& desc, const M44& pose, std::vector<float>& residuals,
                                    std::vector<uint16_t>& valid, int nthreads)
{
  const int N = numPoints();
  if(N == 0) throw std::logic_error("you should call setData before calling computeResiduals");   // :177
  if(fast_warp && params.interp != kLinear) throw std::runtime_error("oracle: the projectPoints f32 formulation is kLinear only");
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
      // DisparitySpaceWarp::operator() (bpvo/disparity_space_warp.h:66-71): pw = H * p, (pw0 * w_i + cx, pw1 * w_i + cy);
      // warp.P holds rows 0, 1, 3 of H
      if(warp.dspace) { xf = xf + warp.K[2]; yf = yf + warp.K[5]; }
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
  const int interp = params.interp;
  const bool two_tap = interp == kLinear || interp == kCosine;
  const int border_lo = two_tap ? 0 : 1, border_hi = two_tap ? 1 : 3;    // photo_error.cc:347-348
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
        if(interp == kLinear) {                                          // :381-388
          const double wx = (1.0 - xf);
          const double Iw = (1.0 - yf) * (I1_ptr[ii] * wx + I1_ptr[ii + 1] * xf) +
                            yf * (I1_ptr[ii + stride] * wx + I1_ptr[ii + stride + 1] * xf);
          r_ptr[i] = (float) (Iw - (double) I0_ptr[i]);
        } else if(interp == kCosine) {                                   // :391-404
          float Cx[2], Cy[2];
          interpolateCosine((float) xf, Cx);
          interpolateCosine((float) yf, Cy);
          const float* p1 = I1_ptr + ii;
          const float* p2 = p1 + stride;
          const float d1 = p1[0] * Cx[0] + p1[1] * Cx[1];                // Eigen 2-vector dot: scalar, index order
          const float d2 = p2[0] * Cx[0] + p2[1] * Cx[1];
          const float Iw = Cy[0] * d1 + Cy[1] * d2;
          r_ptr[i] = Iw - I0_ptr[i];
        } else {
          // The four rows are yi-1 .. yi+2 and the four columns xi .. xi+3 (NOT xi-1 .. xi+2: the Map starts at xi,
          // :415-418 — restated as written).  valid only guarantees yi < rows-1, so for yi == rows-2 the reference reads
          // row `rows`, one past the image (undefined behaviour there); here that row index is clamped to rows-1 (Q21).
          const float* p[4];
          for(int k = 0; k < 4; ++k) p[k] = I1_ptr + (size_t) std::min(yi - 1 + k, rows - 1) * stride + xi;
          float Iw;
          if(interp == kCubic) {                                         // :406-424
            float Cx[4], Cy[4], d[4];
            interpolateCubic((float) xf, Cx);
            interpolateCubic((float) yf, Cy);
            for(int k = 0; k < 4; ++k) d[k] = dot4(p[k], Cx);
            Iw = dot4(Cy, d);
          } else {                                                       // kCubicHermite :426-441
            float V[4];
            for(int k = 0; k < 4; ++k) V[k] = interpolateCubicHermite(p[k], (float) xf);
            Iw = interpolateCubicHermite(V, (float) yf);
          }
          r_ptr[i] = Iw - I0_ptr[i];
        }
      } else {
        r_ptr[i] = 0.0f;
      }
    }
  }
}

}  // namespace orc
