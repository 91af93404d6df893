// ORACLE — test infrastructure only (see orc.h).  Robust scale, M-estimator weights, normal equations, 6x6 solve.
#include "orc.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace orc {

// median (bpvo/utils.h:224-252, Q5): empty -> 0; n < 3 -> data[0]; odd -> middle; even -> (max(lower half) +
// middle) / 2.0 (double) narrowed to float.
float medianOf(std::vector<float>& data)
{
  if(data.empty()) return 0.0f;
  if(data.size() < 3) return data[0];
  const size_t n = data.size();
  auto middle = data.begin() + n / 2;
  std::nth_element(data.begin(), middle, data.end());
  if(n % 2 != 0) return *middle;
  auto m = std::max_element(data.begin(), middle);
  return (float) ((*m + *middle) / 2.0);
}

// ScaleEstimator + AutoScaleEstimator::estimateScale (bpvo/mestimator.cc:452-490, Q5, Q6):
// sigma = (1.4826f * (1.0f + 5.0f / (n - 6))) * median(|r| : valid) with (n - 6) in size_t arithmetic.
float AutoScaleEstimator::estimateScale(const std::vector<float>& r, const std::vector<uint16_t>& valid)
{
  if(delta_scale > tol) {
    buffer.resize(0);
    buffer.reserve(r.size());
    for(size_t i = 0; i < r.size(); ++i)
      if(valid[i] != 0) buffer.push_back(std::fabs(r[i]));
    const size_t nm6 = buffer.size() - 6;   // wraps for n < 6, like the reference
    float s = (1.4826f * (1.0f + 5.0f / nm6)) * medianOf(buffer);
    if(s < 1e-6) s = 1.0;                   // mestimator.cc:480-481
    delta_scale = std::fabs(s - scale);
    scale = s;
  }
  return scale;
}

// MEstimator::ComputeWeights with WITH_SIMD (bpvo/mestimator.cc:390-415 -> huber_simd :242-282, tukey_simd :303-366):
// the vector body ignores `valid` (Q12); 2*8-wide AVX blocks cover n & ~15, the scalar tail (:297-300, :380-384, with
// HuberOp/TukeyOp :36-61) multiplies by valid.  L2 -> 1.
void computeWeights(int loss, const std::vector<float>& r, const std::vector<uint16_t>& valid, float sigma,
                    std::vector<float>& w)
{
  w.resize(valid.size());
  if(loss == kL2) {
    std::fill(w.begin(), w.end(), 1.0f);
    return;
  }
  const float sigma_inv = 1.0f / sigma;
  const size_t N = r.size();
  const size_t n = N & ~(size_t) 15;
  if(loss == kHuber) {
    const float k = 1.345f;
    for(size_t i = 0; i < n; ++i) {
      const float x = std::fabs(r[i] * sigma_inv);
      w[i] = k / std::max(x, k);
    }
    for(size_t i = n; i < N; ++i) {
      const float x = std::fabs(sigma_inv * r[i]);
      w[i] = (float) valid[i] * ((x < k) ? 1.0f : (k / x));
    }
  } else {
    const float t = 4.685f;
    const float t_i = (float) (1.0 / t);
    for(size_t i = 0; i < n; ++i) {
      const float x = r[i] * sigma_inv;
      float q = x * t_i;
      q = 1.0f - q * q;
      q = q * q;
      w[i] = (std::fabs(x) < t) ? q : 0.0f;
    }
    for(size_t i = n; i < N; ++i) {
      const float x = std::fabs(sigma_inv * r[i]);
      const float q = 1.0f - (t_i * x) * (t_i * x);
      w[i] = (float) valid[i] * ((x < 1e-6) ? 1.0f : (x > t) ? 0.0f : q * q);
    }
  }
}

// LinearSystemBuilder::Run -> LinearSystemBuilderReduction::Run serial branch + rankUpdatePoint + toEigen
// (bpvo/linear_system_builder.cc:140-221,239-266,334-350).  w' = w * float(valid); the 24-float packed upper 2x2-block
// buffer accumulates (w' * J[a]) * J[b]; G += (w' * r) * J; e += (w' * r) * r; all f32 in index order (Q15: serial).
// With nthreads > 1 the range is split in contiguous chunks that are summed in chunk order — the reference's
// tbb::parallel_reduce decomposition (:91-131, 233-237) made deterministic; used for the CPU baseline only.
float linearSystemRun(const std::vector<float>& J, const std::vector<float>& r, const std::vector<float>& w,
                      const std::vector<uint16_t>& valid, float H[36], float G[6], int nthreads)
{
  const size_t n = r.size();
  if(nthreads == -1) {      // test instrument (orc.h, PoseEstimator::reduction): the same terms, accumulated in f64
    double h[24] = {0}, g[6] = {0}, e = 0.0;
    for(size_t i = 0; i < n; ++i) {
      const float wi = w[i] * (float) valid[i];
      const float wR = wi * r[i];
      const float* j = J.data() + 6 * i;
      int ii = 0;
      for(int a = 0; a < 6; a += 2)
        for(int b = a; b < 6; b += 2) {
          h[ii++] += (double) ((wi * j[a]) * j[b]);
          h[ii++] += (double) ((wi * j[a]) * j[b + 1]);
          h[ii++] += (double) ((wi * j[a + 1]) * j[b]);
          h[ii++] += (double) ((wi * j[a + 1]) * j[b + 1]);
        }
      for(int a = 0; a < 6; ++a) g[a] += (double) ((wi * r[i]) * j[a]);
      e += (double) (wR * r[i]);
    }
    int ii = 0;
    for(int a = 0; a < 6; a += 2)
      for(int b = a; b < 6; b += 2) {
        H[a * 6 + b] = (float) h[ii++];
        H[a * 6 + b + 1] = (float) h[ii++];
        H[(a + 1) * 6 + b] = (float) h[ii++];
        H[(a + 1) * 6 + b + 1] = (float) h[ii++];
      }
    for(int a = 0; a < 6; ++a)
      for(int b = a + 1; b < 6; ++b) H[b * 6 + a] = H[a * 6 + b];
    for(int a = 0; a < 6; ++a) G[a] = (float) g[a];
    return (float) std::sqrt(e);
  }
  const int nchunks = std::max(1, nthreads);
  std::vector<float> part((size_t) nchunks * 32, 0.0f);
#pragma omp parallel for num_threads(nthreads) if(nthreads > 1)
  for(int ch = 0; ch < nchunks; ++ch) {
    const size_t i0 = n * ch / nchunks, i1 = n * (ch + 1) / nchunks;
    float h[24] = {0}, g[6] = {0}, e = 0.0f;
    for(size_t i = i0; i < i1; ++i) {
      const float wi = w[i] * (float) valid[i];
      const float wR = wi * r[i];
      const float* j = J.data() + 6 * i;
      int ii = 0;
      for(int a = 0; a < 6; a += 2)
        for(int b = a; b < 6; b += 2) {
          h[ii++] += (wi * j[a]) * j[b];
          h[ii++] += (wi * j[a]) * j[b + 1];
          h[ii++] += (wi * j[a + 1]) * j[b];
          h[ii++] += (wi * j[a + 1]) * j[b + 1];
        }
      for(int a = 0; a < 6; ++a) g[a] += (wi * r[i]) * j[a];
      e += wR * r[i];
    }
    float* p = part.data() + (size_t) ch * 32;
    std::memcpy(p, h, sizeof(h));
    std::memcpy(p + 24, g, sizeof(g));
    p[30] = e;
  }
  float h[24], g[6], e;
  std::memcpy(h, part.data(), sizeof(h));
  std::memcpy(g, part.data() + 24, sizeof(g));
  e = part[30];
  for(int ch = 1; ch < nchunks; ++ch) {
    const float* p = part.data() + (size_t) ch * 32;
    for(int k = 0; k < 24; ++k) h[k] += p[k];
    for(int k = 0; k < 6; ++k) g[k] += p[24 + k];
    e += p[30];
  }
  // toEigen (:207-221): unpack upper blocks, then mirror the upper triangle
  int ii = 0;
  for(int a = 0; a < 6; a += 2)
    for(int b = a; b < 6; b += 2) {
      H[a * 6 + b] = h[ii++];
      H[a * 6 + b + 1] = h[ii++];
      H[(a + 1) * 6 + b] = h[ii++];
      H[(a + 1) * 6 + b + 1] = h[ii++];
    }
  for(int a = 0; a < 6; ++a)
    for(int b = a + 1; b < 6; ++b) H[b * 6 + a] = H[a * 6 + b];
  std::memcpy(G, g, sizeof(g));
  return std::sqrt(e);
}

// ---------------------------------------------------------------------------------------------------------------
// Eigen::LDLT<Matrix<T,6,6>> [ext: Eigen 3.2.x Cholesky/LDLT.h, unpinned — restated]: in-place lower LDL^T with
// symmetric pivoting on the largest remaining |diagonal|; compute() + solve(); sums in index order.
template <typename T>
struct LDLT6 {
  T m[36];
  int tr[6];
  void compute(const T* A)
  {
    const int size = 6;
    for(int i = 0; i < 36; ++i) m[i] = A[i];
    T cutoff = 0;
    for(int k = 0; k < size; ++k) {
      int idx = k;
      T biggest = std::fabs(m[k * 6 + k]);
      for(int i = k + 1; i < size; ++i)
        if(std::fabs(m[i * 6 + i]) > biggest) { biggest = std::fabs(m[i * 6 + i]); idx = i; }
      if(k == 0) cutoff = std::fabs(std::numeric_limits<T>::epsilon() * biggest);
      if(biggest < cutoff) {              // not full rank: bail (3.2.x), remaining transpositions = identity
        for(int i = k; i < size; ++i) tr[i] = i;
        break;
      }
      tr[k] = idx;
      if(k != idx) {
        const int s = size - idx - 1;
        for(int c = 0; c < k; ++c) std::swap(m[k * 6 + c], m[idx * 6 + c]);                       // row(k).head(k) <-> row(idx).head(k)
        for(int r = 0; r < s; ++r) std::swap(m[(idx + 1 + r) * 6 + k], m[(idx + 1 + r) * 6 + idx]);  // col(k).tail(s) <-> col(idx).tail(s)
        std::swap(m[k * 6 + k], m[idx * 6 + idx]);
        for(int i = k + 1; i < idx; ++i) std::swap(m[i * 6 + k], m[idx * 6 + i]);
      }
      const int rs = size - k - 1;
      if(k > 0) {
        T temp[6];
        for(int c = 0; c < k; ++c) temp[c] = m[c * 6 + c] * m[k * 6 + c];        // D.head(k) * A10^T
        T dot = 0;
        for(int c = 0; c < k; ++c) dot += m[k * 6 + c] * temp[c];
        m[k * 6 + k] -= dot;
        for(int r = 0; r < rs; ++r) {
          T d2 = 0;
          for(int c = 0; c < k; ++c) d2 += m[(k + 1 + r) * 6 + c] * temp[c];
          m[(k + 1 + r) * 6 + k] -= d2;
        }
      }
      if(rs > 0 && std::fabs(m[k * 6 + k]) > cutoff)
        for(int r = 0; r < rs; ++r) m[(k + 1 + r) * 6 + k] /= m[k * 6 + k];
    }
  }
  void solve(const T* b, T* x) const
  {
    const int size = 6;
    for(int i = 0; i < size; ++i) x[i] = b[i];
    for(int i = 0; i < size; ++i) std::swap(x[i], x[tr[i]]);                     // P b
    for(int i = 0; i < size; ++i) {                                              // L^-1
      T s = x[i];
      for(int c = 0; c < i; ++c) s -= m[i * 6 + c] * x[c];
      x[i] = s;
    }
    const T tolerance = T(1) / std::numeric_limits<T>::max();                     // 3.2.2+: 1/highest
    for(int i = 0; i < size; ++i) {                                              // D^-1 (pseudo inverse)
      if(std::fabs(m[i * 6 + i]) > tolerance) x[i] /= m[i * 6 + i];
      else x[i] = 0;
    }
    for(int i = size - 1; i >= 0; --i) {                                         // L^-T
      T s = x[i];
      for(int c = i + 1; c < size; ++c) s -= m[c * 6 + i] * x[c];
      x[i] = s;
    }
    for(int i = size - 1; i >= 0; --i) std::swap(x[i], x[tr[i]]);                // P^T
  }
};

// (H*dp).isApprox(G) [ext: Eigen isApprox, dummy_precision 1e-5 (float) / 1e-12 (double)]:
// ||a-b||^2 <= prec^2 * min(||a||^2, ||b||^2)
template <typename T>
static bool isApproxHdpG(const T* H, const T* dp, const T* G, T prec)
{
  T a[6];
  for(int i = 0; i < 6; ++i) {
    T s = 0;
    for(int k = 0; k < 6; ++k) s += H[i * 6 + k] * dp[k];
    a[i] = s;
  }
  T d2 = 0, na = 0, nb = 0;
  for(int i = 0; i < 6; ++i) { d2 += (a[i] - G[i]) * (a[i] - G[i]); na += a[i] * a[i]; nb += G[i] * G[i]; }
  return d2 <= prec * prec * std::min(na, nb);
}

// PoseEstimatorData_::solve (bpvo/pose_estimator_base.h:90-111) and solve2Augmented (:136-148):
// f32 LDLT; if (H*dp).isApprox(G) fails, f64 LDLT of H + 1e-3*max(diag H)*I.
bool solveSystem(const float H[36], const float G[6], float dp[6])
{
  LDLT6<float> s;
  s.compute(H);
  s.solve(G, dp);
  if(isApproxHdpG<float>(H, dp, G, 1e-5f)) return true;

  float maxd = H[0];
  for(int i = 1; i < 6; ++i) maxd = std::max(maxd, H[i * 6 + i]);
  const double uu = 0.001 * maxd;
  double Hd[36], Gd[6], dpd[6];
  for(int i = 0; i < 36; ++i) Hd[i] = H[i];
  for(int i = 0; i < 6; ++i) { Gd[i] = G[i]; Hd[i * 6 + i] += uu; }
  LDLT6<double> sd;
  sd.compute(Hd);
  sd.solve(Gd, dpd);
  const bool ok = isApproxHdpG<double>(Hd, dpd, Gd, 1e-12);
  for(int i = 0; i < 6; ++i) dp[i] = (float) dpd[i];
  return ok;
}

}  // namespace orc
