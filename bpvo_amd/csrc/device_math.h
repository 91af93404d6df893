// Small fixed-size math shared by the host driver and the device-side Gauss-Newton step (gn_step kernel).
// Every routine states the reference lines whose arithmetic (order of operations, precision) it follows; the
// library is built with -ffp-contract=off so that no a*b+c is fused on either side.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define BPVO_HD __host__ __device__ inline

namespace bpvo_hip {

struct M44 { float m[16]; };   // row-major

BPVO_HD M44 m44_identity()
{
  M44 r;
  for(int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  return r;
}

// 4x4 f32 product with index-order sums ((a0b0 + a1b1) + a2b2) + a3b3 — how Eigen evaluates the fixed-size products at
// bpvo/rigid_body_warp.h:130 (scalePose) and bpvo/pose_estimator_base.h:371,390 (data.T *= ...).
BPVO_HD M44 m44_mul(const M44& a, const M44& b)
{
  M44 r;
  for(int i = 0; i < 4; ++i)
    for(int j = 0; j < 4; ++j) {
      float s = a.m[i * 4 + 0] * b.m[0 * 4 + j];
      s += a.m[i * 4 + 1] * b.m[1 * 4 + j];
      s += a.m[i * 4 + 2] * b.m[2 * 4 + j];
      s += a.m[i * 4 + 3] * b.m[3 * 4 + j];
      r.m[i * 4 + j] = s;
    }
  return r;
}

// General 4x4 inverse by cofactors in f32 (Matrix44::inverse() at bpvo/vo.cc:153,171).
BPVO_HD M44 m44_inverse(const M44& A)
{
  const float* m = A.m;
  float inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const float idet = 1.0f / det;
  M44 r;
  for(int i = 0; i < 16; ++i) r.m[i] = inv[i] * idet;
  return r;
}

// math::TwistToMatrix<float> (reference: bpvo/math_utils.h:140-168): theta in f32; sin, 1-cos and 1/theta evaluated in
// double on the promoted theta and narrowed; R = I + a*S + b*S^2; t = (I + (b*t_i)*S + ((theta-a)*t_i)*S^2) * v.
BPVO_HD M44 twist_to_matrix(const float p[6])
{
  M44 ret = m44_identity();
  const float theta = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  if((double) theta > 1e-8) {
    // (device: one argument reduction for both — the library's sin and cos are the two halves of its sincos, bit for bit)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BPVO_TWIST_SEPARATE_SIN_COS)
    double sn, cs;
    sincos((double) theta, &sn, &cs);
#else
    const double sn = sin((double) theta), cs = cos((double) theta);
#endif
    const float a = (float) sn;
    const float b = (float) (1.0 - cs);
    const float t_i = (float) (1.0 / (double) theta);
    const float S[9] = {t_i * 0.0f, t_i * -p[2], t_i * p[1], t_i * p[2], t_i * 0.0f, t_i * -p[0], t_i * -p[1], t_i * p[0], t_i * 0.0f};
    float S2[9];
    for(int i = 0; i < 3; ++i)
      for(int j = 0; j < 3; ++j) {
        float v = S[i * 3 + 0] * S[0 * 3 + j];
        v += S[i * 3 + 1] * S[1 * 3 + j];
        v += S[i * 3 + 2] * S[2 * 3 + j];
        S2[i * 3 + j] = v;
      }
    const float bt = b * t_i, ct = (theta - a) * t_i;
    float V[9];
    for(int i = 0; i < 3; ++i)
      for(int j = 0; j < 3; ++j) {
        const float I = (i == j) ? 1.0f : 0.0f;
        ret.m[i * 4 + j] = (I + a * S[i * 3 + j]) + b * S2[i * 3 + j];
        V[i * 3 + j] = (I + bt * S[i * 3 + j]) + ct * S2[i * 3 + j];
      }
    for(int i = 0; i < 3; ++i) {
      float v = V[i * 3 + 0] * p[3];
      v += V[i * 3 + 1] * p[4];
      v += V[i * 3 + 2] * p[5];
      ret.m[i * 4 + 3] = v;
    }
  } else {
    ret.m[3] = p[3]; ret.m[7] = p[4]; ret.m[11] = p[5];
  }
  return ret;
}

// Hartley normalisation T = [sI, -s c; 0 1] is stored as (s, c1, c2, c3); T_inv = [I/s, c; 0 1].
// paramsToPose(p) = T_inv * TwistToMatrix(p) * T, evaluated left to right (reference: bpvo/rigid_body_warp.h:130-138).
BPVO_HD M44 params_to_pose(const float nrm[4], const float p[6])
{
  const float s = nrm[0];
  M44 T = m44_identity(), Ti = m44_identity();
  T.m[0] = s; T.m[5] = s; T.m[10] = s;
  T.m[3] = -s * nrm[1]; T.m[7] = -s * nrm[2]; T.m[11] = -s * nrm[3];
  const float si = 1.0f / s;
  Ti.m[0] = si; Ti.m[5] = si; Ti.m[10] = si;
  Ti.m[3] = nrm[1]; Ti.m[7] = nrm[2]; Ti.m[11] = nrm[3];
  return m44_mul(m44_mul(Ti, twist_to_matrix(p)), T);
}

// ---- Eigen::LDLT<6x6> semantics (symmetric pivoting on the largest remaining |diagonal|, unit-lower L, pseudo-inverse
// of D with tolerance 1/highest), restated; used by PoseEstimatorData_::solve (reference: bpvo/pose_estimator_base.h:90-148).
template <typename T>
struct LDLT6 {
  T m[36];
  T temp[6];
  int tr[6];
  BPVO_HD void compute(const T* A, T eps)
  {
    for(int i = 0; i < 36; ++i) m[i] = A[i];
    T cutoff = 0;
    for(int k = 0; k < 6; ++k) {
      int idx = k;
      T biggest = fabs(m[k * 6 + k]);
      for(int i = k + 1; i < 6; ++i) {
        const T v = fabs(m[i * 6 + i]);
        if(v > biggest) { biggest = v; idx = i; }
      }
      if(k == 0) cutoff = fabs(eps * biggest);
      if(biggest < cutoff) {
        for(int i = k; i < 6; ++i) tr[i] = i;
        break;
      }
      tr[k] = idx;
      if(k != idx) {
        const int s = 6 - idx - 1;
        for(int c = 0; c < k; ++c) { T t = m[k * 6 + c]; m[k * 6 + c] = m[idx * 6 + c]; m[idx * 6 + c] = t; }
        for(int r = 0; r < s; ++r) {
          T t = m[(idx + 1 + r) * 6 + k]; m[(idx + 1 + r) * 6 + k] = m[(idx + 1 + r) * 6 + idx]; m[(idx + 1 + r) * 6 + idx] = t;
        }
        { T t = m[k * 6 + k]; m[k * 6 + k] = m[idx * 6 + idx]; m[idx * 6 + idx] = t; }
        for(int i = k + 1; i < idx; ++i) { T t = m[i * 6 + k]; m[i * 6 + k] = m[idx * 6 + i]; m[idx * 6 + i] = t; }
      }
      const int rs = 6 - k - 1;
      if(k > 0) {
        for(int c = 0; c < k; ++c) temp[c] = m[c * 6 + c] * m[k * 6 + c];
        T dot = 0;
        for(int c = 0; c < k; ++c) dot += m[k * 6 + c] * temp[c];
        m[k * 6 + k] -= dot;
        for(int r = 0; r < rs; ++r) {
          T d2 = 0;
          for(int c = 0; c < k; ++c) d2 += m[(k + 1 + r) * 6 + c] * temp[c];
          m[(k + 1 + r) * 6 + k] -= d2;
        }
      }
      if(rs > 0 && fabs(m[k * 6 + k]) > cutoff)
        for(int r = 0; r < rs; ++r) m[(k + 1 + r) * 6 + k] /= m[k * 6 + k];
    }
  }
  BPVO_HD void solve(const T* b, T* x, T tolerance) const
  {
    for(int i = 0; i < 6; ++i) x[i] = b[i];
    for(int i = 0; i < 6; ++i) { T t = x[i]; x[i] = x[tr[i]]; x[tr[i]] = t; }
    for(int i = 0; i < 6; ++i) {
      T s = x[i];
      for(int c = 0; c < i; ++c) s -= m[i * 6 + c] * x[c];
      x[i] = s;
    }
    for(int i = 0; i < 6; ++i) {
      if(fabs(m[i * 6 + i]) > tolerance) x[i] /= m[i * 6 + i];
      else x[i] = 0;
    }
    for(int i = 5; i >= 0; --i) {
      T s = x[i];
      for(int c = i + 1; c < 6; ++c) s -= m[c * 6 + i] * x[c];
      x[i] = s;
    }
    for(int i = 5; i >= 0; --i) { T t = x[i]; x[i] = x[tr[i]]; x[tr[i]] = t; }
  }
};

template <typename T>
BPVO_HD bool is_approx_Hdp_G(const T* H, const T* dp, const T* G, T prec)
{
  T d2 = 0, na = 0, nb = 0;
  for(int i = 0; i < 6; ++i) {
    T a = 0;
    for(int k = 0; k < 6; ++k) a += H[i * 6 + k] * dp[k];
    d2 += (a - G[i]) * (a - G[i]); na += a * a; nb += G[i] * G[i];
  }
  const T mn = na < nb ? na : nb;
  return d2 <= prec * prec * mn;
}

// The same f32 factorisation + solve with every array index a compile-time constant: loops are fully unrolled and the
// pivot transpositions are applied as conditional swaps over the (statically enumerated) candidates, so the 6x6 matrix
// lives in registers on the GPU (no LDS / scratch round trips in the serial gn_step).  Operation order identical to
// LDLT6<float>::compute + solve above.
#define BPVO_SWAPF(a, b) do { const float t_ = (a); (a) = (b); (b) = t_; } while(0)
BPVO_HD void ldlt6_solve_f32(const float* A, const float* rhs, float* x)
{
  const float eps = 1.1920928955078125e-07f, tolerance = 1.0f / 3.4028234663852886e+38f;
  float m[36];
  int tr[6];
#pragma unroll
  for(int i = 0; i < 36; ++i) m[i] = A[i];
  float cutoff = 0.0f;
  bool stopped = false;
#pragma unroll
  for(int k = 0; k < 6; ++k) {
    int idx = k;
    if(!stopped) {
      float biggest = fabsf(m[k * 6 + k]);
#pragma unroll
      for(int i = k + 1; i < 6; ++i) {
        const float v = fabsf(m[i * 6 + i]);
        if(v > biggest) { biggest = v; idx = i; }
      }
      if(k == 0) cutoff = fabsf(eps * biggest);
      if(biggest < cutoff) { stopped = true; idx = k; }
    }
    tr[k] = idx;
    if(!stopped) {
#pragma unroll
      for(int i = k + 1; i < 6; ++i) {
        if(idx == i) {     // symmetric transposition k <-> i on the lower triangle (static indices)
#pragma unroll
          for(int c = 0; c < k; ++c) BPVO_SWAPF(m[k * 6 + c], m[i * 6 + c]);
#pragma unroll
          for(int r = i + 1; r < 6; ++r) BPVO_SWAPF(m[r * 6 + k], m[r * 6 + i]);
          BPVO_SWAPF(m[k * 6 + k], m[i * 6 + i]);
#pragma unroll
          for(int t = k + 1; t < i; ++t) BPVO_SWAPF(m[t * 6 + k], m[i * 6 + t]);
        }
      }
      float temp[6];
      if(k > 0) {
#pragma unroll
        for(int c = 0; c < k; ++c) temp[c] = m[c * 6 + c] * m[k * 6 + c];
        float dot = 0.0f;
#pragma unroll
        for(int c = 0; c < k; ++c) dot += m[k * 6 + c] * temp[c];
        m[k * 6 + k] -= dot;
#pragma unroll
        for(int r = k + 1; r < 6; ++r) {
          float d2 = 0.0f;
#pragma unroll
          for(int c = 0; c < k; ++c) d2 += m[r * 6 + c] * temp[c];
          m[r * 6 + k] -= d2;
        }
      }
      if(k < 5 && fabsf(m[k * 6 + k]) > cutoff) {
#pragma unroll
        for(int r = k + 1; r < 6; ++r) m[r * 6 + k] /= m[k * 6 + k];
      }
    }
  }
  float v[6];
#pragma unroll
  for(int i = 0; i < 6; ++i) v[i] = rhs[i];
#pragma unroll
  for(int i = 0; i < 6; ++i) {          // P b
#pragma unroll
    for(int q = i + 1; q < 6; ++q) if(tr[i] == q) BPVO_SWAPF(v[i], v[q]);
  }
#pragma unroll
  for(int i = 0; i < 6; ++i) {          // L^-1
    float s = v[i];
#pragma unroll
    for(int c = 0; c < i; ++c) s -= m[i * 6 + c] * v[c];
    v[i] = s;
  }
#pragma unroll
  for(int i = 0; i < 6; ++i) {          // D^-1 (pseudo-inverse)
    if(fabsf(m[i * 6 + i]) > tolerance) v[i] /= m[i * 6 + i];
    else v[i] = 0.0f;
  }
#pragma unroll
  for(int i = 5; i >= 0; --i) {         // L^-T
    float s = v[i];
#pragma unroll
    for(int c = i + 1; c < 6; ++c) s -= m[c * 6 + i] * v[c];
    v[i] = s;
  }
#pragma unroll
  for(int i = 5; i >= 0; --i) {         // P^T
#pragma unroll
    for(int q = i + 1; q < 6; ++q) if(tr[i] == q) BPVO_SWAPF(v[i], v[q]);
  }
#pragma unroll
  for(int i = 0; i < 6; ++i) x[i] = v[i];
}
#undef BPVO_SWAPF

// Working storage of solve_system.  The factorisation indexes its arrays with run-time pivots, which would put
// function-local arrays into (slow) scratch memory on the GPU: the device caller hands in an LDS-resident instance.
struct SolveScratch {
  LDLT6<float> f;
  LDLT6<double> d;
  double Hd[36], Gd[6], dpd[6];
};

// PoseEstimatorData_::solve + solve2Augmented(0.001) (reference: bpvo/pose_estimator_base.h:90-111,136-148).
BPVO_HD bool solve_system(const float H[36], const float G[6], float dp[6], SolveScratch* ws)
{
  ldlt6_solve_f32(H, G, dp);
  if(is_approx_Hdp_G<float>(H, dp, G, 1e-5f)) return true;
  float maxd = H[0];
  for(int i = 1; i < 6; ++i) maxd = H[i * 6 + i] > maxd ? H[i * 6 + i] : maxd;
  const double u = 0.001 * (double) maxd;
  for(int i = 0; i < 36; ++i) ws->Hd[i] = (double) H[i];
  for(int i = 0; i < 6; ++i) { ws->Gd[i] = (double) G[i]; ws->Hd[i * 6 + i] += u; }
  ws->d.compute(ws->Hd, 2.220446049250313e-16);
  ws->d.solve(ws->Gd, ws->dpd, 1.0 / 1.7976931348623157e+308);
  const bool ok = is_approx_Hdp_G<double>(ws->Hd, ws->dpd, ws->Gd, 1e-12);
  for(int i = 0; i < 6; ++i) dp[i] = (float) ws->dpd[i];
  return ok;
}

}  // namespace bpvo_hip
