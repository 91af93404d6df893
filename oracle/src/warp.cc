// ORACLE — test infrastructure only (see orc.h).  RigidBodyWarp, Hartley normalisation, SE(3) exponential.
#include "orc.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace orc {

M44 identity44()
{
  M44 r;
  for(int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  return r;
}

// Eigen fixed-size 4x4 f32 product [ext: Eigen 3.2 CoeffBasedProduct, unrolled k = 0..3, mul+add]:
// each coefficient is ((a0*b0 + a1*b1) + a2*b2) + a3*b3.
M44 mul44(const M44& a, const M44& b)
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

// Matrix4f::inverse() as used at bpvo/vo.cc:153,171 on rigid transforms.  [ext: Eigen uses a cofactor/SSE routine;
// unpinned]  Restated as the plain cofactor expansion in f32.
M44 inverse44(const M44& A)
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

void Warp::init(const float K_[9], float b_)
{
  std::memcpy(K, K_, sizeof(K));
  b = b_;
  T = identity44();
  T_inv = identity44();   // bpvo/rigid_body_warp.cc:27-28
  std::memset(P, 0, sizeof(P));
}

// bpvo/rigid_body_warp.h:47-60: Z = (b*fx) * (1.0 / d) with the reciprocal in double, rest in float.
void Warp::makePoint(float x, float y, float d, float out[4]) const
{
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  if(dspace) {   // DisparitySpaceWarp::makePoint (bpvo/disparity_space_warp.h:31-34)
    out[0] = x - cx; out[1] = y - cy; out[2] = d; out[3] = 1.0f;
    return;
  }
  const float Bf = b * fx;
  const float Z = (float) (Bf * (1.0 / d));
  const float X = (x - cx) * Z * (1.0f / fx);
  const float Y = (y - cy) * Z * (1.0f / fy);
  out[0] = X; out[1] = Y; out[2] = Z; out[3] = 1.0f;
}

// HartlyNormalization (bpvo/warps.cc:27-48) + setNormalization (bpvo/rigid_body_warp.h:62-71).
// Sequential f32 sums over the points in order; (p - c).norm() = sqrt((d0^2 + d1^2) + (d2^2 + d3^2))
// [ext: Eigen SSE3 predux of one Packet4f]; s = sqrt(3.0)/max(m,1e-6f) in double then float.
// T_inv: the reference calls the general Matrix4f::inverse(); for T = [sI, -s c; 0 1] the exact inverse [I/s, c; 0 1]
// is used.
void Warp::setNormalization(const std::vector<float>& pts)
{
  if(dspace) return;   // "no normalization for dspace, we do not need it" (bpvo/disparity_space_warp.h:87-90)
  const size_t N = pts.size() / 4;
  float c[4] = {0, 0, 0, 0};
  for(size_t i = 0; i < N; ++i)
    for(int k = 0; k < 4; ++k) c[k] += pts[4 * i + k];
  for(int k = 0; k < 4; ++k) c[k] /= (float) N;

  float m = 0.0f;
  for(size_t i = 0; i < N; ++i) {
    const float d0 = pts[4 * i + 0] - c[0], d1 = pts[4 * i + 1] - c[1], d2 = pts[4 * i + 2] - c[2], d3 = pts[4 * i + 3] - c[3];
    m += std::sqrt((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
  }
  m /= (float) N;

  const float s = (float) (std::sqrt(3.0) / std::max(m, 1e-6f));
  T = identity44();
  T.m[0] = s; T.m[5] = s; T.m[10] = s;
  T.m[3] = -s * c[0]; T.m[7] = -s * c[1]; T.m[11] = -s * c[2];
  T_inv = identity44();
  const float si = 1.0f / s;
  T_inv.m[0] = si; T_inv.m[5] = si; T_inv.m[10] = si;
  T_inv.m[3] = c[0]; T_inv.m[7] = c[1]; T_inv.m[11] = c[2];
}

// bpvo/rigid_body_warp.h:111-114: P = K * T.block<3,4>(0,0), f32, index-order sums over k = 0..2.
void Warp::setPose(const M44& pose)
{
  if(dspace) {
    // DisparitySpaceWarp::setPose: _H = _G * T * _G_inv (bpvo/disparity_space_warp.h:36), G and G_inv as the constructor
    // fills them (bpvo/disparity_space_warp.cc:26-47: 1.0/fx etc. in double, narrowed by the << initialiser); the two
    // fixed 4x4 f32 products left to right.
    const float fx = K[0], fy = K[4];
    M44 G, Gi;
    for(int i = 0; i < 16; ++i) G.m[i] = Gi.m[i] = 0.0f;
    G.m[0] = fx; G.m[5] = fy; G.m[11] = fx * b; G.m[14] = 1.0f;
    Gi.m[0] = (float) (1.0 / fx); Gi.m[5] = (float) (1.0 / fy); Gi.m[11] = 1.0f; Gi.m[14] = (float) (1.0 / (fx * b));
    const M44 H = mul44(mul44(G, pose), Gi);
    for(int j = 0; j < 4; ++j) { P[0 * 4 + j] = H.m[0 * 4 + j]; P[1 * 4 + j] = H.m[1 * 4 + j]; P[2 * 4 + j] = H.m[3 * 4 + j]; }
    return;
  }
  for(int i = 0; i < 3; ++i)
    for(int j = 0; j < 4; ++j) {
      float s = K[i * 3 + 0] * pose.m[0 * 4 + j];
      s += K[i * 3 + 1] * pose.m[1 * 4 + j];
      s += K[i * 3 + 2] * pose.m[2 * 4 + j];
      P[i * 4 + j] = s;
    }
}

// RigidBodyWarp::computeJacobian (bpvo/rigid_body_warp.cc:60-315), operation order of the six SSE passes.  The
// reference's div_ps(a, b) is _mm_mul_ps(a, _mm_rcp_ps(b)) (:47-58): a multiply by an approximate, vendor-specific
// 12-bit reciprocal.  Q13 deviation: the multiply structure is kept with the correctly rounded reciprocal 1.0f / b in
// place of _mm_rcp_ps.  The formulas equal the scalar jacobian() of bpvo/rigid_body_warp.h:94-106.
// Output [N][6] row-major (rigid_body_warp.cc:304-305).
void Warp::computeJacobian(const float* pts, int N, const float* IxIy, float* J) const
{
  const float fx = K[0], fy = K[4];
  if(dspace) {
    // DisparitySpaceWarp::jacobian (bpvo/disparity_space_warp.h:40-64), scalar f32, C++ evaluation order
    const float t5 = 1.0f / fx, t6 = 1.0f / fy, t7 = 1.0f / b;        // _fx_i, _fy_i, _b_i (disparity_space_warp.cc:32-34)
    for(int i = 0; i < N; ++i) {
      const float x = pts[4 * i + 0], y = pts[4 * i + 1], d = pts[4 * i + 2];
      const float Ix = IxIy[2 * i + 0], Iy = IxIy[2 * i + 1];
      const float t2 = x * Ix, t3 = y * Iy, t4 = t2 + t3;
      float* Ji = J + 6 * (size_t) i;
      Ji[0] = ((-Iy) * fy) - ((t4 * t6) * y);
      Ji[1] = (Ix * fx) + ((t4 * t5) * x);
      Ji[2] = (((Iy * fy) * t5) * x) - (((Ix * fx) * t6) * y);
      Ji[3] = (Ix * d) * t7;
      Ji[4] = (((Iy * d) * fy) * t5) * t7;
      Ji[5] = (((-d) * t4) * t5) * t7;
    }
    return;
  }
  const float s = T.m[0], c1 = T_inv.m[3], c2 = T_inv.m[7], c3 = T_inv.m[11];
  const float s_i = (float) (1.0 / s);                       // _mm_set1_ps(1.0 / s), rigid_body_warp.cc:272
  for(int i = 0; i < N; ++i) {
    const float x = pts[4 * i + 0], y = pts[4 * i + 1], z = pts[4 * i + 2];
    const float Ix = fx * IxIy[2 * i + 0];
    const float Iy = fy * IxIy[2 * i + 1];
    const float xIx_yIy = x * Ix + y * Iy;
    const float rz = 1.0f / z, rz2 = 1.0f / (z * z), rzs = 1.0f / (z * s);
    float* Ji = J + 6 * (size_t) i;
    Ji[0] = (-((Iy * (z - c3)) * rz)) - ((xIx_yIy * (y - c2)) * rz2);        // :80-96
    Ji[1] = ((Ix * (z - c3)) * rz) + ((xIx_yIy * (x - c1)) * rz2);           // :131-137
    Ji[2] = ((Iy * (x - c1)) - (Ix * (y - c2))) * rz;                        // :173-178
    Ji[3] = Ix * rzs;                                                        // :205-206
    Ji[4] = Iy * rzs;                                                        // :232-233
    Ji[5] = -((s_i * xIx_yIy) * rz2);                                        // :295-299
  }
}

// math::TwistToMatrix<float> (bpvo/math_utils.h:140-168): theta = ||w|| in f32; sin/cos/reciprocal evaluated in
// double on the promoted f32 theta and narrowed to f32; S = t_i * skew(w); S2 = S*S;
// R = I + a*S + b*S2; t = (I + (b*t_i)*S + ((theta - a)*t_i)*S2) * v.
M44 twistToMatrix(const float p[6])
{
  M44 ret = identity44();
  const float theta = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  if(theta > 1e-8) {
    const float a = (float) ::sin((double) theta);
    const float b = (float) (1.0 - ::cos((double) theta));
    const float t_i = (float) (1.0 / theta);
    const float w0 = p[0], w1 = p[1], w2 = p[2];
    float S[9] = {t_i * 0.0f, t_i * -w2, t_i * w1, t_i * w2, t_i * 0.0f, t_i * -w0, t_i * -w1, t_i * w0, t_i * 0.0f};
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

// paramsToPose = scalePose(TwistToMatrix(p)) = T_inv * Tw * T (bpvo/rigid_body_warp.h:130-138), left to right.
M44 Warp::paramsToPose(const float p[6]) const
{
  if(dspace) return twistToMatrix(p);   // bpvo/disparity_space_warp.h:79-84; scalePose is the identity (:91)
  return mul44(mul44(T_inv, twistToMatrix(p)), T);
}

}  // namespace orc
