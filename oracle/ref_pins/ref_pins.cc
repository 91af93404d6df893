// ORACLE — TEST INFRASTRUCTURE ONLY.
// Thin C wrappers around the few pieces of the reference on (or next to) the hot path that compile WITHOUT Eigen / OpenCV:
//   bpvo::median              bpvo/utils.h:224-252   (the selection rule of the robust scale)
//   v128 operators            bpvo/v128.h:36-130     (the byte comparisons of the census transform, bpvo/census.cc:42-57)
//   bpvo::ConfigFile, icompare bpvo/config_file.{h,cc}, bpvo/utils.{h,cc}   (the conf/*.cfg reader)
//   simd::dot, simd::abs      bpvo/simd.h:60-80      (4-float dot product of the projectPoints formulation, |x| of the saliency map)
//   Huber, Tukey              bpvo/robust_loss.h:50-72 (the reference's OTHER statement of the two M-estimator weights: w(r / (sigma k));
//                             no translation unit includes it — the running code is the SIMD body of bpvo/mestimator.cc, which needs
//                             Eigen — so it pins the weight FUNCTION of the restatement to rounding, not its bits)
// The reference sources are compiled where they lie under /root/reference (oracle/Makefile, target `ref`); nothing is
// copied.  The wrappers only marshal arguments; census_bytes() composes the reference's operators in the order
// censusOp (bpvo/census.cc:42-57) does, because that function itself sits in a translation unit that needs OpenCV.
#include <algorithm>
#include <cmath>
#include <bpvo/robust_loss.h>   // self-contained but for <cmath> / <algorithm> (std::max, std::fabs), included above
#include <bpvo/config_file.h>
#include <bpvo/simd.h>
#include <bpvo/utils.h>
#include <bpvo/v128.h>

#include <cstring>
#include <string>
#include <vector>

extern "C" {

float ref_median(const float* data, size_t n)
{
  std::vector<float> v(data, data + n);
  return bpvo::median(v);
}

void ref_v128_ge(const uint8_t a[16], const uint8_t b[16], uint8_t out[16])
{
  const bpvo::v128 r = bpvo::v128(a) >= bpvo::v128(b);
  _mm_storeu_si128((__m128i*) out, r);
}

// nbr[k]: 16 bytes of neighbour k in censusOp's order; c: the 16 centre bytes
void ref_census_bytes(const uint8_t nbr[8][16], const uint8_t c16[16], uint8_t out[16])
{
  using bpvo::v128;
  const v128 c(c16);
  const v128 K[8] = {v128(0x01), v128(0x02), v128(0x04), v128(0x08), v128(0x10), v128(0x20), v128(0x40), v128(0x80)};
  v128 r = (v128(nbr[0]) >= c) & K[0];
  for(int k = 1; k < 8; ++k) r = r | ((v128(nbr[k]) >= c) & K[k]);
  _mm_storeu_si128((__m128i*) out, r);
}

float ref_simd_dot(const float a[4], const float b[4]) { return bpvo::simd::dot(_mm_loadu_ps(a), _mm_loadu_ps(b)); }
void ref_simd_abs(const float a[4], float out[4]) { _mm_storeu_ps(out, bpvo::simd::abs(_mm_loadu_ps(a))); }

float ref_huber_weight(float sigma, float r) { return bpvo::Huber(sigma).weight(r); }
float ref_tukey_weight(float sigma, float r) { return bpvo::Tukey(sigma).weight(r); }

int ref_icompare(const char* a, const char* b) { return bpvo::icompare(a, b) ? 1 : 0; }

// 0: ok, 1: key missing (default returned), 2: ConfigFile threw (message in buf)
int ref_config_get(const char* filename, const char* key, const char* def, char* buf, size_t buflen)
{
  try {
    bpvo::ConfigFile cf{std::string(filename)};
    const std::string v = cf.get<std::string>(key, std::string("\x01missing"));
    const bool missing = v == "\x01missing";
    std::strncpy(buf, missing ? def : v.c_str(), buflen - 1);
    buf[buflen - 1] = 0;
    return missing ? 1 : 0;
  } catch(const std::exception& e) {
    std::strncpy(buf, e.what(), buflen - 1);
    buf[buflen - 1] = 0;
    return 2;
  }
}

}  // extern "C"
