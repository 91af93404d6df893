#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "gn_common.h"
using namespace bpvo_hip;
__global__ void k(const unsigned* in, unsigned* scan, unsigned* sum, unsigned* ref)
{
  const unsigned x = in[blockIdx.x * 64 + threadIdx.x];
  scan[blockIdx.x * 64 + threadIdx.x] = wave_incl_scan_u32(x);
  sum[blockIdx.x * 64 + threadIdx.x] = wave_sum_u32(x);
  unsigned incl = x; const int lane = threadIdx.x;
  for(int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if(lane >= o) incl += t; }
  ref[blockIdx.x * 64 + threadIdx.x] = incl;
}
int main()
{
  const int B = 1024; std::vector<unsigned> h(B * 64); std::mt19937 r(3); for(auto& v : h) v = r() % 9;
  unsigned *d, *a, *b, *c; (void) hipMalloc(&d, h.size() * 4); (void) hipMalloc(&a, h.size() * 4); (void) hipMalloc(&b, h.size() * 4); (void) hipMalloc(&c, h.size() * 4);
  (void) hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, d, a, b, c);
  std::vector<unsigned> ha(h.size()), hb(h.size()), hc(h.size());
  (void) hipMemcpy(ha.data(), a, h.size() * 4, hipMemcpyDeviceToHost); (void) hipMemcpy(hb.data(), b, h.size() * 4, hipMemcpyDeviceToHost); (void) hipMemcpy(hc.data(), c, h.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for(size_t i = 0; i < h.size(); ++i) { if(ha[i] != hc[i]) ++bad; if(hb[i] != hc[(i | 63)]) ++bad; }
  printf("wave_incl_scan_u32 / wave_sum_u32 vs shuffle ladder: %zu values, %ld differ\n", h.size(), bad);
  return bad != 0;
}
