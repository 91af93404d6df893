#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
  const unsigned x = __builtin_amdgcn_s_getreg(63508) & 0xf;   // HW_REG_XCC_ID
  asm volatile("buffer_inv sc0" ::: "memory");
  if(threadIdx.x == 0) out[blockIdx.x] = x;
}
int main()
{
  unsigned* d; (void) hipMalloc(&d, 4096 * 4);
  hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d);
  unsigned h[64]; (void) hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for(int i = 0; i < 64; ++i) printf("%u%c", h[i], (i % 8 == 7) ? '\n' : ' ');
  return 0;
}
