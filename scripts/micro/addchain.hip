// How long does a chain of dependent v_add_f32 take on an otherwise idle chip — and at which shader clock?  (The Hartley sums of a
// dense template are such a chain: bpvo_amd/csrc/kernels_frame.hip, normalization_kernel.)  One wave, N dependent adds, plain and
// with a DPP operand; HIP events for the time, s_memtime (shader clock) and s_memrealtime (100 MHz) for the clock.  Optionally a
// second stream keeps the rest of the chip busy (argv[1] = 1) to see whether the clock depends on the load.
//   hipcc --offload-arch=gfx950 -O3 -o addchain scripts/micro/addchain.hip && ./addchain [busy]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void chain(float* out, int n, int dpp, unsigned long long* clk)
{
  float acc = out[threadIdx.x], v = 1.0f + out[64 + threadIdx.x];
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  if(dpp) {
    for(int i = 0; i < n; i += 16)
      asm volatile("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %0, %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                   "v_add_f32_dpp %0, %1, %0 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                   "v_add_f32_dpp %0, %1, %0 row_shl:5 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %0, %1, %0 row_shl:6 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                   "v_add_f32_dpp %0, %1, %0 row_shl:7 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                   "v_add_f32_dpp %0, %1, %0 row_shl:9 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %0, %1, %0 row_shl:10 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                   "v_add_f32_dpp %0, %1, %0 row_shl:11 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %0, %1, %0 row_shl:12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                   "v_add_f32_dpp %0, %1, %0 row_shl:13 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %0, %1, %0 row_shl:14 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                   "v_add_f32_dpp %0, %1, %0 row_shl:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32 %0, %0, %1\n" : "+v"(acc) : "v"(v));
  } else {
    for(int i = 0; i < n; i += 16)
      asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                   "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n" : "+v"(acc) : "v"(v));
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = acc;
  if(threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
__global__ void busy(float* out, int iters)
{
  float a = out[threadIdx.x & 63];
  for(int i = 0; i < iters; ++i) a = a * 1.0000001f + 0.5f;
  if(a == 123.0f) out[0] = a;
}
int main(int argc, char** argv)
{
  const int with_busy = argc > 1 ? atoi(argv[1]) : 0;
  float* d; unsigned long long* clk; unsigned long long h[2];
  hipMalloc(&d, 4096); hipMemset(d, 0, 4096); hipMalloc(&clk, 16);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int n = 1 << 20;
  for(int dpp = 0; dpp < 2; ++dpp)
    for(int rep = 0; rep < 3; ++rep) {
      if(with_busy) hipLaunchKernelGGL(busy, dim3(256 * 8), dim3(256), 0, s2, d + 512, 4000000);
      hipEventRecord(a, s1);
      hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, n, dpp, clk);
      hipEventRecord(b, s1);
      hipStreamSynchronize(s1);
      float ms; hipEventElapsedTime(&ms, a, b);
      hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      printf("%s busy=%d: %d dependent adds in %.1f us = %.2f ns per add; s_memtime ticks %llu (%.2f per add), 100 MHz ticks %llu -> s_memtime at %.0f MHz\n", dpp ? "dpp  " : "plain", with_busy, n,
             1e3 * ms, 1e6 * ms / n, h[0], (double) h[0] / n, h[1], 100.0 * h[0] / h[1]);
      hipDeviceSynchronize();
    }
  return 0;
}
