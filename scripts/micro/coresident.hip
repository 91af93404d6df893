// Can a second kernel run NEXT TO a persistent kernel that holds one 256-thread workgroup on every CU?  (The question behind running the frame
// stage of the finest level under the team kernel)  A spinning kernel of one workgroup per CU with a given dynamic LDS size
// and register footprint on stream 1; 2 ms later a small kernel (256 or 1024 threads per workgroup, 512 workgroups) on stream 2; reported: when
// the small kernel finished relative to the spinner's start and end.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/micro/coresident scripts/micro/coresident.hip && scripts/micro/coresident
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

// SCRATCH: a dynamically indexed private array (the team kernel carries 708 B of scratch per lane)
template <int REGS, bool SCRATCH = false>
__global__ __launch_bounds__(256) void spinner(float* out, long long ticks)
{
  extern __shared__ float lds[];
  float priv[SCRATCH ? 192 : 1];
  if constexpr(SCRATCH) {
    for(int i = 0; i < 192; ++i) priv[i] = (float) i;
    priv[(threadIdx.x * 7 + (int) (ticks & 63)) % 192] += 1.0f;
  }
  float v[REGS];
#pragma unroll
  for(int i = 0; i < REGS; ++i) v[i] = (float) (threadIdx.x + i);
  const long long t0 = wall_clock64();
  lds[threadIdx.x] = 1.0f;
  while(wall_clock64() - t0 < ticks) {
#pragma unroll
    for(int i = 0; i < REGS; ++i) v[i] = v[i] * 1.0001f + lds[(threadIdx.x + i) & 255];
    if constexpr(SCRATCH) {      // the busy form: no sleep, a workgroup barrier and an agent-scope atomic poll per round (what a team barrier does)
      __syncthreads();
      if(threadIdx.x == 0) lds[0] += (float) __hip_atomic_load(reinterpret_cast<unsigned*>(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
  float s = 0.0f;
  if constexpr(SCRATCH) s += priv[(threadIdx.x + (int) (wall_clock64() & 127)) % 192];
#pragma unroll
  for(int i = 0; i < REGS; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void small(float* out)
{
  __shared__ float t[256];
  t[threadIdx.x & 255] = (float) threadIdx.x;
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = t[(threadIdx.x + 1) & 255];
}

template <int REGS, bool SCRATCH = false>
static int run(int lds_kb, int small_threads, int cus)
{
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float *a, *b;
  CK(hipMalloc(&a, sizeof(float) * 256 * 1024));
  CK(hipMalloc(&b, sizeof(float) * 1024 * 1024));
  CK(hipFuncSetAttribute((const void*) spinner<REGS, SCRATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  CK(hipEventRecord(e0, s1));
  hipLaunchKernelGGL((spinner<REGS, SCRATCH>), dim3(cus), dim3(256), lds_kb * 1024, s1, a, 1000000ll);      // 10 ms at 100 MHz
  CK(hipEventRecord(e1, s1));
  std::this_thread::sleep_for(std::chrono::milliseconds(2));
  hipLaunchKernelGGL(small, dim3(512), dim3(small_threads), 0, s2, b);
  CK(hipEventRecord(e2, s2));
  CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
  float t_spin = 0, t_small = 0;
  CK(hipEventElapsedTime(&t_spin, e0, e1));
  CK(hipEventElapsedTime(&t_small, e0, e2));
  if(SCRATCH) std::printf("(with a scratch array) ");
  std::printf("spinner: %3d registers/lane array, %3d KB dynamic LDS, %d workgroups | small kernel of %4d-thread workgroups done at %6.2f ms, spinner at %6.2f ms -> %s\n",
              REGS, lds_kb, cus, small_threads, t_small, t_spin, t_small < t_spin - 1.0f ? "CO-RESIDENT" : "waited for the spinner");
  (void) hipFree(a); (void) hipFree(b);
  return 0;
}

int main()
{
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  std::printf("%s: %d CUs, %zu KB LDS per workgroup max\n", p.name, cus, p.sharedMemPerBlock / 1024);
  for(int lds : {16, 64, 96, 120, 136}) {
    if(run<32>(lds, 256, cus)) return 1;
    if(run<200>(lds, 256, cus)) return 1;
  }
  if(run<200>(120, 1024, cus)) return 1;
  if(run<32>(16, 1024, cus)) return 1;
  if((run<200, true>(120, 256, cus))) return 1;
  if((run<200, true>(120, 1024, cus))) return 1;
  return 0;
}
