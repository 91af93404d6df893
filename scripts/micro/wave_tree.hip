// wave_tree_sums (gn_common.h) against the __shfl_down ladder it replaces: the 30 sums of 64 lanes, bit for bit, on random floats of
// mixed magnitude.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I bpvo_amd/csrc -I include scripts/micro/wave_tree.hip -o /tmp/wave_tree && /tmp/wave_tree
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "gn_common.h"

using namespace bpvo_hip;
constexpr int N = 30;

__global__ void both(const float* in, float* tree, float* ladder)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* p = in + ((size_t) blockIdx.x * 4 + wave) * 64 * N;
  float acc[N];
  for(int k = 0; k < N; ++k) acc[k] = p[k * 64 + lane];
  __shared__ float part[4][32];
  wave_tree_sums_to<N>(acc, lane, part[wave]);
  __syncthreads();
  if(lane < N) tree[((size_t) blockIdx.x * 4 + wave) * 32 + lane] = part[wave][lane];
  for(int k = 0; k < N; ++k) {
    float v = acc[k];
    for(int o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o);
    if(lane == 0) ladder[((size_t) blockIdx.x * 4 + wave) * 32 + k] = v;
  }
}

int main()
{
  const int blocks = 512, waves = blocks * 4;
  std::vector<float> h((size_t) waves * 64 * N);
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.0f, 1.0f);
  std::uniform_int_distribution<int> e(-12, 12);
  for(auto& v : h) v = std::ldexp(u(rng), e(rng));
  float *d_in, *d_a, *d_b;
  (void) hipMalloc(&d_in, h.size() * 4); (void) hipMalloc(&d_a, (size_t) waves * 32 * 4); (void) hipMalloc(&d_b, (size_t) waves * 32 * 4);
  (void) hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(both, dim3(blocks), dim3(256), 0, 0, d_in, d_a, d_b);
  std::vector<float> a((size_t) waves * 32), b((size_t) waves * 32);
  (void) hipMemcpy(a.data(), d_a, a.size() * 4, hipMemcpyDeviceToHost);
  (void) hipMemcpy(b.data(), d_b, b.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for(int w = 0; w < waves; ++w)
    for(int k = 0; k < N; ++k)
      if(std::memcmp(&a[(size_t) w * 32 + k], &b[(size_t) w * 32 + k], 4) != 0) {
        if(bad++ < 5) std::printf("wave %d acc %d: tree %.9g ladder %.9g\n", w, k, a[(size_t) w * 32 + k], b[(size_t) w * 32 + k]);
      }
  std::printf("wave_tree_sums vs __shfl_down ladder: %d waves x %d sums, %ld differ\n", waves, N, bad);
  return bad ? 1 : 0;
}
