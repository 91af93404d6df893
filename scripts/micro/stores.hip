// How fast can 32-byte pixel records be written?  (a) every lane writes both halves of its record (two 16-byte stores, lanes 32 bytes
// apart: the pattern of bitplanes_blur), (b) a lane pair shares a record (one 16-byte store per lane, a wave writes 1 KiB contiguous),
// (c) as (a) plus the 4-byte channel-0 plane.  Non-temporal and plain stores.   hipcc --offload-arch=gfx950 -O3 stores.hip -o stores
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k(float* __restrict__ d, float* __restrict__ ch0, size_t npix)
{
  const size_t stride = (size_t) gridDim.x * 256;
  for(size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < (MODE == 1 ? 2 * npix : npix); i += stride) {
    v4f a = {(float) i, 1.f, 2.f, 3.f}, b = {4.f, 5.f, (float) i, 7.f};
    v4f* p = reinterpret_cast<v4f*>(d);
    if(MODE == 1) { if(NT) __builtin_nontemporal_store(a, p + i); else p[i] = a; }
    else {
      if(NT) { __builtin_nontemporal_store(a, p + 2 * i); __builtin_nontemporal_store(b, p + 2 * i + 1); } else { p[2 * i] = a; p[2 * i + 1] = b; }
      if(MODE == 2) ch0[i] = a.x;
    }
  }
}
template <int MODE, bool NT>
void run(const char* name, float* d, float* c, size_t npix)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int grid : {256 * 8, 256 * 32, 1 << 20}) {
    k<MODE, NT><<<grid, 256>>>(d, c, npix);
    hipEventRecord(e0);
    for(int r = 0; r < 5; ++r) k<MODE, NT><<<grid, 256>>>(d, c, npix);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double) npix * (MODE == 2 ? 36 : 32) * 5;
    printf("%-44s grid %8d: %.2f TB/s\n", name, grid, bytes / (ms * 1e-3) / 1e12);
  }
}
int main()
{
  const size_t npix = (size_t) 128 << 20;   // 4 GiB of records
  float *d, *c; hipMalloc(&d, npix * 32); hipMalloc(&c, npix * 4);
  run<0, true>("two 16-B stores per lane, non-temporal", d, c, npix);
  run<0, false>("two 16-B stores per lane, plain", d, c, npix);
  run<1, true>("one 16-B store per lane (pair), non-temporal", d, c, npix);
  run<1, false>("one 16-B store per lane (pair), plain", d, c, npix);
  run<2, true>("two stores + ch0 plane, non-temporal", d, c, npix);
  hipMemsetAsync(d, 0, npix * 32); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for(int r = 0; r < 5; ++r) hipMemsetAsync(d, 0, npix * 32); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s              : %.2f TB/s\n", "hipMemsetAsync", (double) npix * 32 * 5 / (ms * 1e-3) / 1e12);
  return 0;
}
