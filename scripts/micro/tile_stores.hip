// What does the TILE shape of bitplanes_blur's output cost?  Frames of 1241 x 376 records of 32 bytes; a 256-thread workgroup writes a
// column strip TW pixels wide and ROWS rows tall, row after row, a lane pair per record (1 KB contiguous per wavefront store), non-temporal —
// nothing but the stores.  TW = 64 is the kernel's tile; wider strips make longer runs per row.
//   hipcc --offload-arch=gfx950 -O3 tile_stores.hip -o tile_stores
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int W = 1241, R = 376;
template <int TW, int ROWS>
__global__ __launch_bounds__(256) void k(float* __restrict__ d)
{
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * ROWS;
  v4f* frame = reinterpret_cast<v4f*>(d) + (size_t) blockIdx.z * W * R * 2;
  for(int y = y0; y < min(y0 + ROWS, R); y += (256 * 1) / (2 * TW) > 0 ? (256 / (2 * TW)) : 1) {
    // 256 threads = 128 records per pass: TW = 64 -> two rows per pass, TW = 128 -> one row, TW >= 256 -> several passes per row
    if(2 * TW <= 256) {
      const int row = y + (int) threadIdx.x / (2 * TW), u = threadIdx.x % (2 * TW), x = x0 + (u >> 1);
      if(row < R && row < y0 + ROWS && x < W) { v4f a = {(float) x, (float) row, 2.f, 3.f}; __builtin_nontemporal_store(a, frame + ((size_t) row * W + x) * 2 + (u & 1)); }
    } else {
      for(int u = threadIdx.x; u < 2 * TW; u += 256) {
        const int x = x0 + (u >> 1);
        if(x < W) { v4f a = {(float) x, (float) y, 2.f, 3.f}; __builtin_nontemporal_store(a, frame + ((size_t) y * W + x) * 2 + (u & 1)); }
      }
    }
  }
}
template <int TW, int ROWS>
void run(float* d, int frames)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid((W + TW - 1) / TW, (R + ROWS - 1) / ROWS, frames);
  k<TW, ROWS><<<grid, 256>>>(d);
  hipEventRecord(e0);
  for(int r = 0; r < 5; ++r) k<TW, ROWS><<<grid, 256>>>(d);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("strip %4d px x %3d rows per workgroup: %.2f TB/s\n", TW, ROWS, (double) frames * W * R * 32 * 5 / (ms * 1e-3) / 1e12);
}
int main()
{
  const int frames = 512;
  float* d; hipMalloc(&d, (size_t) frames * W * R * 32);
  run<64, 32>(d, frames); run<64, 64>(d, frames); run<64, 376>(d, frames);
  run<128, 32>(d, frames); run<128, 64>(d, frames);
  run<256, 32>(d, frames); run<256, 64>(d, frames);
  run<1280, 8>(d, frames); run<1280, 32>(d, frames);
  return 0;
}
