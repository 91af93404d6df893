// Micro-benchmark: does HBM throughput of a warp_residual-shaped access pattern depend on how many separate streams the
// per-point data is split over?  Variant A: 4 read streams (16 + 4 + 32 + 128 B/point, wave-tiled) + 2 write streams
// (32 + 1 B/point).  Variant B: one merged read stream (180 B/point, tile-major) + the same writes.  Variant C: pure copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)

__device__ __forceinline__ size_t tile_index(int pieces, size_t i, int piece) { return ((i >> 6) * pieces + piece) * 64 + (i & 63); }

template <int MODE>   // 0 full, 1 no valid store, 2 no key load, 3 neither
__global__ __launch_bounds__(256) void kA(const float4* pts, const unsigned* key, const float4* pix, const float4* taps, float4* r, unsigned char* valid, size_t n)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if(i >= n) return;
  float4 acc = pts[i];
  const unsigned k = (MODE == 2 || MODE == 3) ? 7u : key[i];
  const float4 p0 = pix[tile_index(2, i, 0)], p1 = pix[tile_index(2, i, 1)];
  float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
  for(int q = 0; q < 8; ++q) { const float4 t = taps[tile_index(8, i, q)]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
  if(MODE != 4 || s.x == 1234.5f) {
    r[tile_index(2, i, 0)] = make_float4(s.x - p0.x + acc.x, s.y - p0.y, s.z - p0.z, s.w - p0.w);
    r[tile_index(2, i, 1)] = make_float4(s.x - p1.x, s.y - p1.y + acc.y, s.z - p1.z, s.w - p1.w + (float) k);
  }
  if(MODE == 0 || MODE == 2 || (s.y == 1234.5f)) valid[i] = (unsigned char) (k & 1);
}

// two points per thread (i and i + 64 inside a 128-point span): twice the loads in flight per wave
__global__ __launch_bounds__(256) void kA2(const float4* pts, const unsigned* key, const float4* pix, const float4* taps, float4* r, unsigned char* valid, size_t n)
{
  const size_t base = ((size_t) blockIdx.x * 256 + threadIdx.x);
  const size_t i0 = (base >> 6) * 128 + (base & 63), i1 = i0 + 64;
  if(i1 >= n) return;
  float4 t0[8], t1[8];
  const float4 a0 = pts[i0], a1 = pts[i1];
  const unsigned k0 = key[i0], k1 = key[i1];
  const float4 p00 = pix[tile_index(2, i0, 0)], p01 = pix[tile_index(2, i0, 1)], p10 = pix[tile_index(2, i1, 0)], p11 = pix[tile_index(2, i1, 1)];
#pragma unroll
  for(int q = 0; q < 8; ++q) { t0[q] = taps[tile_index(8, i0, q)]; t1[q] = taps[tile_index(8, i1, q)]; }
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0;
#pragma unroll
  for(int q = 0; q < 8; ++q) { s0.x += t0[q].x; s0.y += t0[q].y; s0.z += t0[q].z; s0.w += t0[q].w; s1.x += t1[q].x; s1.y += t1[q].y; s1.z += t1[q].z; s1.w += t1[q].w; }
  r[tile_index(2, i0, 0)] = make_float4(s0.x - p00.x + a0.x, s0.y - p00.y, s0.z - p00.z, s0.w - p00.w);
  r[tile_index(2, i0, 1)] = make_float4(s0.x - p01.x, s0.y - p01.y + a0.y, s0.z - p01.z, s0.w - p01.w + (float) k0);
  r[tile_index(2, i1, 0)] = make_float4(s1.x - p10.x + a1.x, s1.y - p10.y, s1.z - p10.z, s1.w - p10.w);
  r[tile_index(2, i1, 1)] = make_float4(s1.x - p11.x, s1.y - p11.y + a1.y, s1.z - p11.z, s1.w - p11.w + (float) k1);
  valid[i0] = (unsigned char) (k0 & 1); valid[i1] = (unsigned char) (k1 & 1);
}

// merged: per tile of 64 points 45 pieces of 16 B... 180 B/point = 11.25 float4: use 12 pieces (192 B/point, 16-B key slot)
__global__ __launch_bounds__(256) void kB(const float4* rec, float4* r, unsigned char* valid, size_t n)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if(i >= n) return;
  const float4 acc = rec[tile_index(12, i, 0)];
  const float4 kk = rec[tile_index(12, i, 1)];
  const float4 p0 = rec[tile_index(12, i, 2)], p1 = rec[tile_index(12, i, 3)];
  float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
  for(int q = 0; q < 8; ++q) { const float4 t = rec[tile_index(12, i, 4 + q)]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
  r[tile_index(2, i, 0)] = make_float4(s.x - p0.x + acc.x, s.y - p0.y, s.z - p0.z, s.w - p0.w);
  r[tile_index(2, i, 1)] = make_float4(s.x - p1.x, s.y - p1.y + acc.y, s.z - p1.z, s.w - p1.w + kk.x);
  valid[i] = (unsigned char) (((int) kk.x) & 1);
}

// E: like A with configurable tile size (points per tile) and threads per block, optional nontemporal loads
template <int TILE, int NT>
__device__ __forceinline__ size_t tidx(int pieces, size_t i, int piece) { return ((i / TILE) * pieces + piece) * TILE + (i % TILE); }
template <int TILE, int NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void kE(const float4* pts, const unsigned* key, const float4* pix, const float4* taps, float4* r, unsigned char* valid, size_t n)
{
  const size_t i = (size_t) blockIdx.x * BLOCK + threadIdx.x;
  if(i >= n) return;
  typedef float v4f __attribute__((ext_vector_type(4)));
  auto ld = [&](const float4* p) {
    if(NT & 1) { const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *p;
  };
  float4 acc = ld(pts + i);
  const unsigned k = key[i];
  const float4 p0 = ld(pix + tidx<TILE, NT>(2, i, 0)), p1 = ld(pix + tidx<TILE, NT>(2, i, 1));
  float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
  for(int q = 0; q < 8; ++q) { const float4 t = ld(taps + tidx<TILE, NT>(8, i, q)); s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
  const float4 o0 = make_float4(s.x - p0.x + acc.x, s.y - p0.y, s.z - p0.z, s.w - p0.w);
  const float4 o1 = make_float4(s.x - p1.x, s.y - p1.y + acc.y, s.z - p1.z, s.w - p1.w + (float) k);
  if(NT & 2) {
    v4f a; a.x = o0.x; a.y = o0.y; a.z = o0.z; a.w = o0.w;
    v4f b; b.x = o1.x; b.y = o1.y; b.z = o1.z; b.w = o1.w;
    __builtin_nontemporal_store(a, reinterpret_cast<v4f*>(r + tidx<TILE, NT>(2, i, 0)));
    __builtin_nontemporal_store(b, reinterpret_cast<v4f*>(r + tidx<TILE, NT>(2, i, 1)));
  }
  else { r[tidx<TILE, NT>(2, i, 0)] = o0; r[tidx<TILE, NT>(2, i, 1)] = o1; }
  valid[i] = (unsigned char) (k & 1);
}

__global__ __launch_bounds__(256) void kC(const float4* a, float4* b, size_t n4)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if(i < n4) b[i] = a[i];
}

// read-only stream sum (no write): the read ceiling
__global__ __launch_bounds__(256) void kD(const float4* a, float* out, size_t n4)
{
  size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  float s = 0;
  for(int k = 0; k < 8; ++k, i += (size_t) gridDim.x * 256) if(i < n4) { const float4 t = a[i]; s += t.x + t.y + t.z + t.w; }
  if(s == 1234.5f) out[0] = s;
}

int main()
{
  const size_t n = 12u << 20;   // 12.6 M points, like a 1024-pair launch
  float4 *pts, *pix, *taps, *r, *rec; unsigned* key; unsigned char* valid; float* out;
  CK(hipMalloc(&pts, n * 16)); CK(hipMalloc(&key, n * 4)); CK(hipMalloc(&pix, n * 32)); CK(hipMalloc(&taps, n * 128));
  CK(hipMalloc(&r, n * 32)); CK(hipMalloc(&valid, n)); CK(hipMalloc(&rec, n * 192)); CK(hipMalloc(&out, 4));
  CK(hipMemset(pts, 0, n * 16)); CK(hipMemset(key, 0, n * 4)); CK(hipMemset(pix, 0, n * 32)); CK(hipMemset(taps, 0, n * 128)); CK(hipMemset(rec, 0, n * 192));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (int) ((n + 255) / 256);
  auto time = [&](const char* name, double bytes, auto launch) {
    for(int w = 0; w < 3; ++w) launch();
    hipEventRecord(e0);
    const int reps = 20;
    for(int k = 0; k < reps; ++k) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.1f us  %7.1f GB/s\n", name, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
  };
  time("A: 4 read + 2 write streams (213 B)", n * 213.0, [&] { hipLaunchKernelGGL(kA<0>, dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("A1: no valid store (212 B)", n * 212.0, [&] { hipLaunchKernelGGL(kA<1>, dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("A2: no key load (209 B)", n * 209.0, [&] { hipLaunchKernelGGL(kA<2>, dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("A3: neither (208 B)", n * 208.0, [&] { hipLaunchKernelGGL(kA<3>, dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("A5: 2 points per thread (213 B)", n * 213.0, [&] { hipLaunchKernelGGL(kA2, dim3(grid / 2), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E1: tile 256, block 256", n * 213.0, [&] { hipLaunchKernelGGL((kE<256, 0, 256>), dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E2: tile 64, block 1024", n * 213.0, [&] { hipLaunchKernelGGL((kE<64, 0, 1024>), dim3(grid / 4), dim3(1024), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E3: tile 1024, block 1024", n * 213.0, [&] { hipLaunchKernelGGL((kE<1024, 0, 1024>), dim3(grid / 4), dim3(1024), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E4: tile 64, block 256, NT loads+stores", n * 213.0, [&] { hipLaunchKernelGGL((kE<64, 3, 256>), dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E4a: NT loads only", n * 213.0, [&] { hipLaunchKernelGGL((kE<64, 1, 256>), dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E4b: NT stores only", n * 213.0, [&] { hipLaunchKernelGGL((kE<64, 2, 256>), dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E4c: NT, block 1024", n * 213.0, [&] { hipLaunchKernelGGL((kE<64, 3, 1024>), dim3(grid / 4), dim3(1024), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E4d: NT, block 64", n * 213.0, [&] { hipLaunchKernelGGL((kE<64, 3, 64>), dim3(grid * 4), dim3(64), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E4e: NT, tile 1024 block 256", n * 213.0, [&] { hipLaunchKernelGGL((kE<1024, 3, 256>), dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E4f: NT, tile 2048 block 256", n * 213.0, [&] { hipLaunchKernelGGL((kE<2048, 3, 256>), dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E4g: NT, tile 4096 block 256", n * 213.0, [&] { hipLaunchKernelGGL((kE<4096, 3, 256>), dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E4h: NT, tile 16384 block 256", n * 213.0, [&] { hipLaunchKernelGGL((kE<16384, 3, 256>), dim3(grid), dim3(256), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("E5: tile 64, block 64", n * 213.0, [&] { hipLaunchKernelGGL((kE<64, 0, 64>), dim3(grid * 4), dim3(64), 0, 0, pts, key, pix, taps, r, valid, n); });
  time("B: 1 merged read (192 B) + writes", n * 225.0, [&] { hipLaunchKernelGGL(kB, dim3(grid), dim3(256), 0, 0, rec, r, valid, n); });
  const size_t n4 = n * 8;   // 128 B/point worth of float4
  time("C: float4 copy (r+w)", n4 * 32.0, [&] { hipLaunchKernelGGL(kC, dim3((int) ((n4 + 255) / 256)), dim3(256), 0, 0, taps, rec, n4); });
  time("D: float4 read only", n4 * 16.0, [&] { hipLaunchKernelGGL(kD, dim3((int) ((n4 / 8 + 255) / 256)), dim3(256), 0, 0, taps, out, n4); });
  return 0;
}
