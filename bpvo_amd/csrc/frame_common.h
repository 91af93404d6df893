// Per-frame kernels, shared device helpers: OpenCV's border rule, the launch grid of the one-thread-per-pixel kernels.
#pragma once
#include "kernels.h"

namespace bpvo_hip {

__device__ __forceinline__ int reflect101(int p, int len)
{
  // cv::borderInterpolate(BORDER_REFLECT_101); one reflection suffices for the <= 3 pixel halos used here
  if(p < 0) p = -p;
  if(p >= len) p = 2 * len - 2 - p;
  if(p < 0) p = 0;   // degenerate len == 1
  return p;
}

// the same for halos that may exceed the image (wide smoothing kernels on the coarsest levels): reflect until inside
__device__ __forceinline__ int reflect101_wide(int p, int len)
{
  if(len == 1) return 0;
  while(p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// 64 x 4-pixel workgroups over a W x R level, one grid plane per frame
static inline dim3 grid2d(int W, int R, int nz) { return dim3((W + 63) / 64, (R + 3) / 4, nz); }

}  // namespace bpvo_hip
