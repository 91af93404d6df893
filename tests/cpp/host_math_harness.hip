// Test harness: the serial math of the gn_step kernel (bpvo_amd/csrc/device_math.h is __host__ __device__) compiled for the
// host, so that the CPU suite can compare it with the oracle's restatement of the same reference lines on identical inputs.
#include "device_math.h"

extern "C" {

int host_solve_system(const float* H, const float* G, float* dp)
{
  bpvo_hip::SolveScratch s;
  return bpvo_hip::solve_system(H, G, dp, &s) ? 1 : 0;
}

void host_twist_to_matrix(const float* p, float* T)
{
  const bpvo_hip::M44 m = bpvo_hip::twist_to_matrix(p);
  for(int i = 0; i < 16; ++i) T[i] = m.m[i];
}

void host_params_to_pose(const float* nrm, const float* p, float* T)
{
  const bpvo_hip::M44 m = bpvo_hip::params_to_pose(nrm, p);
  for(int i = 0; i < 16; ++i) T[i] = m.m[i];
}

}
