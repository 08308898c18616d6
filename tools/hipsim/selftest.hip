// Deliberately broken kernels for tests/test_sim_gemm.py: they show what the simulator's modes catch.
//   BUG 0  correct: stage with LDS-DMA, wait for it, barrier, read the NEXT wave's slice, double-buffered
//   BUG 1  reads LDS without waiting for its own LDS-DMA            (caught with DMA landing late)
//   BUG 2  no barrier between the wait and reading another wave's slice   (caught by run-ahead scheduling)
//   BUG 3  one buffer instead of two: a wave restages it while others still read   (same)
//   BUG 4  per-lane (non-uniform) LDS base handed to global_load_lds      (flagged by the simulator)
#include "common.h"

template <int BUG>
__global__ __launch_bounds__(256) void selftest_kernel(const float* __restrict__ src, float* __restrict__ dst, int rounds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float acc = 0.0f;
  for (int r = 0; r < rounds; ++r) {
    char* buf = smem + ((BUG == 3) ? 0 : (r & 1)) * 4096;
    const float* g = src + (size_t)r * 1024 + wave * 256 + lane * 4;      // 16 B per lane, 1 KiB per wave
    char* base = buf + wave * 1024 + (BUG == 4 ? lane : 0);
    __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(g), MNR_LDS_PTR(base), 16, 0, 0);
    if (BUG != 1) {
      MNR_GPU_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"));
      MNR_SIM_HOOK(hipsim::wait_vmcnt(0));
    }
    if (BUG != 2) __builtin_amdgcn_s_barrier();
    const float* nb = (const float*)(buf + ((wave + 1) & 3) * 1024);
    acc += nb[lane * 4] + nb[lane * 4 + 3];
    if (BUG == 1) {
      MNR_GPU_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"));
      MNR_SIM_HOOK(hipsim::wait_vmcnt(0));
    }
  }
  dst[blockIdx.x * 256 + tid] = acc;
}

extern "C" int hipsim_selftest(int bug, const float* src, float* dst, int rounds, int blocks) {
  switch (bug) {
    case 0: hipLaunchKernelGGL(selftest_kernel<0>, dim3(blocks), dim3(256), 8192, nullptr, src, dst, rounds); break;
    case 1: hipLaunchKernelGGL(selftest_kernel<1>, dim3(blocks), dim3(256), 8192, nullptr, src, dst, rounds); break;
    case 2: hipLaunchKernelGGL(selftest_kernel<2>, dim3(blocks), dim3(256), 8192, nullptr, src, dst, rounds); break;
    case 3: hipLaunchKernelGGL(selftest_kernel<3>, dim3(blocks), dim3(256), 8192, nullptr, src, dst, rounds); break;
    case 4: hipLaunchKernelGGL(selftest_kernel<4>, dim3(blocks), dim3(256), 8192, nullptr, src, dst, rounds); break;
    default: return MNR_ERR_INVALID_ARGUMENT;
  }
  return MNR_OK;
}
