// Shader-clock sampler (probe only, not part of the library): a handful of single-wave workgroups, no LDS, < 16 VGPRs, that sit
// next to whatever else runs on the chip and record (s_memrealtime, s_memtime) pairs every `period` ticks of the 100 MHz
// real-time counter.  s_memtime counts shader-clock cycles, so the ratio of the differences is the shader clock the CU ran at
// (DESIGN.md section 6 used the same pair from inside the GEMM kernels).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/_bin/libclock_probe.so tools/clock_probe.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void clock_sampler(unsigned long long* out, int n, unsigned period, const unsigned* stop) {
  unsigned long long* o = out + (size_t)blockIdx.x * 2 * n;
  unsigned long long next = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    unsigned long long rt;
    do {
      __builtin_amdgcn_s_sleep(8);
      rt = __builtin_amdgcn_s_memrealtime();
    } while (rt < next);
    next = rt + period;
    const unsigned long long st = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
      o[2 * i] = rt;
      o[2 * i + 1] = st;
    }
    if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
      if (threadIdx.x == 0)
        for (int j = i + 1; j < n; ++j) o[2 * j] = o[2 * j + 1] = 0;
      return;
    }
  }
}

extern "C" int clock_sampler_launch(void* stream, unsigned long long* out, int wgs, int n, unsigned period, const unsigned* stop) {
  hipLaunchKernelGGL(clock_sampler, dim3(wgs), dim3(64), 0, (hipStream_t)stream, out, n, period, stop);
  return (int)hipGetLastError();
}
