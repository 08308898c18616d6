// How fast can one CU, and the whole chip, pull GEMM operand tiles out of L2 / HBM, and does the path matter?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ingest_probe tools/ingest_probe.hip && /tmp/ingest_probe
//
// The NT / TN GEMM loops move 64 KiB per K step into each CU and measure ~3.8k cycles per step against 2.05k of MFMA
// issue: ~23-30 B/clk per CU, ~9.5 TB/s over the chip, the same in cycles whether the chip runs
// at 1.65 or 2.1 GHz.  This probe separates the candidates for that wall.  Every workgroup (512 threads, as the GEMMs)
// streams 64 KiB "steps" with the GEMM loops' structure (issue step s+1, wait for step s with a counted vmcnt, one
// barrier per step) and nothing else:
//   mode 0  LDS-DMA, 16 B per lane, 128-byte rows of a row-major matrix with the NT kernel's source-side XOR swizzle
//   mode 1  LDS-DMA, 16 B per lane, fully linear source (1 KiB contiguous per wave instruction)
//   mode 2  global -> VGPR loads, 16 B per lane, addresses of mode 0
//   mode 3  global -> VGPR loads, linear
//   mode 4  half the bytes by LDS-DMA (waves 0-3), half by VGPR loads (waves 4-7): the direct-weights split
//   mode 5  LDS-DMA, 4 B per lane (the pre-gfx950 width), linear
// over grids of 1 workgroup per CU on 1/8, 1/2 and all CUs and 4 workgroups per CU, on a footprint that stays in L2 /
// MALL (all workgroups of an XCD read the same 2 MiB) and one that streams from HBM (every workgroup its own panel).
// Output: one line per case with us, GB/s per CU, B/clk per CU (shader clock from s_memtime / s_memrealtime) and TB/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int STEP_BYTES = 64 * 1024;
constexpr int THREADS = 512;
constexpr int LD_BYTES = 2048;                      // row pitch of the "matrix" of modes 0 / 2 (K = 1024 bf16)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// Source address of 16-byte chunk c (0 .. 4095) of a step: swizzled rows or linear.
template <bool LINEAR>
__device__ __forceinline__ const char* chunk_src(const char* base, int step, int c) {
  if (LINEAR) return base + (size_t)step * STEP_BYTES + (size_t)c * 16;
  const int r = c >> 3;                             // 512 rows of 128 B per step
  const int slot = (c & 7) ^ ((r >> 1) & 7);
  return base + (size_t)r * LD_BYTES + (size_t)step * 128 + slot * 16;   // K advances by 64 elements per step
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void ingest_kernel(const char* __restrict__ src, size_t panel_stride, int panels, int steps,
                                                         unsigned long long* __restrict__ times, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)(blockIdx.x % panels) * panel_stride;
  constexpr bool LINEAR = (MODE == 1 || MODE == 3 || MODE == 5);
  constexpr bool DMA_ALL = (MODE == 0 || MODE == 1 || MODE == 5);
  constexpr bool VGPR_ALL = (MODE == 2 || MODE == 3);
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  u32x4 acc = {0, 0, 0, 0};
  // Per wave and step: 8 instructions of 1 KiB (16 B per lane); mode 5: 32 of 256 B.  A wave is either a DMA wave or a
  // register wave for the whole kernel (mode 4: waves 0-3 / 4-7), each with its own loop: the two kinds of load never
  // share a vmcnt, and hipcc never sees them mixed (it would drain to 0).  Both loops hold one barrier per step.
  const bool dma_wave = DMA_ALL || (MODE == 4 && wave < 4);
  if (dma_wave) {
    constexpr int N_INSTR = (MODE == 5) ? 32 : 8;
    auto issue = [&](int s) {
      char* buf = smem + (s & 1) * STEP_BYTES;
#pragma unroll
      for (int i = 0; i < N_INSTR; ++i) {
        const int cbase = (i * 8 + wave) * 64;
        if (MODE == 5) {
          __builtin_amdgcn_global_load_lds(GLOBAL_PTR(base + (size_t)s * STEP_BYTES + (size_t)(cbase + lane) * 4), LDS_PTR(buf + cbase * 4), 4, 0, 0);
        } else {
          __builtin_amdgcn_global_load_lds(GLOBAL_PTR(chunk_src<LINEAR>(base, s, cbase + lane)), LDS_PTR(buf + cbase * 16), 16, 0, 0);
        }
      }
    };
    issue(0);
    for (int s = 0; s < steps; ++s) {
      if (s + 1 < steps) {
        issue(s + 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_INSTR) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
  } else {
    auto issue = [&](int s, u32x4 (&regs)[8]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) regs[i] = *(const u32x4*)chunk_src<LINEAR>(base, s, (i * 8 + wave) * 64 + lane);
    };
    u32x4 ra[8], rb[8];
    issue(0, ra);
    for (int s = 0; s < steps; s += 2) {               // step s in ra, step s + 1 in rb; hipcc counts these waits
      if (s + 1 < steps) issue(s + 1, rb);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= ra[i];
      __builtin_amdgcn_s_barrier();
      if (s + 1 >= steps) break;
      if (s + 2 < steps) issue(s + 2, ra);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= rb[i];
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) {
    times[2 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - t0;
    times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  const unsigned v = acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ (unsigned)smem[tid * 16];
  if (v == 0x12345677u) sink[0] = v;                // keeps the loads alive
}

template <int MODE>
static void run_mode(const char* src, size_t panel_stride, int panels, int grid, int steps, int lds_bytes, const char* label,
                     unsigned long long* times_d, unsigned* sink_d, int cus) {
  CHECK(hipFuncSetAttribute((const void*)ingest_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {                // first launch warms L2 / MALL and the clocks
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(ingest_kernel<MODE>, dim3(grid), dim3(THREADS), lds_bytes, 0, src, panel_stride, panels, steps, times_d, sink_d);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> t(2 * grid);
  CHECK(hipMemcpy(t.data(), times_d, t.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int b = 0; b < grid; ++b) {
    cyc += (double)t[2 * b];
    rt += (double)t[2 * b + 1];
  }
  cyc /= grid;
  rt /= grid;                                        // 100 MHz ticks
  const double bytes_wg = (double)steps * STEP_BYTES;
  const double ghz = cyc / (rt * 10.0);              // cycles per ns
  const int active_cus = grid < cus ? grid : cus;
  const double wg_per_cu = (double)grid / active_cus;
  printf("mode %d %-28s grid %5d steps %4d lds %3dK: %8.1f us  wg: %7.0f cyc @ %.2f GHz  %6.1f B/clk/wg  per-CU %6.1f GB/s (%5.1f B/clk)  chip %6.2f TB/s\n",
         MODE, label, grid, steps, lds_bytes / 1024, ms * 1e3, cyc, ghz, bytes_wg / cyc,
         bytes_wg * wg_per_cu / (ms * 1e-3) / 1e9, bytes_wg * wg_per_cu / (ms * 1e-3) / 1e9 / ghz, bytes_wg * grid / (ms * 1e-3) / 1e12);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.gcnArchName, cus);
  // HBM case: 4096 panels of 512 rows x 2 KiB (1 MiB each, 4 GiB); L2 case: 16 panels (16 MiB).
  const size_t panel = (size_t)512 * LD_BYTES;
  const int max_panels = 4096;
  char* src;
  CHECK(hipMalloc(&src, panel * max_panels + STEP_BYTES));
  CHECK(hipMemset(src, 1, panel * max_panels + STEP_BYTES));
  unsigned long long* times_d;
  unsigned* sink_d;
  CHECK(hipMalloc(&times_d, 2 * 8192 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink_d, 4));
  // A panel holds 1024 B of K per row = 8 swizzled steps of 128 B (modes 0 / 2) or 16 linear steps of 64 KiB.
  struct Case {
    const char* label;
    int grid, panels, steps, lds;
  };
  const Case cases[] = {
      {"L2 1wg/CU on 1/8 of the CUs", cus / 8, 16, 8, 128 * 1024},   {"L2 1wg/CU on half the CUs", cus / 2, 16, 8, 128 * 1024},
      {"L2 1wg/CU all CUs", cus, 16, 8, 128 * 1024},                 {"L2 4wg/CU (serial)", 4 * cus, 16, 8, 128 * 1024},
      {"HBM 1wg/CU all CUs", cus, 4096, 8, 128 * 1024},              {"HBM 4wg/CU (serial)", 4 * cus, 4096, 8, 128 * 1024},
      {"HBM 16wg/CU (serial)", 16 * cus, 4096, 8, 128 * 1024},
  };
  for (const Case& c : cases) {
    run_mode<0>(src, panel, c.panels, c.grid, c.steps, c.lds, c.label, times_d, sink_d, cus);
    run_mode<1>(src, panel, c.panels, c.grid, c.steps, c.lds, c.label, times_d, sink_d, cus);
    run_mode<2>(src, panel, c.panels, c.grid, c.steps, c.lds, c.label, times_d, sink_d, cus);
    run_mode<3>(src, panel, c.panels, c.grid, c.steps, c.lds, c.label, times_d, sink_d, cus);
    run_mode<4>(src, panel, c.panels, c.grid, c.steps, c.lds, c.label, times_d, sink_d, cus);
    run_mode<5>(src, panel, c.panels, c.grid, c.steps, c.lds, c.label, times_d, sink_d, cus);
  }
  // Longer streams (the dW kernel's regime: hundreds of steps per workgroup), linear only.
  run_mode<1>(src, panel, 4096, cus, 16, 128 * 1024, "HBM 1wg/CU 16 linear steps", times_d, sink_d, cus);
  run_mode<3>(src, panel, 4096, cus, 16, 128 * 1024, "HBM 1wg/CU 16 linear steps", times_d, sink_d, cus);
  return 0;
}
