// Microbenchmark: per-CU rate of the L2 -> CU operand stream by instruction flavour (probe only, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/dma_probe tools/dma_probe.hip
// Every workgroup (one per CU: 128 KiB of LDS) streams the same 2-MiB buffer (L2-resident, larger than L1) in 64-KiB
// steps; the flavours: 0 global_load_lds_dwordx4 (LDS-DMA), 1 global_load_dwordx4 into registers, 2 the same + ds_write_b128,
// 3 LDS-DMA of 4 B per lane (global_load_lds_dword).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(const char* __restrict__ src, int steps, int row_stride, unsigned long long* out, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int PER_STEP = 65536 / 16 / (WAVES * 64);       // 16-B chunks per thread per 64-KiB step
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    const char* base = src + (size_t)(s & 31) * 65536;
    char* lds = smem + (s & 1) * 65536;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < PER_STEP; ++i) {
        const int c = (i * WAVES + wave) * 64 + lane;          // 16-B chunk index; rows of 128 B, row_stride apart in memory
        const char* g = base + (size_t)(c >> 3) * row_stride + (c & 7) * 16;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(lds + (i * WAVES + wave) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < PER_STEP * 4; ++i) {
        const int c = (i * WAVES + wave) * 64 + lane;          // 4-B chunk index
        const char* g = base + (size_t)(c >> 5) * row_stride + (c & 31) * 4;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(lds + (i * WAVES + wave) * 256), 4, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[PER_STEP];
#pragma unroll
      for (int i = 0; i < PER_STEP; ++i) {
        const int c = (i * WAVES + wave) * 64 + lane;
        v[i] = *(const u32x4*)(base + (size_t)(c >> 3) * row_stride + (c & 7) * 16);
      }
#pragma unroll
      for (int i = 0; i < PER_STEP; ++i) {
        if (MODE == 2) *(u32x4*)(lds + ((i * WAVES + wave) * 64 + lane) * 16) = v[i];
        else acc ^= v[i][0] ^ v[i][3];
      }
    }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc + smem[tid];
}

// The NT GEMM's operand streams without its MFMAs: per 64-k step a workgroup brings in a [256 rows][128 B] slice of A (row
// stride 2 KiB, from a 1-GiB matrix: HBM) and 32 KiB of the L2-resident weights through a 2-stage LDS-DMA pipeline, then
// "computes" for `sleep64` x 64 cycles.  share: workgroups per A tile (consecutive workgroups of one XCD, as in the GEMM's
// tile order); pf > 0: each of the `share` siblings also touches its 1/share of the A slice of step kt + pf with one
// 4-byte load per cache line (an L2 prefetch), so that the DMA of that step hits in L2.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_stream(const char* __restrict__ A, const char* __restrict__ Wt, int tiles_per_wg,
                                                          int share, int pf, int sleep64, unsigned long long* out, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int PER = 32768 / 16 / (WAVES * 64);
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int sib = q % share;
  unsigned acc = 0;
  const int nk = 16, total = tiles_per_wg * nk;
  auto a_base = [&](int it) {                            // slice `it` of this workgroup's stream
    const int t = it / nk, kt = it % nk;
    const size_t m_tile = (size_t)xcd + 8 * (size_t)(q / share) + 8 * (size_t)(gridDim.x / 8 / share) * t;
    return A + m_tile * 256 * 2048 + (size_t)kt * 128;
  };
  auto stage = [&](int it) {
    const char* ab = a_base(it);
    const char* wb = Wt + (size_t)(it & 31) * 32768;
    char* lds = smem + (it & 1) * 65536;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = (i * WAVES + wave) * 64 + lane;
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ab + (size_t)(c >> 3) * 2048 + (c & 7) * 16), LDS_PTR(lds + (i * WAVES + wave) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = (i * WAVES + wave) * 64 + lane;
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wb + c * 16), LDS_PTR(lds + 32768 + (i * WAVES + wave) * 1024), 16, 0, 0);
    }
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  stage(0);
  for (int it = 0; it < total; ++it) {
    if (pf > 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 1 < total) stage(it + 1);
    if (pf > 0) {
      // one line per lane: rows [sib * 256 / share, +256 / share) of slice it + pf, spread over the waves
      const int rows = 256 / share;
      const int r = sib * rows + (wave * 64 + lane) % rows;
      const int itp = it + pf < total ? it + pf : total - 1;
      unsigned v;
      const char* g = a_base(itp) + (size_t)r * 2048;
      if (wave * 64 < rows) {
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(g) : "memory");
        asm volatile("" ::"v"(v));
      }
    }
    for (int z = 0; z < sleep64; z += 16) __builtin_amdgcn_s_sleep(16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc + smem[tid];
}

template <int WAVES>
static void run_stream(const char* A, const char* Wt, int share, int pf, int sleep64, unsigned long long* out, unsigned* sink) {
  const int grid = 256, tiles = 8;
  hipFuncSetAttribute((const void*)gemm_stream<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((gemm_stream<WAVES>), dim3(grid), dim3(WAVES * 64), 131072, 0, A, Wt, tiles, share, pf, sleep64, out, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto x : h) sum += (double)x;
  printf("GEMM streams, waves %d share %d prefetch %2d compute %4d: %7.0f cycles per 64-k step\n", WAVES, share, pf, sleep64 * 64, sum / grid / (tiles * 16));
}

template <int MODE, int WAVES>
static void run(const char* name, const char* src, int row_stride, unsigned long long* out, unsigned* sink, int grid) {
  const int steps = 512;
  hipFuncSetAttribute((const void*)probe<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 131072, 0, src, steps, row_stride, out, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto x : h) sum += (double)x;
  const double cyc = sum / grid / steps;
  printf("%-34s waves %d grid %3d stride %5d: %7.0f cycles per 64 KiB = %5.1f B/clk/CU\n", name, WAVES, grid, row_stride, cyc, 65536.0 / cyc);
}

int main() {
  char* src;
  unsigned long long* out;
  unsigned* sink;
  hipMalloc(&src, 64 << 20);
  hipMemset(src, 1, 64 << 20);
  hipMalloc(&out, 8 * 1024);
  hipMalloc(&sink, 1024);
  {
    char* A;
    hipMalloc(&A, (size_t)1 << 30);
    hipMemset(A, 1, (size_t)1 << 30);
    for (int sleep64 : {0, 32, 48})
      for (int share : {1, 4})
        for (int pf : {0, 2, 4}) {
          if (share == 1 && pf) continue;
          run_stream<8>(A, src, share, pf, sleep64, out, sink);
          run_stream<4>(A, src, share, pf, sleep64, out, sink);
        }
  }
  for (int grid : {256}) {
    for (int stride : {128, 2048}) {
      if (stride == 2048 && false) continue;
      run<0, 4>("LDS-DMA 16 B/lane", src, stride, out, sink, grid);
      run<0, 8>("LDS-DMA 16 B/lane", src, stride, out, sink, grid);
      run<3, 8>("LDS-DMA 4 B/lane", src, stride, out, sink, grid);
      run<1, 4>("global_load_dwordx4 -> VGPR", src, stride, out, sink, grid);
      run<1, 8>("global_load_dwordx4 -> VGPR", src, stride, out, sink, grid);
      run<2, 8>("global_load_dwordx4 + ds_write_b128", src, stride, out, sink, grid);
    }
  }
  return 0;
}
