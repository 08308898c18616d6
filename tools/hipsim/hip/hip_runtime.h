// hipsim: a host-side FUNCTIONAL simulator for the kernels in multinerf_amd/csrc (development / test infrastructure).
//
// This header stands in for <hip/hip_runtime.h> when a .hip file is compiled for the HOST by tools/hipsim/build.py
// (plain clang++, `-I tools/hipsim` first on the include path).  The kernel SOURCE is the product's, unchanged; what
// is replaced is the machine: every thread of a workgroup is a fiber, workgroups run one after the other, LDS is a
// host array, and the gfx950 instructions the kernels reach through builtins / inline asm (MFMA 32x32x16 bf16,
// LDS-DMA global_load_lds, ds_read_b64_tr_b16, DPP quad_perm, s_barrier, s_waitcnt vmcnt) are restated in
// hipsim.cpp with their documented lane layouts.
//
// It exists so that index arithmetic, tile layouts, swizzles, pipeline bookkeeping and barrier placement of a new
// kernel can be checked HERE (no GPU) before GPU minutes are spent on it:
//   * HIPSIM_DMA=late : an LDS-DMA lands only when its wave executes the s_waitcnt vmcnt(N) / __syncthreads() that
//     covers it, i.e. as late as the hardware may deliver it: a missing or too-weak wait reads stale LDS;
//   * HIPSIM_DMA=eager (default): it lands at issue, and fibers run until they block, so a wave runs as far ahead of
//     the others as the barriers allow: a missing barrier lets it overwrite a buffer others still read;
//   * HIPSIM_ORDER=reverse|<seed>: fibers are visited in reverse / shuffled order.
// It is NOT a product path and is never loaded by the multinerf_amd package: it says nothing about performance and
// the GPU parity tests (tests/test_gpu_*.py) remain the only evidence of correctness on hardware.
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <functional>

#define MNR_HIPSIM 1

// Seams declared by csrc/common.h: GPU-only statements vanish, simulator hooks appear.
#define MNR_GPU_ONLY(...)
#define MNR_SIM_HOOK(...) __VA_ARGS__

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
// One OS thread runs every fiber, so thread_local storage is shared by all threads of the (single resident) workgroup.
#define __shared__ thread_local

using std::max;
using std::min;

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// HIP's built-in vector structs (only the ones the kernels name)
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct float3 { float x, y, z; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
  int multiProcessorCount;
  size_t maxSharedMemoryPerMultiProcessor;
  char gcnArchName[64];
};

inline const char* hipGetErrorString(hipError_t) { return "hipsim"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
// (streams: the simulator runs every launch to completion at the call; a "CU-masked stream" is just another null handle)
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) {
  *s = nullptr;
  return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  p->multiProcessorCount = 256;
  p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  strcpy(p->gcnArchName, "hipsim-gfx950");
  return hipSuccess;
}
// host-side runtime calls of stand-alone probes (tools/cabi_probe.cpp built against the simulator)
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
struct hipsimEvent { double t; };
typedef hipsimEvent* hipEvent_t;
template <class T>
inline hipError_t hipMalloc(T** p, size_t n) {
  *p = (T*)aligned_alloc(256, (n + 255) / 256 * 256);
  return *p ? hipSuccess : 1;
}
inline hipError_t hipFree(void* p) {
  free(p);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s_, size_t n, hipMemcpyKind) {
  memcpy(d, s_, n);
  return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) {
  memset(d, v, n);
  return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new hipsimEvent{0.0};
  return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
  e->t = (double)clock() / CLOCKS_PER_SEC;
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = (float)((b->t - a->t) * 1e3) + 1e-3f;
  return hipSuccess;
}
#define HIP_SYMBOL(x) x
#define hipMemcpyToSymbol(sym, src, n) (memcpy((void*)&(sym), (src), (n)), hipSuccess)

namespace hipsim {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct ThreadCtx {
  dim3 tid3, bid3, bdim3, gdim3;
  int lane, wave, linear;
};
extern ThreadCtx* cur;                           // the running fiber

void launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes, const std::function<void()>& body);

void barrier();                                  // s_barrier
void syncthreads();                              // s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier
void wait_vmcnt(int n);                          // s_waitcnt vmcnt(n)
void vm_store();                                 // a global store enters the in-order vmcnt queue (kernels that count it by hand)
void global_load_lds(const void* gptr, void* lds_wave_base, int size, int offset);
s16x4_t ds_read_tr16_b64(const void* lds_ptr);   // ds_read_b64_tr_b16
f32x16_t mfma_f32_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c);
int update_dpp(int old, int src, int dpp_ctrl, int row_mask, int bank_mask, bool bound_ctrl);
int readfirstlane(int v);
u32x2_t permlane32_swap(unsigned vdst, unsigned src);   // v_permlane32_swap_b32: lanes 32-63 of vdst <-> lanes 0-31 of src
float shfl(float v, int src_lane, int width);
int shfl_i(int v, int src_lane, int width);
unsigned long long ballot(int pred);
unsigned long long clock64();

}  // namespace hipsim

#define threadIdx (hipsim::cur->tid3)
#define blockIdx (hipsim::cur->bid3)
#define blockDim (hipsim::cur->bdim3)
#define gridDim (hipsim::cur->gdim3)

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  hipsim::launch(dim3(grid), dim3(block), (size_t)(shmem), [&]() { (kern)(__VA_ARGS__); })

#define __syncthreads() hipsim::syncthreads()
#define __builtin_amdgcn_s_barrier() hipsim::barrier()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_memtime() hipsim::clock64()
#define __builtin_amdgcn_s_memrealtime() hipsim::clock64()
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_readfirstlane(x) hipsim::readfirstlane(x)
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipsim::permlane32_swap(a, b)
#define __builtin_amdgcn_sbfe(src, off, width) ((int)((int32_t)((uint32_t)(src) << (32 - (off) - (width))) >> (32 - (width))))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipsim::update_dpp(old, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipsim::mfma_f32_32x32x16_bf16(a, b, c)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) hipsim::global_load_lds((const void*)(g), (void*)(l), size, off)
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipsim::ds_read_tr16_b64((const void*)(p))

inline float unsafeAtomicAdd(float* p, float v) {
  const float o = *p;
  *p = o + v;
  return o;
}
inline float atomicAdd(float* p, float v) { return unsafeAtomicAdd(p, v); }
inline int atomicAdd(int* p, int v) {
  const int o = *p;
  *p = o + v;
  return o;
}
inline unsigned atomicAdd(unsigned* p, unsigned v) {
  const unsigned o = *p;
  *p = o + v;
  return o;
}
inline float __shfl(float v, int src, int width = 64) { return hipsim::shfl(v, src, width); }
inline float __shfl_down(float v, unsigned d, int width = 64) {
  const int l = hipsim::cur->lane;
  const int src = ((l % width) + (int)d < width) ? l + (int)d : l;
  return hipsim::shfl(v, src, 64);
}
inline float __shfl_up(float v, unsigned d, int width = 64) {
  const int l = hipsim::cur->lane;
  const int src = ((l % width) >= (int)d) ? l - (int)d : l;
  return hipsim::shfl(v, src, 64);
}
inline float __shfl_xor(float v, int m, int width = 64) {
  (void)width;
  return hipsim::shfl(v, hipsim::cur->lane ^ m, 64);
}
inline int __shfl(int v, int src, int width = 64) { return hipsim::shfl_i(v, src, width); }
inline int __shfl_down(int v, unsigned d, int width = 64) {
  const int l = hipsim::cur->lane;
  const int src = ((l % width) + (int)d < width) ? l + (int)d : l;
  return hipsim::shfl_i(v, src, 64);
}
inline int __shfl_up(int v, unsigned d, int width = 64) {
  const int l = hipsim::cur->lane;
  const int src = ((l % width) >= (int)d) ? l - (int)d : l;
  return hipsim::shfl_i(v, src, 64);
}
inline int __shfl_xor(int v, int m, int width = 64) {
  (void)width;
  return hipsim::shfl_i(v, hipsim::cur->lane ^ m, 64);
}
inline unsigned long long __ballot(int pred) { return hipsim::ballot(pred); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline unsigned __float_as_uint(float f) { return __builtin_bit_cast(unsigned, f); }
inline float __uint_as_float(unsigned u) { return __builtin_bit_cast(float, u); }
inline int __float_as_int(float f) { return __builtin_bit_cast(int, f); }
inline float __int_as_float(int u) { return __builtin_bit_cast(float, u); }
