// Shared device/host helpers for the MI355X (gfx950) MultiNeRF hot-path kernels.
// gfx950 only: 64-lane wavefronts, bf16 MFMA 32x32x16, LDS-DMA, LDS transpose reads.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mnerf.h"
#include "../../include/mnerf_debug.h"

// `bf16` is the STORAGE type of every activation / gradient / packed-weight matrix the Dense layers read or write.  The product
// library is built with it as __bf16 (the MFMA operand type).  The fp32-Dense debug build (multinerf_amd/build.py: libmnerf_hip_f32.so,
// -DMNR_DENSE_F32: Model(dense_precision='fp32'), SURVEY.md section 7 hard parts 2 / 10) compiles the SAME kernel sources with the
// storage type float and csrc/dense_f32.inc in place of the MFMA GEMMs, so that parity against the plain fp32 oracle can be held
// at ~1e-4 instead of at the bf16 rounding of the operands; a uint16_t* in include/mnerf.h is a float* there.
#ifdef MNR_DENSE_F32
typedef float bf16;
#else
typedef __bf16 bf16;
#endif
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define MNR_F32_EPS 1.1920928955078125e-07f      // jnp.finfo(jnp.float32).eps
#define MNR_F32_MAX 3.4028234663852886e+38f

#define MNR_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MNR_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// Seams for tools/hipsim, the host-side functional simulator the CPU tests run this source through (it stands in
// for <hip/hip_runtime.h> and defines both macros the other way round): tokens that only exist on the GPU
// (inline asm, amdgpu attributes) go through MNR_GPU_ONLY, their restated semantics through MNR_SIM_HOOK.  Here: asm as written, no hook.
#ifndef MNR_GPU_ONLY
#define MNR_GPU_ONLY(...) __VA_ARGS__
#define MNR_SIM_HOOK(...)
#endif

// Error plumbing: thread-local message owned by the library (mnr_last_error()).
void mnr_set_error(const char* fmt, ...);

#define MNR_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      mnr_set_error(__VA_ARGS__);                \
      return MNR_ERR_INVALID_ARGUMENT;           \
    }                                            \
  } while (0)

#define MNR_CHECK_LAUNCH()                                                   \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      mnr_set_error("%s:%d: HIP launch failed: %s", __FILE__, __LINE__,      \
                    hipGetErrorString(e_));                                  \
      return MNR_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)

static inline int mnr_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// CUs of the current device (cached: the GPUs of one node are identical); persistent kernels size their grids with it.
static inline int mnr_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cus;
}

// Lane-per-ray kernels (resample, compositing, the fused level backward) run one long dependent instruction stream per
// wave whatever the number of active lanes, so a batch is spread over `mnr_ray_wave_target()` single-wave workgroups =
// one per SIMD of the chip (4 per CU), each walking B / target rays.  Measured (round 2, 16384 rays, ms per step of the
// fused level backward at 1024 / 2048 / 4096 / 8192 / 16384 waves): 0.64 / 0.81 / 1.13 / 2.2 / 3.9: beyond one wave per
// SIMD the streams queue behind each other (issue-bound, not latency-bound).
static inline long long mnr_ray_wave_target() { return 4ll * mnr_cu_count(); }

// hipFuncSetAttribute applies per device: a launcher's "attribute already set" flag is a bit mask over device ordinals
// (a process that drives a second GPU sets the attribute there too).  True when the current device still needs it.
static inline bool mnr_attr_needed(unsigned long long* mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (*mask & bit) return false;
  *mask |= bit;
  return true;
}

// nan_to_num(x, nan=0) followed by clip to [0,1] (reference internal/math.py:125).
__device__ __forceinline__ float mnr_nan0_clip01(float x) {
  if (x != x) return 0.0f;
  return fminf(fmaxf(x, 0.0f), 1.0f);
}

// (element > 0) flags of 8 bf16 values held as 4 dwords -> one byte, bit e = element e.  Clamp the halves at 0 as signed
// 16-bit (a no-op after a ReLU), then "> 0" is "bits != 0" = min(half, 1) as unsigned 16-bit: a flag in bit 0 and bit 16 of
// each dword (the packed min as inline asm: hipcc turns min(max(x, 0), 1) into a compare + select per half).
__device__ __forceinline__ unsigned mnr_relu_mask_byte(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
  typedef short mnr_s16x2 __attribute__((ext_vector_type(2)));
  const unsigned w[4] = {w0, w1, w2, w3};
  unsigned m[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const mnr_s16x2 z = {0, 0};
    const unsigned pos = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(mnr_s16x2, w[d]), z));
    MNR_GPU_ONLY(asm("v_pk_min_u16 %0, %1, %2" : "=v"(m[d]) : "v"(pos), "v"(0x00010001u)));
    MNR_SIM_HOOK(m[d] = ((pos & 0xffffu) ? 1u : 0u) | ((pos >> 16) ? 0x10000u : 0u));
  }
  const unsigned t = (((m[3] << 2) | m[2]) << 4) | ((m[1] << 2) | m[0]);      // element 2d at bit 2d, 2d+1 at bit 16+2d
  return ((t >> 15) & 0xaau) | (t & 0x55u);
}

// ---- helpers of the hand-pipelined GEMM loops (gemm.hip, gemm_blk.hip) ----
__device__ __forceinline__ void nt_wait_lgkmcnt0() { MNR_GPU_ONLY(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")); }

// A value the optimiser cannot see through: address arithmetic derived from it is recomputed where it is used instead of
// being hoisted out of the tile loop and kept (or spilled) across it.
__device__ __forceinline__ int mnr_opaque(int v) {
  MNR_GPU_ONLY(asm volatile("" : "+v"(v)));
  return v;
}

// lane index 0..63 from the exec-mask prefix count over an opaque zero (not derived from threadIdx.x, not hoistable)
__device__ __forceinline__ int mnr_lane_id() {
#ifdef MNR_HIPSIM
  return (int)threadIdx.x & 63;
#else
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)mnr_opaque(0)));
#endif
}
__device__ __forceinline__ int mnr_opaque_s(int v) {
  MNR_GPU_ONLY(asm volatile("" : "+s"(v)));
  return v;
}
__device__ __forceinline__ void nt_launder(bf16x8& f) { MNR_GPU_ONLY(asm volatile("" : "+v"(f))); }

template <int N>
__device__ __forceinline__ void nt_wait_vmcnt() {
  MNR_GPU_ONLY(asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"));
  MNR_SIM_HOOK(hipsim::wait_vmcnt(N));
}

__device__ __forceinline__ float mnr_softplus(float x) {
  // jax.nn.softplus = logaddexp(x, 0) = max(x,0) + log1p(exp(-|x|)).
  return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
}

__device__ __forceinline__ float mnr_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
