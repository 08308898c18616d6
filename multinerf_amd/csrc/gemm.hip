// bf16 MFMA GEMMs for the Dense stacks (gfx950, v_mfma_f32_32x32x16_bf16).
//
//   gemm_nt : C[M,N] = epi([A1|A2][M,K] * Bt[N,K]^T)   forward layers and dX
//   gemm_tn : C[K,N] += A[M,K]^T * B[M,N]               weight gradients
//
// Both: 256x256 output tile per 512-thread workgroup (8 waves, each a 128x64 block = 4x2 MFMA tiles of
// 32x32, fp32 accumulators in 128 VGPRs; a 128x128 / 4-wave variant serves shapes that are not multiples
// of 256), operand tiles streamed L2 -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4, double-buffered,
// one barrier per K step), XOR-swizzled through the SOURCE address so the linear DMA image is conflict-free
// for the fragment reads.  Workgroup ids are remapped so that the tiles sharing an operand panel run together
// on one XCD (block b lands on XCD b%8; each XCD has a private L2).
//
// What the measurements in DESIGN.md section 6 say about these loops, for whoever tunes them next:
//   * one K step (64 KiB of operands, 256x256x64 MACs) costs ~3.8k cycles; the MFMAs alone 2.05k, the LDS-DMA
//     alone ~2.8k (23-30 B/clk per CU, independent of prefetch depth), and they overlap only partially;
//   * hipcc puts `s_waitcnt vmcnt(0)` in front of LDS reads it cannot separate from an in-flight LDS-DMA
//     (seen with the ds_read_tr builtins in the TN loop: fixed by issuing them from inline asm) and forces
//     vmcnt(0) when register loads and LDS-DMA are in flight together; always check the ISA of the K loop;
//   * per-tile fixed costs matter: the NT epilogue was 15.6k cycles of a ~75k-cycle tile before it was reworked,
//     the TN atomic epilogue makes a second round of workgroups a loss; persistent NT launches (one workgroup per CU
//     walking the tiles) let a tile's stores drain under the next tile's K loop: +2 % end to end.
// Shipped configurations (round-2 A/B on hardware, profiles/r2_gemm_ab.md): NtBig / NtSmall, the weights-resident
// kernel for N = 256, K <= 256 layers, TnBig / TnSmall.  The thirteen other loop structures tried in rounds 1-2
// (phase-interleaved, split operand paths, direct weights, 4-wave 128x128, 256x128 two-per-CU, 3/4-stage BK=32)
// were within +-3 % of these or slower and were removed; DESIGN.md section 6 keeps their numbers.
// The NT kernel's body lives in gemm_nt_body.inc (textually inside the tile loop of the kernel function).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "common.h"

#ifdef MNR_DENSE_F32
// fp32-Dense debug build (common.h): the two Dense entry points on plain fp32 FMAs; everything below the #endif is shared.
#include "dense_f32.inc"
#else

// ---------------------------------------------------------------------------
// NT kernel.
//
// LDS image of one operand tile: [rows][64 k] bf16 = 128 B per row = eight 16-B
// slots.  Slot s of row r is stored at slot position s ^ ((r >> 1) & 7): a
// ds_read_b128 lane group covers 16 distinct rows at one k-slot, and
// (r & 1, (r >> 1) & 7) is distinct for them, so the 16 reads hit 16 distinct
// 16-B slots of the 256-B bank row.
// MFMA roles are swapped (A-operand = weights rows n, B-operand = activations
// rows m) so that a lane's 4 consecutive accumulators are 4 consecutive n of
// one output row m.
//
// Two tile configurations (template NtCfg):
//   small: 128x128 tile, 4 waves (2x2), 64x64 per wave   -- N not a multiple of 256 (heads)
//   big  : 256x256 tile, 8 waves (2 along M x 4 along N), 128x64 per wave -- the trunk layers.
// The big tile halves both the L2->LDS bytes and the LDS fragment reads per MFMA
// (6 ds_read_b128 per 8 MFMAs instead of 4 per 4); at 128x128 the LDS array is
// busy ~as long as the matrix pipe (measured 400-560 TF/s).

#define NT_CPAD 16                                   // epilogue staging: 16 B pad per row

template <int MI_, int NJ_, int WM_, int WN_, int EPI_BATCH_ = 0, int BIAS_LDS_ = 0, int PIPE_ = 0, int DBG_ = 0, int BK_ = 64, int STAGES_ = 2>
struct NtCfg {
  static constexpr int DBG = DBG_;                    // probe builds: 1 no DMA, 2 no MFMA, 3 no fragment reads, 4 MFMA only
  // 1: K loop with explicitly double-buffered fragments (gemm_nt_body.inc), for the one-wave-per-SIMD configuration
  static constexpr bool PIPE = PIPE_ != 0;
  // n > 0: the epilogue's store loop reads its staged chunks from LDS n at a time (n ds_read_b128 in flight per thread)
  // instead of one read per iteration waited for on the spot, and the fp32 side outputs (which read the accumulators)
  // are written before the store loop instead of after it, so that the accumulators' 128 registers are free for the
  // batch (with them alive, 16 reads at once spilled 85 registers).
  static constexpr int EPI_BATCH = EPI_BATCH_;
  static constexpr int MI = MI_, NJ = NJ_, WM = WM_, WN = WN_, BK = BK_, STAGES = STAGES_;
  static constexpr int MINW = 1;                      // __launch_bounds__ min waves per SIMD
  static constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
  static constexpr int THREADS = 64 * WM * WN;
  static constexpr int ROWB = BK * 2;                 // bytes per staged operand row
  static constexpr int SLOTS = ROWB / 16;             // 16-B slots per row (8 at BK=64)
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_OPERANDS = STAGES * STAGE_BYTES;
  static constexpr int LOADS_PER_STAGE = STAGE_BYTES / 16 / THREADS;   // LDS-DMA instructions per wave per stage
  static constexpr int CPITCH = BN * 2 + NT_CPAD;     // bytes per staged output row
  // Rows staged per epilogue pass: the whole tile when it fits the 160 KiB of LDS (one pass, every wave converts
  // and writes at once: 135 KiB for the 256x256 tile), else 128 or 64 rows inside the operand buffers.
  static constexpr int EPI_ROWS = (BM * CPITCH <= 160 * 1024) ? BM : (128 * CPITCH <= LDS_OPERANDS) ? 128 : 64;
  // 1: the tile's bias row waits in LDS during the K loop: one coalesced load per thread in the prologue instead of
  // NJ x 16 scalar loads per lane, and NJ x 16 fewer registers alive across the K loop.
  static constexpr bool BIAS_LDS = BIAS_LDS_ != 0;
  static constexpr int LDS_MAIN = LDS_OPERANDS > EPI_ROWS * CPITCH ? LDS_OPERANDS : EPI_ROWS * CPITCH;
  static constexpr int BIAS_OFF = (LDS_MAIN + 15) / 16 * 16;
  // Pipelined dX layers park the tile's ReLU-mask bits ([BM][BN / 8] bytes) in LDS: one 16-byte load per thread behind the
  // K loop instead of 16 byte loads held in registers across the epilogue (the 256-register kernel spilled them).
  static constexpr bool BITS_LDS = PIPE && BM * BN / 8 == THREADS * 16;
  static constexpr int BITS_OFF = BIAS_LDS ? BIAS_OFF + BN * 4 : LDS_MAIN;
  static constexpr int LDS_BYTES = BITS_OFF + (BITS_LDS ? BM * BN / 8 : 0);
  static_assert(EPI_ROWS * CPITCH <= LDS_BYTES, "epilogue staging must fit in the operand buffers");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(BM % EPI_ROWS == 0 && (32 * MI) <= EPI_ROWS && EPI_ROWS % (32 * MI) == 0, "epilogue pass shape");
  static_assert((BM * SLOTS) % THREADS == 0 && (BN * SLOTS) % THREADS == 0, "stage loop shape");
};

// Stage one operand tile [ROWS][BK] with 16-B LDS-DMA.  LDS image: row r, slot s stored at slot
// position s ^ swz(r) with swz(r) = (r >> 1) & 7 for 128-B rows (BK=64) and (r >> 2) & 3 for 64-B
// rows (BK=32): the 16 rows of a ds_read_b128 lane group then cover 16 distinct 16-B slots of the
// 256-B bank row.  The swizzle is applied to the SOURCE address; the DMA image itself is linear.
template <class CFG, int ROWS>
__device__ __forceinline__ void nt_stage_tile(const bf16* __restrict__ g, int ld, int64_t row0,
                                              int k0, char* lds_tile, int wave, int lane) {
  constexpr int NWAVES = CFG::THREADS / 64;
  constexpr int SLOTS = CFG::SLOTS;
  constexpr int ITERS = ROWS * SLOTS / CFG::THREADS;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int cbase = (i * NWAVES + wave) * 64;
    const int c = cbase + lane;
    const int r = c / SLOTS;
    const int swz = (SLOTS == 8) ? ((r >> 1) & 7) : ((r >> 2) & 3);
    const int slot = (c % SLOTS) ^ swz;             // global k-slot stored at this position
    const bf16* src = g + (row0 + r) * (int64_t)ld + k0 + slot * 8;
    __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(src), MNR_LDS_PTR(lds_tile + cbase * 16), 16, 0, 0);
  }
}

template <class CFG>
__device__ __forceinline__ bf16x8 nt_read_frag(const char* lds_tile, int row, int kslot) {
  const int swz = (CFG::SLOTS == 8) ? ((row >> 1) & 7) : ((row >> 2) & 3);
  const int off = row * CFG::ROWB + ((kslot ^ swz) << 4);
  return *(const bf16x8*)(lds_tile + off);
}

// Profiling hook (tools/step_timeline.py): when set, wave 0 of every workgroup records s_memtime at
// kernel entry, K-loop start, K-loop end and kernel exit into g_nt_timeline[16 * blockIdx.x + 0..3], s_memrealtime
// (100 MHz) at entry / exit into [4], [5], XCC_ID << 32 | HW_ID into [6]; epilogue pass h: staged [8+2h], stored [9+2h].
__device__ unsigned long long* g_nt_timeline = nullptr;
__device__ unsigned long long* g_tn_timeline = nullptr;

extern "C" int mnr_debug_gemm_timeline(unsigned long long* device_buffer) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_nt_timeline), &device_buffer, sizeof(device_buffer));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_tn_timeline), &device_buffer, sizeof(device_buffer));
  if (e != hipSuccess) {
    mnr_set_error("mnr_debug_gemm_timeline: %s", hipGetErrorString(e));
    return MNR_ERR_HIP;
  }
  return MNR_OK;
}

// VCOL (mnr_gemm_nt_args.vcol): one more output column supplied as a vector, kept in LDS behind everything else for the whole
// walk: [K1 + K2] bf16 (at most NT_VCOL_MAX_K) followed by 16 bytes of zeros (the fragment of the lanes that hold other columns)
#define NT_VCOL_MAX_K 1536
#define NT_VCOL_BYTES (NT_VCOL_MAX_K * 2 + 16)

template <class CFG, bool BITS_IN, bool A1_PANEL = false, bool VCOL = false>
__global__ __launch_bounds__(CFG::THREADS, CFG::MINW) void gemm_nt_kernel(mnr_gemm_nt_args p, int fast_epi, long long vtotal) {
  // the wave index lives in an SGPR across the tile loop and the lane index is re-derived per tile (mbcnt): with
  // threadIdx.x itself kept alive across the loop, hipcc spills it and reloads it (behind a vmcnt(0)) at every tile
  const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  unsigned long long* const tl = g_nt_timeline;          // read once: a load per tile is a vmcnt(0) behind the previous tile's stores
  if constexpr (VCOL) {
    // the vector goes to LDS once per workgroup (the first tile's prologue barrier publishes it)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    const int t = (int)threadIdx.x, n16 = (p.K1 + p.K2) / 8;
    if (t < n16) *(u32x4v*)(smem + CFG::LDS_BYTES + t * 16) = *(const u32x4v*)(p.vcol + t * 8);
    if (t == CFG::THREADS - 1) *(u32x4v*)(smem + CFG::LDS_BYTES + NT_VCOL_MAX_K * 2) = u32x4v{0u, 0u, 0u, 0u};
  }
  for (int64_t vbid = blockIdx.x; vbid < vtotal; vbid += gridDim.x) {
#include "gemm_nt_body.inc"
    if (vbid + gridDim.x < vtotal) __syncthreads();      // (persistent launch) this tile's LDS is free for the next one
  }
}

static int g_nt_persist = 1;                             // workgroups per CU of a persistent launch (round-2 A/B: +2 % end to end,
                                                         // bitwise equal); 0: one workgroup per tile

// A/B switch: 0 = one workgroup per output tile; n > 0 = persistent launches of n workgroups per CU for the
// single-resident-workgroup (> 80 KiB LDS) configurations; n < 0 = at most -n workgroups in total (tests).
static int g_nt_pipe = 1;                                // round-2 A/B (profiles/r2_nt_pipe_probe.txt): K loop 3.7k -> 3.1k cycles per 64 k
extern "C" int mnr_gemm_nt_set_pipelined(int on) {
  g_nt_pipe = on;
  return MNR_OK;
}

extern "C" int mnr_gemm_nt_set_persistent(int wgs_per_cu) {
  g_nt_persist = wgs_per_cu;
  return MNR_OK;
}

template <class CFG, bool A1_PANEL = false>
static int nt_launch(const mnr_gemm_nt_args* a, int fast_epi, void* stream) {
  MNR_CHECK_ARG(a->M % CFG::BM == 0 && a->N % CFG::BN == 0, "mnr_gemm_nt_bf16: M=%lld / N=%d not multiples of the %dx%d tile",
                (long long)a->M, a->N, CFG::BM, CFG::BN);
  MNR_CHECK_ARG(a->K1 % CFG::BK == 0 && a->K2 % CFG::BK == 0, "mnr_gemm_nt_bf16: K segments must be multiples of %d", CFG::BK);
  const int nt = a->N / CFG::BN;
  const int64_t mt = a->M / CFG::BM;
  const int64_t groups = (mt + 7) / 8;
  const int64_t vtotal = groups * 8 * nt;                  // virtual workgroups = tiles (M tiles padded to 8 per XCD group)
  MNR_CHECK_ARG(vtotal < (1ll << 31), "mnr_gemm_nt_bf16: grid too large");
  // Persistent launch: one resident workgroup per CU walks the tiles vbid = blockIdx.x, + gridDim.x, ... (a multiple of 8,
  // so a workgroup keeps its XCD's share of the tile order): the stores of a tile's epilogue drain under the next
  // tile's K loop instead of in front of the workgroup's retirement, and the per-workgroup launch gap goes away.
  int64_t grid = vtotal;
  if (g_nt_persist != 0 && CFG::LDS_BYTES > 80 * 1024) {
    const int64_t cap = g_nt_persist > 0 ? (int64_t)g_nt_persist * mnr_cu_count() : -(int64_t)g_nt_persist;
    if (grid > cap && cap >= 8) grid = cap / 8 * 8;
  }
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if constexpr (A1_PANEL) {
    if (a->vcol) {
      // one more column as a vector (the density head next to the bottleneck): a flavour of its own, LDS + NT_VCOL_BYTES
      static unsigned long long attr_set_v = 0;
      constexpr int lds = CFG::LDS_BYTES + NT_VCOL_BYTES;
      static_assert(lds <= 160 * 1024, "LDS");
      if (mnr_attr_needed(&attr_set_v))
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<CFG, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((gemm_nt_kernel<CFG, false, true, true>), dim3((unsigned)grid), dim3(CFG::THREADS), lds, (hipStream_t)stream, *a,
                         fast_epi, (long long)vtotal);
      MNR_CHECK_LAUNCH();
      return MNR_OK;
    }
    if (mnr_attr_needed(&attr_set))
      (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<CFG, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    hipLaunchKernelGGL((gemm_nt_kernel<CFG, false, true>), dim3((unsigned)grid), dim3(CFG::THREADS), CFG::LDS_BYTES,
                       (hipStream_t)stream, *a, fast_epi, (long long)vtotal);
    MNR_CHECK_LAUNCH();
    return MNR_OK;
  }
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<CFG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<CFG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
  }
  if (a->mask_bits_in) {
    hipLaunchKernelGGL((gemm_nt_kernel<CFG, true>), dim3((unsigned)grid), dim3(CFG::THREADS), CFG::LDS_BYTES,
                       (hipStream_t)stream, *a, fast_epi, (long long)vtotal);
  } else {
    hipLaunchKernelGGL((gemm_nt_kernel<CFG, false>), dim3((unsigned)grid), dim3(CFG::THREADS), CFG::LDS_BYTES,
                       (hipStream_t)stream, *a, fast_epi, (long long)vtotal);
  }
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

//            MI NJ WM WN EPI_BATCH BIAS_LDS
typedef NtCfg<2, 2, 2, 2> NtSmall;                  // 128x128, 4 waves, 64 KiB: N not a multiple of 256 (heads, view MLP)
typedef NtCfg<4, 2, 2, 4, 16, 1> NtBig;             // 256x256, 8 waves (2 x 4, 128x64 per wave), 128 KiB + bias row: the trunk layers
// the same tile with the hand-pipelined K loop (gemm_nt_body.inc): BK = 32 x 4 stages, register double-buffered fragments,
// the LDS-DMA pieces of a K-tile issued BETWEEN the MFMAs (default; bitwise equal to NtBig)
typedef NtCfg<4, 2, 2, 4, 8, 1, 1, 0, 32, 4> NtBigP;

// ---------------------------------------------------------------------------
// Weights-resident NT kernel for the short-K layers (N = 256, K <= 256: the proposal MLP's hidden layers and their dX).
//
// There a 256x256 tile of the tiled kernel brings in as many weight bytes as activation bytes, runs only four K steps and
// spends ~40 % of its time in prologue and epilogue.  Here the whole weight matrix (<= 128 KiB) is held in REGISTERS: one
// wave per 32 output columns (the 1x8 layout), 4 fragments per K tile = 64 VGPRs for K = 256, loaded once per workgroup.
// The workgroups are persistent (one per CU) and walk over the M tiles; only activation tiles are streamed (LDS-DMA, two
// 32-KiB slots), as ONE pipeline across tile boundaries: the first activation tile of the next M tile is in flight while
// this tile's epilogue runs, because the epilogue stages through its own LDS region (two passes of 128 rows).
// Round-2 A/B on hardware (bitwise equal to the tiled kernel): the 256-wide layers 282 -> 230 us at M = 2^20; llff_raw +7 %,
// blender_256 +2.5 % end to end.  On by default for eligible launches (mnr_gemm_nt_set_wres(0) switches it off).
typedef NtCfg<8, 1, 1, 8> WresCfg;                  // 256 rows, 8 waves as 1 x 8, for nt_stage_tile / nt_read_frag
#define WRES_A_BYTES (2 * 32768)
#define WRES_STAGE_ROWS 128
#define WRES_CPITCH (256 * 2 + NT_CPAD)
#define WRES_BIAS_OFF (WRES_A_BYTES + WRES_STAGE_ROWS * WRES_CPITCH)
#define WRES_LDS_BYTES (WRES_BIAS_OFF + 256 * 4)
// dX flavour: the tile's ReLU-mask bits ([256 rows][32 bytes] = 8 KiB, contiguous in the bit matrix) ride along with the tile's
// first activation K-tile as one more 16-byte LDS-DMA piece per thread, two buffers by tile parity
#define WRES_BITS_OFF WRES_LDS_BYTES
#define WRES_LDS_BYTES_BITS (WRES_BITS_OFF + 2 * 8192)

template <bool BITS_IN>
__global__ __launch_bounds__(512) void gemm_nt_wres_kernel(mnr_gemm_nt_args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, khalf = lane >> 5;
  const bf16* A = (const bf16*)p.A1;
  const bf16* Bt = (const bf16*)p.Bt;
  bf16* Cb = (bf16*)p.Cb;
  const int nk = p.K1 / 64;                               // 1 .. 4
  const int64_t mt = p.M / 256;
  const int64_t my_tiles = (mt - blockIdx.x + gridDim.x - 1) / gridDim.x;      // tiles blockIdx.x, + gridDim.x, ...
  if (my_tiles <= 0) return;

  // bias row -> LDS (read back per tile in the epilogue)
  {
    const float* bp = p.bias ? p.bias : reinterpret_cast<const float*>(p.Bt);
    const int nbias = p.bias ? p.n_bias : 1;
    if (tid < 256) {
      const float keep = (p.bias != nullptr && tid < nbias) ? 1.0f : 0.0f;
      ((float*)(smem + WRES_BIAS_OFF))[tid] = bp[min(tid, nbias - 1)] * keep;
    }
  }
  // this wave's weights: columns 32 * wave + frow, all of K
  bf16x8 wreg[4][4];
  {
    const bf16* wsrc = Bt + (int64_t)(wave * 32 + frow) * p.ldb + khalf * 8;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16)0.0f;
        wreg[kt][ks] = kt < nk ? *(const bf16x8*)(wsrc + kt * 64 + ks * 16) : z;
      }
    // Make hipcc wait for these loads HERE: left pending into the loop, its waitcnt pass re-waits (vmcnt(0)) at every
    // use, i.e. behind the just-issued LDS-DMA of every step, and serialises the prefetch with the MFMAs (seen in the ISA).
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) MNR_GPU_ONLY(asm volatile("" : "+v"(wreg[kt][ks])));
  }

  f32x16 acc[8];
  const int64_t steps = my_tiles * nk;
  // BITS_IN: the mask bits of a tile through LDS: the tile's 256 rows are one contiguous 8-KiB block of the bit matrix
  // (32 bytes per row, and a row modulus that is a multiple of the tile).  As byte loads in the epilogue they were
  // vector-memory operations issued BEHIND the next tile's LDS-DMA: vmcnt retires in order, so every epilogue waited for the
  // next tile's 32 KiB to arrive from HBM before it could mask its first row (the dX layers of blender_refnerf's tangent
  // network: 3.2 GB per launch at 2.6 TB/s, where the forward flavour of the same kernel streams 6 TB/s).
  // (nt_wres_eligible sends a dX launch here only then; anything else takes the tiled kernel)
  auto stage = [&](int64_t g) {                           // activation tile of global step g into slot g & 1
    const int64_t tile = blockIdx.x + (g / nk) * gridDim.x;
    const int kt = (int)(g % nk);
    nt_stage_tile<WresCfg, 256>(A, p.lda1, tile * 256, kt * 64, smem + (g & 1) * 32768, wave, lane);
    if constexpr (BITS_IN) {
      if (kt == 0) {
        int64_t brow = tile * 256;
        if (p.bits_row_mod > 0) brow %= p.bits_row_mod;
        const uint8_t* src = p.mask_bits_in + brow * 32 + (wave * 64 + lane) * 16;
        __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(src), MNR_LDS_PTR(smem + WRES_BITS_OFF + ((g / nk) & 1) * 8192 + wave * 1024), 16, 0, 0);
      }
    }
  };
  stage(0);
  for (int64_t g = 0; g < steps; ++g) {
    const int kt = (int)(g % nk);
    const int64_t m0 = (blockIdx.x + (g / nk) * gridDim.x) * 256;
    if (kt == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    }
    nt_wait_vmcnt<0>();                                   // A(g) has landed (A(g+1) is not issued yet)
    MNR_GPU_ONLY(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (g + 1 < steps) stage(g + 1);
    const char* As = smem + (g & 1) * 32768;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // (the register array is indexed by a loop-carried kt: select with a uniform switch, not a dynamic index)
      bf16x8 w;
      switch (kt) {
        case 0: w = wreg[0][ks]; break;
        case 1: w = wreg[1][ks]; break;
        case 2: w = wreg[2][ks]; break;
        default: w = wreg[3][ks]; break;
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {               // four row blocks at a time: 16 fragment registers, not 32
        bf16x8 fa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = nt_read_frag<WresCfg>(As, (half * 4 + i) * 32 + frow, ks * 2 + khalf);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[half * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, fa[i], acc[half * 4 + i], 0, 0, 0);
      }
    }
    if (kt != nk - 1) continue;

    // ---- epilogue of this M tile: two passes of 128 rows through the staging region (the next tile's DMA flies on)
    // (the dX flavour has no bias: the launcher only sends bias-free calls to it, which keeps 16 registers out of an
    // epilogue that otherwise makes hipcc spill weight fragments across the whole loop)
    float bias_r[16];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 b4 = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (!BITS_IN) b4 = *(const f32x4*)(smem + WRES_BIAS_OFF + (wave * 32 + rq * 8 + khalf * 4) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) bias_r[rq * 4 + e] = b4[e];
    }
    char* cs = smem + WRES_A_BYTES;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      constexpr int ITERS = WRES_STAGE_ROWS * 32 / 512;     // 8 chunks of 16 B per thread and pass
      const int row0 = tid / 32, ch = tid % 32;
      const int64_t mfirst = m0 + h * WRES_STAGE_ROWS + row0;
      unsigned mbits[BITS_IN ? ITERS : 1];
      if (BITS_IN) {
        const uint8_t* bl = (const uint8_t*)smem + WRES_BITS_OFF + ((g / nk) & 1) * 8192 + (h * WRES_STAGE_ROWS + row0) * 32 + ch;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) mbits[it] = bl[it * 16 * 32];
      }
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = h * 4 + ii;
        const int ml = ii * 32 + frow;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int nl = wave * 32 + rq * 8 + khalf * 4;
          const f32x2 s0 = f32x2{acc[i][rq * 4 + 0], acc[i][rq * 4 + 1]} + f32x2{bias_r[rq * 4 + 0], bias_r[rq * 4 + 1]};
          const f32x2 s1 = f32x2{acc[i][rq * 4 + 2], acc[i][rq * 4 + 3]} + f32x2{bias_r[rq * 4 + 2], bias_r[rq * 4 + 3]};
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          s16x2 h0 = __builtin_bit_cast(s16x2, __builtin_convertvector(s0, bf16x2));
          s16x2 h1 = __builtin_bit_cast(s16x2, __builtin_convertvector(s1, bf16x2));
          if (p.relu) {
            const s16x2 z = {0, 0};
            h0 = __builtin_elementwise_max(h0, z);
            h1 = __builtin_elementwise_max(h1, z);
          }
          typedef int i32x2 __attribute__((ext_vector_type(2)));
          const i32x2 pk = {__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1)};
          *(i32x2*)(cs + ml * WRES_CPITCH + nl * 2) = pk;
        }
      }
      MNR_GPU_ONLY(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      bf16* cptr = Cb + mfirst * p.ldcb + ch * 8;
      uint8_t* bptr = p.mask_bits_out ? p.mask_bits_out + mfirst * p.ld_bits_out + ch : nullptr;
      const char* lptr = cs + row0 * WRES_CPITCH + ch * 16;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        u32x4 w = *(const u32x4*)(lptr + it * 16 * WRES_CPITCH);
        if (BITS_IN) {
          const int mb = (int)mbits[BITS_IN ? it : 0];
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const unsigned lo = (unsigned)__builtin_amdgcn_sbfe(mb, 2 * d, 1);
            const unsigned hi = (unsigned)__builtin_amdgcn_sbfe(mb, 2 * d + 1, 1);
            w[d] &= (lo & 0xffffu) | (hi & 0xffff0000u);
          }
        }
        if (bptr) {                                        // kernel-uniform
          unsigned mb = mnr_relu_mask_byte(w[0], w[1], w[2], w[3]);
          mb |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xF5, 0xf, 0xf, false) << 8;
          mb |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xAA, 0xf, 0xf, false) << 16;
          if ((ch & 3) == 0) *(unsigned*)(bptr + (int64_t)it * 16 * p.ld_bits_out) = mb;
        }
        *(u32x4*)(cptr + (int64_t)it * 16 * p.ldcb) = w;
      }
      MNR_GPU_ONLY(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"));
      __builtin_amdgcn_s_barrier();                         // the staging region is free again
      asm volatile("" ::: "memory");
    }
  }
}

static int g_nt_wres = 1;

// A/B switch: 0 = off; 1 (default) = eligible short-K launches (N = 256, K <= 256, plain epilogues) go to the weights-resident
// persistent kernel, one workgroup per CU; n > 1 = the same with at most n workgroups.
extern "C" int mnr_gemm_nt_set_wres(int max_wgs) {
  g_nt_wres = max_wgs > 0 ? max_wgs : 0;
  return MNR_OK;
}

static bool nt_wres_eligible(const mnr_gemm_nt_args* a, int fast_epi) {
  return a->N == 256 && a->K2 == 0 && a->K1 <= 256 && a->M % 256 == 0 && !a->mask && !a->Cf && a->Cb && a->nb == a->N && (fast_epi & 1) &&
         (!a->mask_bits_out || (a->ld_bits_out % 4 == 0)) && (!a->mask_bits_in || ((a->bits_row_mod == 0 || a->bits_row_mod % 256 == 0) && a->ld_bits_in == 32 && ((uintptr_t)a->mask_bits_in % 16) == 0 &&
                               !a->bias && !a->relu && !a->mask_bits_out));
}

static int nt_wres_launch(const mnr_gemm_nt_args* a, int max_wgs, void* stream) {
  const int cus = mnr_cu_count();
  const int64_t mt = a->M / 256;
  const int cap = max_wgs > 1 ? max_wgs : cus;            // MNR_NT_WRES = 1: one workgroup per CU; n > 1: at most n workgroups
  const int grid = (int)(mt < cap ? mt : cap);
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_wres_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WRES_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_nt_wres_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WRES_LDS_BYTES_BITS);
  }
  if (a->mask_bits_in) hipLaunchKernelGGL(gemm_nt_wres_kernel<true>, dim3(grid), dim3(512), WRES_LDS_BYTES_BITS, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(gemm_nt_wres_kernel<false>, dim3(grid), dim3(512), WRES_LDS_BYTES, (hipStream_t)stream, *a);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

int mnr_gemm_nt_panel_launch(const mnr_gemm_nt_args* a, void* stream);      // gemm_blk.hip

extern "C" int mnr_gemm_nt_bf16(const mnr_gemm_nt_args* a, void* stream) {
  MNR_CHECK_ARG(a != nullptr, "mnr_gemm_nt_bf16: null args");
  MNR_CHECK_ARG((a->a1_layout == MNR_LAYOUT_ROWMAJOR || a->a1_layout == MNR_LAYOUT_PANEL) &&
                    (a->c_layout == MNR_LAYOUT_ROWMAJOR || a->c_layout == MNR_LAYOUT_PANEL),
                "mnr_gemm_nt_bf16: unknown layout %d / %d", a->a1_layout, a->c_layout);
  MNR_CHECK_ARG(a->M > 0 && a->M % 128 == 0, "mnr_gemm_nt_bf16: M=%lld must be a positive multiple of 128", (long long)a->M);
  MNR_CHECK_ARG(a->N > 0 && a->N % 128 == 0, "mnr_gemm_nt_bf16: N=%d must be a positive multiple of 128", a->N);
  // (a panel-layout result runs the BK = 32 kernel of gemm_blk.hip: multiples of 32 there, e.g. the merged head's dX at K = 288)
  const int kgran = a->c_layout == MNR_LAYOUT_PANEL ? 32 : 64;
  MNR_CHECK_ARG(a->K1 > 0 && a->K1 % kgran == 0 && a->K2 >= 0 && a->K2 % kgran == 0,
                "mnr_gemm_nt_bf16: K1=%d, K2=%d must be multiples of %d", a->K1, a->K2, kgran);
  MNR_CHECK_ARG(a->A1 && a->Bt && (a->K2 == 0 || a->A2), "mnr_gemm_nt_bf16: null operand");
  MNR_CHECK_ARG(a->lda1 % 8 == 0 && a->ldb % 8 == 0 && (a->K2 == 0 || a->lda2 % 8 == 0),
                "mnr_gemm_nt_bf16: leading dimensions must be multiples of 8 elements (16 B)");
  MNR_CHECK_ARG(!a->Cb || a->ldcb % 4 == 0, "mnr_gemm_nt_bf16: ldcb must be a multiple of 4");
  MNR_CHECK_ARG(!a->mask || a->ldmask % 4 == 0, "mnr_gemm_nt_bf16: ldmask must be a multiple of 4");
  MNR_CHECK_ARG(a->Cb || a->Cf, "mnr_gemm_nt_bf16: no output");
  MNR_CHECK_ARG(!(a->mask_bits_out && a->mask), "mnr_gemm_nt_bf16: mask_bits_out cannot be combined with a bf16 mask");
  MNR_CHECK_ARG(!a->bias || a->n_bias >= 1, "mnr_gemm_nt_bf16: bias needs n_bias >= 1");
  MNR_CHECK_ARG(!a->mask_bits_in || (!a->bias && !a->relu), "mnr_gemm_nt_bf16: mask_bits_in (a dX layer) takes no bias and no ReLU");
  MNR_CHECK_ARG(!a->mask_bits_in || a->bits_row_mod == 0 || a->bits_row_mod >= 256, "mnr_gemm_nt_bf16: bits_row_mod must be 0 or >= 256");
  MNR_CHECK_ARG(!a->mask_bits_out || a->c_layout == MNR_LAYOUT_PANEL ||
                    (a->Cb && a->nb == a->N && a->ldcb % 8 == 0 && ((uintptr_t)a->Cb % 16) == 0 &&
                     a->ld_bits_out % 4 == 0 && ((uintptr_t)a->mask_bits_out % 4) == 0),
                "mnr_gemm_nt_bf16: mask_bits_out needs a full-width, 16-byte-aligned bf16 output and a 4-byte-aligned bit matrix");
  // 16-byte row segments in the epilogue need 8-element-aligned output / mask pitches and bases.
  const int fast_epi = (int)((!a->Cb || (a->ldcb % 8 == 0 && ((uintptr_t)a->Cb % 16) == 0)) &&
                             (!a->mask || (a->ldmask % 8 == 0 && ((uintptr_t)a->mask % 16) == 0)));
  MNR_CHECK_ARG(!a->vcol || (a->a1_layout == MNR_LAYOUT_PANEL && a->c_layout == MNR_LAYOUT_ROWMAJOR),
                "mnr_gemm_nt_bf16: vcol goes with a panel-layout A1 and a row-major result");
  if (a->c_layout == MNR_LAYOUT_PANEL) return mnr_gemm_nt_panel_launch(a, stream);
  if (a->a1_layout == MNR_LAYOUT_PANEL) {
    // a panel-layout activation into a row-major result (the merged head behind the trunk): the pipelined tiled kernel
    MNR_CHECK_ARG(a->M % 256 == 0 && a->N % 256 == 0 && a->lda1 == a->K1 && a->K1 % 32 == 0 && !a->mask_bits_in,
                  "mnr_gemm_nt_bf16: a panel-layout A1 needs M, N multiples of 256, lda1 == K1 and a forward epilogue");
    MNR_CHECK_ARG(!a->vcol || (a->vcol_out && a->N == 256 && a->K1 + a->K2 <= NT_VCOL_MAX_K && !a->Cf && a->Cb && a->nb == a->N &&
                               (fast_epi & 1) && ((uintptr_t)a->vcol % 16) == 0 && ((uintptr_t)a->vcol_out % 4) == 0),
                  "mnr_gemm_nt_bf16: vcol needs vcol_out, N == 256, K1 + K2 <= %d, a full-width 16-byte-aligned bf16 result, no fp32 side output",
                  NT_VCOL_MAX_K);
    return nt_launch<NtBigP, true>(a, fast_epi, stream);
  }
  if (g_nt_wres > 0 && nt_wres_eligible(a, fast_epi)) return nt_wres_launch(a, g_nt_wres, stream);
#ifdef MNR_NT_DEBUG_VARIANTS      // probe build (tools/nt_pipe_probe.py): the pipelined loop with one ingredient removed
  if (a->M % 256 == 0 && a->N % 256 == 0) {
    if (g_nt_pipe == 11) return nt_launch<NtCfg<4, 2, 2, 4, 8, 1, 1, 1, 32, 4>>(a, fast_epi, stream);      // no DMA
    if (g_nt_pipe == 12) return nt_launch<NtCfg<4, 2, 2, 4, 8, 1, 1, 2, 32, 4>>(a, fast_epi, stream);      // no MFMA
    if (g_nt_pipe == 13) return nt_launch<NtCfg<4, 2, 2, 4, 8, 1, 1, 3, 32, 4>>(a, fast_epi, stream);      // no fragment reads
    if (g_nt_pipe == 14) return nt_launch<NtCfg<4, 2, 2, 4, 8, 1, 1, 4, 32, 4>>(a, fast_epi, stream);      // MFMA only
  }
#endif
  // (the pipelined dX flavour loads 16-byte rows of the bit tile)
  const bool bits16 = !a->mask_bits_in || (a->ld_bits_in % 16 == 0 && ((uintptr_t)a->mask_bits_in % 16) == 0);
  if (g_nt_pipe && a->M % 256 == 0 && a->N % 256 == 0 && bits16) return nt_launch<NtBigP>(a, fast_epi, stream);
  if (a->M % 256 == 0 && a->N % 256 == 0) return nt_launch<NtBig>(a, fast_epi, stream);
  return nt_launch<NtSmall>(a, fast_epi, stream);
}

// ---------------------------------------------------------------------------
// TN kernel (weight gradients): C[k,n] += sum_m A[m,k] B[m,n].
//
// The reduction index m is the slow (row) index of both operands, so MFMA
// fragments (8 consecutive reduction elements per lane) are produced with the
// gfx950 LDS transpose read ds_read_b64_tr_b16: per 16-lane group it reads a
// [4 m][16 cols] block (lane p supplies the address of row p>>2, 8-byte chunk
// p&3) and hands lane c the 4 m-values of column c.  Two reads give the 8
// m-values of one MFMA k-step half.  LDS image of a tile: [64 m][128 cols] bf16
// = 256 B per row = four 64-B blocks; block b of row r is stored at block
// position b ^ (r & 3), so the 4 rows of one transpose read (r & 3 = 0..3) use
// four different 64-B bank ranges.

#define TN_BM 64       // reduction rows per step

template <int KI_, int NJ_, int WK_, int WN_>
struct TnCfg {
  static constexpr int KI = KI_, NJ = NJ_, WK = WK_, WN = WN_;
  static constexpr int BKO = 32 * KI * WK;            // output rows (columns of A)
  static constexpr int BNO = 32 * NJ * WN;            // output cols (columns of B)
  static constexpr int THREADS = 64 * WK * WN;
  static constexpr int SBM = 32, STAGES = 4;          // reduction rows per stage buffer, stage buffers
  static constexpr int A_BYTES = SBM * BKO * 2, B_BYTES = SBM * BNO * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static constexpr int LOADS_PER_STAGE = STAGE_BYTES / 16 / THREADS;
};
typedef TnCfg<2, 2, 2, 2> TnSmall;   // 128x128 output tile, 4 waves,  64 KiB
typedef TnCfg<4, 2, 2, 4> TnBig;     // 256x256 output tile, 8 waves, 128 KiB


static inline int mnr_gcd(int a, int b) {
  while (b) {
    const int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

template <class CFG, bool A_PANEL = false, bool B_PANEL = false>
__global__ __launch_bounds__(CFG::THREADS) void gemm_tn_kernel(mnr_gemm_tn_args p, int splits, int steps_per_split) {
  constexpr bool GCOL = false, RANK1 = false;
#include "gemm_tn_body.inc"
}

// The same with one more column of B supplied as an fp32 vector (mnr_gemm_tn_args.gcol): a kernel of its own, so that the
// weight-gradient GEMMs without it keep their registers and schedule.
template <class CFG, bool A_PANEL = false, bool B_PANEL = false>
__global__ __launch_bounds__(CFG::THREADS) void gemm_tn_gcol_kernel(mnr_gemm_tn_args p, int splits, int steps_per_split) {
  constexpr bool GCOL = true, RANK1 = false;
#include "gemm_tn_body.inc"
}

// B built inside the kernel from its rank-1 factors and the ReLU mask bits (mnr_gemm_tn_args.rank1_*): the proposal MLP's last dY
// is never stored (1 GB written by mlp_chain_bwd and 1 GB read back here per step of 360.gin).
template <class CFG>
__global__ __launch_bounds__(CFG::THREADS) void gemm_tn_rank1_kernel(mnr_gemm_tn_args p, int splits, int steps_per_split) {
  constexpr bool GCOL = false, RANK1 = true, A_PANEL = false, B_PANEL = false;
#include "gemm_tn_body.inc"
}

template <class CFG, bool G = false, bool AP = false, bool BP = false, bool R1 = false>
static int tn_launch(const mnr_gemm_tn_args* a, int target_wgs, void* stream) {
  const int tiles = (a->K / CFG::BKO) * (a->N / CFG::BNO);
  const int total_steps = (int)(a->M / TN_BM);
  // Enough M-splits for ~target_wgs workgroups.  The grid (splits x tiles) is a multiple of 8 and workgroup b belongs to
  // XCD b % 8: the kernel hands XCD x the (split, tile) pairs [x * grid / 8, (x + 1) * grid / 8) in split-major order, so
  // the tiles of one split (which share its operand rows) run on one XCD (two when a split straddles a boundary).
  const int unit = 8 / mnr_gcd(tiles, 8);                  // splits come in multiples of this
  int splits = ((target_wgs + tiles - 1) / tiles + 7) / 8 * 8;
  if (splits < unit) splits = unit;
  // (every workgroup ends with an atomic epilogue of its whole fp32 tile: a split must be long enough to pay for it)
  constexpr int min_steps = 4;
  while (splits > unit && (total_steps + splits - 1) / splits < min_steps) splits -= unit;
  // block-cyclic M-tiles (m_interleave) need every split to get the same whole number of 256-row M-tiles (4 steps of 64 rows).  The
  // split count depends on the shape and on the CU count (8 at W = 1024 on 256 CUs, 16 at 768, up to 32 at 512), so a caller cannot
  // know it: where it does not divide, the launch runs with contiguous splits (the flag is performance only, include/mnerf.h).
  mnr_gemm_tn_args local = *a;
  if (local.m_interleave && total_steps % (4 * splits) != 0) local.m_interleave = 0;
  a = &local;
  const int steps_per_split = (total_steps + splits - 1) / splits;
  const int64_t grid = (int64_t)splits * tiles;
  // LDS: a panel operand's stage image carries 128 bytes of padding per 1-KiB block (gemm_tn_body.inc)
  constexpr int lds = CFG::STAGES * ((AP ? (CFG::BKO / 16) * 1152 : CFG::A_BYTES) + (BP ? (CFG::BNO / 16) * 1152 : CFG::B_BYTES)) +
                      (R1 ? CFG::STAGES * 1280 + 128 : 0);      // (rank1: the ring of factor slots + the mask table, gemm_tn_body.inc)
  static_assert(lds <= 160 * 1024, "LDS");
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set)) {
    if constexpr (R1) (void)hipFuncSetAttribute((const void*)gemm_tn_rank1_kernel<CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else if constexpr (G) (void)hipFuncSetAttribute((const void*)gemm_tn_gcol_kernel<CFG, AP, BP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    else (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<CFG, AP, BP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  if constexpr (R1)
    hipLaunchKernelGGL((gemm_tn_rank1_kernel<CFG>), dim3((unsigned)grid), dim3(CFG::THREADS), lds, (hipStream_t)stream, *a, splits,
                       steps_per_split);
  else if constexpr (G)
    hipLaunchKernelGGL((gemm_tn_gcol_kernel<CFG, AP, BP>), dim3((unsigned)grid), dim3(CFG::THREADS), lds, (hipStream_t)stream,
                       *a, splits, steps_per_split);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<CFG, AP, BP>), dim3((unsigned)grid), dim3(CFG::THREADS), lds, (hipStream_t)stream,
                       *a, splits, steps_per_split);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_gemm_tn_bf16(const mnr_gemm_tn_args* a, void* stream) {
  MNR_CHECK_ARG(a != nullptr, "mnr_gemm_tn_bf16: null args");
  MNR_CHECK_ARG(a->M > 0 && a->M % TN_BM == 0, "mnr_gemm_tn_bf16: M=%lld must be a positive multiple of 64", (long long)a->M);
  MNR_CHECK_ARG(a->K > 0 && a->K % 128 == 0 && a->N > 0 && a->N % 128 == 0,
                "mnr_gemm_tn_bf16: K=%d, N=%d must be multiples of 128", a->K, a->N);
  const bool rank1 = a->rank1_g != nullptr;
  MNR_CHECK_ARG(a->A && (a->B || rank1) && a->C, "mnr_gemm_tn_bf16: null operand");
  MNR_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0, "mnr_gemm_tn_bf16: lda/ldb must be multiples of 8");
  // The 256x256 tile halves operand traffic per MFMA and issues 4x the atomics per workgroup: it is used whenever K and N are
  // multiples of 256 (the 128x128 tile for the 256-wide layers measured slower in round 3, profiles/HISTORY.md).
  const bool big = (a->K % 256 == 0) && (a->N % 256 == 0);
  // one workgroup per CU: every workgroup ends with a 256 KiB fp32 atomic epilogue (21-88k cycles, bound by the
  // L2's atomic rate, tools/step_timeline.py), so a second round of workgroups only adds epilogues:
  // 512 -> 256 workgroups = 398k -> 407k rays/s end to end, 1024: 390k; 128 / 512 re-measured in round 3 (profiles/HISTORY.md).
  const int tn_target = (a->max_wgs > 0 && a->max_wgs < mnr_cu_count()) ? a->max_wgs : mnr_cu_count();
  MNR_CHECK_ARG(!a->gcol || (a->gcol_out && a->K % 256 == 0 && a->N % 256 == 0 && ((uintptr_t)a->gcol % 32) == 0),
                "mnr_gemm_tn_bf16: gcol needs gcol_out, K and N multiples of 256 and a 32-byte-aligned vector");
  const bool ap = a->a_layout == MNR_LAYOUT_PANEL, bp = a->b_layout == MNR_LAYOUT_PANEL;
  if (rank1) {
    MNR_CHECK_ARG(!a->B && !a->gcol && a->a_layout == MNR_LAYOUT_ROWMAJOR && a->b_layout == MNR_LAYOUT_ROWMAJOR && big,
                  "mnr_gemm_tn_bf16: rank1 goes with B == NULL, no gcol, row-major A, K and N multiples of 256");
    MNR_CHECK_ARG(a->rank1_w && a->rank1_bits && ((uintptr_t)a->rank1_g % 4) == 0 && ((uintptr_t)a->rank1_w % 16) == 0 &&
                      ((uintptr_t)a->rank1_bits % 4) == 0 && a->ld_rank1_bits % 4 == 0 && a->ld_rank1_bits * 8 >= a->N,
                  "mnr_gemm_tn_bf16: rank1 needs g (fp32 [M]), w (fp32 [N], 16-byte aligned) and mask rows of >= N bits with a pitch that is a multiple of 4");
    return tn_launch<TnBig, false, false, false, true>(a, tn_target, stream);
  }
  MNR_CHECK_ARG((ap || a->a_layout == MNR_LAYOUT_ROWMAJOR) && (bp || a->b_layout == MNR_LAYOUT_ROWMAJOR), "mnr_gemm_tn_bf16: unknown layout");
  if (ap || bp) {
    // panel operands: the 256 x 256 tile only (the 1024-wide trunk's weight gradients)
    MNR_CHECK_ARG(a->K % 256 == 0 && a->N % 256 == 0 && (!ap || a->lda % 16 == 0) && (!bp || a->ldb % 16 == 0) &&
                      ((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->B % 16) == 0,
                  "mnr_gemm_tn_bf16: panel operands need K, N multiples of 256, leading dimensions multiples of 16, 16-byte-aligned bases");
    if (a->gcol) {
      MNR_CHECK_ARG(ap && !bp, "mnr_gemm_tn_bf16: the extra column goes with a panel A and a row-major B");
      return tn_launch<TnBig, true, true, false>(a, tn_target, stream);
    }
    if (ap && bp) return tn_launch<TnBig, false, true, true>(a, tn_target, stream);
    if (bp) return tn_launch<TnBig, false, false, true>(a, tn_target, stream);
    return tn_launch<TnBig, false, true, false>(a, tn_target, stream);
  }
  if (a->gcol) return tn_launch<TnBig, true>(a, tn_target, stream);
  if (big) return tn_launch<TnBig>(a, tn_target, stream);
  return tn_launch<TnSmall>(a, 3 * mnr_cu_count(), stream);         // (the 64-KiB tile: up to three workgroups per CU)
}

#endif  // MNR_DENSE_F32

// ---------------------------------------------------------------------------
// Bias gradient: out[n] += sum_m X[m,n].

__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ X, int ld, int64_t M,
                                                      int n_valid, int rows_per_block, float* out) {
  __shared__ float red[256 * 8];
  const int groups = (n_valid + 7) / 8;              // 16-B column groups
  const int lanes_per_row = groups;                  // threads across columns
  const int row_lanes = 256 / lanes_per_row;         // >= 1 (host guarantees groups <= 256)
  const int cg = threadIdx.x % lanes_per_row;
  const int rl = threadIdx.x / lanes_per_row;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.0f;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  if (rl < row_lanes) {
    for (int64_t r = r0 + rl; r < r1; r += row_lanes) {
      const bf16x8 v = *(const bf16x8*)(X + r * ld + cg * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < row_lanes; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += red[(k * lanes_per_row + cg) * 8 + e];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (cg * 8 + e < n_valid) unsafeAtomicAdd(out + cg * 8 + e, s[e]);
  }
}

extern "C" int mnr_colsum_bf16(const uint16_t* X, int ld, int64_t M, int n_valid, float* out, void* stream) {
  MNR_CHECK_ARG(X && out && M > 0 && n_valid > 0, "mnr_colsum_bf16: bad arguments");
  MNR_CHECK_ARG(ld % 8 == 0 && (n_valid + 7) / 8 <= 256 && ((n_valid + 7) / 8) * 8 <= ld,
                "mnr_colsum_bf16: need ld %% 8 == 0 and n_valid <= min(ld, 2048)");
  const int rows_per_block = 1024;
  const int grid = mnr_cdiv(M, rows_per_block);
  hipLaunchKernelGGL(colsum_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)X, ld, M,
                     n_valid, rows_per_block, out);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// Weight packing: fp32 flax kernels -> padded bf16 GEMM operands (one launch for the whole table).

__global__ void pack_weights_kernel(const float* __restrict__ params, const mnr_pack_desc* __restrict__ descs,
                                    bf16* __restrict__ dst) {
  const mnr_pack_desc d = descs[blockIdx.y];
  const int total = d.rows_in * d.cols_out;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int r, c;
    if (d.transpose) {          // consecutive threads walk the destination row (= source column)
      c = e / d.rows_in;
      r = e % d.rows_in;
    } else {
      r = e / d.cols_out;
      c = e % d.cols_out;
    }
    const float v = params[d.src_off + (int64_t)r * d.cols_out + c];
    const int64_t o = d.transpose ? d.dst_off + (int64_t)(d.row0 + c) * d.ld + d.col0 + r
                                  : d.dst_off + (int64_t)(d.row0 + r) * d.ld + d.col0 + c;
    dst[o] = (bf16)v;
  }
}

extern "C" int mnr_pack_weights_bf16(const float* params, const mnr_pack_desc* descs_device, int n_desc,
                                     int max_elems, uint16_t* dst, void* stream) {
  MNR_CHECK_ARG(params && descs_device && dst && n_desc > 0 && max_elems > 0, "mnr_pack_weights_bf16: bad arguments");
  int gx = mnr_cdiv(max_elems, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(gx, n_desc), dim3(256), 0, (hipStream_t)stream, params,
                     descs_device, (bf16*)dst);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

__global__ void scatter_add_kernel(const float* __restrict__ src, int ld_src, int row0, int col0, int rows,
                                   int cols, float* __restrict__ dst, int ld_dst) {
  const int total = rows * cols;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int k = e / cols, c = e % cols;
    dst[(int64_t)k * ld_dst + c] += src[(int64_t)(row0 + k) * ld_src + col0 + c];
  }
}

extern "C" int mnr_scatter_add_f32(const float* src, int ld_src, int row0, int col0, int rows, int cols,
                                   float* dst, int ld_dst, void* stream) {
  MNR_CHECK_ARG(src && dst && rows > 0 && cols > 0, "mnr_scatter_add_f32: bad arguments");
  int grid = mnr_cdiv((long long)rows * cols, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(scatter_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, row0,
                     col0, rows, cols, dst, ld_dst);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, int ld_src, int64_t M, int n,
                                     bf16* __restrict__ dst, int ld_dst, int col0) {
  const int64_t total = M * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / n;
    const int c = (int)(e % n);
    dst[r * ld_dst + col0 + c] = (bf16)src[r * ld_src + c];
  }
}

extern "C" int mnr_cast_f32_to_bf16(const float* src, int ld_src, int64_t M, int n, uint16_t* dst, int ld_dst,
                                    int col0, void* stream) {
  MNR_CHECK_ARG(src && dst && M > 0 && n > 0, "mnr_cast_f32_to_bf16: bad arguments");
  int grid = mnr_cdiv(M * n, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, M, n,
                     (bf16*)dst, ld_dst, col0);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// Non-ReLU net_activation (reference internal/models.py:348,457,578; the reference registers jax.nn.softplus and jax.nn.silu,
// internal/configs.py:29-31): the Dense GEMM stores the bf16 PRE-activation z, these two kernels apply the activation
// (forward) and its derivative (backward, in place on the gradient) in fp32.  kind: 1 = softplus, 2 = silu.
__device__ __forceinline__ float mnr_act_apply(int kind, float z) {
  if (kind == 1) return mnr_softplus(z);
  return z * mnr_sigmoid(z);
}
__device__ __forceinline__ float mnr_act_deriv(int kind, float z) {
  const float sg = mnr_sigmoid(z);
  if (kind == 1) return sg;
  return sg + z * sg * (1.0f - sg);
}

template <bool BWD>
__global__ __launch_bounds__(256) void act_bf16_kernel(int kind, int64_t chunks, const bf16* __restrict__ z, bf16* __restrict__ io) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < chunks; e += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8 zv = *(const bf16x8*)(z + e * 8);
    bf16x8 v;
    if constexpr (BWD) v = *(const bf16x8*)(io + e * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (BWD) v[i] = (bf16)((float)v[i] * mnr_act_deriv(kind, (float)zv[i]));
      else v[i] = (bf16)mnr_act_apply(kind, (float)zv[i]);
    }
    *(bf16x8*)(io + e * 8) = v;
  }
}

static int act_launch(bool bwd, int kind, int64_t n, const uint16_t* z, uint16_t* io, void* stream) {
  MNR_CHECK_ARG((kind == 1 || kind == 2) && z && io && n > 0 && n % 8 == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)io % 16) == 0,
                "mnr_act_*_bf16: kind 1 (softplus) / 2 (silu), n a multiple of 8, 16-byte-aligned pointers");
  const int64_t chunks = n / 8;
  const int64_t want = (chunks + 255) / 256;
  const int grid = (int)(want > 16384 ? 16384 : want);
  if (bwd) hipLaunchKernelGGL(act_bf16_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, kind, chunks, (const bf16*)z, (bf16*)io);
  else hipLaunchKernelGGL(act_bf16_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, kind, chunks, (const bf16*)z, (bf16*)io);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_act_fwd_bf16(int kind, int64_t n, const uint16_t* z, uint16_t* a, void* stream) {
  return act_launch(false, kind, n, z, a, stream);
}

// The tangent network of the density-gradient normals (models.py:478-492 in forward mode) behind a non-ReLU
// activation.  With z [M, W] the primal pre-activation of a layer and U [3 M, W] the tangent pre-activation (rows c * M + s =
// direction c of sample s):
//   forward : T = act'(z) * U
//   backward: given G = d loss / d T:  S = G * act'(z)  (in place: what the weight-gradient and dX GEMMs take), and the term a
//             ReLU network does not have,  extra[s] = sum_c G[c M + s] * U[c M + s] * act''(z[s])  = d loss / d z through act',
//             which joins the PRIMAL backward pass's gradient of that layer (autodiff's second-order term of the "double backward").
__device__ __forceinline__ float mnr_act_deriv2(int kind, float z) {
  const float sg = mnr_sigmoid(z);
  const float d = sg * (1.0f - sg);
  if (kind == 1) return d;                                         // softplus'' = sigmoid'
  return d * (2.0f + z * (1.0f - 2.0f * sg));                      // silu'' = sigmoid' (2 + z (1 - 2 sigmoid))
}

template <bool BWD>
__global__ __launch_bounds__(256) void act_tangent_kernel(int kind, int64_t chunks, const bf16* __restrict__ z, const bf16* __restrict__ U,
                                                          bf16* __restrict__ io, bf16* __restrict__ extra) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < chunks; e += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8 zv = *(const bf16x8*)(z + e * 8);
    float d1[8], d2[8], acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      d1[i] = mnr_act_deriv(kind, (float)zv[i]);
      d2[i] = BWD ? mnr_act_deriv2(kind, (float)zv[i]) : 0.0f;
      acc[i] = 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int64_t o = (c * chunks + e) * 8;
      const bf16x8 u = *(const bf16x8*)(U + o);
      bf16x8 v;
      if constexpr (BWD) {
        const bf16x8 g = *(const bf16x8*)(io + o);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] += (float)g[i] * (float)u[i] * d2[i];
          v[i] = (bf16)((float)g[i] * d1[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16)((float)u[i] * d1[i]);
      }
      *(bf16x8*)(io + o) = v;
    }
    if constexpr (BWD) {
      bf16x8 x;
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = (bf16)acc[i];
      *(bf16x8*)(extra + e * 8) = x;
    }
  }
}

static int act_tangent_launch(bool bwd, int kind, int64_t n, const uint16_t* z, const uint16_t* U, uint16_t* io, uint16_t* extra, void* stream) {
  MNR_CHECK_ARG((kind == 1 || kind == 2) && z && U && io && (!bwd || extra) && n > 0 && n % 8 == 0 && ((uintptr_t)z % 16) == 0 &&
                    ((uintptr_t)U % 16) == 0 && ((uintptr_t)io % 16) == 0 && ((uintptr_t)extra % 16) == 0,
                "mnr_act_tangent_*_bf16: kind 1 (softplus) / 2 (silu), n = M * W a multiple of 8, 16-byte-aligned pointers");
  const int64_t chunks = n / 8;
  const int64_t want = (chunks + 255) / 256;
  const int grid = (int)(want > 16384 ? 16384 : want);
  if (bwd) hipLaunchKernelGGL(act_tangent_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, kind, chunks, (const bf16*)z, (const bf16*)U, (bf16*)io, (bf16*)extra);
  else hipLaunchKernelGGL(act_tangent_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, kind, chunks, (const bf16*)z, (const bf16*)U, (bf16*)io, (bf16*)extra);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_act_tangent_fwd_bf16(int kind, int64_t n, const uint16_t* z, const uint16_t* U, uint16_t* T, void* stream) {
  return act_tangent_launch(false, kind, n, z, U, T, nullptr, stream);
}

extern "C" int mnr_act_tangent_bwd_bf16(int kind, int64_t n, const uint16_t* z, const uint16_t* U, uint16_t* G, uint16_t* extra, void* stream) {
  return act_tangent_launch(true, kind, n, z, U, G, extra, stream);
}

extern "C" int mnr_act_bwd_bf16(int kind, int64_t n, const uint16_t* z, uint16_t* d, void* stream) {
  return act_launch(true, kind, n, z, d, stream);
}

// X[m, c] = bf16(float(X[m, c]) + scale * noise[m, c]) for c < cols (a multiple of 8): the bottleneck noise of
// reference internal/models.py:530-533, added to the bf16 bottleneck columns of the view-MLP input in fp32.
__global__ __launch_bounds__(256) void add_noise_bf16_kernel(int64_t M, int cols, bf16* __restrict__ X, int ld,
                                                             const float* __restrict__ noise, float scale) {
  const int cpr = cols / 8;
  const int64_t total = M * cpr;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cpr;
    const int c = (int)(e % cpr) * 8;
    bf16x8 v = *(const bf16x8*)(X + r * ld + c);
    const f32x4 n0 = *(const f32x4*)(noise + r * cols + c), n1 = *(const f32x4*)(noise + r * cols + c + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = (bf16)((float)v[i] + scale * n0[i]);
      v[4 + i] = (bf16)((float)v[4 + i] + scale * n1[i]);
    }
    *(bf16x8*)(X + r * ld + c) = v;
  }
}

extern "C" int mnr_add_noise_bf16(int64_t M, int cols, uint16_t* X, int ld, const float* noise, float scale, void* stream) {
  MNR_CHECK_ARG(X && noise && M > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols &&
                    ((uintptr_t)X % 16) == 0 && ((uintptr_t)noise % 16) == 0,
                "mnr_add_noise_bf16: cols and ld must be multiples of 8, pointers 16-byte aligned");
  int64_t want = (M * (cols / 8) + 255) / 256;
  const int grid = (int)(want > 8192 ? 8192 : want);
  hipLaunchKernelGGL(add_noise_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, cols, (bf16*)X, ld, noise, scale);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------
// Small-N head VJP (e.g. rgb Dense(3)): dX = relu'(H) * (g W^T), dW += H^T g, db += sum g.

template <int U>
__global__ __launch_bounds__(256) void small_head_bwd_kernel(int64_t M, int K, int C, const bf16* __restrict__ H,
                                                              int ldh, const float* __restrict__ g,
                                                              const float* __restrict__ W, bf16* __restrict__ dX,
                                                              int lddx, int relu_mask, float* dW, float* db,
                                                              int rows_per_block, const uint8_t* __restrict__ mbits,
                                                              int ld_bits, int64_t bits_row_mod,
                                                              float* __restrict__ partials) {
  // Thread (rl, cg): column group cg of 8 consecutive k (16-byte accesses), row lane rl; the block's
  // rows are strided over the row lanes.  dW partials are reduced across row lanes through LDS.
  __shared__ float red[256 * 8];
  const int groups = K / 8;                       // host guarantees K % 8 == 0 and groups <= 256
  const int row_lanes = 256 / groups;
  const int cg = threadIdx.x % groups;
  const int rl = threadIdx.x / groups;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  float w[8][4], aw[8][4];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      w[e][c] = (c < C) ? W[(int64_t)(cg * 8 + e) * C + c] : 0.0f;
      aw[e][c] = 0.0f;
    }
  float sb[4] = {0.0f, 0.0f, 0.0f, 0.0f};       // bias gradient partial (row lanes of column group 0)
  if (rl < row_lanes) {
    // 4 rows per trip, all loads issued before the arithmetic: one row per trip leaves a single 16-byte load
    // in flight per thread and the kernel runs at ~1/5 of HBM bandwidth (profiles/r1_h: 475 us per call).
    for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)row_lanes * U) {
      bf16x8 h[U];
      float gc[U][4];
      unsigned mb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = min(rb + (int64_t)u * row_lanes, r1 - 1);        // clamped: loads stay in range
        h[u] = *(const bf16x8*)(H + r * ldh + cg * 8);
#pragma unroll
        for (int c = 0; c < 4; ++c) gc[u][c] = (c < C) ? g[r * C + c] : 0.0f;
        mb[u] = 0xffu;
        if (mbits) mb[u] = mbits[(bits_row_mod > 0 ? r % bits_row_mod : r) * ld_bits + cg];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = rb + (int64_t)u * row_lanes;
        if (r >= r1) {
#pragma unroll
          for (int c = 0; c < 4; ++c) gc[u][c] = 0.0f;                      // clamped duplicate: contributes nothing
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) sb[c] += gc[u][c];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float hv = (float)h[u][e];
          float gx = 0.0f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            gx += gc[u][c] * w[e][c];
            aw[e][c] += hv * gc[u][c];
          }
          o[e] = (bf16)(((relu_mask && !(hv > 0.0f)) || !((mb[u] >> e) & 1u)) ? 0.0f : gx);
        }
        if (dX && r < r1) *(bf16x8*)(dX + r * lddx + cg * 8) = o;
      }
    }
  }
  if (dW) {
    for (int c = 0; c < C; ++c) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = aw[e][c];
      __syncthreads();
      if (rl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float sum = 0.0f;
          for (int k2 = 0; k2 < row_lanes; ++k2) sum += red[(k2 * groups + cg) * 8 + e];
          // with a scratch buffer: this workgroup's partial (reduced by small_head_reduce_kernel); without: K*C
          // atomics per workgroup onto the same few cache lines, which bounds the kernel once there are >~512 of them
          if (partials) partials[(int64_t)blockIdx.x * (K * C + C) + (cg * 8 + e) * C + c] = sum;
          else unsafeAtomicAdd(dW + (int64_t)(cg * 8 + e) * C + c, sum);
        }
      }
    }
  }
  if (partials && db) {
    // bias partial of this workgroup: reduce the row lanes' sums through LDS
    __syncthreads();
    if (cg == 0 && rl < row_lanes)
      for (int c = 0; c < C; ++c) red[rl * 4 + c] = sb[c];
    __syncthreads();
    if (threadIdx.x < C) {
      float sum = 0.0f;
      for (int k2 = 0; k2 < row_lanes; ++k2) sum += red[k2 * 4 + threadIdx.x];
      partials[(int64_t)blockIdx.x * (K * C + C) + K * C + threadIdx.x] = sum;
    }
  } else if (db && cg == 0 && rl < row_lanes) {
    for (int c = 0; c < C; ++c) unsafeAtomicAdd(db + c, sb[c]);
  }
}

// out[i] += sum_b partials[b * n + i]: blockIdx.y splits the workgroup partials 16 ways.
__global__ void small_head_reduce_kernel(int nblk, int n, int n_dw, const float* __restrict__ partials, float* dW,
                                         float* db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float sum = 0.0f;
  for (int b = b0; b < b1; ++b) sum += partials[(int64_t)b * n + i];
  if (i < n_dw) {
    if (dW) unsafeAtomicAdd(dW + i, sum);
  } else if (db) {
    unsafeAtomicAdd(db + (i - n_dw), sum);
  }
}

extern "C" int mnr_small_head_bwd(int64_t M, int K, int C, const uint16_t* H, int ldh, const float* g,
                                  const float* W, uint16_t* dX, int lddx, int apply_relu_mask, float* dW,
                                  float* db, const uint8_t* mask_bits, int ld_bits, int64_t bits_row_mod,
                                  float* scratch, int64_t scratch_floats, void* stream) {
  MNR_CHECK_ARG(M > 0 && K > 0 && C >= 1 && C <= 4 && H && g && W, "mnr_small_head_bwd: bad arguments (1 <= C <= 4)");
  MNR_CHECK_ARG(K % 8 == 0 && K / 8 <= 256 && ldh % 8 == 0 && (!dX || lddx % 8 == 0),
                "mnr_small_head_bwd: K must be a multiple of 8 (<= 2048) and the pitches multiples of 8");
  // Every workgroup ends with K*C fp32 atomics onto the same few cache lines of dW: more than ~512 workgroups
  // and the kernel is bound by that serialisation (2048 workgroups: 508 us, 512: 291 us at M = 2^20, K = 256);
  // fewer than ~256 and it loses memory parallelism.  tools/head_probe.py.
  int rows_auto = (int)(((M + 511) / 512 + 63) / 64 * 64);
  if (rows_auto < 512) rows_auto = 512;
  int rows_per_block = rows_auto;
  // With scratch for per-workgroup partials (reduced by small_head_reduce_kernel) instead of K*C atomics per
  // workgroup: 295 -> 253-267 us (K = 256, C = 1), 245 -> 98 us (K = 128, C = 3).
  float* partials = nullptr;
  if (scratch && dW) {
    constexpr int blocks_env = 512;                // tools/head_probe.py: 256: 347 us, 512: 253-267, 1024: 291, 2048: 313 (M = 2^20, K = 256)
    int rows_p = (int)(((M + blocks_env - 1) / blocks_env + 63) / 64 * 64);
    if (rows_p < 256) rows_p = 256;
    if ((int64_t)mnr_cdiv(M, rows_p) * (K * C + C) <= scratch_floats) {
      rows_per_block = rows_p;
      partials = scratch;
    }
  }
  const int grid = mnr_cdiv(M, rows_per_block);
  hipLaunchKernelGGL(small_head_bwd_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, K, C, (const bf16*)H, ldh, g, W,
                     (bf16*)dX, lddx, apply_relu_mask, dW, db, rows_per_block, mask_bits, ld_bits, bits_row_mod, partials);
  MNR_CHECK_LAUNCH();
  if (partials) {
    const int n = K * C + C;
    hipLaunchKernelGGL(small_head_reduce_kernel, dim3(mnr_cdiv(n, 128), 16), dim3(128), 0, (hipStream_t)stream, grid, n,
                       K * C, partials, dW, db);
    MNR_CHECK_LAUNCH();
  }
  return MNR_OK;
}
