// Fused Dense chain for the proposal MLP (reference internal/models.py:441-465 with net_depth <= skip_layer, i.e. no
// skip concat, and the Dense(1) density head of :460): ONE persistent kernel per sampling level instead of one GEMM
// launch per layer.
//
//   forward : x_0 = relu(feat W_0 + b_0), x_i = relu(x_{i-1} W_i + b_i), raw = x_last . w_head + b_head
//   backward: dY_last = mask_last * (g (x) w_head), dY_{i-1} = mask_{i-1} * (dY_i W_i^T)        (the dX chain; the weight
//             gradients dW_i = x_{i-1}^T dY_i stay with gemm_tn_kernel, which reads the x_i / dY_i this kernel leaves)
//
// A workgroup (512 threads = 8 waves, one per CU, persistent over the 256-row tiles of the level) keeps the tile's
// activation [256, W] in LDS for the whole chain: layer i reads it as the MFMA B-operand, accumulates [256, W] fp32 in
// registers (every wave: 32 output columns x 256/RG rows), and after a barrier overwrites it with layer i's bf16 output.
// The weights of a layer are private to a wave (its 32 columns x K) and go global -> registers once per tile and layer
// (16 B per lane and k-step, L2-resident: <= 128 KiB per layer); they never touch LDS.  Only layer 0 of the forward
// pass streams from HBM: its [256, K0] feature tile and its weights come in 64-wide K steps through a two-stage LDS-DMA
// pipeline that lives in the (still empty) activation buffer.
//
// HBM traffic per level and direction is what training has to keep for dW anyway: the bf16 activations / gradients
// (written once, straight from the LDS tile in full 2W-byte rows, 1-bit ReLU masks alongside); nothing is read back
// between layers.  Inference (acts = bits = NULL) writes only the head output.  The rows leave with NONTEMPORAL stores
// (global_store_dwordx4 ... nt): nothing on the chip reads them again before they have left every cache, and without the
// hint they push the lines the next tile's feature stream and the following launches want out of the L2 / Infinity Cache:
// forward chain -2 to -4 %, backward chain -5 %, blender_256 / llff_raw steps -2 to -3 % (profiles/r5_ab.md (i)).
//
// LDS tile format (shared with gemm.hip's NT kernel): a [256 rows][W] bf16 activation is W/64 K-tiles of
// [256][64] = 32 KiB, row pitch 128 B, 16-byte slot s of row r stored at slot position s ^ ((r >> 1) & 7).
#include <stdio.h>

#include <type_traits>

#include "common.h"

#define FM_ROWS 256
#define FM_KT_BYTES 32768                     // one [256][64] bf16 K-tile

// Address arithmetic derived from fm_opaque(tid) cannot be hoisted out of the enclosing loops: hipcc otherwise keeps the
// lane-dependent addresses of EVERY phase (stage, fragments, epilogue, copy-out) live across the MFMA loops, runs out of
// its 256 registers and spills them; a reload is a vector-memory operation, and the wait for it (vmcnt(0)) also waits
// for every LDS-DMA / store issued before it.
__device__ __forceinline__ int fm_opaque(int v) {
  MNR_GPU_ONLY(asm volatile("" : "+v"(v)));
  return v;
}

__device__ __forceinline__ int fm_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// LDS-DMA of a [ROWS][64] bf16 tile (rows row0.., columns k0.. of the row-major matrix g) into lds_tile; the swizzle is
// applied to the SOURCE address, the DMA image itself is lane-linear (see gemm.hip nt_stage_tile).
template <int ROWS>
__device__ __forceinline__ void fm_stage_tile(const bf16* __restrict__ tile_base, int ld, char* lds_tile, int wave, int lane) {
  // tile_base (wave-uniform) = &g[row0][k0]; per lane only a 32-bit element offset
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 512; ++i) {
    const int cbase = (i * 8 + wave) * 64;
    const int c = cbase + lane;
    const int r = c >> 3;
    const int slot = (c & 7) ^ ((r >> 1) & 7);
    const bf16* src = tile_base + (unsigned)(r * ld + slot * 8);
    __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(src), MNR_LDS_PTR(lds_tile + cbase * 16), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 fm_read_frag(const char* kt_tile, int row, int kslot) {
  return *(const bf16x8*)(kt_tile + fm_off(row, kslot));
}

// Profiling hook (tools/chain_probe.py --timeline): when set, thread 0 of every workgroup stamps s_memtime into
// g_fm_timeline[32 * blockIdx.x + slot] during its SECOND tile (steady state): slot 0 tile start, 1 layer-0 stream done,
// 2 + 3*li: layer li's MFMAs done, 3 + 3*li: its epilogue done (tile in LDS), 4 + 3*li: its copy-out issued;
// s_memrealtime (100 MHz) in slots 30 / 31 at tile start / end.
__device__ unsigned long long* g_fm_timeline = nullptr;

extern "C" int mnr_debug_chain_timeline(unsigned long long* device_buffer) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_fm_timeline), &device_buffer, sizeof(device_buffer));
  if (e != hipSuccess) {
    mnr_set_error("mnr_debug_chain_timeline: %s", hipGetErrorString(e));
    return MNR_ERR_HIP;
  }
  return MNR_OK;
}

#define FM_STAMP(slot)                                                                                  \
  do {                                                                                                  \
    if (tl_on) g_fm_timeline[32 * (int64_t)blockIdx.x + (slot)] = __builtin_amdgcn_s_memtime();         \
  } while (0)

template <int W>
struct FmCfg {
  static constexpr int NW = W / 32;                    // waves along the output columns
  static constexpr int RG = 8 / NW;                    // row groups
  static constexpr int RB = 8 / RG;                    // 32-row blocks per wave
  static constexpr int NKT = W / 64;                   // K-tiles of the resident activation
  static constexpr int X_BYTES = NKT * FM_KT_BYTES;
  static constexpr int STAGE_BYTES = FM_KT_BYTES + W * 128;       // layer 0: feature tile + weight tile [W][64]
  static constexpr int LDS_MAIN = X_BYTES > 2 * STAGE_BYTES ? X_BYTES : 2 * STAGE_BYTES;
  static constexpr int BIAS_OFF = LDS_MAIN;                        // [MNR_CHAIN_MAX_DEPTH][W] fp32 bias rows
  // skip-concat layer (models.py:458-459): its feature segment streams through two [256 rows][16 k] stages = one MFMA
  // k-step each (the activation tile occupies the main buffer at that point)
  static constexpr int SKIP_STAGE_BYTES = FM_ROWS * 32;
  static constexpr int SKIP_OFF = BIAS_OFF + MNR_CHAIN_MAX_DEPTH * W * 4;
  static constexpr int LDS_BYTES = SKIP_OFF + 2 * SKIP_STAGE_BYTES;
  static constexpr int CPR = W / 8;                    // 16-byte chunks per activation row
  static constexpr int COPY_ITERS = FM_ROWS * CPR / 512;
  static constexpr int ROW_STEP = 512 / CPR;
  static_assert(W == 128 || W == 256, "fused chain: width 128 or 256");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only (all of a wave's LDS operations retired, then s_barrier).  __syncthreads()
// also waits for vmcnt(0), i.e. for every global store in flight: with the copy-out stores deferred into the MFMA pass that
// would put their drain time back in front of the epilogue.  Nothing inside these kernels reads what they store to global
// memory, so the stores may stay in flight across the barrier.
__device__ __forceinline__ void fm_lds_barrier() {
  MNR_GPU_ONLY(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// One layer whose input sits in LDS: acc[rb] = X[rows of this wave] * Bt[cols of this wave]^T.
// The wave's weight fragments come global -> registers in chunks of four k-steps (16 registers), the next chunk in
// flight while this one multiplies; chunk 0 arrives preloaded (`w0`, issued before the previous layer's epilogue).
// The activation fragments are double-buffered by hand, four row blocks (16 registers) at a time: left to itself hipcc
// (at ~250 live registers) reads one fragment, waits, multiplies, reads the next.
#define FM_WCHUNK 4
__device__ __forceinline__ void fm_load_wchunk(const bf16* __restrict__ Bt, int ldb, int cw, int frow, int khalf, int chunk,
                                               bf16x8 (&w)[FM_WCHUNK]) {
  // (wave-uniform row block + 32-bit lane offset)
  const bf16* wsrc = Bt + (int64_t)(cw * 32) * ldb + chunk * (FM_WCHUNK * 16) + (unsigned)(frow * ldb + khalf * 8);
#pragma unroll
  for (int j = 0; j < FM_WCHUNK; ++j) w[j] = *(const bf16x8*)(wsrc + j * 16);
}

// DEFER (bit 0: activation rows, bit 1: mask bits): the copy-out of the tile this pass READS (the previous layer's output,
// intact in LDS for the whole pass) is issued from inside the pass, in COPY_ITERS / 4 batches placed AFTER the layer's last
// weight-chunk request.  vmcnt retires in order on gfx950, loads and stores alike: with the copy-out in front of the pass
// (its place in round 2) the wait for weight chunk 1 also waited for every store of the copy-out, 8-10k cycles of every
// layer (tools/chain_probe.py --timeline: MFMA phase 14.7k cycles in training against 5.9k in inference); behind the last
// chunk request nothing in the pass waits on vmcnt any more and the stores drain under the epilogue.
template <int W, int DEFER>
__device__ __forceinline__ void fm_layer_mfma(const char* X, const bf16* __restrict__ Bt, int ldb, int cw, int rg, int lane_,
                                              const bf16x8 (&w0)[FM_WCHUNK], f32x16 (&acc)[FmCfg<W>::RB], int tid_ = 0,
                                              int64_t m0 = 0, bf16* __restrict__ dst = nullptr, uint8_t* __restrict__ bits = nullptr) {
  typedef FmCfg<W> C;
  const int lane = fm_opaque(lane_), frow = lane & 31, khalf = lane >> 5;
  constexpr int NKS = W / 16;
  constexpr int NCH = NKS / FM_WCHUNK;
  constexpr int HB = C::RB > 4 ? 4 : C::RB;            // row blocks per fragment batch
  constexpr int NH = C::RB / HB;
  constexpr int STEPS = NKS * NH;
  // deferred copy-out: batches of UB chunks, ds_reads at step CP_FIRST + b * CP_STRIDE, the stores one step later
  constexpr int UB = 4;
  constexpr int NB = C::COPY_ITERS / UB;
  constexpr int LAST_LOAD = NCH >= 2 ? ((NCH - 2) * FM_WCHUNK + 1) * NH : 0;
  constexpr int CP_FIRST = LAST_LOAD + 1;
  constexpr int CP_STRIDE = (STEPS - LAST_LOAD - 2) / NB;
  static_assert(DEFER == 0 || (C::COPY_ITERS % UB == 0 && CP_STRIDE >= 2 && CP_FIRST + (NB - 1) * CP_STRIDE + 1 < STEPS),
                "deferred copy-out does not fit behind the last weight-chunk request");
#pragma unroll
  for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
  // lane part of the fragment address: row = (rg*RB + rb)*32 + frow, slot = (ks&3)*2 + khalf; the swizzle term
  // (row >> 1) & 7 = (frow >> 1) & 7 does not depend on the row block
  const int sw = (frow >> 1) & 7;
  const char* xrow = X + (rg * C::RB * 32 + frow) * 128;
  auto read_batch = [&](int step, bf16x8 (&fa)[HB]) {
    const int ks = step / NH, h = step % NH;
    const char* base = xrow + (ks >> 2) * FM_KT_BYTES + h * HB * 4096 + ((((ks & 3) * 2 + khalf) ^ sw) << 4);
#pragma unroll
    for (int i = 0; i < HB; ++i) fa[i] = *(const bf16x8*)(base + i * 4096);
  };
  bf16x8 wq[2][FM_WCHUNK];
#pragma unroll
  for (int j = 0; j < FM_WCHUNK; ++j) wq[0][j] = w0[j];
  bf16x8 fa[2][HB];
  u32x4 cpw[UB];                                        // one copy-out batch in flight (DEFER)
  read_batch(0, fa[0]);
#pragma unroll
  for (int step = 0; step < STEPS; ++step) {
    const int ks = step / NH, h = step % NH;
    const int c = ks / FM_WCHUNK, j = ks % FM_WCHUNK;
    if (j == 1 && h == 0 && c + 1 < NCH) {
      // issue the next chunk's four loads HERE, one k-step into this chunk (at the chunk's first step hipcc would wait
      // vmcnt(0), i.e. for these loads too, before the first MFMA; they have three k-steps to land) (left to the scheduler they sink to their first use, one L2 round trip
      // per k-step in front of its MFMAs)
      __builtin_amdgcn_sched_barrier(0);
      fm_load_wchunk(Bt, ldb, cw, frow, khalf, c + 1, wq[(c + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (DEFER != 0) {
      const int rel = step - CP_FIRST;
      if (rel >= 0 && rel % CP_STRIDE == 0 && rel / CP_STRIDE < NB) {
        // batch b: UB 16-byte chunks of this thread's rows, LDS -> registers
        const int b = rel / CP_STRIDE;
        const int tid = fm_opaque(tid_);
        const int row0 = tid / C::CPR, ch = tid % C::CPR;
        const char* lptr = X + (ch >> 3) * FM_KT_BYTES + row0 * 128 + (((ch & 7) ^ ((row0 >> 1) & 7)) << 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) cpw[u] = *(const u32x4*)(lptr + (b * UB + u) * C::ROW_STEP * 128);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (rel >= 1 && (rel - 1) % CP_STRIDE == 0 && (rel - 1) / CP_STRIDE < NB) {
        const int b = (rel - 1) / CP_STRIDE;
        const int tid = fm_opaque(tid_);
        const int row0 = tid / C::CPR, ch = tid % C::CPR;
        const int64_t r = m0 + row0 + (int64_t)b * UB * C::ROW_STEP;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          if constexpr ((DEFER & 2) != 0) {
            unsigned mb = mnr_relu_mask_byte(cpw[u][0], cpw[u][1], cpw[u][2], cpw[u][3]);
            mb |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xF5, 0xf, 0xf, false) << 8;     // quad_perm [1,1,3,3]
            mb |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xAA, 0xf, 0xf, false) << 16;    // quad_perm [2,2,2,2]
            if ((ch & 3) == 0) *(unsigned*)(bits + (r + (int64_t)u * C::ROW_STEP) * (W / 8) + ch) = mb;
          }
          if constexpr ((DEFER & 1) != 0) __builtin_nontemporal_store(cpw[u], (u32x4*)(dst + (r + (int64_t)u * C::ROW_STEP) * W + ch * 8));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (step + 1 < STEPS) {
      read_batch(step + 1, fa[(step + 1) & 1]);
      __builtin_amdgcn_sched_group_barrier(0x100, HB, 0);          // this step's (next-batch) ds_reads first ...
    }
#pragma unroll
    for (int i = 0; i < HB; ++i)
      acc[h * HB + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[c & 1][j], fa[step & 1][i], acc[h * HB + i], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, HB, 0);            // ... then its MFMAs
  }
}

// (kernel-uniform dispatch on which outputs of the previous layer are deferred into this pass)
template <int W>
__device__ __forceinline__ void fm_layer_mfma_defer(const char* X, const bf16* __restrict__ Bt, int ldb, int cw, int rg, int lane,
                                                    const bf16x8 (&w0)[FM_WCHUNK], f32x16 (&acc)[FmCfg<W>::RB], int tid, int64_t m0,
                                                    bf16* dst, uint8_t* bits) {
  if (dst && bits) fm_layer_mfma<W, 3>(X, Bt, ldb, cw, rg, lane, w0, acc, tid, m0, dst, bits);
  else if (dst) fm_layer_mfma<W, 1>(X, Bt, ldb, cw, rg, lane, w0, acc, tid, m0, dst, bits);
  else if (bits) fm_layer_mfma<W, 2>(X, Bt, ldb, cw, rg, lane, w0, acc, tid, m0, dst, bits);
  else fm_layer_mfma<W, 0>(X, Bt, ldb, cw, rg, lane, w0, acc);
}

// The feature segment of a skip-concat layer: acc[rb] += feat[rows of this wave, 0..K0) * Bt[cols of this wave, kcol0..kcol0+K0)^T.
// The [256][K0] feature tile streams HBM -> LDS one MFMA k-step ([256 rows][16 k] = 8 KiB = one 16-byte LDS-DMA piece per
// thread) at a time through two stages; the wave's weight fragment of a k-step comes global -> registers one step ahead.
// Stage image: row r at byte r * 32, its two 16-byte k-halves swapped when (r >> 3) & 1 (rows 8 apart would otherwise
// share banks in a ds_read_b128 of 16 consecutive rows); the swizzle is applied to the SOURCE address, the DMA image is
// lane-linear.
template <int W>
__device__ __forceinline__ void fm_skip_segment(char* stages, const bf16* __restrict__ feat_tile, int ld_feat, int K0,
                                                const bf16* __restrict__ Bt, int ldb, int kcol0, int cw, int rg, int wave,
                                                int lane_, f32x16 (&acc)[FmCfg<W>::RB]) {
  typedef FmCfg<W> C;
  static_assert(FM_ROWS * 2 == 512, "one 16-byte piece per thread and k-step");
  const int lane = fm_opaque(lane_), frow = lane & 31, khalf = lane >> 5;
  const int nks = K0 / 16;
  const int c = wave * 64 + lane;                        // piece index: row c >> 1, position c & 1
  const int prow = c >> 1;
  const unsigned src_off = (unsigned)(prow * ld_feat + (((c & 1) ^ ((prow >> 3) & 1)) << 3));
  auto stage = [&](int ks) {
    __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(feat_tile + ks * 16 + src_off),
                                     MNR_LDS_PTR(stages + (ks & 1) * C::SKIP_STAGE_BYTES + wave * 1024), 16, 0, 0);
  };
  const bf16* wsrc = Bt + (int64_t)(cw * 32) * ldb + kcol0 + (unsigned)(frow * ldb + khalf * 8);
  bf16x8 wq[2];
  stage(0);
  wq[0] = *(const bf16x8*)wsrc;
  for (int ks = 0; ks < nks; ++ks) {
    MNR_GPU_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"));
    MNR_SIM_HOOK(hipsim::wait_vmcnt(0));
    __builtin_amdgcn_s_barrier();                        // k-step ks has landed everywhere; the other stage is free
    asm volatile("" ::: "memory");
    if (ks + 1 < nks) {
      stage(ks + 1);
      wq[(ks + 1) & 1] = *(const bf16x8*)(wsrc + (ks + 1) * 16);
    }
    const char* sb = stages + (ks & 1) * C::SKIP_STAGE_BYTES;
    bf16x8 fa[C::RB];
#pragma unroll
    for (int rb = 0; rb < C::RB; ++rb) {
      const int row = (rg * C::RB + rb) * 32 + frow;
      fa[rb] = *(const bf16x8*)(sb + row * 32 + ((khalf ^ ((row >> 3) & 1)) << 4));
    }
#pragma unroll
    for (int rb = 0; rb < C::RB; ++rb)
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks & 1], fa[rb], acc[rb], 0, 0, 0);
  }
}

// acc -> bf16 -> the LDS activation tile.  Forward: + bias, ReLU.  Backward: ReLU mask bits of the layer below.
// acc[rb][r]: column n = cw*32 + (r&3) + 8*(r>>2) + 4*khalf, row m = (rg*RB + rb)*32 + frow.
template <int W, bool BWD>
__device__ __forceinline__ void fm_epilogue(char* X, int cw, int rg, int lane_, f32x16 (&acc)[FmCfg<W>::RB],
                                            const float* __restrict__ bias, const unsigned (&mbits)[FmCfg<W>::RB]) {
  typedef FmCfg<W> C;
  const int lane = fm_opaque(lane_), frow = lane & 31, khalf = lane >> 5;
  float bias_r[16];
  if constexpr (!BWD) {                                  // `bias`: this layer's row in LDS (parked there once per kernel)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 b4 = *(const f32x4*)(bias + cw * 32 + rq * 8 + khalf * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) bias_r[rq * 4 + e] = b4[e];
    }
  }
  const int sw = (frow >> 1) & 7;
#pragma unroll
  for (int rb = 0; rb < C::RB; ++rb) {
    const int ml = (rg * C::RB + rb) * 32 + frow;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int nl = cw * 32 + rq * 8 + khalf * 4;
      f32x2 s0 = {acc[rb][rq * 4 + 0], acc[rb][rq * 4 + 1]};
      f32x2 s1 = {acc[rb][rq * 4 + 2], acc[rb][rq * 4 + 3]};
      if constexpr (!BWD) {
        s0 += f32x2{bias_r[rq * 4 + 0], bias_r[rq * 4 + 1]};
        s1 += f32x2{bias_r[rq * 4 + 2], bias_r[rq * 4 + 3]};
      }
      s16x2 h0 = __builtin_bit_cast(s16x2, __builtin_convertvector(s0, bf16x2));
      s16x2 h1 = __builtin_bit_cast(s16x2, __builtin_convertvector(s1, bf16x2));
      unsigned d0 = __builtin_bit_cast(unsigned, h0), d1 = __builtin_bit_cast(unsigned, h1);
      if constexpr (!BWD) {
        // ReLU on the bf16 bit patterns: packed signed 16-bit max with 0 (rounding commutes with the clamp)
        const s16x2 z = {0, 0};
        d0 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(h0, z));
        d1 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(h1, z));
      } else {
        const int mb = (int)(mbits[rb] >> (rq * 8 + khalf * 4));     // bits 0..3: the four columns of this group
        const unsigned b0 = (unsigned)__builtin_amdgcn_sbfe(mb, 0, 1), b1 = (unsigned)__builtin_amdgcn_sbfe(mb, 1, 1);
        const unsigned b2 = (unsigned)__builtin_amdgcn_sbfe(mb, 2, 1), b3 = (unsigned)__builtin_amdgcn_sbfe(mb, 3, 1);
        d0 &= (b0 & 0xffffu) | (b1 & 0xffff0000u);
        d1 &= (b2 & 0xffffu) | (b3 & 0xffff0000u);
      }
      const i32x2 pk = {(int)d0, (int)d1};
      *(i32x2*)(X + (nl >> 6) * FM_KT_BYTES + ml * 128 + ((((nl & 63) >> 3) ^ sw) << 4) + (nl & 7) * 2) = pk;
    }
  }
}

// The LDS activation tile -> global memory in full rows (16 B per lane), the 1-bit "> 0" masks, and (forward, last
// layer) the Dense(1) head: 8-element partial dot products reduced over the W/8 lanes that share a row.
// A rolled loop over batches of four rows with running pointers: fully unrolled, hipcc hoists the 16 x 2 addresses out
// of the tile loop, spills them, and every reload (a vmcnt operation) then waits for the stores before it.
template <int W, bool DST, bool BITS, bool HEAD>
__device__ __forceinline__ void fm_copy_out(const char* X, int tid_, int64_t m0, bf16* __restrict__ dst, uint8_t* __restrict__ bits,
                                            const bf16* __restrict__ w_head, float b_head, float* __restrict__ head_out) {
  typedef FmCfg<W> C;
  constexpr int UB = 4;
  static_assert(C::COPY_ITERS % UB == 0, "copy-out batch");
  const int tid = fm_opaque(tid_);
  const int row0 = tid / C::CPR, ch = tid % C::CPR;
  // (row >> 1) & 7 is the same for every row this thread touches (they are ROW_STEP = 16 or 32 apart)
  const char* lptr = X + (ch >> 3) * FM_KT_BYTES + row0 * 128 + (((ch & 7) ^ ((row0 >> 1) & 7)) << 4);
  int64_t e0 = (m0 + row0) * (int64_t)W + ch * 8;                  // element offset of this thread's first chunk
  int64_t bo = (m0 + row0) * (int64_t)(W / 8) + ch;
  int64_t ro = m0 + row0;
  float whead[8];
  if constexpr (HEAD) {
    const bf16x8 wh = *(const bf16x8*)(w_head + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) whead[e] = (float)wh[e];
  }
#pragma unroll 1
  for (int it0 = 0; it0 < C::COPY_ITERS; it0 += UB) {
    u32x4 w[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) w[u] = *(const u32x4*)(lptr + (it0 + u) * C::ROW_STEP * 128);
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if constexpr (BITS) {
        unsigned mb = mnr_relu_mask_byte(w[u][0], w[u][1], w[u][2], w[u][3]);
        mb |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xF5, 0xf, 0xf, false) << 8;     // quad_perm [1,1,3,3]
        mb |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xAA, 0xf, 0xf, false) << 16;    // quad_perm [2,2,2,2]
        if ((ch & 3) == 0) *(unsigned*)(bits + bo + (int64_t)u * C::ROW_STEP * (W / 8)) = mb;
      }
      if constexpr (DST) __builtin_nontemporal_store(w[u], (u32x4*)(dst + e0 + (int64_t)u * C::ROW_STEP * W));
      if constexpr (HEAD) {
        const bf16x8 v = __builtin_bit_cast(bf16x8, w[u]);
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[e] * whead[e];
#pragma unroll
        for (int d = C::CPR / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        if (ch == 0) head_out[ro + u * C::ROW_STEP] = s + b_head;
      }
    }
    e0 += (int64_t)UB * C::ROW_STEP * W;
    bo += (int64_t)UB * C::ROW_STEP * (W / 8);
    ro += UB * C::ROW_STEP;
  }
}

template <int W>
__device__ __forceinline__ void fm_copy_out_dispatch(const char* X, int tid, int64_t m0, bf16* dst, uint8_t* bits,
                                                     const bf16* w_head, float b_head, float* head_out) {
  // (all flags are kernel-uniform)
  if (w_head) {
    if (dst && bits) fm_copy_out<W, true, true, true>(X, tid, m0, dst, bits, w_head, b_head, head_out);
    else if (dst) fm_copy_out<W, true, false, true>(X, tid, m0, dst, bits, w_head, b_head, head_out);
    else if (bits) fm_copy_out<W, false, true, true>(X, tid, m0, dst, bits, w_head, b_head, head_out);
    else fm_copy_out<W, false, false, true>(X, tid, m0, dst, bits, w_head, b_head, head_out);
  } else {
    if (dst && bits) fm_copy_out<W, true, true, false>(X, tid, m0, dst, bits, w_head, b_head, head_out);
    else if (dst) fm_copy_out<W, true, false, false>(X, tid, m0, dst, bits, w_head, b_head, head_out);
    else if (bits) fm_copy_out<W, false, true, false>(X, tid, m0, dst, bits, w_head, b_head, head_out);
  }
}

template <int W>
__global__ __launch_bounds__(512) void mlp_chain_fwd_kernel(mnr_mlp_chain_fwd_args p, int defer) {
  typedef FmCfg<W> C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave % C::NW, rg = wave / C::NW;
  const int64_t tiles = p.M / FM_ROWS;
  const bf16* feat = (const bf16*)p.feat;
  const int nk0 = p.K0 / 64;
  const unsigned no_bits[C::RB] = {};
  const float b_head = (p.w_head && p.b_head) ? p.b_head[0] : 0.0f;
  // bias rows -> LDS, once (an epilogue then reads its 16 values with four ds_read_b128 instead of waiting on L2)
  for (int i = tid; i < p.depth * W; i += 512) ((float*)(smem + C::BIAS_OFF))[i] = p.bias[i / W][i % W];

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * FM_ROWS;
    const bool tl_on = g_fm_timeline != nullptr && tid == 0 && tile == (int64_t)blockIdx.x + gridDim.x;
    FM_STAMP(0);
    if (tl_on) g_fm_timeline[32 * (int64_t)blockIdx.x + 30] = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[C::RB];
#pragma unroll
    for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;

    // ---- layer 0: K0 streamed through two LDS stages (feature tile [256][64] + weight tile [W][64] per step)
    __syncthreads();                                    // the previous tile's copy-out is done reading the buffer
    {
      const bf16* Bt0 = (const bf16*)p.Bt[0];
      const bf16* feat_tile = feat + m0 * (int64_t)p.ld_feat;
      auto stage = [&](int kt) {
        char* base = smem + (kt & 1) * C::STAGE_BYTES;
        const int ln = fm_opaque(lane);
        fm_stage_tile<FM_ROWS>(feat_tile + kt * 64, p.ld_feat, base, wave, ln);
        fm_stage_tile<W>(Bt0 + kt * 64, p.ldb[0], base + FM_KT_BYTES, wave, ln);
      };
      constexpr int HB = C::RB > 4 ? 4 : C::RB;
      constexpr int NH = C::RB / HB;
      const int ln0 = fm_opaque(lane), frow = ln0 & 31, khalf = ln0 >> 5;
      const int sw = (frow >> 1) & 7;
      stage(0);
      for (int kt = 0; kt < nk0; ++kt) {
        MNR_GPU_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"));
        MNR_SIM_HOOK(hipsim::wait_vmcnt(0));
        __builtin_amdgcn_s_barrier();                   // step kt has landed everywhere; step kt-1's buffer is free
        asm volatile("" ::: "memory");
        if (kt + 1 < nk0) stage(kt + 1);
        const char* As = smem + (kt & 1) * C::STAGE_BYTES + (rg * C::RB * 32 + frow) * 128;
        const char* Ws = smem + (kt & 1) * C::STAGE_BYTES + FM_KT_BYTES + (cw * 32 + frow) * 128;
        bf16x8 wf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[ks] = *(const bf16x8*)(Ws + (((ks * 2 + khalf) ^ sw) << 4));
        bf16x8 fa[2][HB];
        auto read_batch = [&](int step, bf16x8 (&f)[HB]) {
          const int ks = step / NH, h = step % NH;
          const char* base = As + h * HB * 4096 + (((ks * 2 + khalf) ^ sw) << 4);
#pragma unroll
          for (int i = 0; i < HB; ++i) f[i] = *(const bf16x8*)(base + i * 4096);
        };
        read_batch(0, fa[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + HB, 0);
#pragma unroll
        for (int step = 0; step < 4 * NH; ++step) {
          const int ks = step / NH, h = step % NH;
          if (step + 1 < 4 * NH) {
            read_batch(step + 1, fa[(step + 1) & 1]);
            __builtin_amdgcn_sched_group_barrier(0x100, HB, 0);
          }
#pragma unroll
          for (int i = 0; i < HB; ++i)
            acc[h * HB + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], fa[step & 1][i], acc[h * HB + i], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, HB, 0);
        }
      }
    }

    FM_STAMP(1);
    bf16x8 w0[FM_WCHUNK];
    if (p.depth > 1) {
      const int ln = fm_opaque(lane);
      fm_load_wchunk((const bf16*)p.Bt[1], p.ldb[1], cw, ln & 31, ln >> 5, 0, w0);
    }
    for (int li = 0; li < p.depth; ++li) {
      if (li > 0) {
        // (defer: the previous layer's copy-out rides inside this pass instead of standing in front of it)
        if (defer) fm_layer_mfma_defer<W>(smem, (const bf16*)p.Bt[li], p.ldb[li], cw, rg, lane, w0, acc, tid, m0, (bf16*)p.acts[li - 1], p.bits[li - 1]);
        else fm_layer_mfma<W, 0>(smem, (const bf16*)p.Bt[li], p.ldb[li], cw, rg, lane, w0, acc);
        if (li == p.skip_layer)                             // input = [x_{li-1} | features] (models.py:458-459)
          fm_skip_segment<W>(smem + C::SKIP_OFF, feat + m0 * (int64_t)p.ld_feat, p.ld_feat, p.K0, (const bf16*)p.Bt[li],
                             p.ldb[li], W, cw, rg, wave, lane, acc);
        if (li + 1 < p.depth) {
          const int ln = fm_opaque(lane);
          fm_load_wchunk((const bf16*)p.Bt[li + 1], p.ldb[li + 1], cw, ln & 31, ln >> 5, 0, w0);
        }
      }
      FM_STAMP(2 + 3 * li);
      fm_lds_barrier();                                 // every wave is done reading this layer's input
      fm_epilogue<W, false>(smem, cw, rg, lane, acc, (const float*)(smem + C::BIAS_OFF) + li * W, no_bits);
      fm_lds_barrier();                                 // the layer's output is complete in LDS
      FM_STAMP(3 + 3 * li);
      const bool last = li == p.depth - 1;
      if (last || !defer)
        fm_copy_out_dispatch<W>(smem, tid, m0, (bf16*)p.acts[li], p.bits[li], last ? (const bf16*)p.w_head : nullptr, b_head,
                                p.head_out);
      FM_STAMP(4 + 3 * li);
    }
    if (tl_on) g_fm_timeline[32 * (int64_t)blockIdx.x + 31] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int W>
__global__ __launch_bounds__(512) void mlp_chain_bwd_kernel(mnr_mlp_chain_bwd_args p, int defer) {
  typedef FmCfg<W> C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave % C::NW, rg = wave / C::NW;
  const int64_t tiles = p.M / FM_ROWS;

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * FM_ROWS;
    const bool tl_on = g_fm_timeline != nullptr && tid == 0 && tile == (int64_t)blockIdx.x + gridDim.x;
    FM_STAMP(0);
    if (tl_on) g_fm_timeline[32 * (int64_t)blockIdx.x + 30] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();                                    // the previous tile's copy-out is done reading the buffer
    if (p.dY_in) {
      // the chain starts from a dY_last the caller computed (an MLP with heads: the merged head's dX GEMM wrote it)
      const bf16* src = (const bf16*)p.dY_in + m0 * (int64_t)W;
      const int ln = fm_opaque(lane);
#pragma unroll
      for (int kt = 0; kt < C::NKT; ++kt) fm_stage_tile<FM_ROWS>(src + kt * 64, W, smem + kt * FM_KT_BYTES, wave, ln);
      MNR_GPU_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"));
      MNR_SIM_HOOK(hipsim::wait_vmcnt(0));
    } else {
      // rank-1 start: dY_last = mask_last * (g (x) w_head), this thread's 8 columns of its COPY_ITERS rows
      const int last = p.depth - 1;
      const int t_ = fm_opaque(tid);
      const int row0 = t_ / C::CPR, ch = t_ % C::CPR;
      float wh[8];
      {
        const f32x4 a = *(const f32x4*)(p.w_head + ch * 8), b = *(const f32x4*)(p.w_head + ch * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) wh[e] = a[e], wh[4 + e] = b[e];
      }
      char* lptr = smem + (ch >> 3) * FM_KT_BYTES + row0 * 128 + (((ch & 7) ^ ((row0 >> 1) & 7)) << 4);
      const int64_t ro = m0 + row0;
      const uint8_t* bl = p.bits[last] + ro * (W / 8) + (ch & ~3);          // the dword holding this chunk's byte
      const float* gp = p.g_head + ro;
      bf16* dst = p.dY[last] ? (bf16*)p.dY[last] + ro * W + ch * 8 : nullptr;
      // every row's head gradient and mask word requested up front (nothing else is live here: 2 x COPY_ITERS registers),
      // one HBM round trip for the whole start instead of one per batch of rows
      float g[C::COPY_ITERS];
      unsigned mb[C::COPY_ITERS];
#pragma unroll
      for (int u = 0; u < C::COPY_ITERS; ++u) {
        g[u] = gp[u * C::ROW_STEP];
        mb[u] = *(const unsigned*)(bl + (int64_t)u * C::ROW_STEP * (W / 8)) >> ((ch & 3) * 8);
      }
#pragma unroll
      for (int u = 0; u < C::COPY_ITERS; ++u) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)(((mb[u] >> e) & 1u) ? g[u] * wh[e] : 0.0f);
        *(bf16x8*)(lptr + u * C::ROW_STEP * 128) = v;
        if (dst) *(bf16x8*)(dst + (int64_t)u * C::ROW_STEP * W) = v;
      }
    }
    bf16x8 w0[FM_WCHUNK];
    if (p.depth > 1) {
      const int ln = fm_opaque(lane);
      fm_load_wchunk((const bf16*)p.Bw[p.depth - 1], p.ldb[p.depth - 1], cw, ln & 31, ln >> 5, 0, w0);
    }
    __syncthreads();
    FM_STAMP(1);
    for (int li = p.depth - 1; li >= 1; --li) {
      // ReLU masks of layer li-1 for this lane's rows (one dword = the 32 columns of this wave), requested ahead of the
      // MFMAs that hide their latency
      unsigned mbits[C::RB];
      {
        const int frow = fm_opaque(lane) & 31;
        const uint8_t* bl = p.bits[li - 1] + (m0 + rg * C::RB * 32) * (int64_t)(W / 8) + cw * 4 + (unsigned)(frow * (W / 8));
#pragma unroll
        for (int rb = 0; rb < C::RB; ++rb) mbits[rb] = *(const unsigned*)(bl + (int64_t)rb * 32 * (W / 8));
      }
      f32x16 acc[C::RB];
      // (defer: dY_li, this pass's input, is copied out from inside the pass; the rank-1 start wrote dY_last itself)
      if (defer && li < p.depth - 1) fm_layer_mfma<W, 1>(smem, (const bf16*)p.Bw[li], p.ldb[li], cw, rg, lane, w0, acc, tid, m0, (bf16*)p.dY[li], nullptr);
      else fm_layer_mfma<W, 0>(smem, (const bf16*)p.Bw[li], p.ldb[li], cw, rg, lane, w0, acc);
      FM_STAMP(2 + 3 * li);
      if (li > 1) {
        const int ln = fm_opaque(lane);
        fm_load_wchunk((const bf16*)p.Bw[li - 1], p.ldb[li - 1], cw, ln & 31, ln >> 5, 0, w0);
      }
      fm_lds_barrier();
      fm_epilogue<W, true>(smem, cw, rg, lane, acc, nullptr, mbits);
      fm_lds_barrier();
      FM_STAMP(3 + 3 * li);
      if (!defer || li == 1) fm_copy_out<W, true, false, false>(smem, tid, m0, (bf16*)p.dY[li - 1], nullptr, nullptr, 0.0f, nullptr);
      FM_STAMP(4 + 3 * li);
    }
    if (tl_on) g_fm_timeline[32 * (int64_t)blockIdx.x + 31] = __builtin_amdgcn_s_memrealtime();
  }
}

// A/B switch: 1 (default) = a layer's copy-out rides inside the next layer's MFMA pass behind its last weight-chunk request
// (fm_layer_mfma DEFER); 0 = in front of the pass (round 2).  Bitwise equal results.
static int g_fm_defer = 1;
extern "C" int mnr_mlp_chain_set_deferred(int on) {
  g_fm_defer = on;
  return MNR_OK;
}

// Test hook (include/mnerf_debug.h): at most n persistent workgroups, so that small sizes walk several tiles per workgroup.
static int g_fm_max_wgs = 0;
extern "C" int mnr_mlp_chain_set_max_wgs(int n) {
  g_fm_max_wgs = n > 0 ? n : 0;
  return MNR_OK;
}

static int fm_grid(int64_t tiles) {
  int cus = mnr_cu_count();
  if (g_fm_max_wgs > 0 && g_fm_max_wgs < cus) cus = g_fm_max_wgs;
  return (int)(tiles < cus ? tiles : cus);
}

static int fm_check_common(const char* who, int64_t M, int W, int depth) {
  MNR_CHECK_ARG(M > 0 && M % FM_ROWS == 0, "%s: M=%lld must be a positive multiple of 256", who, (long long)M);
  MNR_CHECK_ARG(W == 128 || W == 256, "%s: width %d is not 128 or 256", who, W);
  MNR_CHECK_ARG(depth >= 1 && depth <= MNR_CHAIN_MAX_DEPTH, "%s: depth %d out of range", who, depth);
  return MNR_OK;
}

extern "C" int mnr_mlp_chain_fwd(const mnr_mlp_chain_fwd_args* a, void* stream) {
  MNR_CHECK_ARG(a != nullptr, "mnr_mlp_chain_fwd: null args");
  if (int s = fm_check_common("mnr_mlp_chain_fwd", a->M, a->W, a->depth)) return s;
  MNR_CHECK_ARG(a->feat && a->K0 > 0 && a->K0 % 64 == 0 && a->ld_feat % 8 == 0 && a->ld_feat >= a->K0 && a->ld_feat <= (1 << 20),
                "mnr_mlp_chain_fwd: feature matrix needs K0 %% 64 == 0 and ld_feat %% 8 == 0");
  for (int i = 0; i < a->depth; ++i) {
    MNR_CHECK_ARG(a->Bt[i] && a->bias[i] && a->ldb[i] % 8 == 0 && a->ldb[i] >= (i == 0 ? a->K0 : a->W),
                  "mnr_mlp_chain_fwd: layer %d operand", i);
    MNR_CHECK_ARG(((uintptr_t)a->bias[i] % 16) == 0, "mnr_mlp_chain_fwd: bias[%d] must be 16-byte aligned", i);
    MNR_CHECK_ARG(((uintptr_t)a->Bt[i] % 16) == 0 && (!a->acts[i] || ((uintptr_t)a->acts[i] % 16) == 0) &&
                      (!a->bits[i] || ((uintptr_t)a->bits[i] % 4) == 0),
                  "mnr_mlp_chain_fwd: layer %d pointers must be 16-byte (bits: 4-byte) aligned", i);
  }
  MNR_CHECK_ARG(((uintptr_t)a->feat % 16) == 0, "mnr_mlp_chain_fwd: feat must be 16-byte aligned");
  MNR_CHECK_ARG(!a->w_head || (a->head_out && ((uintptr_t)a->w_head % 16) == 0), "mnr_mlp_chain_fwd: head needs head_out");
  MNR_CHECK_ARG(a->skip_layer <= 0 || (a->skip_layer < a->depth && a->K0 % 16 == 0 && a->ldb[a->skip_layer] >= a->W + a->K0),
                "mnr_mlp_chain_fwd: skip_layer %d needs 0 < skip_layer < depth and an operand of W + K0 columns", a->skip_layer);
  const int grid = fm_grid(a->M / FM_ROWS);
  if (a->W == 256) {
    (void)hipFuncSetAttribute((const void*)mlp_chain_fwd_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, FmCfg<256>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_fwd_kernel<256>, dim3(grid), dim3(512), FmCfg<256>::LDS_BYTES, (hipStream_t)stream, *a, g_fm_defer);
  } else {
    (void)hipFuncSetAttribute((const void*)mlp_chain_fwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, FmCfg<128>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_fwd_kernel<128>, dim3(grid), dim3(512), FmCfg<128>::LDS_BYTES, (hipStream_t)stream, *a, g_fm_defer);
  }
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_mlp_chain_bwd(const mnr_mlp_chain_bwd_args* a, void* stream) {
  MNR_CHECK_ARG(a != nullptr, "mnr_mlp_chain_bwd: null args");
  if (int s = fm_check_common("mnr_mlp_chain_bwd", a->M, a->W, a->depth)) return s;
  MNR_CHECK_ARG(a->dY_in ? ((uintptr_t)a->dY_in % 16) == 0 : (a->g_head && a->w_head && ((uintptr_t)a->w_head % 16) == 0),
                "mnr_mlp_chain_bwd: needs dY_in (16-byte aligned) or a head gradient with a 16-byte-aligned head kernel");
  for (int i = 0; i < a->depth; ++i) {
    MNR_CHECK_ARG(a->bits[i] && ((uintptr_t)a->bits[i] % 4) == 0, "mnr_mlp_chain_bwd: layer %d needs its ReLU mask bits", i);
    MNR_CHECK_ARG(a->dY[i] || i == a->depth - 1, "mnr_mlp_chain_bwd: dY[%d] missing", i);
    MNR_CHECK_ARG(!(a->dY_in && i == a->depth - 1 && a->dY[i]), "mnr_mlp_chain_bwd: with dY_in the last dY is the input itself");
    MNR_CHECK_ARG(!a->dY[i] || ((uintptr_t)a->dY[i] % 16) == 0, "mnr_mlp_chain_bwd: dY[%d] must be 16-byte aligned", i);
    if (i >= 1)
      MNR_CHECK_ARG(a->Bw[i] && a->ldb[i] % 8 == 0 && a->ldb[i] >= a->W && ((uintptr_t)a->Bw[i] % 16) == 0,
                    "mnr_mlp_chain_bwd: layer %d operand", i);
  }
  const int grid = fm_grid(a->M / FM_ROWS);
  if (a->W == 256) {
    (void)hipFuncSetAttribute((const void*)mlp_chain_bwd_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, FmCfg<256>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_bwd_kernel<256>, dim3(grid), dim3(512), FmCfg<256>::LDS_BYTES, (hipStream_t)stream, *a, g_fm_defer);
  } else {
    (void)hipFuncSetAttribute((const void*)mlp_chain_bwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, FmCfg<128>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_bwd_kernel<128>, dim3(grid), dim3(512), FmCfg<128>::LDS_BYTES, (hipStream_t)stream, *a, g_fm_defer);
  }
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Inference chain with layer 0's features produced in the kernel (mnr_mlp_chain_fwd_ipe).
//
// mnr_cast_rays_ipe writes the level's [M, 2KL] bf16 feature matrix (1 GB per 64-sample proposal level of 360.gin) only for
// layer 0 to stream it back in: rendering needs it nowhere else.  Here a workgroup evaluates the encoding of its 256 samples
// itself: one thread per sample for the (warped) Gaussian, then, four encoding degrees (one "group" = 8K feature columns,
// zero-padded to 192 = three LDS K-tiles, so that every loop below has a compile-time trip count) at a time, one thread per (sample, basis direction) exactly as in cast_rays_ipe_kernel (anchor
// sin / cos / attenuation at the group's first degree, three double-angle steps), the bf16 features written straight into the
// LDS K-tiles the MFMAs read; the group's slice of the (group-major reordered) layer-0 weights goes global -> registers
// while the VALUs work.  Layer 0 costs ~3.5k VALU instructions per thread and tile instead of a 256 KiB HBM stream.
//
// No floating-point contraction from here on: the encoding must round like features.hip (see ipe_math.h).
#pragma clang fp contract(off)
#include "ipe_math.h"

#define FM_IPE_MAX_KS (MNR_CHAIN_IPE_GROUP_COLS / 16)      // k-steps of one group (192 columns = three K-tiles)
#define FM_IPE_MAX_K 24

template <int W>
struct FmIpeCfg {
  typedef FmCfg<W> C;
  static constexpr int GROUP_BYTES = 3 * FM_KT_BYTES;
  static constexpr int MAIN = C::X_BYTES > GROUP_BYTES ? C::X_BYTES : GROUP_BYTES;
  static constexpr int BIAS_OFF = MAIN;                                            // [MNR_CHAIN_MAX_DEPTH][W] fp32
  static constexpr int GS_OFF = BIAS_OFF + MNR_CHAIN_MAX_DEPTH * W * 4;            // FeSample[256]
  static constexpr int BASIS_OFF = GS_OFF + FM_ROWS * (int)sizeof(FeSample);       // float[3 * FM_IPE_MAX_K]
  static constexpr int LDS_BYTES = BASIS_OFF + 3 * FM_IPE_MAX_K * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(sizeof(FeSample) % 4 == 0 && GS_OFF % 16 == 0, "alignment");
};

template <int W>
__global__ __launch_bounds__(512) void mlp_chain_fwd_ipe_kernel(mnr_mlp_chain_fwd_args p, mnr_chain_ipe_args q) {
  typedef FmCfg<W> C;
  typedef FmIpeCfg<W> I;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave % C::NW, rg = wave / C::NW;
  const int64_t tiles = p.M / FM_ROWS;
  const int K = q.cfg.basis_k;
  const int ngroups = (q.cfg.max_deg - q.cfg.min_deg) / 4;
  const int pad_slots = MNR_CHAIN_IPE_GROUP_COLS / 8 - K;   // 16-byte slots [K, 24) of a row: never written by the encoding
  const unsigned no_bits[C::RB] = {};
  const float b_head = (p.w_head && p.b_head) ? p.b_head[0] : 0.0f;
  FeSample* gs = (FeSample*)(smem + I::GS_OFF);
  float* bs = (float*)(smem + I::BASIS_OFF);
  for (int i = tid; i < p.depth * W; i += 512) ((float*)(smem + I::BIAS_OFF))[i] = p.bias[i / W][i % W];
  for (int i = tid; i < K * 3; i += 512) bs[i] = q.basis[i];
  const bf16* Bt0 = (const bf16*)p.Bt[0];
  const int rpi = 512 / K;                      // encoding: rows per pass of the 512 threads (one thread per row and direction)

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * FM_ROWS;
    const bool tl_on = g_fm_timeline != nullptr && tid == 0 && tile == (int64_t)blockIdx.x + gridDim.x;
    FM_STAMP(0);
    if (tl_on) g_fm_timeline[32 * (int64_t)blockIdx.x + 30] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();                                    // the previous tile's head is done reading the activation tile
    // ---- the tile's 256 Gaussians (render.cast_rays + the contraction), one thread per sample
    if (tid < FM_ROWS) {
      const int64_t s = m0 + tid;
      const int64_t ray = s / q.n;
      const int j = (int)(s - ray * q.n);
      const float t0 = q.tdist[ray * (q.n + 1) + j], t1 = q.tdist[ray * (q.n + 1) + j + 1];
      float o[3], d[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        o[i] = q.origins[ray * 3 + i];
        d[i] = q.directions[ray * 3 + i];
      }
      FeSample g;
      fe_gaussian(q.cfg, t0, t1, o, d, q.radii[ray], g);
      gs[tid] = g;
    }
    for (int e = tid; e < FM_ROWS * pad_slots; e += 512) {
      const int r = e / pad_slots, sl = K + e % pad_slots;
      const u32x4 z = {0u, 0u, 0u, 0u};
      *(u32x4*)(smem + (sl >> 3) * FM_KT_BYTES + fm_off(r, sl & 7)) = z;
    }
    f32x16 acc[C::RB];
#pragma unroll
    for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    __syncthreads();

    // ---- layer 0, one group of four degrees at a time
    // coord.py:129-133 + :102-126 for degrees 4g .. 4g+3 of rows [r_lo, r_hi) (the loop body of cast_rays_ipe_kernel, same
    // operations in the same order, math.safe_sin's wrap through fe_wrap_100pi; feature (dl, sin/cos, k) lands in group
    // column dl*2K + {0, K} + k)
    auto encode = [&](int g, int r_lo, int r_hi) {
      const float sc0 = ldexpf(1.0f, q.cfg.min_deg + 4 * g);
      // thread et handles direction ek = et % K of rows r_lo + er0, + rpi, ... (re-derived per call: kept across the tile loop,
      // these and everything hipcc hoists with them would be spilled around the Gaussian phase)
      const int et = fm_opaque(tid);
      const int er0 = et / K, ek = et - er0 * K;
      if (et < rpi * K) {
        const float px = bs[ek * 3 + 0], py = bs[ek * 3 + 1], pz = bs[ek * 3 + 2];
        // byte offsets of this thread's eight feature columns inside a row of the group tile, before the row's swizzle:
        // K-tile (bits 15+), 16-byte slot (bits 4-6), element (bits 1-3).  The swizzle XORs the slot with (row >> 1) & 7, i.e.
        // the offset with that value << 4: one XOR per store instead of rebuilding the address from the column.
        unsigned cb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = ek + j * K;                      // sin of degree j / 2 for even j, its cos for odd j
          cb[j] = (unsigned)((c >> 6) * FM_KT_BYTES + (((c & 63) >> 3) << 4) + (c & 7) * 2);
        }
        for (int si = r_lo + er0; si < r_hi; si += rpi) {
          const FeSample gsm = gs[si];
          const float lm = gsm.mean[0] * px + gsm.mean[1] * py + gsm.mean[2] * pz;
          const float cx = gsm.cov[0] * px + gsm.cov[1] * py + gsm.cov[2] * pz;
          const float cy = gsm.cov[1] * px + gsm.cov[3] * py + gsm.cov[4] * pz;
          const float cz = gsm.cov[2] * px + gsm.cov[4] * py + gsm.cov[5] * pz;
          const float lv = px * cx + py * cy + pz * cz;
          const float vscale = -0.5f * 1.44269504088896340736f * lv;
          float sc = sc0;
          float sn, cs;
          fe_sincos_wrapped(fe_wrap_100pi(lm * sc), &sn, &cs);
          float att = exp2f(vscale * sc * sc);
          char* rowp = smem + si * 128;
          const unsigned sw4 = (unsigned)((si >> 1) & 7) << 4;
#pragma unroll
          for (int dl = 0; dl < 4; ++dl) {
            const float fs = att * sn;
            const float fc = att * cs;
            const f32x2 pr = {fs, fc};
            const bf16x2 pb = __builtin_convertvector(pr, bf16x2);
            *(bf16*)(rowp + (cb[2 * dl] ^ sw4)) = pb[0];
            *(bf16*)(rowp + (cb[2 * dl + 1] ^ sw4)) = pb[1];
            const float s2 = 2.0f * sn * cs;
            cs = 1.0f - 2.0f * sn * sn;
            sn = s2;
            const float a2 = att * att;
            att = a2 * a2;
            sc *= 2.0f;
          }
        }
      }
    };
    // the group's slice of the (group-major) layer-0 weights for this wave's 32 columns: global -> registers, L2-resident
    // (k-steps [0, 4) are requested in front of the encoding, the other eight behind it, under the first MFMAs: twelve fragments
    // do not fit next to the encoding's registers)
    auto load_w = [&](int g, auto lo_c, auto hi_c, bf16x8 (&wq)[FM_IPE_MAX_KS]) {
      const int ln = fm_opaque(lane);
      const bf16* wsrc = Bt0 + (int64_t)(cw * 32) * p.ldb[0] + g * MNR_CHAIN_IPE_GROUP_COLS + (unsigned)((ln & 31) * p.ldb[0] + (ln >> 5) * 8);
#pragma unroll
      for (int ks = decltype(lo_c)::value; ks < decltype(hi_c)::value; ++ks) wq[ks] = *(const bf16x8*)(wsrc + ks * 16);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 4> I4;
    typedef std::integral_constant<int, FM_IPE_MAX_KS> IN;
    for (int g = 0; g < ngroups; ++g) {
      bf16x8 wq[FM_IPE_MAX_KS];
      load_w(g, I0{}, I4{}, wq);
      encode(g, 0, FM_ROWS);
      fm_lds_barrier();                               // the group's feature tile is complete
      {
        load_w(g, I4{}, IN{}, wq);
        const int ln = fm_opaque(lane), frow = ln & 31, khalf = ln >> 5;
        const int sw = (frow >> 1) & 7;
        const char* xrow = smem + (rg * C::RB * 32 + frow) * 128;
#pragma unroll
        for (int ks = 0; ks < FM_IPE_MAX_KS; ++ks) {
          const char* base = xrow + (ks >> 2) * FM_KT_BYTES + ((((ks & 3) * 2 + khalf) ^ sw) << 4);
          bf16x8 fa[C::RB];
#pragma unroll
          for (int rb = 0; rb < C::RB; ++rb) fa[rb] = *(const bf16x8*)(base + rb * 4096);
#pragma unroll
          for (int rb = 0; rb < C::RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks], fa[rb], acc[rb], 0, 0, 0);
        }
      }
      fm_lds_barrier();                               // every wave is done reading the group: the next one overwrites it
    }

    FM_STAMP(1);
    bf16x8 w0[FM_WCHUNK];
    if (p.depth > 1) {
      const int ln = fm_opaque(lane);
      fm_load_wchunk((const bf16*)p.Bt[1], p.ldb[1], cw, ln & 31, ln >> 5, 0, w0);
    }
    for (int li = 0; li < p.depth; ++li) {
      if (li > 0) {
        fm_layer_mfma<W, 0>(smem, (const bf16*)p.Bt[li], p.ldb[li], cw, rg, lane, w0, acc);
        if (li + 1 < p.depth) {
          const int ln = fm_opaque(lane);
          fm_load_wchunk((const bf16*)p.Bt[li + 1], p.ldb[li + 1], cw, ln & 31, ln >> 5, 0, w0);
        }
      }
      FM_STAMP(2 + 3 * li);
      fm_lds_barrier();
      fm_epilogue<W, false>(smem, cw, rg, lane, acc, (const float*)(smem + I::BIAS_OFF) + li * W, no_bits);
      fm_lds_barrier();
      FM_STAMP(3 + 3 * li);
      if (li == p.depth - 1)
        fm_copy_out_dispatch<W>(smem, tid, m0, (bf16*)p.acts[li], nullptr, (const bf16*)p.w_head, b_head, p.head_out);
      FM_STAMP(4 + 3 * li);
    }
    if (tl_on) g_fm_timeline[32 * (int64_t)blockIdx.x + 31] = __builtin_amdgcn_s_memrealtime();
  }
}

extern "C" int mnr_mlp_chain_fwd_ipe(const mnr_mlp_chain_fwd_args* a, const mnr_chain_ipe_args* q, void* stream) {
  MNR_CHECK_ARG(a != nullptr && q != nullptr, "mnr_mlp_chain_fwd_ipe: null args");
  if (int s = fm_check_common("mnr_mlp_chain_fwd_ipe", a->M, a->W, a->depth)) return s;
  const int K = q->cfg.basis_k, L = q->cfg.max_deg - q->cfg.min_deg;
  MNR_CHECK_ARG(q->cfg.ray_shape == 0 || q->cfg.ray_shape == 1, "ray_shape must be 'cone' or 'cylinder'");   // render.py:124
  MNR_CHECK_ARG(K >= 1 && K <= FM_IPE_MAX_K && L >= 4 && L <= 32 && L % 4 == 0,
                "mnr_mlp_chain_fwd_ipe: basis_k=%d (1..%d) / degrees=%d (a multiple of 4) out of range", K, FM_IPE_MAX_K, L);
  MNR_CHECK_ARG(q->n > 0 && a->M % q->n == 0 && q->tdist && q->origins && q->directions && q->radii && q->basis,
                "mnr_mlp_chain_fwd_ipe: rays missing, or M is not a whole number of rays of n samples");
  MNR_CHECK_ARG(a->skip_layer <= 0, "mnr_mlp_chain_fwd_ipe: no skip concat");
  for (int i = 0; i < a->depth; ++i) {
    MNR_CHECK_ARG(a->Bt[i] && a->bias[i] && a->ldb[i] % 8 == 0 && a->ldb[i] >= (i == 0 ? (L / 4) * MNR_CHAIN_IPE_GROUP_COLS : a->W),
                  "mnr_mlp_chain_fwd_ipe: layer %d operand", i);
    MNR_CHECK_ARG(((uintptr_t)a->bias[i] % 16) == 0 && ((uintptr_t)a->Bt[i] % 16) == 0,
                  "mnr_mlp_chain_fwd_ipe: layer %d pointers must be 16-byte aligned", i);
    MNR_CHECK_ARG(!a->bits[i] && (!a->acts[i] || (i == a->depth - 1 && ((uintptr_t)a->acts[i] % 16) == 0)),
                  "mnr_mlp_chain_fwd_ipe: inference only (per-layer outputs are mnr_mlp_chain_fwd's)");
  }
  MNR_CHECK_ARG(a->w_head ? (a->head_out && ((uintptr_t)a->w_head % 16) == 0) : a->acts[a->depth - 1] != nullptr,
                "mnr_mlp_chain_fwd_ipe: needs a head (with head_out) or acts[depth-1]");
  const int grid = fm_grid(a->M / FM_ROWS);
  if (a->W == 256) {
    (void)hipFuncSetAttribute((const void*)mlp_chain_fwd_ipe_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, FmIpeCfg<256>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_fwd_ipe_kernel<256>, dim3(grid), dim3(512), FmIpeCfg<256>::LDS_BYTES, (hipStream_t)stream, *a, *q);
  } else {
    (void)hipFuncSetAttribute((const void*)mlp_chain_fwd_ipe_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, FmIpeCfg<128>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_fwd_ipe_kernel<128>, dim3(grid), dim3(512), FmIpeCfg<128>::LDS_BYTES, (hipStream_t)stream, *a, *q);
  }
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}
