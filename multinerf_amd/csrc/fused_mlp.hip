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
// between layers.  Inference (acts = bits = NULL) writes only the head output.
//
// LDS tile format (shared with gemm.hip's NT kernel): a [256 rows][W] bf16 activation is W/64 K-tiles of
// [256][64] = 32 KiB, row pitch 128 B, 16-byte slot s of row r stored at slot position s ^ ((r >> 1) & 7).
#include <stdio.h>

#include "common.h"

#define FM_ROWS 256
#define FM_KT_BYTES 32768                     // one [256][64] bf16 K-tile

__device__ __forceinline__ int fm_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// LDS-DMA of a [ROWS][64] bf16 tile (rows row0.., columns k0.. of the row-major matrix g) into lds_tile; the swizzle is
// applied to the SOURCE address, the DMA image itself is lane-linear (see gemm.hip nt_stage_tile).
template <int ROWS>
__device__ __forceinline__ void fm_stage_tile(const bf16* __restrict__ g, int ld, int64_t row0, int k0, char* lds_tile,
                                              int wave, int lane) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 512; ++i) {
    const int cbase = (i * 8 + wave) * 64;
    const int c = cbase + lane;
    const int r = c >> 3;
    const int slot = (c & 7) ^ ((r >> 1) & 7);
    const bf16* src = g + (row0 + r) * (int64_t)ld + k0 + slot * 8;
    __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(src), MNR_LDS_PTR(lds_tile + cbase * 16), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 fm_read_frag(const char* kt_tile, int row, int kslot) {
  return *(const bf16x8*)(kt_tile + fm_off(row, kslot));
}

template <int W>
struct FmCfg {
  static constexpr int NW = W / 32;                    // waves along the output columns
  static constexpr int RG = 8 / NW;                    // row groups
  static constexpr int RB = 8 / RG;                    // 32-row blocks per wave
  static constexpr int NKT = W / 64;                   // K-tiles of the resident activation
  static constexpr int X_BYTES = NKT * FM_KT_BYTES;
  static constexpr int STAGE_BYTES = FM_KT_BYTES + W * 128;       // layer 0: feature tile + weight tile [W][64]
  static constexpr int LDS_BYTES = X_BYTES > 2 * STAGE_BYTES ? X_BYTES : 2 * STAGE_BYTES;
  static constexpr int CPR = W / 8;                    // 16-byte chunks per activation row
  static constexpr int COPY_ITERS = FM_ROWS * CPR / 512;
  static constexpr int ROW_STEP = 512 / CPR;
  static_assert(W == 128 || W == 256, "fused chain: width 128 or 256");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// One layer whose input sits in LDS: acc[rb] += X[rows of this wave] * Bt[cols of this wave]^T.
// The wave's weight fragments come global -> registers in chunks of four k-steps (16 registers), the next chunk in
// flight while this one multiplies; chunk 0 arrives preloaded (`w0`, issued before the previous layer's epilogue).
#define FM_WCHUNK 4
__device__ __forceinline__ void fm_load_wchunk(const bf16* __restrict__ Bt, int ldb, int cw, int frow, int khalf, int chunk,
                                               bf16x8 (&w)[FM_WCHUNK]) {
  const bf16* wsrc = Bt + (int64_t)(cw * 32 + frow) * ldb + khalf * 8 + chunk * (FM_WCHUNK * 16);
#pragma unroll
  for (int j = 0; j < FM_WCHUNK; ++j) w[j] = *(const bf16x8*)(wsrc + j * 16);
}

template <int W>
__device__ __forceinline__ void fm_layer_mfma(const char* X, const bf16* __restrict__ Bt, int ldb, int cw, int rg, int frow,
                                              int khalf, const bf16x8 (&w0)[FM_WCHUNK], f32x16 (&acc)[FmCfg<W>::RB]) {
  typedef FmCfg<W> C;
  constexpr int NCH = W / 16 / FM_WCHUNK;
  bf16x8 wq[2][FM_WCHUNK];
#pragma unroll
  for (int j = 0; j < FM_WCHUNK; ++j) wq[0][j] = w0[j];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) fm_load_wchunk(Bt, ldb, cw, frow, khalf, c + 1, wq[(c + 1) & 1]);
#pragma unroll
    for (int j = 0; j < FM_WCHUNK; ++j) {
      const int ks = c * FM_WCHUNK + j;
      const char* kt = X + (ks >> 2) * FM_KT_BYTES;
      constexpr int HB = C::RB > 4 ? 4 : C::RB;          // four row blocks at a time: 16 fragment registers
#pragma unroll
      for (int h = 0; h < C::RB / HB; ++h) {
        bf16x8 fa[HB];
#pragma unroll
        for (int i = 0; i < HB; ++i) fa[i] = fm_read_frag(kt, (rg * C::RB + h * HB + i) * 32 + frow, (ks & 3) * 2 + khalf);
#pragma unroll
        for (int i = 0; i < HB; ++i)
          acc[h * HB + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[c & 1][j], fa[i], acc[h * HB + i], 0, 0, 0);
      }
    }
  }
}

// acc -> bf16 -> the LDS activation tile.  Forward: + bias, ReLU.  Backward: ReLU mask bits of the layer below.
// acc[rb][r]: column n = cw*32 + (r&3) + 8*(r>>2) + 4*khalf, row m = (rg*RB + rb)*32 + frow.
template <int W, bool BWD>
__device__ __forceinline__ void fm_epilogue(char* X, int cw, int rg, int frow, int khalf, f32x16 (&acc)[FmCfg<W>::RB],
                                            const float (&bias_r)[16], const unsigned (&mbits)[FmCfg<W>::RB]) {
  typedef FmCfg<W> C;
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
      *(i32x2*)(X + (nl >> 6) * FM_KT_BYTES + fm_off(ml, (nl & 63) >> 3) + (nl & 7) * 2) = pk;
    }
  }
}

// The LDS activation tile -> global memory in full rows (16 B per lane), the 1-bit "> 0" masks, and (forward, last
// layer) the Dense(1) head: 8-element partial dot products reduced over the W/8 lanes that share a row.
template <int W>
__device__ __forceinline__ void fm_copy_out(const char* X, int tid, int64_t m0, bf16* __restrict__ dst, uint8_t* __restrict__ bits,
                                            const float (&whead)[8], bool do_head, float b_head, float* __restrict__ head_out) {
  typedef FmCfg<W> C;
  const int row0 = tid / C::CPR, ch = tid % C::CPR;
  const char* lptr = X + (ch >> 3) * FM_KT_BYTES;
#pragma unroll
  for (int it = 0; it < C::COPY_ITERS; ++it) {
    const int row = row0 + it * C::ROW_STEP;
    const u32x4 w = *(const u32x4*)(lptr + fm_off(row, ch & 7));
    if (bits) {                                           // (kernel-uniform)
      unsigned f = 0;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned wd = w[d];
        const s16x2 z = {0, 0};
        const unsigned pos = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, wd), z));
        f |= ((pos + 0x7fff7fffu) & 0x80008000u) >> (15 - 2 * d);
      }
      unsigned mb = (f | (f >> 15)) & 0xffu;
      mb |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xF5, 0xf, 0xf, false) << 8;     // quad_perm [1,1,3,3]
      mb |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mb, 0xAA, 0xf, 0xf, false) << 16;    // quad_perm [2,2,2,2]
      if ((ch & 3) == 0) *(unsigned*)(bits + (m0 + row) * (int64_t)(W / 8) + ch) = mb;
    }
    if (dst) *(u32x4*)(dst + (m0 + row) * (int64_t)W + ch * 8) = w;
    if (do_head) {
      const bf16x8 v = __builtin_bit_cast(bf16x8, w);
      float s = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[e] * whead[e];
#pragma unroll
      for (int d = C::CPR / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
      if (ch == 0) head_out[m0 + row] = s + b_head;
    }
  }
}

template <int W>
__global__ __launch_bounds__(512) void mlp_chain_fwd_kernel(mnr_mlp_chain_fwd_args p) {
  typedef FmCfg<W> C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave % C::NW, rg = wave / C::NW;
  const int frow = lane & 31, khalf = lane >> 5;
  const int64_t tiles = p.M / FM_ROWS;
  const bf16* feat = (const bf16*)p.feat;
  const int nk0 = p.K0 / 64;
  const unsigned no_bits[C::RB] = {};

  const float b_head = (p.w_head && p.b_head) ? p.b_head[0] : 0.0f;

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * FM_ROWS;
    f32x16 acc[C::RB];
#pragma unroll
    for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;

    // ---- layer 0: K0 streamed through two LDS stages (feature tile [256][64] + weight tile [W][64] per step)
    __syncthreads();                                    // the previous tile's copy-out is done reading the buffer
    {
      const bf16* Bt0 = (const bf16*)p.Bt[0];
      auto stage = [&](int kt) {
        char* base = smem + (kt & 1) * C::STAGE_BYTES;
        fm_stage_tile<FM_ROWS>(feat, p.ld_feat, m0, kt * 64, base, wave, lane);
        fm_stage_tile<W>(Bt0, p.ldb[0], 0, kt * 64, base + FM_KT_BYTES, wave, lane);
      };
      stage(0);
      for (int kt = 0; kt < nk0; ++kt) {
        MNR_GPU_ONLY(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"));
        MNR_SIM_HOOK(hipsim::wait_vmcnt(0));
        __builtin_amdgcn_s_barrier();                   // step kt has landed everywhere; step kt-1's buffer is free
        asm volatile("" ::: "memory");
        if (kt + 1 < nk0) stage(kt + 1);
        const char* As = smem + (kt & 1) * C::STAGE_BYTES;
        const char* Ws = As + FM_KT_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 wf = fm_read_frag(Ws, cw * 32 + frow, ks * 2 + khalf);
          constexpr int HB = C::RB > 4 ? 4 : C::RB;
#pragma unroll
          for (int h = 0; h < C::RB / HB; ++h) {
            bf16x8 fa[HB];
#pragma unroll
            for (int i = 0; i < HB; ++i) fa[i] = fm_read_frag(As, (rg * C::RB + h * HB + i) * 32 + frow, ks * 2 + khalf);
#pragma unroll
            for (int i = 0; i < HB; ++i)
              acc[h * HB + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, fa[i], acc[h * HB + i], 0, 0, 0);
          }
        }
      }
    }

    bf16x8 w0[FM_WCHUNK];
    if (p.depth > 1) fm_load_wchunk((const bf16*)p.Bt[1], p.ldb[1], cw, frow, khalf, 0, w0);
    for (int li = 0; li < p.depth; ++li) {
      if (li > 0) fm_layer_mfma<W>(smem, (const bf16*)p.Bt[li], p.ldb[li], cw, rg, frow, khalf, w0, acc);
      if (li + 1 < p.depth && li > 0) fm_load_wchunk((const bf16*)p.Bt[li + 1], p.ldb[li + 1], cw, frow, khalf, 0, w0);
      float bias_r[16];
      {
        const float* bp = p.bias[li];
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[r] = bp[cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf];
      }
      __syncthreads();                                  // every wave is done reading this layer's input
      fm_epilogue<W, false>(smem, cw, rg, frow, khalf, acc, bias_r, no_bits);
#pragma unroll
      for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
      __syncthreads();                                  // the layer's output is complete in LDS
      const bool last = li == p.depth - 1;
      bf16* dst = p.acts[li] ? (bf16*)p.acts[li] : nullptr;
      if (last && p.w_head) {
        float whead[8];
        const bf16x8 wh = *(const bf16x8*)((const bf16*)p.w_head + (tid % C::CPR) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) whead[e] = (float)wh[e];
        fm_copy_out<W>(smem, tid, m0, dst, p.bits[li], whead, true, b_head, p.head_out);
      } else if (dst || p.bits[li]) {
        const float none[8] = {};
        fm_copy_out<W>(smem, tid, m0, dst, p.bits[li], none, false, 0.0f, nullptr);
      }
    }
  }
}

template <int W>
__global__ __launch_bounds__(512) void mlp_chain_bwd_kernel(mnr_mlp_chain_bwd_args p) {
  typedef FmCfg<W> C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave % C::NW, rg = wave / C::NW;
  const int frow = lane & 31, khalf = lane >> 5;
  const int64_t tiles = p.M / FM_ROWS;
  const float no_bias[16] = {};
  const float no_head[8] = {};

  // head kernel: this thread's 8 columns of the rank-1 start dY_last = mask * (g (x) w_head)
  const int row0 = tid / C::CPR, ch = tid % C::CPR;
  float wh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) wh[e] = p.w_head[ch * 8 + e];

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * FM_ROWS;
    __syncthreads();                                    // the previous tile's copy-out is done reading the buffer
    {
      const int last = p.depth - 1;
      const uint8_t* bl = p.bits[last];
      bf16* dst = (bf16*)p.dY[last];
#pragma unroll
      for (int it = 0; it < C::COPY_ITERS; ++it) {
        const int row = row0 + it * C::ROW_STEP;
        const float g = p.g_head[m0 + row];
        const unsigned mb = bl[(m0 + row) * (int64_t)(W / 8) + ch];
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)(((mb >> e) & 1u) ? g * wh[e] : 0.0f);
        *(bf16x8*)(smem + (ch >> 3) * FM_KT_BYTES + fm_off(row, ch & 7)) = v;
        if (dst) *(bf16x8*)(dst + (m0 + row) * (int64_t)W + ch * 8) = v;
      }
    }
    bf16x8 w0[FM_WCHUNK];
    if (p.depth > 1) fm_load_wchunk((const bf16*)p.Bw[p.depth - 1], p.ldb[p.depth - 1], cw, frow, khalf, 0, w0);
    __syncthreads();
    for (int li = p.depth - 1; li >= 1; --li) {
      f32x16 acc[C::RB];
#pragma unroll
      for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
      fm_layer_mfma<W>(smem, (const bf16*)p.Bw[li], p.ldb[li], cw, rg, frow, khalf, w0, acc);
      if (li > 1) fm_load_wchunk((const bf16*)p.Bw[li - 1], p.ldb[li - 1], cw, frow, khalf, 0, w0);
      // ReLU masks of layer li-1 for this lane's rows: one dword = the 32 columns of this wave
      unsigned mbits[C::RB];
      {
        const uint8_t* bl = p.bits[li - 1];
#pragma unroll
        for (int rb = 0; rb < C::RB; ++rb)
          mbits[rb] = *(const unsigned*)(bl + (m0 + (rg * C::RB + rb) * 32 + frow) * (int64_t)(W / 8) + cw * 4);
      }
      __syncthreads();
      fm_epilogue<W, true>(smem, cw, rg, frow, khalf, acc, no_bias, mbits);
      __syncthreads();
      fm_copy_out<W>(smem, tid, m0, (bf16*)p.dY[li - 1], nullptr, no_head, false, 0.0f, nullptr);
    }
  }
}

static int fm_grid(int64_t tiles) {
  static int cus = 0;
  if (cus == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
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
  MNR_CHECK_ARG(a->feat && a->K0 > 0 && a->K0 % 64 == 0 && a->ld_feat % 8 == 0 && a->ld_feat >= a->K0,
                "mnr_mlp_chain_fwd: feature matrix needs K0 %% 64 == 0 and ld_feat %% 8 == 0");
  for (int i = 0; i < a->depth; ++i) {
    MNR_CHECK_ARG(a->Bt[i] && a->bias[i] && a->ldb[i] % 8 == 0 && a->ldb[i] >= (i == 0 ? a->K0 : a->W),
                  "mnr_mlp_chain_fwd: layer %d operand", i);
    MNR_CHECK_ARG(((uintptr_t)a->Bt[i] % 16) == 0 && (!a->acts[i] || ((uintptr_t)a->acts[i] % 16) == 0) &&
                      (!a->bits[i] || ((uintptr_t)a->bits[i] % 4) == 0),
                  "mnr_mlp_chain_fwd: layer %d pointers must be 16-byte (bits: 4-byte) aligned", i);
  }
  MNR_CHECK_ARG(((uintptr_t)a->feat % 16) == 0, "mnr_mlp_chain_fwd: feat must be 16-byte aligned");
  MNR_CHECK_ARG(!a->w_head || (a->head_out && ((uintptr_t)a->w_head % 16) == 0), "mnr_mlp_chain_fwd: head needs head_out");
  const int grid = fm_grid(a->M / FM_ROWS);
  if (a->W == 256) {
    (void)hipFuncSetAttribute((const void*)mlp_chain_fwd_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, FmCfg<256>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_fwd_kernel<256>, dim3(grid), dim3(512), FmCfg<256>::LDS_BYTES, (hipStream_t)stream, *a);
  } else {
    (void)hipFuncSetAttribute((const void*)mlp_chain_fwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, FmCfg<128>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_fwd_kernel<128>, dim3(grid), dim3(512), FmCfg<128>::LDS_BYTES, (hipStream_t)stream, *a);
  }
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

extern "C" int mnr_mlp_chain_bwd(const mnr_mlp_chain_bwd_args* a, void* stream) {
  MNR_CHECK_ARG(a != nullptr, "mnr_mlp_chain_bwd: null args");
  if (int s = fm_check_common("mnr_mlp_chain_bwd", a->M, a->W, a->depth)) return s;
  MNR_CHECK_ARG(a->g_head && a->w_head, "mnr_mlp_chain_bwd: head gradient / head kernel missing");
  for (int i = 0; i < a->depth; ++i) {
    MNR_CHECK_ARG(a->bits[i] && ((uintptr_t)a->bits[i] % 4) == 0, "mnr_mlp_chain_bwd: layer %d needs its ReLU mask bits", i);
    MNR_CHECK_ARG(a->dY[i] || i == a->depth - 1, "mnr_mlp_chain_bwd: dY[%d] missing", i);
    MNR_CHECK_ARG(!a->dY[i] || ((uintptr_t)a->dY[i] % 16) == 0, "mnr_mlp_chain_bwd: dY[%d] must be 16-byte aligned", i);
    if (i >= 1)
      MNR_CHECK_ARG(a->Bw[i] && a->ldb[i] % 8 == 0 && a->ldb[i] >= a->W && ((uintptr_t)a->Bw[i] % 16) == 0,
                    "mnr_mlp_chain_bwd: layer %d operand", i);
  }
  const int grid = fm_grid(a->M / FM_ROWS);
  if (a->W == 256) {
    (void)hipFuncSetAttribute((const void*)mlp_chain_bwd_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, FmCfg<256>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_bwd_kernel<256>, dim3(grid), dim3(512), FmCfg<256>::LDS_BYTES, (hipStream_t)stream, *a);
  } else {
    (void)hipFuncSetAttribute((const void*)mlp_chain_bwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, FmCfg<128>::LDS_BYTES);
    hipLaunchKernelGGL(mlp_chain_bwd_kernel<128>, dim3(grid), dim3(512), FmCfg<128>::LDS_BYTES, (hipStream_t)stream, *a);
  }
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}
