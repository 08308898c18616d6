// Panel-layout NT GEMM for the wide trunk layers (gfx950): C = epi([A1 | A2] * Bt^T) with the bf16 result, and
// optionally A1, in MNR_LAYOUT_PANEL (include/mnerf.h: 1-KiB blocks of 32 rows x 16 columns).
//
// Why a second NT kernel.  The tiled kernel of gemm.hip spends 22k of the 59k cycles of a 256 x 256 x 1024 tile at the
// tile boundary: its row-major result has to be transposed through LDS (a lane's accumulators are 4
// consecutive columns of 32 different rows), the staging pass reuses the operand stage buffers, so the next tile's
// first K-tiles cannot be requested before the epilogue is over (pipeline fill), and the staged rows leave as 136 KiB of
// stores behind two barriers.  Here the result matrix is stored in the order the MFMA hands it out:
//   * after four v_permlane32_swap per 32 x 32 block a lane holds 2 x 8 consecutive columns of one row, and the 64 lanes of
//     a store instruction cover one whole 1-KiB block of the panel layout: the tile leaves in 16 stores per wave straight
//     from registers, no LDS staging, no barrier;
//   * so the stage buffers are never reused: the K loop is ONE LDS-DMA pipeline over all the tiles a persistent workgroup
//     walks (the next tile's first three K-tiles are requested during this tile's last three);
//   * the epilogue of tile t is interleaved, block by block, with the first k-step of tile t+1 (each block's first MFMA takes
//     C = 0, so a block's accumulators are free as soon as its own 16 values are converted): the matrix pipe runs on under
//     the conversion / mask / store instructions;
//   * the 1-bit ReLU masks are kept in TILE order (16 bytes per thread and tile, include/mnerf.h): the forward layer
//     writes them with one store per thread, the dX layer of the same tile shape gets its 16 bytes through LDS-DMA.
// The next GEMM reads a panel-layout A1 with source-swizzled LDS-DMA as before (a piece = 8 consecutive k of one row
// is 16 contiguous bytes in either layout): 512-byte runs instead of 64-byte row segments.  The weight-gradient kernel
// takes panel operands too (gemm.hip, gemm_tn_body.inc).
//
// K loop: the hand-pipelined loop of gemm_nt_body.inc (BK = 32, four 32-KiB stages, fragments double-buffered in
// registers, DMA pieces between the MFMAs, counted vmcnt + one raw barrier per K-tile).  vmcnt retires in order, stores
// included; the counted waits of the first two K-tiles after an epilogue allow the tile's 16 stores to stay outstanding
// (they were issued behind the pieces those waits are for), the third one is the first that needs them acknowledged.
#include <type_traits>

#include "common.h"

// Probe builds (tools/panel_probe.py with MNR_LIB_PATH): 1 = the epilogue without its global stores, 2 = no epilogue at all
// (K loop only), 3 = the epilogue without the mask bytes, 4 = no MFMAs (operand movement only), 5 = the counted waits of K-tiles 2
// and 3 behind an epilogue also leave its stores outstanding (a RACE: timing only), 6 = every second store, 7 / 8 = every M-tile
// reads the A1 rows of M-tile (m & 7) / (m & 63): one 512-KiB tile per XCD (L2-resident, not L1) / 32 MiB in all (Infinity-Cache-
// resident): what an A operand that does not come from HBM is worth; 9 / 10 = (m & 511) / (m & 1023): 256 / 512 MiB, re-read 4 / 2
// times a launch (past the Infinity Cache, still few pages); 11 / 12 = (m & 127) / (m & 255): 64 / 128 MiB; 13 / 14 = 1/6 / 1/3
// fewer LDS fragment reads (read_frags).  0 in the product.
#ifndef PN_DBG
#define PN_DBG 0
#endif

namespace {

constexpr int PN_BM = 256, PN_BN = 256, PN_BK = 32, PN_STAGES = 4, PN_KS = PN_BK / 16;
constexpr int PN_MI = 4, PN_NJ = 2;                      // 8 waves as 2 (M) x 4 (N), 128 x 64 per wave
constexpr int PN_ROWB = PN_BK * 2;                       // bytes per staged operand row
constexpr int PN_A_BYTES = PN_BM * PN_ROWB;
constexpr int PN_STAGE_BYTES = PN_A_BYTES + PN_BN * PN_ROWB;
constexpr int PN_LPS = PN_STAGE_BYTES / 16 / 512;        // LDS-DMA instructions per wave and K-tile (4)
constexpr int PN_PPK = PN_LPS / PN_KS;                   // pieces issued per k-step (2)
constexpr int PN_EXTRA_OFF = PN_STAGES * PN_STAGE_BYTES; // per-tile extras behind the stage buffers, two parities
constexpr int PN_NST = 16;                               // stores of one tile's epilogue per wave that the counted waits may leave
                                                         // outstanding (the forward layer's mask store is not counted: one
                                                         // older piece is waited for instead)
constexpr int PN_MIN_NK = 6;                             // K-tiles per output tile: {0, 1} and {nk-3, nk-2, nk-1} are special

typedef unsigned pn_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pn_u32x2 __attribute__((ext_vector_type(2)));

// (post-ReLU bf16 > 0) flags of 8 values held as 4 dwords -> one byte, bit e = element e (element 2d = low half of dword d).
// The values are not negative here (ReLU applied), so "> 0" is "bits != 0" = min(half, 1) as unsigned 16-bit: a 0 / 1 in
// bytes 0 and 2 of each dword, which one v_dot4_u32_u8 per dword weighs with (1 << 2d, 1 << (2d + 1)) and adds up
// (8 instructions per byte; shifts, ors and ands took 13).
__device__ __forceinline__ unsigned pn_nonzero_byte(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
  const unsigned w[4] = {w0, w1, w2, w3};
  unsigned byte = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    unsigned m;
    MNR_GPU_ONLY(asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(w[d]), "v"(0x00010001u));
                 byte = __builtin_amdgcn_udot4(m, (1u << (2 * d)) | (1u << (2 * d + 17)), byte, false));
    MNR_SIM_HOOK(m = ((w[d] & 0xffffu) ? 1u : 0u) | ((w[d] >> 16) ? 0x10000u : 0u);
                 byte += (m & 1u) << (2 * d) | (m >> 16) << (2 * d + 1));
  }
  return byte;
}

}  // namespace

template <bool A1_PANEL, bool BITS_IN>
__global__ __launch_bounds__(512) void gemm_nt_panel_kernel(mnr_gemm_nt_args p, long long vtotal, int rev) {
  constexpr int BM = PN_BM, BN = PN_BN, BK = PN_BK, STAGES = PN_STAGES, KS = PN_KS, MI = PN_MI, NJ = PN_NJ;
  constexpr int ROWB = PN_ROWB, A_BYTES = PN_A_BYTES, STAGE_BYTES = PN_STAGE_BYTES, LPS = PN_LPS, PPK = PN_PPK;
  constexpr int EXTRA = BITS_IN ? 8192 : 2048;           // per tile: the mask bits of a dX tile / two copies of the bias row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = p.N / BN;
  const int64_t mt = p.M / BM;
  const int K = p.K1 + p.K2;
  const int nk = K / BK;
  const bf16* const A1 = (const bf16*)p.A1;
  const bf16* const A2 = (const bf16*)p.A2;
  const bf16* const Bt = (const bf16*)p.Bt;
  bf16* const Cb = (bf16*)p.Cb;

  // XCD-aware tile order (as gemm_nt_kernel): virtual workgroup v = tile; the nt N-tiles of one M-tile run consecutively on one
  // XCD; M-tiles are padded to 8 per group, padded slots are skipped.  32-bit arithmetic (the launcher checks vtotal < 2^31),
  // results pinned to SGPRs: hipcc evaluates integer divisions on the vector unit, and everything derived from them (six tile
  // base pointers, the epilogue's addresses) would live in VGPR pairs.
  const int vtot = (int)vtotal, gstep = (int)gridDim.x, mt32 = (int)mt;
  // rev: the M-tiles in descending order (the launcher alternates it between consecutive launches: a layer then starts with the
  // rows the previous layer wrote last, which are the ones the 256 MB Infinity Cache still holds)
  auto decode = [&](int v_, int& m_tile, int& n_tile) -> bool {
    const int v = rev ? vtot - 1 - v_ : v_;
    const int q = v >> 3;
    m_tile = __builtin_amdgcn_readfirstlane((v & 7) + 8 * (q / nt));
    n_tile = __builtin_amdgcn_readfirstlane(q % nt);
    return m_tile < mt32;
  };
  int v = (int)blockIdx.x, cm = 0, cn = 0;
  while (v < vtot && !decode(v, cm, cn)) v += gstep;
  if (v >= vtot) return;

  // Per-lane constants (live across the whole walk).  DMA piece c = (i * 8 + wave) * 64 + lane of an operand tile
  // [256 rows][4 slots of 16 B]: row c / 4, slot position c % 4 holds global k-slot (c % 4) ^ ((row >> 2) & 3).
  const int lane = mnr_lane_id();
  const int fr = lane & 31, kh = lane >> 5;
  const int sw = (fr >> 2) & 3;
  const int a_lane = (wm * 32 * MI + fr) * ROWB;
  const int b_lane = A_BYTES + (wn * 32 * NJ + fr) * ROWB;
  const int slot = (lane & 3) ^ ((lane >> 4) & 3);
  const int prow = wave * 16 + (lane >> 2);                // row of piece 0; piece 1: + 128 rows (a wave-uniform term)
  const int pcol = slot * 8;
  const int lda1 = p.lda1, lda2 = p.lda2, ldb = p.ldb, K1 = p.K1;
  // per-lane element offsets of a piece inside its segment's tile; panel source of A1: block (row >> 5, k >> 4) of 512
  // elements, row pitch 16 inside
  const unsigned offA1 = A1_PANEL ? (unsigned)((wave >> 1) * (lda1 * 32) + ((wave & 1) * 16 + (lane >> 2)) * 16 + (slot >> 1) * 512 + (slot & 1) * 8)
                                  : (unsigned)(prow * lda1 + pcol);
  const unsigned offA2 = (unsigned)(prow * lda2 + pcol);
  const unsigned offB = (unsigned)(prow * ldb + pcol);

  // Source of the K-tile at k0 of the tile with base pointers (a1t, a2t, bt), resolved ONCE per K-tile in scalar registers
  // (segment select, k offset): its four pieces (one 1-KiB LDS-DMA instruction per wave each: 2 of the activations, then 2 of
  // the weights) differ by a wave-uniform 128-row stride.  k0 passes through an empty asm: with a constant k0 (the peeled
  // K-tiles) hipcc hoists the segment select of every call site out of the walk as a 64-bit per-lane offset and spills them.
  struct Src {
    const bf16* a;
    const bf16* b;
    int64_t astep;
    unsigned aoff;
  };
  auto resolve = [&](const bf16* a1t, const bf16* a2t, const bf16* bt, int k0_) {
    const int k0 = mnr_opaque_s(k0_);
    const bool first = k0 < K1;
    const int64_t ka1 = A1_PANEL ? (int64_t)k0 * 32 : (int64_t)k0;
    Src r;
    r.a = first ? a1t + ka1 : a2t + (k0 - K1);
    r.astep = (int64_t)128 * (first ? lda1 : lda2);
    r.aoff = first ? offA1 : offA2;
    r.b = bt + k0;
    return r;
  };
  auto stage_piece = [&](int buf, const Src& sr, int i) {
    char* base = smem + buf * STAGE_BYTES;
    if (i < 2) {
      __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(sr.a + i * sr.astep + sr.aoff), MNR_LDS_PTR(base + (i * 8 + wave) * 1024), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(sr.b + (int64_t)(i - 2) * 128 * ldb + offB),
                                       MNR_LDS_PTR(base + A_BYTES + ((i - 2) * 8 + wave) * 1024), 16, 0, 0);
    }
  };
  // per-tile extras of tile (m_tile, n_tile) into parity `par`: one more DMA instruction per wave
  auto stage_extra = [&](int m_tile, int n_tile, int par) {
    char* dst = smem + PN_EXTRA_OFF + par * EXTRA;
    if constexpr (BITS_IN) {
      const uint8_t* src = p.mask_bits_in + ((int64_t)(m_tile * nt + n_tile) * 512 + wave * 64) * 16 + (unsigned)(lane * 16);
      __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(src), MNR_LDS_PTR(dst + wave * 1024), 16, 0, 0);
    } else {
      const float* src = p.bias + (n_tile * BN + (wave & 3) * 64) + (unsigned)lane;
      __builtin_amdgcn_global_load_lds(MNR_GLOBAL_PTR(src), MNR_LDS_PTR(dst + wave * 256), 4, 0, 0);
    }
  };

  f32x16 acc[NJ][MI];
  bf16x8 fa[2][MI], fb[2][NJ];
  auto read_frags = [&](const char* st, int ks, bf16x8 (&A)[MI], bf16x8 (&Bf)[NJ]) {
    const int ko = ((ks * 2 + kh) ^ sw) << 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // (probe builds 13 / 14: one / two of the four A-fragment reads of a k-step are skipped, the registers of another row block
      // reused -- wrong values, timing only: what the launch costs with 1/6 / 1/3 fewer LDS fragment reads per MFMA)
      if ((PN_DBG == 13 && i == 3) || (PN_DBG == 14 && i >= 2)) A[i] = A[i - 2 + (PN_DBG == 13 ? 1 : 0)];
      else A[i] = *(const bf16x8*)(st + a_lane + ko + i * 32 * ROWB);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) Bf[j] = *(const bf16x8*)(st + b_lane + ko + j * 32 * ROWB);
  };

  // Epilogue of one 32 x 32 block (j, i) of the tile described by (ec_*): accumulators -> (+ bias, ReLU | mask bits) -> bf16
  // pairs -> half-wave exchange -> two 16-byte stores = two whole panel blocks per wave.
  // acc[j][i][r]: n = n0 + wn*64 + j*32 + (r&3) + 8*(r>>2) + 4*kh, m = m0 + wm*128 + i*32 + fr.
  // Blocks are visited i-major (the order of the first k-step's MFMAs of the next tile): ONE 64-bit per-lane address walks
  // the tile, + immediates inside a row block, + the row-block stride after each i.  Register budget of the interleaved form:
  // 128 accumulators + 24 fragment registers of the running k-step + 32 bias values + the block in flight; the next k-step's
  // fragments are requested behind the epilogue (with them in front of it hipcc went past 256 registers, spilled the per-lane
  // constants of the K loop and reloaded them, behind a vmcnt(0), in every K-tile).
  char* ec_c = nullptr;                                    // byte address of the wave's first block of the tile (uniform)
  const char* ec_x = nullptr;                              // the tile's extras in LDS (uniform)
  char* ec_bits = nullptr;                                 // the wave's 1 KiB of the tile's mask bits (forward; uniform)
  int64_t ec_istride = 0;                                  // bytes between the row blocks i and i + 1
  char* ec_p = nullptr;                                    // per lane: this lane's 16 bytes of block (j = 0, g = 0) of row block i
  pn_u32x4 bitsv = {0u, 0u, 0u, 0u}, mout = {0u, 0u, 0u, 0u};
  f32x4 bias_v[NJ][4];
  auto epi_set = [&](int m_tile, int n_tile, int par) {
    const int64_t blk0 = (int64_t)(m_tile * (BM / 32) + wm * 4) * (p.N / 16) + n_tile * (BN / 16) + wn * 4;
    ec_c = (char*)Cb + blk0 * 1024;
    ec_istride = (int64_t)p.N * 64;
    ec_x = smem + PN_EXTRA_OFF + par * EXTRA;
    ec_bits = p.mask_bits_out ? (char*)p.mask_bits_out + ((int64_t)(m_tile * nt + n_tile) * 512 + wave * 64) * 16 : nullptr;
  };
  auto epi_block = [&](int j, int i) {
    if constexpr (PN_DBG == 2) {                          // no epilogue: the accumulators are only kept alive
      mout[0] ^= __builtin_bit_cast(unsigned, acc[j][i][0]);
      if (j == NJ - 1 && i == MI - 1) *(pn_u32x4*)(ec_c + (unsigned)(mnr_opaque(lane) * 16)) = mout;
      return;
    }
    const int lo = mnr_opaque(lane);
    if (j == 0 && i == 0) {
      ec_p = ec_c + (unsigned)((lo & 31) * 32 + (lo >> 5) * 16);
      if constexpr (BITS_IN) {
        bitsv = *(const pn_u32x4*)(ec_x + (wave * 64 + lo) * 16);
      } else {
        mout = pn_u32x4{0u, 0u, 0u, 0u};
        // this lane's 2 x 16 bias values, eight 16-byte LDS reads in one go (fetched four at a time where they are added, each
        // read was waited for on the spot: 32 LDS round trips per tile in the epilogue's dependent chain)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) bias_v[jj][rq] = *(const f32x4*)(ec_x + (wn * 64 + jj * 32 + rq * 8 + (lo >> 5) * 4) * 4);
      }
    }
    unsigned d[8];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x2 a0 = {acc[j][i][rq * 4 + 0], acc[j][i][rq * 4 + 1]};
      const f32x2 a1 = {acc[j][i][rq * 4 + 2], acc[j][i][rq * 4 + 3]};
      f32x2 s0 = a0, s1 = a1;
      if constexpr (!BITS_IN) {
        const f32x4 b4 = bias_v[j][rq];
        s0 = a0 + f32x2{b4[0], b4[1]};
        s1 = a1 + f32x2{b4[2], b4[3]};
      }
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      s16x2 h0 = __builtin_bit_cast(s16x2, __builtin_convertvector(s0, bf16x2));
      s16x2 h1 = __builtin_bit_cast(s16x2, __builtin_convertvector(s1, bf16x2));
      if constexpr (!BITS_IN) {
        // ReLU on the bf16 bit patterns as a packed signed 16-bit max with 0 (rounding commutes with the clamp)
        const s16x2 z = {0, 0};
        h0 = __builtin_elementwise_max(h0, z);
        h1 = __builtin_elementwise_max(h1, z);
      }
      d[2 * rq] = __builtin_bit_cast(unsigned, h0);
      d[2 * rq + 1] = __builtin_bit_cast(unsigned, h1);
    }
    // Half-wave exchange (cdna guide T21): column groups rq = 2g (vdst) and 2g + 1 (src): afterwards lanes 0-31 hold columns
    // 16g .. 16g+7 and lanes 32-63 columns 16g+8 .. 16g+15 of their row, as (d[4g], d[4g+1], d[4g+2], d[4g+3]).
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const pn_u32x2 r = __builtin_amdgcn_permlane32_swap(d[4 * g + h], d[4 * g + 2 + h], false, false);
        d[4 * g + h] = r[0];
        d[4 * g + 2 + h] = r[1];
      }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      pn_u32x4 w = {d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]};
      const int bidx = (j * MI + i) * 2 + g;              // byte of the thread's 16 mask bytes
      if constexpr (BITS_IN) {
        // 1 bit per element, written by the forward layer of the same tile: dword e holds elements 2e (low half), 2e + 1
        const int mb = (int)(bitsv[bidx >> 2] >> ((bidx & 3) * 8));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned lo16 = (unsigned)__builtin_amdgcn_sbfe(mb, 2 * e, 1);
          const unsigned hi16 = (unsigned)__builtin_amdgcn_sbfe(mb, 2 * e + 1, 1);
          w[e] &= (lo16 & 0xffffu) | (hi16 & 0xffff0000u);
        }
      } else {
        unsigned byte = PN_DBG == 3 ? 0u : pn_nonzero_byte(w[0], w[1], w[2], w[3]);
        MNR_GPU_ONLY(asm volatile("" : "+v"(byte)));      // folded now: eight flag registers per block are not kept for later
        mout[bidx >> 2] |= byte << ((bidx & 3) * 8);
      }
      if constexpr (PN_DBG == 1 || (PN_DBG == 6 && true)) {
        if (PN_DBG == 1 || g == 1) {
          MNR_GPU_ONLY(asm volatile("" ::"v"(w)));
        } else {
          *(pn_u32x4*)(ec_p + (j * 2 + g) * 1024) = w;
        }
      } else {
        *(pn_u32x4*)(ec_p + (j * 2 + g) * 1024) = w;
      }
      MNR_SIM_HOOK(hipsim::vm_store());
    }
    if (j == NJ - 1) ec_p += ec_istride;
    if constexpr (!BITS_IN) {
      if (j == NJ - 1 && i == MI - 1 && ec_bits) {
        *(pn_u32x4*)(ec_bits + (unsigned)(lo * 16)) = mout;
        MNR_SIM_HOOK(hipsim::vm_store());
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // One K-tile.  gk: global K-tile counter of this workgroup (stage buffer gk & 3).  ZERO: first K-tile of an output tile
  // (its first k-step's MFMAs take C = 0); EPI: with the previous tile's epilogue interleaved block by block; NEXT: another
  // K-tile follows (counted wait with AHEAD younger operations in flight / barrier / its first fragments in the last k-step);
  // ISSUE: the pieces of K-tile gk + 3 = (it1, it2, itb, ik0) go out between the MFMAs; XTRA: then the extras of the next tile.
  auto ktile = [&](int gk, auto zero_c, auto epi_c, auto next_c, auto issue_c, auto ahead_c, auto xtra_c, const bf16* it1,
                   const bf16* it2, const bf16* itb, int ik0, int xm, int xn, int xpar) {
    constexpr bool ZERO = decltype(zero_c)::value, EPI = decltype(epi_c)::value, NEXT = decltype(next_c)::value;
    constexpr bool ISSUE = decltype(issue_c)::value, XTRA = decltype(xtra_c)::value;
    constexpr int AHEAD = decltype(ahead_c)::value;
    const char* st = smem + (gk & (STAGES - 1)) * STAGE_BYTES;
    const char* st_next = smem + ((gk + 1) & (STAGES - 1)) * STAGE_BYTES;
    const int ibuf = (gk + STAGES - 1) & (STAGES - 1);
    Src sr = {};
    if constexpr (ISSUE) sr = resolve(it1, it2, itb, ik0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks < KS - 1 || NEXT) {
#pragma unroll
        for (int i = 0; i < MI; ++i) nt_launder(fa[ks & 1][i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) nt_launder(fb[ks & 1][j]);
      }
      // (behind an interleaved epilogue the next k-step's fragments are requested after it: 24 registers less under it)
      constexpr bool LATE = ZERO && EPI;
      if (ks < KS - 1) {
        if (!(LATE && ks == 0)) read_frags(st, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
      } else if (NEXT) {
        nt_wait_vmcnt<AHEAD>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(st_next, 0, fa[0], fb[0]);
      }
      constexpr int NM = MI * NJ;
#pragma unroll
      for (int q = 0; q < NM; ++q) {
        // the first k-step of an output tile walks the blocks i-major (the epilogue's address walk), the others j-major
        const bool zstep = ZERO && ks == 0;
        const int j = zstep ? q % NJ : q / MI, i = zstep ? q / NJ : q % MI;
        if constexpr (PN_DBG == 4) {
          MNR_GPU_ONLY(asm volatile("" ::"v"(fb[ks & 1][j]), "v"(fa[ks & 1][i])));
        } else if (zstep) {
          if constexpr (EPI) epi_block(j, i);
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.0f;
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks & 1][j], fa[ks & 1][i], z, 0, 0, 0);
        } else {
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks & 1][j], fa[ks & 1][i], acc[j][i], 0, 0, 0);
        }
        if (ISSUE && q % (NM / PPK) == 1) {
          __builtin_amdgcn_sched_barrier(0);
          stage_piece(ibuf, sr, ks * PPK + q / (NM / PPK));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (LATE && ks == 0) read_frags(st, 1, fa[1], fb[1]);
      if (XTRA && ks == KS - 1) {
        __builtin_amdgcn_sched_barrier(0);
        stage_extra(xm, xn, xpar);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  typedef std::true_type T;
  typedef std::false_type F;
#define PN_I(n) std::integral_constant<int, (n)>()

  // Prologue: the first tile's extras and K-tiles 0 .. 2 in flight, K-tile 0 landed, its first fragments requested.
  int par = 0;
  auto amap = [&](int m) { return PN_DBG == 7 ? (m & 7) : PN_DBG == 8 ? (m & 63) : PN_DBG == 9 ? (m & 511) : PN_DBG == 10 ? (m & 1023) : PN_DBG == 11 ? (m & 127) : PN_DBG == 12 ? (m & 255) : m; };
  const bf16* a1t = A1 + (int64_t)amap(cm) * BM * lda1;
  const bf16* a2t = A2 + (int64_t)cm * BM * lda2;
  const bf16* bt = Bt + (int64_t)cn * BN * ldb;
  stage_extra(cm, cn, par);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) {
    const Src sr = resolve(a1t, a2t, bt, s * BK);
#pragma unroll
    for (int i = 0; i < LPS; ++i) stage_piece(s, sr, i);
  }
  nt_wait_vmcnt<(STAGES - 2) * LPS>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(smem, 0, fa[0], fb[0]);

  // The walk.  One loop body per output tile, entered at its K-tile 2: steady K-tiles, the last three (which request the next
  // tile's first three and its extras), then the next tile's K-tiles 0 and 1 with this tile's epilogue interleaved.  A workgroup's
  // last tile requests its own first K-tiles once more instead of a next tile's (32 KiB x 3 of L2 reads per workgroup buy a
  // loop without a second set of tail flavours: with them hipcc stopped accumulating in place and spilled whole accumulator
  // blocks).
  constexpr int STD = (STAGES - 3) * LPS + (KS - 1) * PPK;           // younger operations behind a K-tile in steady state
  int gk = 0;
  ktile(gk, T(), F(), T(), T(), PN_I(STD), F(), a1t, a2t, bt, 3 * BK, 0, 0, 0);
  ktile(gk + 1, F(), F(), T(), T(), PN_I(STD), F(), a1t, a2t, bt, 4 * BK, 0, 0, 0);
  for (;;) {
    int nv = v + gstep, nm = 0, nn = 0;
    while (nv < vtot && !decode(nv, nm, nn)) nv += gstep;
    const bool has_next = nv < vtot;
    if (!has_next) {
      nm = cm;
      nn = cn;
    }
    const bf16* n1t = A1 + (int64_t)amap(nm) * BM * lda1;
    const bf16* n2t = A2 + (int64_t)nm * BM * lda2;
    const bf16* nbt = Bt + (int64_t)nn * BN * ldb;
    int kt = 2;
    if constexpr (PN_DBG == 5) {
      if (gk > 0) {
        ktile(gk + 2, F(), F(), T(), T(), PN_I(STD + PN_NST), F(), a1t, a2t, bt, 5 * BK, 0, 0, 0);
        ktile(gk + 3, F(), F(), T(), T(), PN_I(STD + PN_NST), F(), a1t, a2t, bt, 6 * BK, 0, 0, 0);
        kt = 4;
      }
    }
    for (; kt < nk - 3; ++kt) ktile(gk + kt, F(), F(), T(), T(), PN_I(STD), F(), a1t, a2t, bt, (kt + 3) * BK, 0, 0, 0);
    // nk-3: pieces of (next, 0), then the next tile's extras; nk-2: (next, 1), the extras among the younger ones; nk-1: (next, 2)
    ktile(gk + kt, F(), F(), T(), T(), PN_I(STD), T(), n1t, n2t, nbt, 0, nm, nn, par ^ 1);
    ktile(gk + kt + 1, F(), F(), T(), T(), PN_I(STD + 1), F(), n1t, n2t, nbt, BK, 0, 0, 0);
    ktile(gk + kt + 2, F(), F(), T(), T(), PN_I(STD), F(), n1t, n2t, nbt, 2 * BK, 0, 0, 0);
    gk += nk;
    epi_set(cm, cn, par);
    if (!has_next) break;
    v = nv;
    cm = nm;
    cn = nn;
    a1t = n1t;
    a2t = n2t;
    bt = nbt;
    par ^= 1;
    // K-tiles 0, 1 of the next tile: the epilogue's stores are younger than the pieces their waits are for
    ktile(gk, T(), T(), T(), T(), PN_I(STD + PN_NST), F(), a1t, a2t, bt, 3 * BK, 0, 0, 0);
    ktile(gk + 1, F(), F(), T(), T(), PN_I(STD + PN_NST), F(), a1t, a2t, bt, 4 * BK, 0, 0, 0);
  }
  // the last tile's epilogue on its own (the re-requested K-tiles land in stage buffers nobody reads again)
  if constexpr (PN_DBG != 4) {
#pragma unroll
    for (int q = 0; q < MI * NJ; ++q) epi_block(q % NJ, q / NJ);
  }
  nt_wait_vmcnt<0>();
#undef PN_I
}

// The caller alternates mnr_gemm_nt_args.walk_descending between consecutive layers (same-box A/B 31.21 / 31.21 -> 31.12 / 31.12 ms
// per step at 360.gin: the rows a layer reads first are the ones the previous layer wrote last).  A per-call argument: round 4
// toggled a process-global here, which two streams raced on.
static int g_panel_max_wgs = 0;                           // tests: at most this many workgroups (0: one per CU)
extern "C" int mnr_gemm_nt_panel_set_max_wgs(int n) {
  g_panel_max_wgs = n;
  return MNR_OK;
}

template <bool A1_PANEL, bool BITS_IN>
static int panel_launch_t(const mnr_gemm_nt_args* a, int64_t grid, int64_t vtotal, void* stream) {
  constexpr int lds = PN_EXTRA_OFF + 2 * (BITS_IN ? 8192 : 2048);
  static unsigned long long attr_set = 0;                 // per device (mnr_attr_needed)
  if (mnr_attr_needed(&attr_set))
    (void)hipFuncSetAttribute((const void*)gemm_nt_panel_kernel<A1_PANEL, BITS_IN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((gemm_nt_panel_kernel<A1_PANEL, BITS_IN>), dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, *a,
                     (long long)vtotal, a->walk_descending ? 1 : 0);
  MNR_CHECK_LAUNCH();
  return MNR_OK;
}

// mnr_gemm_nt_bf16 with c_layout = MNR_LAYOUT_PANEL (called by the dispatcher in gemm.hip after its common checks)
int mnr_gemm_nt_panel_launch(const mnr_gemm_nt_args* a, void* stream) {
  const int K = a->K1 + a->K2;
  MNR_CHECK_ARG(a->M % PN_BM == 0 && a->N % PN_BN == 0, "mnr_gemm_nt_bf16 (panel result): M=%lld, N=%d must be multiples of 256",
                (long long)a->M, a->N);
  MNR_CHECK_ARG(a->K1 % PN_BK == 0 && a->K2 % PN_BK == 0 && K / PN_BK >= PN_MIN_NK,
                "mnr_gemm_nt_bf16 (panel result): K1=%d, K2=%d must be multiples of 32 with K1 + K2 >= %d", a->K1, a->K2, PN_MIN_NK * PN_BK);
  MNR_CHECK_ARG(a->Cb && a->nb == a->N && !a->Cf && !a->mask && a->bits_row_mod == 0 && ((uintptr_t)a->Cb % 16) == 0,
                "mnr_gemm_nt_bf16 (panel result): needs a full-width 16-byte-aligned bf16 result, no fp32 side output, no bf16 mask, no row modulus");
  MNR_CHECK_ARG(a->a1_layout == MNR_LAYOUT_ROWMAJOR || (a->a1_layout == MNR_LAYOUT_PANEL && a->lda1 == a->K1 && a->K1 % 16 == 0),
                "mnr_gemm_nt_bf16: a panel-layout A1 needs lda1 == K1");
  MNR_CHECK_ARG(((uintptr_t)a->A1 % 16) == 0 && ((uintptr_t)a->Bt % 16) == 0 && (a->K2 == 0 || ((uintptr_t)a->A2 % 16) == 0),
                "mnr_gemm_nt_bf16 (panel result): operands must be 16-byte aligned");
  if (a->mask_bits_in) {
    MNR_CHECK_ARG(!a->bias && !a->relu && !a->mask_bits_out && ((uintptr_t)a->mask_bits_in % 16) == 0,
                  "mnr_gemm_nt_bf16 (panel result): a dX layer takes no bias / ReLU / mask output, tile-order bits 16-byte aligned");
  } else {
    MNR_CHECK_ARG(a->bias && a->n_bias == a->N && a->relu && ((uintptr_t)a->bias % 4) == 0,
                  "mnr_gemm_nt_bf16 (panel result): a forward layer needs a full bias row (n_bias == N) and ReLU");
    MNR_CHECK_ARG(!a->mask_bits_out || ((uintptr_t)a->mask_bits_out % 16) == 0, "mnr_gemm_nt_bf16 (panel result): tile-order bits 16-byte aligned");
  }
  const int nt = a->N / PN_BN;
  const int64_t mt = a->M / PN_BM;
  const int64_t vtotal = (mt + 7) / 8 * 8 * nt;
  MNR_CHECK_ARG(vtotal < (1ll << 31), "mnr_gemm_nt_bf16: grid too large");
  int64_t cap = g_panel_max_wgs > 0 ? g_panel_max_wgs : mnr_cu_count();
  if (a->max_wgs > 0 && a->max_wgs < cap) cap = a->max_wgs;
  cap = cap / 8 * 8;
  if (cap < 8) cap = 8;
  const int64_t grid = vtotal < cap ? vtotal : cap;
  const bool pan = a->a1_layout == MNR_LAYOUT_PANEL;
  if (a->mask_bits_in) return pan ? panel_launch_t<true, true>(a, grid, vtotal, stream) : panel_launch_t<false, true>(a, grid, vtotal, stream);
  return pan ? panel_launch_t<true, false>(a, grid, vtotal, stream) : panel_launch_t<false, false>(a, grid, vtotal, stream);
}
