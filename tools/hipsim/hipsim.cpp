// hipsim runtime: fibers, the workgroup scheduler and the gfx950 instructions the kernels use (see hip/hip_runtime.h).
//
// Execution model.  launch() runs the workgroups of a grid one after the other on the calling OS thread.  Inside a
// workgroup every thread is a ucontext fiber that runs until it blocks: at a workgroup barrier, at a wave-collective
// instruction (MFMA, transpose read, DPP, shuffles: the 64 lanes deposit their operands, the last one to arrive
// evaluates the instruction for the whole wave) or at its end.  Lanes that have returned from the kernel no longer
// count for barriers (as terminated waves on the hardware) and contribute zeros to collectives.
//
// Instruction semantics restated here (each is exercised by the GPU-validated kernels, whose simulated results must
// match the fp32 reference in tests/test_sim_gemm.py; that is the check on this file):
//   v_mfma_f32_32x32x16_bf16  D[i][j] = C[i][j] + sum_k A[i][k] B[k][j];  lane l holds A[l%32][8*(l/32)..+7],
//                             B[8*(l/32)..+7][l%32]; register r of lane l is D[8*(r/4) + 4*(l/32) + r%4][l%32]
//   global_load_lds (16 B)    LDS[base + offset + 16*lane] <- global[per-lane address]; base is wave-uniform (M0)
//   ds_read_b64_tr_b16        per 16-lane group: lane p supplies the address of 4 consecutive 16-bit elements,
//                             row p/4, columns 4*(p%4)..+3 of a [4][16] block; lane c receives column c (4 rows)
//   DPP quad_perm             lane l reads lane (l & ~3) | perm[l & 3]
//   v_permlane32_swap_b32     lanes 32-63 of vdst are exchanged with lanes 0-31 of src, the other two halves stay
#include <stdarg.h>
#include <stdio.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <deque>
#include <random>
#include <vector>

#include "hip/hip_runtime.h"

#define HIPSIM_LDS_BYTES (160 * 1024)
#define HIPSIM_STACK_BYTES (256 * 1024)
#define HIPSIM_MAX_THREADS 1024

// Dynamic LDS, under the names the kernels declare it with (`extern __shared__ ... smem[]` / `lds[]`).
thread_local __attribute__((aligned(64))) char smem[HIPSIM_LDS_BYTES + 64];
thread_local __attribute__((aligned(64))) float lds[HIPSIM_LDS_BYTES / 4 + 16];

namespace hipsim {

ThreadCtx* cur = nullptr;

namespace {

enum State { READY = 0, AT_BARRIER, AT_WAVE, DONE };

struct PendingDma {
  char* dst;
  char data[16];
  int size;
};

struct Fiber {
  ucontext_t ctx;
  ThreadCtx tc;
  State state;
  std::deque<PendingDma> dma;
  unsigned dma_seq;
};

struct Wave;
typedef void (*WaveOp)(Wave&);

struct Wave {
  int live = 0, arrived = 0;
  WaveOp op = nullptr;
  const char* op_name = "";
  bool present[64];
  alignas(64) char in[64][128];
  alignas(64) char out[64][64];
  // wave-uniformity check of the LDS-DMA base: entry s % 256 holds the base lane-first-to-issue used for DMA number s
  struct {
    unsigned seq;
    const void* base;
    bool valid;
  } dma_base[256];
};

struct Block {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int live = 0, at_barrier = 0;
  size_t dyn_lds = 0;
};

Block g_blk;
Fiber* g_fiber = nullptr;
ucontext_t g_sched;
const std::function<void()>* g_body = nullptr;
char* g_stacks = nullptr;
bool g_failed = false;
char g_err[512] = "";
int g_dma_late = -1;
long g_order = -2;                                   // 0 forward, -1 reverse, > 0 shuffle seed
unsigned long long g_clock = 0;
unsigned long long g_n_switch = 0, g_n_mfma = 0, g_n_dma = 0, g_n_barrier = 0;

void read_env() {
  if (g_dma_late < 0) {
    const char* e = getenv("HIPSIM_DMA");
    g_dma_late = (e && !strcmp(e, "late")) ? 1 : 0;
  }
  if (g_order == -2) {
    const char* e = getenv("HIPSIM_ORDER");
    g_order = !e ? 0 : !strcmp(e, "reverse") ? -1 : atol(e);
  }
}

void yield() {
  ++g_n_switch;
  swapcontext(&g_fiber->ctx, &g_sched);
}

__attribute__((format(printf, 1, 2))) void fail(const char* fmt, ...) {
  if (!g_failed) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    g_failed = true;
  }
  if (g_fiber) {                                     // abandon this fiber; the scheduler stops the launch
    g_fiber->state = DONE;
    yield();
  }
}

void land(Fiber& f, size_t keep) {
  while (f.dma.size() > keep) {
    const PendingDma& d = f.dma.front();
    if (d.size) memcpy(d.dst, d.data, d.size);
    f.dma.pop_front();
  }
}

void release_barrier() {
  for (Fiber& f : g_blk.fibers)
    if (f.state == AT_BARRIER) f.state = READY;
  g_blk.at_barrier = 0;
}

void resolve(Wave& w) {
  w.op(w);
  w.arrived = 0;
  const int wi = (int)(&w - g_blk.waves.data());
  for (int l = 0; l < 64; ++l) {
    w.present[l] = false;
    const size_t t = (size_t)wi * 64 + l;
    if (t < g_blk.fibers.size() && g_blk.fibers[t].state == AT_WAVE) g_blk.fibers[t].state = READY;
  }
}

void fiber_exit() {
  Fiber& f = *g_fiber;
  land(f, 0);
  f.state = DONE;
  Wave& w = g_blk.waves[f.tc.wave];
  --w.live;
  --g_blk.live;
  if (w.live > 0 && w.arrived == w.live) resolve(w);
  if (g_blk.live > 0 && g_blk.at_barrier == g_blk.live) release_barrier();
}

void trampoline() {
  (*g_body)();
  fiber_exit();
  yield();
  abort();                                           // a finished fiber is never resumed
}

void wave_op(WaveOp op, const char* name, const void* in, int in_bytes, void* out, int out_bytes) {
  Fiber& f = *g_fiber;
  Wave& w = g_blk.waves[f.tc.wave];
  if (w.arrived == 0) {
    w.op = op;
    w.op_name = name;
  } else if (w.op != op) {
    fail("wave %d of block %u: lanes diverge at a wave-wide instruction (%s vs %s)", f.tc.wave, f.tc.bid3.x, w.op_name, name);
    return;
  }
  memcpy(w.in[f.tc.lane], in, in_bytes);
  w.present[f.tc.lane] = true;
  ++w.arrived;
  if (w.arrived == w.live) {
    resolve(w);
  } else {
    f.state = AT_WAVE;
    yield();
  }
  memcpy(out, w.out[f.tc.lane], out_bytes);
}

float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

void op_mfma(Wave& w) {
  ++g_n_mfma;
  float A[32][16], B[16][32];
  for (int l = 0; l < 64; ++l) {
    uint16_t a[8] = {0}, b[8] = {0};
    if (w.present[l]) {
      memcpy(a, w.in[l], 16);
      memcpy(b, w.in[l] + 16, 16);
    }
    for (int e = 0; e < 8; ++e) {
      A[l & 31][8 * (l >> 5) + e] = bf16_to_f32(a[e]);
      B[8 * (l >> 5) + e][l & 31] = bf16_to_f32(b[e]);
    }
  }
  for (int l = 0; l < 64; ++l) {
    float c[16] = {0};
    if (w.present[l]) memcpy(c, w.in[l] + 32, 64);
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
      const int i = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
      float s = 0.0f;
      for (int k = 0; k < 16; ++k) s += A[i][k] * B[k][j];
      c[r] += s;
    }
    memcpy(w.out[l], c, 64);
  }
}

bool in_lds(const void* p, size_t bytes) {
  const char* c = (const char*)p;
  if (c >= smem && c + bytes <= smem + g_blk.dyn_lds) return true;
  const char* l0 = (const char*)lds;
  return c >= l0 && c + bytes <= l0 + g_blk.dyn_lds;
}

void op_ds_read_tr16_b64(Wave& w) {
  for (int g = 0; g < 4; ++g) {
    uint16_t blk[4][16];
    memset(blk, 0, sizeof(blk));
    for (int p = 0; p < 16; ++p) {
      const int l = g * 16 + p;
      if (!w.present[l]) continue;
      const void* addr;
      memcpy(&addr, w.in[l], sizeof(addr));
      memcpy(&blk[p >> 2][4 * (p & 3)], addr, 8);
    }
    for (int c = 0; c < 16; ++c) {
      uint16_t v[4] = {blk[0][c], blk[1][c], blk[2][c], blk[3][c]};
      memcpy(w.out[g * 16 + c], v, 8);
    }
  }
}

void op_dpp(Wave& w) {
  for (int l = 0; l < 64; ++l) {
    int in[3] = {0, 0, 0};                           // old, src, ctrl
    memcpy(in, w.in[l], 12);
    const int ctrl = in[2];
    const int srcl = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    int v = 0;
    if (w.present[srcl]) memcpy(&v, w.in[srcl] + 4, 4);
    memcpy(w.out[l], &v, 4);
  }
}

void op_readfirstlane(Wave& w) {
  int v = 0;
  for (int l = 0; l < 64; ++l)
    if (w.present[l]) {
      memcpy(&v, w.in[l], 4);
      break;
    }
  for (int l = 0; l < 64; ++l) memcpy(w.out[l], &v, 4);
}

void op_permlane32_swap(Wave& w) {
  // in[l] = {vdst, src}; out[l] = {new vdst, new src}: vdst[32 + k] <-> src[k]
  for (int l = 0; l < 64; ++l) {
    unsigned mine[2] = {0, 0}, other[2] = {0, 0};
    if (w.present[l]) memcpy(mine, w.in[l], 8);
    const int o = l ^ 32;
    if (w.present[o]) memcpy(other, w.in[o], 8);
    unsigned out[2];
    if (l < 32) {
      out[0] = mine[0];            // lower half of vdst stays
      out[1] = other[0];           // lower half of src <- upper half of vdst
    } else {
      out[0] = other[1];           // upper half of vdst <- lower half of src
      out[1] = mine[1];            // upper half of src stays
    }
    memcpy(w.out[l], out, 8);
  }
}

void op_shfl(Wave& w) {
  for (int l = 0; l < 64; ++l) {
    int in[2] = {0, 0};                              // value bits, source lane
    memcpy(in, w.in[l], 8);
    int src = in[1] & 63;
    int v = in[0];                                   // an inactive source returns the lane's own value
    if (w.present[src]) memcpy(&v, w.in[src], 4);
    memcpy(w.out[l], &v, 4);
  }
}

void op_ballot(Wave& w) {
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) {
    int p = 0;
    if (w.present[l]) memcpy(&p, w.in[l], 4);
    if (p) m |= 1ull << l;
  }
  for (int l = 0; l < 64; ++l) memcpy(w.out[l], &m, 8);
}

}  // namespace

void barrier() {
  ++g_n_barrier;
  Fiber& f = *g_fiber;
  ++g_blk.at_barrier;
  if (g_blk.at_barrier == g_blk.live) {
    release_barrier();
  } else {
    f.state = AT_BARRIER;
    yield();
  }
}

void wait_vmcnt(int n) { land(*g_fiber, (size_t)n); }

// A global store of a kernel that counts its vector-memory queue by hand: on gfx950 stores retire through the same in-order
// vmcnt as loads and LDS-DMA, so a counted wait behind them has to allow for them (gemm_blk.hip).
void vm_store() {
  if (!g_dma_late) return;
  PendingDma d;
  d.dst = nullptr;
  d.size = 0;
  g_fiber->dma.push_back(d);
}

void syncthreads() {
  wait_vmcnt(0);
  barrier();
}

void global_load_lds(const void* gptr, void* lds_wave_base, int size, int offset) {
  ++g_n_dma;
  Fiber& f = *g_fiber;
  Wave& w = g_blk.waves[f.tc.wave];
  if (size != 16 && size != 4) {
    fail("global_load_lds: size %d", size);
    return;
  }
  auto& slot = w.dma_base[f.dma_seq & 255];
  if (!slot.valid || slot.seq != f.dma_seq) {
    slot.valid = true;
    slot.seq = f.dma_seq;
    slot.base = lds_wave_base;
  } else if (slot.base != lds_wave_base) {
    fail("global_load_lds #%u of wave %d, block %u: the LDS base differs between lanes (it is taken from M0, i.e. must be wave-uniform)",
         f.dma_seq, f.tc.wave, f.tc.bid3.x);
    return;
  }
  ++f.dma_seq;
  char* dst = (char*)lds_wave_base + offset + f.tc.lane * size;
  if (!in_lds(dst, size)) {
    fail("global_load_lds of wave %d lane %d, block %u: destination outside the %zu bytes of dynamic LDS", f.tc.wave, f.tc.lane,
         f.tc.bid3.x, g_blk.dyn_lds);
    return;
  }
  if (g_dma_late) {
    PendingDma d;
    d.dst = dst;
    d.size = size;
    memcpy(d.data, gptr, size);
    f.dma.push_back(d);
  } else {
    memcpy(dst, gptr, size);
  }
}

s16x4_t ds_read_tr16_b64(const void* lds_ptr) {
  if (!in_lds(lds_ptr, 8)) {
    fail("ds_read_b64_tr_b16 of wave %d lane %d: address outside dynamic LDS", g_fiber->tc.wave, g_fiber->tc.lane);
    return s16x4_t{0, 0, 0, 0};
  }
  s16x4_t out;
  wave_op(op_ds_read_tr16_b64, "ds_read_b64_tr_b16", &lds_ptr, sizeof(lds_ptr), &out, 8);
  return out;
}

f32x16_t mfma_f32_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  char in[96];
  memcpy(in, &a, 16);
  memcpy(in + 16, &b, 16);
  memcpy(in + 32, &c, 64);
  f32x16_t d;
  wave_op(op_mfma, "v_mfma_f32_32x32x16_bf16", in, 96, &d, 64);
  return d;
}

int update_dpp(int old, int src, int dpp_ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  if (dpp_ctrl < 0 || dpp_ctrl > 0xff || row_mask != 0xf || bank_mask != 0xf) {
    fail("update_dpp: only quad_perm with full row/bank masks is restated (ctrl 0x%x)", dpp_ctrl);
    return old;
  }
  (void)bound_ctrl;
  int in[3] = {old, src, dpp_ctrl}, out = 0;
  wave_op(op_dpp, "dpp quad_perm", in, 12, &out, 4);
  return out;
}

u32x2_t permlane32_swap(unsigned vdst, unsigned src) {
  unsigned in[2] = {vdst, src};
  u32x2_t out = {0u, 0u};
  wave_op(op_permlane32_swap, "v_permlane32_swap", in, 8, &out, 8);
  return out;
}

int readfirstlane(int v) {
  int out = 0;
  wave_op(op_readfirstlane, "v_readfirstlane", &v, 4, &out, 4);
  return out;
}

int shfl_i(int v, int src_lane, int width) {
  if (width != 64) {
    const int l = g_fiber->tc.lane;
    src_lane = (l / width) * width + (src_lane % width);
  }
  int in[2] = {v, src_lane}, out = 0;
  wave_op(op_shfl, "shuffle", in, 8, &out, 4);
  return out;
}

float shfl(float v, int src_lane, int width) {
  return __builtin_bit_cast(float, shfl_i(__builtin_bit_cast(int, v), src_lane, width));
}

unsigned long long ballot(int pred) {
  unsigned long long out = 0;
  wave_op(op_ballot, "ballot", &pred, 4, &out, 8);
  return out;
}

unsigned long long clock64() { return ++g_clock; }

void launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes, const std::function<void()>& body) {
  read_env();
  if (g_failed) return;
  const size_t nthreads = (size_t)block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > HIPSIM_MAX_THREADS) {
    fail("launch: %zu threads per workgroup", nthreads);
    return;
  }
  if (dynamic_lds_bytes > HIPSIM_LDS_BYTES) {
    fail("launch: %zu bytes of dynamic LDS (the CU has %d)", dynamic_lds_bytes, HIPSIM_LDS_BYTES);
    return;
  }
  if (!g_stacks) {
    g_stacks = (char*)mmap(nullptr, (size_t)HIPSIM_MAX_THREADS * HIPSIM_STACK_BYTES, PROT_READ | PROT_WRITE,
                           MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_stacks == MAP_FAILED) {
      g_stacks = nullptr;
      fail("launch: cannot map fiber stacks");
      return;
    }
  }
  const size_t nwaves = (nthreads + 63) / 64;
  std::vector<size_t> order(nthreads);
  for (size_t i = 0; i < nthreads; ++i) order[i] = g_order == -1 ? nthreads - 1 - i : i;
  if (g_order > 0) {
    // shuffle whole waves and the lanes inside them separately: lanes of a wave stay together in time
    std::mt19937 rng((unsigned)g_order);
    std::vector<size_t> wv(nwaves);
    for (size_t i = 0; i < nwaves; ++i) wv[i] = i;
    std::shuffle(wv.begin(), wv.end(), rng);
    size_t o = 0;
    for (size_t wi : wv) {
      std::vector<size_t> ln;
      for (size_t l = 0; l < 64 && wi * 64 + l < nthreads; ++l) ln.push_back(wi * 64 + l);
      std::shuffle(ln.begin(), ln.end(), rng);
      for (size_t t : ln) order[o++] = t;
    }
  }
  g_body = &body;
  for (unsigned bz = 0; bz < grid.z && !g_failed; ++bz)
    for (unsigned by = 0; by < grid.y && !g_failed; ++by)
      for (unsigned bx = 0; bx < grid.x && !g_failed; ++bx) {
        Block& b = g_blk;
        b.fibers.resize(nthreads);
        b.waves.assign(nwaves, Wave());
        b.live = (int)nthreads;
        b.at_barrier = 0;
        b.dyn_lds = dynamic_lds_bytes;
        // uninitialised LDS reads as bf16 / fp32 NaNs
        memset(smem, 0xff, dynamic_lds_bytes);
        memset(lds, 0xff, dynamic_lds_bytes);
        for (size_t t = 0; t < nthreads; ++t) {
          Fiber& f = b.fibers[t];
          f.state = READY;
          f.dma.clear();
          f.dma_seq = 0;
          f.tc.linear = (int)t;
          f.tc.lane = (int)(t & 63);
          f.tc.wave = (int)(t >> 6);
          f.tc.tid3 = dim3((unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y)));
          f.tc.bid3 = dim3(bx, by, bz);
          f.tc.bdim3 = block;
          f.tc.gdim3 = grid;
          b.waves[t >> 6].live++;
          for (int l = 0; l < 64; ++l) b.waves[t >> 6].present[l] = false;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = g_stacks + t * HIPSIM_STACK_BYTES;
          f.ctx.uc_stack.ss_size = HIPSIM_STACK_BYTES;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, trampoline, 0);
        }
        while (b.live > 0 && !g_failed) {
          bool any = false;
          for (size_t t : order) {
            Fiber& f = b.fibers[t];
            if (f.state != READY) continue;
            any = true;
            g_fiber = &f;
            cur = &f.tc;
            swapcontext(&g_sched, &f.ctx);
            g_fiber = nullptr;
            cur = nullptr;
            if (g_failed) break;
          }
          if (!any && b.live > 0 && !g_failed) {
            int nb = 0, nw = 0;
            for (Fiber& f : b.fibers) {
              nb += f.state == AT_BARRIER;
              nw += f.state == AT_WAVE;
            }
            fail("block (%u,%u,%u): deadlock, %d threads at a barrier, %d inside a wave-wide instruction, %d live", bx, by, bz, nb,
                 nw, b.live);
          }
        }
      }
  g_body = nullptr;
}

}  // namespace hipsim

extern "C" {

// 0 when every launch since the last hipsim_reset() completed; else the first failure's message is in hipsim_error().
int hipsim_failed(void) { return hipsim::g_failed ? 1 : 0; }
const char* hipsim_error(void) { return hipsim::g_err; }

// dma_late: 1 = LDS-DMA lands at the covering wait, 0 = at issue; order: 0 forward, -1 reverse, > 0 shuffle seed.
void hipsim_reset(int dma_late, long order) {
  hipsim::g_failed = false;
  hipsim::g_err[0] = 0;
  hipsim::g_dma_late = dma_late ? 1 : 0;
  hipsim::g_order = order;
  hipsim::g_n_switch = hipsim::g_n_mfma = hipsim::g_n_dma = hipsim::g_n_barrier = 0;
}

void hipsim_stats(unsigned long long* out4) {
  out4[0] = hipsim::g_n_switch;
  out4[1] = hipsim::g_n_mfma;
  out4[2] = hipsim::g_n_dma;
  out4[3] = hipsim::g_n_barrier;
}
}
