// Infinity-Cache warmer (see warm.h).
#include "warm.h"

#include <string.h>

namespace {

struct WarmArgs {
  const char* base[2];
  long long unit_bytes[2];
  long long stream_stride[2];
  const unsigned* slot;
  unsigned id;
  int units, nstreams, rev, ahead, ntensors, stride;
};

constexpr int WARM_WAVES = 256;
constexpr int WARM_SLOTS = 64;                            // ring of 256-byte progress slots, one per launch in flight

}  // namespace

// One wave per workgroup.  Unit u may be fetched once the GEMM has started unit u - ahead (or has not started at all and
// u < ahead).  Every lane reads ONE dword of its own 128-byte line: the line is filled from HBM through the Infinity Cache (and
// the L2 of whichever XCD this wave runs on), 4 of its 128 bytes travel on to the CU.  All loads target the same register (they
// return in order); it stays tied to the asm statements until the final wait, so the compiler cannot hand it to anything else
// while loads are in flight.
template <int POL>
__global__ __launch_bounds__(64) void mnr_warm_kernel(WarmArgs a) {
#ifndef MNR_HIPSIM
  if constexpr (POL == 9) return;                          // (probe: the orchestration alone)
  const int w = (int)blockIdx.x;
  const int lane = (int)threadIdx.x;
  const int sh = a.stride == 32 ? 11 : a.stride == 64 ? 12 : 13;        // log2 of the bytes one instruction covers
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
  unsigned sink = 0;
  int cur = 0, fetched = 0, total_polls = 0, reason = 0, skipped = 0;
  unsigned* const stats = const_cast<unsigned*>(a.slot) + 64;
  for (int u = 0; u < a.units; ++u) {
    int polls = 0;
    while (u >= cur + a.ahead) {
      const unsigned v = __hip_atomic_load(a.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((v >> 12) == a.id) {
        const int c = (int)(v & 0xfffu);
        if (c == (int)MNR_WARM_DONE) {                     // the GEMM is over
          reason = 1;
          goto out;
        }
        cur = c;
      }
      if (u < cur + a.ahead) break;
      // bounded: 20000 polls of ~0.5 us, 6 ms of wall time
      ++total_polls;
      if (++polls > 20000 || __builtin_amdgcn_s_memrealtime() - t0 > 600000ull) {
        reason = 2;
        goto out;
      }
      __builtin_amdgcn_s_sleep(16);
    }
    if (u < cur) {                                         // behind the GEMM: what it has already read is of no use
      skipped += cur - u;
      u = cur - 1;
      continue;
    }
    const long long uu = a.rev ? a.units - 1 - u : u;
    for (int t = 0; t < a.ntensors; ++t) {
      const int n = (int)(a.unit_bytes[t] >> sh);          // instructions per stream of this unit
      for (int s = 0; s < a.nstreams; ++s) {
        const char* p = a.base[t] + s * a.stream_stride[t] + uu * a.unit_bytes[t] + lane * a.stride;
        for (int i = w; i < n; i += WARM_WAVES) {
          const char* q = p + ((long long)i << sh);
          // cache policy of the warming loads (MNR_WARM_POLICY): the lines are wanted in the memory-side Infinity Cache, not in the L2
          if constexpr (POL == 0) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(q) : "memory");
          if constexpr (POL == 1) asm volatile("global_load_dword %0, %1, off nt" : "+v"(sink) : "v"(q) : "memory");
          if constexpr (POL == 2) asm volatile("global_load_dword %0, %1, off sc1" : "+v"(sink) : "v"(q) : "memory");
          if constexpr (POL == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "+v"(sink) : "v"(q) : "memory");
          if constexpr (POL == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "+v"(sink) : "v"(q) : "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory");
    ++fetched;
  }
out:
  if (w == 0 && lane == 0) {                              // (debugging aid: what this warmer did, next to its progress word)
    stats[0] = (unsigned)fetched;
    stats[1] = (unsigned)reason;
    stats[2] = (unsigned)total_polls;
    stats[3] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);
    stats[4] = (unsigned)skipped;
    stats[5] = a.id;
  }
  if (sink == 0x7fc12345u && a.ahead < 0) *(volatile unsigned*)a.slot = sink;      // (never: keeps the loads' register observable)
#endif
}

static unsigned* g_warm_slots = nullptr;
// tools: the 64 x 128 dwords of the progress ring (dword 0: progress word, 64..69: the warmer's statistics) copied to the host
extern "C" int mnr_warm_debug(unsigned* host_out) {
#ifndef MNR_HIPSIM
  if (!g_warm_slots) return MNR_ERR_INVALID_ARGUMENT;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, g_warm_slots, WARM_SLOTS * 512, hipMemcpyDeviceToHost) != hipSuccess) return MNR_ERR_HIP;
#endif
  return MNR_OK;
}

static const void* g_warm_decoy = nullptr;
// tools: the warmers read this buffer (>= the operand's size) instead of the operand: what the warming traffic costs by itself
extern "C" int mnr_warm_set_decoy(const void* p) {
  g_warm_decoy = p;
  return MNR_OK;
}

static int g_warm_on = -1, g_warm_ahead = 2;
static long long g_warm_min_bytes = 256ll << 20;
extern "C" int mnr_set_warm(int on, int ahead) {
  g_warm_on = on ? 1 : 0;
  g_warm_min_bytes = on == 2 ? 0 : (256ll << 20);
  if (ahead > 0) g_warm_ahead = ahead;
  return MNR_OK;
}
long long mnr_warm_min_bytes() { return g_warm_min_bytes; }

mnr_warm_ticket mnr_warm_begin(const mnr_warm_tensor* t, int ntensors, int units, int nstreams, int rev, void* stream) {
  mnr_warm_ticket none = {nullptr, 0};
  if (g_warm_on < 0) {
    const char* e = getenv("MNR_WARM");
    g_warm_on = e ? (atoi(e) != 0) : 1;
    const char* a = getenv("MNR_WARM_AHEAD");
    if (a && atoi(a) > 0) g_warm_ahead = atoi(a);
  }
  if (!g_warm_on || ntensors < 1 || ntensors > 2 || units < 2 || units >= (int)MNR_WARM_DONE || nstreams < 1) return none;
  for (int i = 0; i < ntensors; ++i)
    if (!t[i].base || t[i].unit_bytes <= 0 || (t[i].unit_bytes & 8191) || ((uintptr_t)t[i].base & 3)) return none;
  static unsigned next_id = 1;
#ifdef MNR_HIPSIM
  // the simulator runs no warmer; the GEMM's publishing path is exercised against a host slot
  static unsigned host_slots[WARM_SLOTS * 128];
  const unsigned id = next_id++ & 0xfffffu;
  (void)rev;
  (void)stream;
  return mnr_warm_ticket{host_slots + (id % WARM_SLOTS) * 128, id};
#else
  // per process: one side stream, a ring of events, the progress slots (a process drives one GPU: bench.py, train.py, dist.py)
  static hipStream_t ws = nullptr;
  static hipEvent_t evs[16];
  static unsigned* slots = nullptr;
  static int ei = 0, dev0 = -1, broken = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (broken) return none;
  if (!ws) {
    dev0 = dev;
    bool ok = hipStreamCreateWithFlags(&ws, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 16; ++i) ok = hipEventCreateWithFlags(&evs[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&slots, WARM_SLOTS * 512) == hipSuccess && hipMemset(slots, 0, WARM_SLOTS * 512) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      broken = 1;
      ws = nullptr;
      return none;
    }
    g_warm_slots = slots;
  }
  if (dev != dev0) return none;
  const unsigned id = next_id++ & 0xfffffu;
  WarmArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < ntensors; ++i) {
    a.base[i] = (const char*)(g_warm_decoy && i == 0 ? g_warm_decoy : t[i].base);
    a.unit_bytes[i] = t[i].unit_bytes;
    a.stream_stride[i] = t[i].stream_stride;
  }
  a.slot = slots + (id % WARM_SLOTS) * 128;
  a.id = id;
  a.units = units;
  a.nstreams = nstreams;
  a.rev = rev;
  a.ahead = g_warm_ahead;
  a.ntensors = ntensors;
  static int stride = 0;
  if (!stride) {
    const char* e = getenv("MNR_WARM_STRIDE");
    stride = e ? atoi(e) : 128;
    if (stride != 32 && stride != 64) stride = 128;
  }
  a.stride = stride;
  // the warmer starts when everything in front of the GEMM on its stream is done (the operand has been written)
  hipEvent_t ev = evs[ei];
  ei = (ei + 1) & 15;
  if (hipEventRecord(ev, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(ws, ev, 0) != hipSuccess) {
    (void)hipGetLastError();
    return none;
  }
  static int pol = -1;
  if (pol < 0) {
    const char* e = getenv("MNR_WARM_POLICY");
    pol = e ? atoi(e) : 0;
  }
  switch (pol) {
    case 1: hipLaunchKernelGGL(mnr_warm_kernel<1>, dim3(WARM_WAVES), dim3(64), 0, ws, a); break;
    case 2: hipLaunchKernelGGL(mnr_warm_kernel<2>, dim3(WARM_WAVES), dim3(64), 0, ws, a); break;
    case 3: hipLaunchKernelGGL(mnr_warm_kernel<3>, dim3(WARM_WAVES), dim3(64), 0, ws, a); break;
    case 4: hipLaunchKernelGGL(mnr_warm_kernel<4>, dim3(WARM_WAVES), dim3(64), 0, ws, a); break;
    case 9: hipLaunchKernelGGL(mnr_warm_kernel<9>, dim3(WARM_WAVES), dim3(64), 0, ws, a); break;
    default: hipLaunchKernelGGL(mnr_warm_kernel<0>, dim3(WARM_WAVES), dim3(64), 0, ws, a); break;
  }
  if (hipGetLastError() != hipSuccess) return none;
  return mnr_warm_ticket{(unsigned*)a.slot, id};
#endif
}
