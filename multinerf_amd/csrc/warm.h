// Infinity-Cache warmer for the streamed operand of the wide trunk GEMMs (internal to the library, not part of the C ABI).
//
// Measured (tools/panel_probe.py, probe builds -DPN_DBG=7..10, profiles/r4_ab.md): a 524288 x 1024 x 1024 trunk layer takes
// 1015-1030 us when its A operand (1 GB, written by the previous layer) arrives from HBM and 880-890 us when the same bytes sit
// in the 256 MB Infinity Cache, in the L2 or in the L1 alike; a 256 / 512 MB footprint is as slow as 1 GB.  The K loop keeps three
// K-tiles (2.5 us of work) in flight, which covers the on-die latencies and not the tail of the HBM's.  Freshly WRITTEN lines are
// not retained by the Infinity Cache (tools/chunk_probe.py: the layers run chunk by chunk gain 3 %), lines that were READ are.
//
// So a second, tiny kernel reads the operand ahead of the GEMM: `mnr_warm_kernel`, 256 single-wave workgroups without LDS and
// with < 16 VGPRs (they fit next to the GEMM's two waves per SIMD), every lane touching one 128-byte line per instruction (one
// instruction covers 8 KiB of the operand and returns 256 bytes to the CU).  It is paced by the GEMM itself: wave 0 of
// workgroup 0 publishes the index of the progress unit (NT: the round of output tiles) it has started, and the warmer stays at
// most `ahead` units in front of it; every wait is bounded (polls and wall time), a warmer that has lost its GEMM exits.
// The bytes come from HBM once either way: through the warmer instead of through the GEMM's own LDS-DMA.
#pragma once

#include <stdio.h>

#include "common.h"

struct mnr_warm_tensor {
  const void* base;         // first byte of progress unit 0 (of stream 0)
  long long unit_bytes;     // contiguous bytes per progress unit and stream (a multiple of 8192)
  long long stream_stride;  // bytes between the streams of one unit
};

struct mnr_warm_ticket {
  unsigned* slot;           // where the GEMM publishes (64 dwords); nullptr: this launch is not warmed
  unsigned id;              // launch id (20 bits)
};

#define MNR_WARM_DONE 0xfffu

// Enqueues the warmer of the launch the caller is about to put on `stream`: unit u of tensor t covers, for s in
// [0, nstreams), the bytes [base + s * stream_stride + u' * unit_bytes, + unit_bytes) with u' = rev ? units - 1 - u : u.
// Returns {nullptr, 0} when warming is off or the request does not qualify.
mnr_warm_ticket mnr_warm_begin(const mnr_warm_tensor* t, int ntensors, int units, int nstreams, int rev, void* stream);

// Operands below this many bytes are not worth a warmer (mnr_set_warm(2, .): 0, the tests' way to run the path at small sizes).
long long mnr_warm_min_bytes();

// The GEMM side: called by ONE wave (all 64 lanes) of one workgroup.  A relaxed system-scope store: it sits in that wave's vmcnt
// queue as one more store, which makes the counted waits around it more conservative by one operation, never less.
__device__ __forceinline__ void mnr_warm_publish(unsigned* slot, unsigned id, unsigned count) {
  const unsigned v = (id << 12) | (count & 0xfffu);
  MNR_GPU_ONLY(__hip_atomic_store(slot + mnr_lane_id(), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  MNR_SIM_HOOK(slot[mnr_lane_id()] = v; hipsim::vm_store(); if (mnr_lane_id() == 0 && getenv("MNR_TRACE_PUBLISH")) printf("PUBLISH %x\n", v));
}
