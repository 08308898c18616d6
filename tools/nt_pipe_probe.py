"""Hand-pipelined K loop (BK = 32 x 4 stages, DMA between the MFMAs) against the two-stage loop of the 256x256 NT tiles:
bitwise screen, timing, per-tile cycle breakdown.  PIPES=0,1,11,12,13,14 with a -DMNR_NT_DEBUG_VARIANTS build adds the
pipelined loop with one ingredient removed (11 no DMA, 12 no MFMA, 13 no fragment reads, 14 MFMA only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import ops  # noqa: E402

dev = 'cuda'
PIPES = [int(x) for x in os.environ.get('PIPES', '0,1,0,1').split(',')]
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=10):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


def case(M, N, K1, K2=0, fwd=True, sparse=True):
  A1 = (torch.rand((M, K1), generator=g, device=dev) * 2 - 1)
  if sparse:
    A1 = torch.relu(A1)                       # post-ReLU activations: half the operand is zero, as in the trunk
  A1 = A1.to(bf)
  A2 = (torch.rand((M, K2), generator=g, device=dev) * 2 - 1).to(bf) if K2 else None
  Bt = ((torch.rand((N, K1 + K2), generator=g, device=dev) * 2 - 1) * (6.0 / (K1 + K2)) ** 0.5).to(bf)
  bias = 0.05 * torch.randn((N,), generator=g, device=dev)
  bits = torch.randint(0, 256, (M, N // 8), generator=g, device=dev, dtype=torch.uint8)
  outs = {}
  for wide in PIPES:
    if wide > 1 and not fwd:
      continue                                  # the probe variants exist for the forward kernel only
    ops.L.check(ops.L.debug().mnr_gemm_nt_set_pipelined(wide))
    C = torch.zeros((M, N), dtype=bf, device=dev)
    bo = torch.zeros((M, N // 8), dtype=torch.uint8, device=dev)
    if fwd:
      fn = lambda: ops.gemm_nt(A1, Bt, M=M, N=N, K1=K1, A2=A2, K2=K2, bias=bias, n_bias=N, relu=True, Cb=C, ldcb=N, nb=N, bits_out=bo)
    else:
      fn = lambda: ops.gemm_nt(A1, Bt, M=M, N=N, K1=K1, Cb=C, ldcb=N, nb=N, bits_in=bits)
    us = timed(fn)
    torch.cuda.synchronize()
    tl_text = None
    if M >= 65536:
      import numpy as np
      buf = torch.zeros((8192 * 2, 16), dtype=torch.int64, device=dev)
      ops.L.check(ops.L.debug().mnr_debug_gemm_timeline(buf.data_ptr()))
      fn()
      torch.cuda.synchronize()
      ops.L.check(ops.L.debug().mnr_debug_gemm_timeline(None))
      t = buf.cpu().numpy()
      first = (np.arange(t.shape[0]) < 256)[t[:, 3] != 0]          # a workgroup's first tile: no epilogue stores in front of its fill
      full = t[t[:, 3] != 0].astype(np.float64)
      t = full[:, :4]
      med = lambda x: float(np.median(x))
      tl_text = (f'      cycles/tile: prologue {med(t[:, 1] - t[:, 0]):.0f}  K-loop {med(t[:, 2] - t[:, 1]):.0f} ({med(t[:, 2] - t[:, 1]) / ((K1 + K2) / 64):.0f}/K-tile)  '
            f'epilogue {med(t[:, 3] - t[:, 2]):.0f}' + (f'  (staging {med(full[:, 8] - full[:, 2]):.0f} + store loop {med(full[:, 9] - full[:, 8]):.0f})' if full[:, 8].any() else '') + (f'  (pipeline fill {med(full[:, 12] - full[:, 1]):.0f}; first tiles {med((full[:, 12] - full[:, 1])[first]):.0f}, later {med((full[:, 12] - full[:, 1])[~first]):.0f})' if full[:, 12].any() else ''))
    outs.setdefault(wide, (C.view(torch.int16).clone(), bo.clone()))
    print(f'M={M} N={N} K={K1}+{K2} {"fwd" if fwd else "dX "} pipe={wide}: {us:8.1f} us  {2.0 * M * N * (K1 + K2) / us / 1e6:7.1f} TF/s', flush=True)
    if tl_text:
      print(tl_text, flush=True)
  for w in outs:
    if w != 0 and 0 in outs:
      same = torch.equal(outs[0][0], outs[w][0]) and torch.equal(outs[0][1], outs[w][1])
      print(f'   pipe={w}: ' + ('bitwise equal' if same else 'MISMATCH'), flush=True)
  ops.L.check(ops.L.debug().mnr_gemm_nt_set_pipelined(1))


if os.environ.get('PIPES') and not os.environ.get('ALL_CASES'):
  case(524288, 1024, 1024)
else:
  case(8192, 1024, 1024)
  case(524288, 1024, 1024)
  case(524288, 1024, 1024, fwd=False)
  case(524288, 1024, 1024, 512)
  case(524288, 1024, 512)
  case(524288, 512, 1024)
  case(1048576, 256, 512)
