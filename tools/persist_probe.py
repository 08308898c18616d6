"""Persistent NT launches (mnr_gemm_nt_set_persistent) against one-workgroup-per-tile launches: bitwise screen + timing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import ops  # noqa: E402

dev = 'cuda'
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


def case(M, N, K1, K2=0, fwd=True):
  A1 = (torch.rand((M, K1), generator=g, device=dev) * 2 - 1).to(bf)
  A2 = (torch.rand((M, K2), generator=g, device=dev) * 2 - 1).to(bf) if K2 else None
  Bt = ((torch.rand((N, K1 + K2), generator=g, device=dev) * 2 - 1) * (6.0 / (K1 + K2)) ** 0.5).to(bf)
  bias = 0.05 * torch.randn((N,), generator=g, device=dev)
  bits = torch.randint(0, 256, (M, N // 8), generator=g, device=dev, dtype=torch.uint8)
  outs = {}
  for persist in (0, 1, 0, 1):
    ops.L.check(ops.L.debug().mnr_gemm_nt_set_persistent(persist))
    C = torch.zeros((M, N), dtype=bf, device=dev)
    bo = torch.zeros((M, N // 8), dtype=torch.uint8, device=dev)
    if fwd:
      fn = lambda: ops.gemm_nt(A1, Bt, M=M, N=N, K1=K1, A2=A2, K2=K2, bias=bias, n_bias=N, relu=True, Cb=C, ldcb=N, nb=N, bits_out=bo)
    else:
      fn = lambda: ops.gemm_nt(A1, Bt, M=M, N=N, K1=K1, Cb=C, ldcb=N, nb=N, bits_in=bits)
    us = timed(fn)
    torch.cuda.synchronize()
    key = (C.view(torch.int16).clone(), bo.clone())
    if persist in outs:
      pass
    outs.setdefault(persist, key)
    print(f'M={M} N={N} K={K1}+{K2} {"fwd" if fwd else "dX "} persist={persist}: {us:8.1f} us  {2.0 * M * N * (K1 + K2) / us / 1e6:7.1f} TF/s', flush=True)
  same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
  print('   bitwise equal' if same else '   MISMATCH', flush=True)
  ops.L.check(ops.L.debug().mnr_gemm_nt_set_persistent(0))


case(524288, 1024, 1024)
case(524288, 1024, 1024, fwd=False)
case(524288, 1024, 1024, 512)
case(524288, 1024, 512)
case(524288, 512, 1024)
case(524288, 256, 384)
case(1048576, 256, 256)
