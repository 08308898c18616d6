#!/usr/bin/env python
"""Micro-benchmark of the merged NeRF head's weight-gradient GEMM at the 360.gin shape (M = 524288 rows, K = 1024):
N = 384 on the 128x128 tile (the merged [bottleneck | density] operand), N = 256 on the 256x256 tile, the same with the
density column as a vector (gcol), and a contiguous B for comparison."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multinerf_amd import ops

dev = 'cuda'
M, K = 1 << 19, 1024
g = torch.Generator(device=dev).manual_seed(0)
A = torch.relu(torch.randn((M, K), generator=g, device=dev)).to(torch.bfloat16)
B384 = (torch.randn((M, 384), generator=g, device=dev) * 0.01).to(torch.bfloat16)
B384[:, 257:] = 0
B256 = B384[:, :256].contiguous()
gv = B384[:, 256].contiguous()


def timed(fn, reps=10):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  ts = []
  for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
  ts.sort()
  return ts[len(ts) // 2]


C384 = torch.zeros((K, 384), device=dev)
C256 = torch.zeros((K, 256), device=dev)
b384 = torch.zeros((384,), device=dev)
b256 = torch.zeros((256,), device=dev)
go = torch.zeros((K,), device=dev)
runs = [
    ('N = 384, 128x128 tile (merged head)', lambda: ops.gemm_tn(A, B384, C384, M=M, K=K, N=384, bias_out=b384, bias_n_valid=257)),
    ('N = 256 of the 384-wide B, 256x256 tile', lambda: ops.gemm_tn(A, B384, C256, M=M, K=K, N=256, ldb=384, bias_out=b256, bias_n_valid=256)),
    ('the same + density column as a vector (gcol)', lambda: ops.gemm_tn(A, B384, C256, M=M, K=K, N=256, ldb=384, bias_out=b256, bias_n_valid=256, gcol=gv, gcol_out=go)),
    ('N = 256, contiguous B', lambda: ops.gemm_tn(A, B256, C256, M=M, K=K, N=256, bias_out=b256, bias_n_valid=256)),
    ('N = 256, contiguous B, no bias', lambda: ops.gemm_tn(A, B256, C256, M=M, K=K, N=256)),
]
for name, fn in runs:
  us = timed(fn)
  print(f'{name:50s} {us:8.1f} us   ({2.0 * M * K * 256 / us / 1e6:6.1f} TFLOP/s useful, {(M * K * 2 + M * 256 * 2) / us / 1e6:5.2f} TB/s)', flush=True)
