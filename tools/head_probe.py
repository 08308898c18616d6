#!/usr/bin/env python
"""Micro-benchmark of mnr_small_head_bwd at the 360.gin shapes (env MNR_SHB_ROWS / MNR_SHB_UNROLL select variants)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multinerf_amd import ops

dev = 'cuda'
for (M, K, C, what) in ((1 << 20, 256, 1, 'prop density head'), (1 << 19, 128, 3, 'rgb head, 128-wide view MLP'), (1 << 19, 256, 3, 'rgb head of 360.gin')):
  H = torch.rand((M, K), device=dev).to(torch.bfloat16)
  g = torch.randn((M, C), device=dev)
  W = torch.randn((K, C), device=dev)
  dX = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
  dW = torch.zeros((K, C), device=dev)
  db = torch.zeros((C,), device=dev)
  fn = lambda: ops.small_head_bwd(H, K, g, W, M=M, K=K, Cn=C, dX=dX, lddx=K, relu_mask=True, dW=dW.view(-1), db=db)
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10):
    fn()
  e1.record()
  torch.cuda.synchronize()
  us = e0.elapsed_time(e1) * 100
  gb = 2 * M * K * 2 / 1e9
  print(f'{what}: M={M} K={K} C={C}: {us:.1f} us, {gb / us * 1e6 / 1e3:.2f} TB/s (H read + dX write)', flush=True)
