#!/usr/bin/env python
"""What the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the trunk shapes: a ceiling reference for
DESIGN.md, not part of the product path."""
import torch

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)


def t(fn, flops, name, reps=10):
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  print(f'{name}: {ms * 1e3:.1f} us  {flops / ms / 1e9:.1f} TFLOP/s', flush=True)


for (M, N, K) in ((524288, 1024, 1024), (1048576, 256, 256), (8192, 8192, 8192)):
  A = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
  B = ((torch.rand((N, K), generator=g, device=dev) * 2 - 1) * 0.05).to(torch.bfloat16)
  C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
  t(lambda: torch.matmul(A, B.t(), out=C), 2.0 * M * N * K, f'torch.matmul NT  M={M} N={N} K={K}')
  if M != N:
    G = (torch.rand((M, N), generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    W = torch.empty((K, N), dtype=torch.bfloat16, device=dev)
    t(lambda: torch.matmul(A.t(), G, out=W), 2.0 * M * N * K, f'torch.matmul TN  (dW) M={M} N={N} K={K}')
