"""Weight-gradient kernel at the trunk's shape: row-major vs panel operands, and both with cache-resident operands (leading
dimension 8: every 32-row stage re-reads the same few KiB; values meaningless, timing only) to see what the operands' trip from
HBM costs the K loop."""
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


M, K, N = 524288, 1024, 1024
A = torch.relu(torch.rand((M, K), generator=g, device=dev) * 2 - 1).to(bf)
B = (torch.rand((M, N), generator=g, device=dev) * 2 - 1).to(bf)
C = torch.zeros((K, N), device=dev)
db = torch.zeros((N,), device=dev)
PAN = ops.LAYOUT_PANEL
arms = {
    'row-major': lambda: ops.gemm_tn(A, B, C, M=M, K=K, N=N, bias_out=db, bias_n_valid=N),
    'panel': lambda: ops.gemm_tn(A, B, C, M=M, K=K, N=N, bias_out=db, bias_n_valid=N, a_layout=PAN, b_layout=PAN),
    'row-major, resident operands': lambda: ops.gemm_tn(A, B, C, M=M, K=K, N=N, lda=8, ldb=8, bias_out=db, bias_n_valid=N),
    'row-major, resident B': lambda: ops.gemm_tn(A, B, C, M=M, K=K, N=N, ldb=8, bias_out=db, bias_n_valid=N),
}
for rep in range(2):
  for name, fn in arms.items():
    us = timed(fn)
    print(f'TN {M}x{K}x{N} {name:32s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s', flush=True)
