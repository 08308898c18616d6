#!/usr/bin/env python
"""Micro-benchmark of the Dense-layer GEMMs at the 360.gin shapes (for rocprofv3 --pmc runs).

  python tools/gemm_probe.py [--reps 10] [--which nt,tn]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multinerf_amd import ops


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=10)
  ap.add_argument('--which', default='nt,ntmask,tn,prop')
  ap.add_argument('--M', type=int, default=524288)
  ap.add_argument('--cfgs', default='', help='comma list of NT configuration ids to sweep (csrc/gemm.hip NtC*)')
  args = ap.parse_args()
  dev = 'cuda'
  M = args.M
  g = torch.Generator(device=dev).manual_seed(0)
  bf = torch.bfloat16

  def rnd(*shape):
    return (torch.rand(shape, generator=g, device=dev) * 2 - 1).to(bf)

  def time_it(name, fn, flops):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
      fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    print(f'{name}: {ms*1e3:.1f} us  {flops / ms / 1e9:.1f} TFLOP/s', flush=True)

  which = args.which.split(',')
  K = N = 1024
  A = rnd(M, K)
  Bt = rnd(N, K) * 0.05
  bias = torch.zeros(N, device=dev)
  if args.cfgs:
    C = torch.empty((M, N), dtype=bf, device=dev)
    mask = rnd(M, N)
    M2, K2 = 2 * M, 256
    A2 = rnd(M2, K2)
    B2 = rnd(K2, K2) * 0.05
    C2 = torch.empty((M2, K2), dtype=bf, device=dev)
    b2 = torch.zeros(K2, device=dev)
    for c in [int(x) for x in args.cfgs.split(',')]:
      ops.L.check(ops.lib().mnr_gemm_nt_set_config(c, c if c in (0, 1, 5, 6) else 0))
      time_it(f'cfg {c} nt fwd  1024', lambda: ops.gemm_nt(A, Bt, M=M, N=N, K1=K, bias=bias, n_bias=N, relu=True, Cb=C, ldcb=N, nb=N), 2.0 * M * N * K)
      time_it(f'cfg {c} nt dX   1024', lambda: ops.gemm_nt(A, Bt, M=M, N=N, K1=K, mask=mask, ldmask=N, Cb=C, ldcb=N, nb=N), 2.0 * M * N * K)
      time_it(f'cfg {c} nt prop  256', lambda: ops.gemm_nt(A2, B2, M=M2, N=K2, K1=K2, bias=b2, n_bias=K2, relu=True, Cb=C2, ldcb=K2, nb=K2), 2.0 * M2 * K2 * K2)
    return
  if 'nt' in which:
    C = torch.empty((M, N), dtype=bf, device=dev)
    time_it(f'nt  fwd  M={M} N={N} K={K}', lambda: ops.gemm_nt(A, Bt, M=M, N=N, K1=K, bias=bias, n_bias=N, relu=True, Cb=C, ldcb=N, nb=N), 2.0 * M * N * K)
  if 'ntmask' in which:
    C = torch.empty((M, N), dtype=bf, device=dev)
    mask = rnd(M, N)
    time_it(f'nt  dX   M={M} N={N} K={K} +mask', lambda: ops.gemm_nt(A, Bt, M=M, N=N, K1=K, mask=mask, ldmask=N, Cb=C, ldcb=N, nb=N), 2.0 * M * N * K)
  if 'tncfg' in which:
    Bm = rnd(M, N)
    Cw = torch.zeros((K, N), device=dev)
    bb = torch.zeros(N, device=dev)
    A2 = rnd(2 * M, 256)
    Bm2 = rnd(2 * M, 256)
    Cw2 = torch.zeros((256, 256), device=dev)
    A3 = rnd(2 * M, 512)
    Cw3 = torch.zeros((512, 256), device=dev)
    for mt in (4, 1, 1 << 30):
      ops.L.check(ops.lib().mnr_gemm_tn_set_config(mt))
      time_it(f'tn big_min_tiles={mt} dW 1024x1024', lambda: ops.gemm_tn(A, Bm, Cw, M=M, K=K, N=N, bias_out=bb, bias_n_valid=N), 2.0 * M * N * K)
      time_it(f'tn big_min_tiles={mt} dW prop 256x256', lambda: ops.gemm_tn(A2, Bm2, Cw2, M=2 * M, K=256, N=256), 2.0 * 2 * M * 256 * 256)
      time_it(f'tn big_min_tiles={mt} dW prop 512x256', lambda: ops.gemm_tn(A3, Bm2, Cw3, M=2 * M, K=512, N=256), 2.0 * 2 * M * 512 * 256)
    ops.L.check(ops.lib().mnr_gemm_tn_set_config(1))
  if 'tn' in which:
    Bm = rnd(M, N)
    Cw = torch.zeros((K, N), device=dev)
    bb = torch.zeros(N, device=dev)
    time_it(f'tn  dW   M={M} K={K} N={N}', lambda: ops.gemm_tn(A, Bm, Cw, M=M, K=K, N=N, bias_out=bb, bias_n_valid=N), 2.0 * M * N * K)
  if 'prop' in which:
    M2, K2 = 2 * M, 256
    A2 = rnd(M2, K2)
    B2 = rnd(K2, K2) * 0.05
    C2 = torch.empty((M2, K2), dtype=bf, device=dev)
    b2 = torch.zeros(K2, device=dev)
    time_it(f'nt  prop M={M2} N={K2} K={K2}', lambda: ops.gemm_nt(A2, B2, M=M2, N=K2, K1=K2, bias=b2, n_bias=K2, relu=True, Cb=C2, ldcb=K2, nb=K2), 2.0 * M2 * K2 * K2)
    Bm2 = rnd(M2, K2)
    Cw2 = torch.zeros((K2, K2), device=dev)
    time_it(f'tn  prop M={M2} K={K2} N={K2}', lambda: ops.gemm_tn(A2, Bm2, Cw2, M=M2, K=K2, N=K2), 2.0 * M2 * K2 * K2)


if __name__ == '__main__':
  main()
