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
  ap.add_argument('--pad', type=int, default=0, help='extra elements in the leading dimension of A / Bt / C (channel-conflict probe)')
  ap.add_argument('--timeline', action='store_true', help='per-workgroup s_memtime breakdown of the 1024-wide forward GEMM')
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
  if args.timeline:
    import numpy as np
    pad = args.pad
    A = rnd(M, K + pad)
    Bt = rnd(N, K + pad) * 0.05
    bias = torch.zeros(N, device=dev)
    C = torch.empty((M, N + pad), dtype=bf, device=dev)
    nwg = (M // 256) * (N // 256)
    buf = torch.zeros((nwg, 16), dtype=torch.int64, device=dev)
    for c in [int(x) for x in (args.cfgs or '2,18').split(',')]:
      ops.L.check(ops.lib().mnr_gemm_nt_set_config(c, 0))
      fn = lambda: ops.gemm_nt(A, Bt, M=M, N=N, K1=K, bias=bias, n_bias=N, relu=True, Cb=C, ldcb=N + pad, nb=N)
      for _ in range(5):
        fn()
      torch.cuda.synchronize()
      ops.L.check(ops.lib().mnr_debug_gemm_timeline(buf.data_ptr()))
      fn()
      torch.cuda.synchronize()
      ops.L.check(ops.lib().mnr_debug_gemm_timeline(None))
      t = buf.cpu().numpy()
      t = t[t[:, 3] != 0]
      tm = t[:, :4].astype(np.float64)
      rt = t[:, 4:6].astype(np.float64)
      span_rt = rt[:, 1].max() - rt[:, 0].min()              # 100 MHz ticks
      span_mt = tm[:, 3].max() - tm[:, 0].min()
      pro, loop, epi = tm[:, 1] - tm[:, 0], tm[:, 2] - tm[:, 1], tm[:, 3] - tm[:, 2]
      tot = tm[:, 3] - tm[:, 0]
      hw = t[:, 6]
      cu_key = ((hw >> 32) & 0xf) * 4096 + (hw & 0xffff & ~0xff)   # xcc, se/sh/cu fields of HW_ID (wave/simd/pipe masked)
      ncu = len(np.unique(cu_key))
      gaps = []
      for k in np.unique(cu_key):
        sel = tm[cu_key == k]
        sel = sel[np.argsort(sel[:, 0])]
        gaps += list(sel[1:, 0] - sel[:-1, 3])
      gaps = np.array(gaps)
      wg_rt = rt[:, 1] - rt[:, 0]
      ghz = np.median(tot[wg_rt > 0] / wg_rt[wg_rt > 0]) * 100 / 1e3
      q = lambda x: f'{np.median(x):.0f} [{np.percentile(x, 10):.0f} {np.percentile(x, 90):.0f}]'
      ep = t[:, 8:12].astype(np.float64)
      if (ep > 0).all():
        e = [ep[:, 0] - tm[:, 2], ep[:, 1] - ep[:, 0], ep[:, 2] - ep[:, 1], ep[:, 3] - ep[:, 2], tm[:, 3] - ep[:, 3]]
        print('   epilogue: stage0 ' + q(e[0]) + '  store0 ' + q(e[1]) + '  stage1 ' + q(e[2]) + '  store1 ' + q(e[3]) + '  tail ' + q(e[4]))
      print(f'pad {pad} cfg {c}: {len(t)} workgroups on {ncu} CUs; span {span_rt / 100:.1f} us '
            f'-> {ghz:.3f} GHz tick rate (per-workgroup memtime / realtime)\n'
            f'   per workgroup ticks median [p10 p90]: prologue {q(pro)}  K-loop {q(loop)}  epilogue {q(epi)}  total {q(tot)}\n'
            f'   gap between consecutive workgroups of one CU: {q(gaps)}; busy fraction {tot.sum() / ncu / span_mt:.3f}', flush=True)
    return
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
    # reference outputs from the default configuration (same MFMA order per accumulator -> bitwise equal)
    Kc = 512
    Ac = rnd(M, Kc)
    Btc = rnd(N, K + Kc) * 0.05
    biasc = torch.randn(N, device=dev, generator=g) * 0.1
    bits = torch.empty((M, N // 8), dtype=torch.uint8, device=dev)
    def run_checks(tag):
      outs = []
      Cx = torch.empty((M, N), dtype=bf, device=dev)
      ops.gemm_nt(A, Btc, M=M, N=N, K1=K, A2=Ac, K2=Kc, bias=biasc, n_bias=N, relu=True, Cb=Cx, ldcb=N, nb=N, bits_out=bits)
      outs.append(Cx.clone()); outs.append(bits.clone())
      Cy = torch.empty((M, N), dtype=bf, device=dev)
      ops.gemm_nt(A, Bt, M=M, N=N, K1=K, Cb=Cy, ldcb=N, nb=N, bits_in=bits)
      outs.append(Cy.clone())
      Cz = torch.empty((M, N), dtype=bf, device=dev)
      ops.gemm_nt(A, Bt, M=M, N=N, K1=64, lda1=K, ldb=K, Cb=Cz, ldcb=N, nb=N)   # single K tile
      outs.append(Cz.clone())
      ops.gemm_nt(A, Bt, M=M, N=N, K1=128, lda1=K, ldb=K, Cb=Cz, ldcb=N, nb=N)  # two K tiles
      outs.append(Cz.clone())
      torch.cuda.synchronize()
      return outs
    ops.L.check(ops.lib().mnr_gemm_nt_set_config(2, 0))
    ref = run_checks('ref')
    # spot check of the reference itself against torch on a slice
    sl = slice(0, 2048)
    want = torch.relu(torch.cat([A[sl], Ac[sl]], 1).float() @ Btc.float().t() + biasc)
    print('cfg 2 vs torch max abs err', (ref[0][sl].float() - want).abs().max().item(), flush=True)
    for c in [int(x) for x in args.cfgs.split(',')]:
      ops.L.check(ops.lib().mnr_gemm_nt_set_config(c, 0))
      if c >= 18:
        for rep in range(3):
          got = run_checks(f'cfg{c}')
          bad = [i for i, (x, y) in enumerate(zip(got, ref)) if not torch.equal(x, y)]
          print(f'cfg {c} check rep {rep}: ' + ('bitwise equal to cfg 2' if not bad else f'MISMATCH in outputs {bad}: '
                + ', '.join(f'{(got[i].float() - ref[i].float()).abs().max().item():.3e}' for i in bad)), flush=True)
      time_it(f'cfg {c} nt fwd  1024', lambda: ops.gemm_nt(A, Bt, M=M, N=N, K1=K, bias=bias, n_bias=N, relu=True, Cb=C, ldcb=N, nb=N), 2.0 * M * N * K)
      time_it(f'cfg {c} nt dX   1024', lambda: ops.gemm_nt(A, Bt, M=M, N=N, K1=K, mask=mask, ldmask=N, Cb=C, ldcb=N, nb=N), 2.0 * M * N * K)
      time_it(f'cfg {c} nt prop  256', lambda: ops.gemm_nt(A2, B2, M=M2, N=K2, K1=K2, bias=b2, n_bias=K2, relu=True, Cb=C2, ldcb=K2, nb=K2), 2.0 * M2 * K2 * K2)
      if c in (36, 37):
        # the direct-weights loop reading its weights from a fragment-major image (one contiguous KiB per instruction)
        import ctypes
        lib = ops.lib()
        img = torch.empty(N * K, dtype=bf, device=dev)
        ops.L.check(lib.mnr_pack_w_frag_bf16(ctypes.c_void_p(Bt.data_ptr()), K, N, K, ctypes.c_void_p(img.data_ptr()), None))
        img2 = torch.empty(K2 * K2, dtype=bf, device=dev)
        ops.L.check(lib.mnr_pack_w_frag_bf16(ctypes.c_void_p(B2.data_ptr()), K2, K2, K2, ctypes.c_void_p(img2.data_ptr()), None))
        torch.cuda.synchronize()
        ops.L.check(lib.mnr_debug_gemm_wfrag(ctypes.c_void_p(img.data_ptr())))
        Cw = torch.empty((M, N), dtype=bf, device=dev)
        ops.gemm_nt(A, Bt, M=M, N=N, K1=K, bias=bias, n_bias=N, relu=True, Cb=Cw, ldcb=N, nb=N)
        ops.L.check(lib.mnr_debug_gemm_wfrag(None))
        ops.gemm_nt(A, Bt, M=M, N=N, K1=K, bias=bias, n_bias=N, relu=True, Cb=C, ldcb=N, nb=N)
        torch.cuda.synchronize()
        print(f'cfg {c} fragment-major weights: ' + ('bitwise equal to the row-major run' if torch.equal(Cw, C) else 'MISMATCH'), flush=True)
        ops.L.check(lib.mnr_debug_gemm_wfrag(ctypes.c_void_p(img.data_ptr())))
        time_it(f'cfg {c} nt fwd  1024 wfrag', lambda: ops.gemm_nt(A, Bt, M=M, N=N, K1=K, bias=bias, n_bias=N, relu=True, Cb=C, ldcb=N, nb=N), 2.0 * M * N * K)
        time_it(f'cfg {c} nt dX   1024 wfrag', lambda: ops.gemm_nt(A, Bt, M=M, N=N, K1=K, mask=mask, ldmask=N, Cb=C, ldcb=N, nb=N), 2.0 * M * N * K)
        ops.L.check(lib.mnr_debug_gemm_wfrag(ctypes.c_void_p(img2.data_ptr())))
        time_it(f'cfg {c} nt prop  256 wfrag', lambda: ops.gemm_nt(A2, B2, M=M2, N=K2, K1=K2, bias=b2, n_bias=K2, relu=True, Cb=C2, ldcb=K2, nb=K2), 2.0 * M2 * K2 * K2)
        ops.L.check(lib.mnr_debug_gemm_wfrag(None))
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
    # split operand paths (TnBigSplit): dY through registers + ds_write, activations by LDS-DMA; screened against the default
    Cs = torch.zeros((K, N), device=dev)
    Cd = torch.zeros((K, N), device=dev)
    ops.gemm_tn(A, Bm, Cd, M=M, K=K, N=N)
    ops.L.check(ops.lib().mnr_gemm_tn_set_split(1))
    ops.gemm_tn(A, Bm, Cs, M=M, K=K, N=N)
    torch.cuda.synchronize()
    rel = ((Cs - Cd).norm() / Cd.norm()).item()          # fp32 atomics: the order differs from run to run, not the terms
    print(f'tn split vs default: relative difference {rel:.2e} ' + ('(ok)' if rel < 1e-5 else '(MISMATCH)'), flush=True)
    time_it(f'tn  dW   M={M} K={K} N={N} split', lambda: ops.gemm_tn(A, Bm, Cw, M=M, K=K, N=N, bias_out=bb, bias_n_valid=N), 2.0 * M * N * K)
    ops.L.check(ops.lib().mnr_gemm_tn_set_split(0))
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
