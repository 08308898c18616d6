"""Timing decomposition of the fused Dense chain (csrc/fused_mlp.hip) at the 360.gin proposal-level shape.

    python tools/chain_probe.py [--M 1048576] [--W 256] [--K0 512]

Variants: training forward (activations + masks written), inference forward (head only), depth 1 (layer-0 stream only),
short layer 0 (K0 = 64), backward chain; next to the per-layer GEMMs they replace.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import ops  # noqa: E402


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


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--M', type=int, default=1 << 20)
  ap.add_argument('--W', type=int, default=256)
  ap.add_argument('--K0', type=int, default=512)
  ap.add_argument('--depth', type=int, default=4)
  ap.add_argument('--timeline', action='store_true')
  a = ap.parse_args()
  M, W, K0, D = a.M, a.W, a.K0, a.depth
  dev = 'cuda'
  g = torch.Generator(device=dev).manual_seed(0)
  bf = torch.bfloat16
  feat = (torch.rand((M, K0), generator=g, device=dev) * 2 - 1).to(bf)
  Bt = [((torch.rand((W, K0 if i == 0 else W), generator=g, device=dev) * 2 - 1) * (6.0 / (K0 if i == 0 else W)) ** 0.5).to(bf) for i in range(D)]
  bias = [0.05 * torch.randn((W,), generator=g, device=dev) for _ in range(D)]
  wh = ((torch.rand((W,), generator=g, device=dev) * 2 - 1) * 0.15).to(bf)
  bh = torch.zeros((1,), device=dev)
  out = torch.empty((M,), device=dev)
  acts = [torch.empty((M, W), dtype=bf, device=dev) for _ in range(D)]
  bits = [torch.empty((M, W // 8), dtype=torch.uint8, device=dev) for _ in range(D)]
  layers = list(zip(Bt, bias))
  flops = 2.0 * M * (K0 * W + (D - 1) * W * W + W)

  def show(name, us, fl=None, nbytes=None):
    s = f'{name:44s} {us:9.1f} us'
    if fl:
      s += f'  {fl / us / 1e6:7.1f} TF/s'
    if nbytes:
      s += f'  {nbytes / us / 1e6:6.2f} TB/s'
    print(s, flush=True)

  act_bytes = M * W * 2
  show('fwd train (acts + bits)', timed(lambda: ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out, acts=acts, bits=bits)),
       flops, M * K0 * 2 + D * act_bytes)
  show('fwd train (acts, no bits)', timed(lambda: ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out, acts=acts)),
       flops, M * K0 * 2 + D * act_bytes)
  show('fwd inference (head only)', timed(lambda: ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out)), flops, M * K0 * 2)
  show('fwd depth 1 inference', timed(lambda: ops.mlp_chain_fwd(feat, K0, layers[:1], M=M, W=W, w_head=wh, b_head=bh, head_out=out)), 2.0 * M * K0 * W, M * K0 * 2)
  show('fwd depth 1 train', timed(lambda: ops.mlp_chain_fwd(feat, K0, layers[:1], M=M, W=W, w_head=wh, b_head=bh, head_out=out, acts=acts[:1], bits=bits[:1])),
       2.0 * M * K0 * W, M * K0 * 2 + act_bytes)
  l64 = [(Bt[0][:, :64].contiguous(), bias[0])] + layers[1:]
  show('fwd K0 = 64 inference', timed(lambda: ops.mlp_chain_fwd(feat, 64, l64, M=M, W=W, w_head=wh, b_head=bh, head_out=out)), 2.0 * M * (64 * W + (D - 1) * W * W))
  show('fwd K0 = 64 train', timed(lambda: ops.mlp_chain_fwd(feat, 64, l64, M=M, W=W, w_head=wh, b_head=bh, head_out=out, acts=acts, bits=bits)),
       2.0 * M * (64 * W + (D - 1) * W * W), D * act_bytes)
  # per-layer path
  def per_layer():
    x = feat
    for i in range(D):
      ops.gemm_nt(x, Bt[i], M=M, N=W, K1=K0 if i == 0 else W, bias=bias[i], n_bias=W, relu=True, Cb=acts[i], ldcb=W, nb=W, bits_out=bits[i])
      x = acts[i]
  show('per-layer NT GEMMs (no head)', timed(per_layer), flops)
  # backward chain
  gh = torch.randn((M,), generator=g, device=dev) * 0.01
  whf = wh.float()
  Bw = [None] + [Bt[i].t().contiguous() for i in range(1, D)]
  dY = [torch.empty((M, W), dtype=bf, device=dev) for _ in range(D)]
  bflops = 2.0 * M * (D - 1) * W * W
  show('bwd chain', timed(lambda: ops.mlp_chain_bwd(gh, whf, bits, Bw, dY, M=M, W=W)), bflops, D * act_bytes)
  show('bwd chain (dY_last not stored)', timed(lambda: ops.mlp_chain_bwd(gh, whf, bits, Bw, dY[:-1] + [None], M=M, W=W)), bflops, (D - 1) * act_bytes)


  if a.timeline:
    import numpy as np
    buf = torch.zeros((256, 32), dtype=torch.int64, device=dev)

    def timeline(name, fn, slots):
      buf.zero_()
      ops.L.check(ops.L.debug().mnr_debug_chain_timeline(buf.data_ptr()))
      fn()
      torch.cuda.synchronize()
      ops.L.check(ops.L.debug().mnr_debug_chain_timeline(None))
      t = buf.cpu().numpy().astype(np.float64)
      ok = t[:, 31] > 0
      t = t[ok]
      clk = ((t[:, max(slots)] - t[:, 0]) / ((t[:, 31] - t[:, 30]) * 10.0)).mean()      # cycles per ns (s_memrealtime: 100 MHz)
      print(f'{name}: {ok.sum()} workgroups, shader clock ~{clk:.2f} GHz; cycles since tile start (mean over workgroups):')
      prev = 0.0
      for sl in slots:
        v = (t[:, sl] - t[:, 0]).mean()
        print(f'   slot {sl:2d}: {v:9.0f}  (+{v - prev:8.0f})')
        prev = v

    fwd_slots = [1] + [x for li in range(D) for x in (2 + 3 * li, 3 + 3 * li, 4 + 3 * li)]
    timeline('fwd train', lambda: ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out, acts=acts, bits=bits), fwd_slots)
    timeline('fwd train, no masks', lambda: ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out, acts=acts), fwd_slots)
    timeline('fwd inference', lambda: ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out), fwd_slots)
    bwd_slots = [1] + [x for li in range(D - 1, 0, -1) for x in (2 + 3 * li, 3 + 3 * li, 4 + 3 * li)]
    timeline('bwd chain', lambda: ops.mlp_chain_bwd(gh, whf, bits, Bw, dY, M=M, W=W), bwd_slots)


if __name__ == '__main__':
  main()
