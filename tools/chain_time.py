"""One line of chain-kernel timings at the 360.gin proposal shape for the loaded library (MNR_LIB_PATH selects a variant build):
forward training (activations + masks), forward without masks, forward inference, backward over both proposal levels (last dY not
stored).  Used by tools/variant_probe.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import ops  # noqa: E402


def timed(fn, reps=12):
  for _ in range(3):
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
  M, W, K0, D = 1 << 20, 256, int(os.environ.get('CHAIN_K0', '512')), int(os.environ.get('CHAIN_DEPTH', '4'))
  dev, bf = 'cuda', torch.bfloat16
  g = torch.Generator(device=dev).manual_seed(0)
  feat = (torch.rand((M, K0), generator=g, device=dev) * 2 - 1).to(bf)
  Bt = [((torch.rand((W, K0 if i == 0 else W), generator=g, device=dev) * 2 - 1) * (6.0 / (K0 if i == 0 else W)) ** 0.5).to(bf) for i in range(D)]
  bias = [0.05 * torch.randn((W,), generator=g, device=dev) for _ in range(D)]
  wh = ((torch.rand((W,), generator=g, device=dev) * 2 - 1) * 0.15).to(bf)
  bh = torch.zeros((1,), device=dev)
  out = torch.empty((M,), device=dev)
  acts = [torch.empty((M, W), dtype=bf, device=dev) for _ in range(D)]
  bits = [torch.empty((M, W // 8), dtype=torch.uint8, device=dev) for _ in range(D)]
  layers = list(zip(Bt, bias))
  gh = torch.randn((2 * M,), generator=g, device=dev) * 0.01
  bits2 = [torch.randint(0, 256, (2 * M, W // 8), dtype=torch.uint8, device=dev, generator=g) for _ in range(D)]
  Bw = [None] + [Bt[i].t().contiguous() for i in range(1, D)]
  dY = [torch.empty((2 * M, W), dtype=bf, device=dev) for _ in range(D - 1)] + [None]
  t = [timed(lambda: ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out, acts=acts, bits=bits)),
       timed(lambda: ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out, acts=acts)),
       timed(lambda: ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out)),
       timed(lambda: ops.mlp_chain_bwd(gh, wh.float(), bits2, Bw, dY, M=2 * M, W=W))]
  chk = (acts[-1].float().sum().item(), sum(int(b.sum().item()) for b in bits), out.sum().item())
  print(f'fwd train {t[0]:8.1f}  no masks {t[1]:8.1f}  inference {t[2]:8.1f}  bwd {t[3]:8.1f} us   checksum {chk[0]:.6e} {chk[1]} {chk[2]:.6e}', flush=True)


if __name__ == '__main__':
  main()
