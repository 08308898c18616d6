"""A trunk layer's dX (panel NT) and dW (TN) GEMMs, which read the same dY matrix: one after the other on the whole chip against
side by side on two streams with half the chip each, the dW kernel's M-splits contiguous or block-cyclic
(mnr_gemm_tn_args.m_interleave: both launches then walk M from top to bottom, and part of the second reads of dY is served by the
Infinity Cache).  Timing only; results in profiles/r5_ab.md (d).

    python tools/dxdw_probe.py [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import ops  # noqa: E402

dev, bf, PAN = 'cuda', torch.bfloat16, ops.LAYOUT_PANEL
M, W = 524288, 1024
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator(device=dev).manual_seed(0)
act = ops.to_panel(torch.relu(torch.rand((M, W), generator=g, device=dev) * 2 - 1).to(bf))
dys = [ops.to_panel(((torch.rand((M, W), generator=g, device=dev) * 2 - 1) * (torch.rand((M, W), generator=g, device=dev) > 0.5)).to(bf)) for _ in range(2)]
outs = [torch.zeros((M, W), dtype=bf, device=dev) for _ in range(2)]
Bw = ((torch.rand((W, W), generator=g, device=dev) * 2 - 1) * (6.0 / W) ** 0.5).to(bf)
bits = ops.bits_to_tile_order(torch.randint(0, 256, (M, W // 8), generator=g, device=dev, dtype=torch.uint8), W)
dW = torch.zeros((W, W), dtype=torch.float32, device=dev)
db = torch.zeros((W,), dtype=torch.float32, device=dev)
s2 = torch.cuda.Stream()


def dx(i, cap=0):
  ops.gemm_nt(dys[i % 2], Bw, M=M, N=W, K1=W, Cb=outs[i % 2], ldcb=W, nb=W, bits_in=bits, a1_layout=PAN, c_layout=PAN,
              walk_descending=bool(i & 1) and cap == 0, max_wgs=cap)


def dw(i, cap=0, il=False):
  ops.gemm_tn(act, dys[i % 2], dW, M=M, K=W, N=W, lda=W, ldb=W, ldc=W, bias_out=db, bias_n_valid=W, a_layout=PAN, b_layout=PAN,
              m_interleave=il, max_wgs=cap)


def run(mode, cap_x, cap_w, il, layers=7):
  cur = torch.cuda.current_stream()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

  def once():
    prev = None
    for i in range(layers):
      if mode == 'seq':
        dw(i)
        dx(i)
      else:
        ready = torch.cuda.Event()
        ready.record(cur)
        s2.wait_event(ready)
        with torch.cuda.stream(s2):
          dw(i, cap_w, il)
          done = torch.cuda.Event()
          done.record(s2)
        if prev is not None:
          cur.wait_event(prev)
        dx(i, cap_x)
        prev = done
    if prev is not None:
      cur.wait_event(prev)
  once()
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    once()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps / layers


def alone(which, cap, il=False):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  f = (lambda i: dx(i, cap)) if which == 'dx' else (lambda i: dw(i, cap, il))
  for i in range(2):
    f(i)
  torch.cuda.synchronize()
  e0.record()
  for i in range(reps * 2):
    f(i)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / (reps * 2)


for name, args in (('one after the other, whole chip', ('seq', 0, 0, False)),
                   ('side by side 128 + 128, dW splits contiguous', ('par', 128, 128, False)),
                   ('side by side 128 + 128, dW splits block-cyclic', ('par', 128, 128, True)),
                   ('side by side 112 (dX) + 128 (dW), block-cyclic', ('par', 112, 128, True)),
                   ('side by side, no caps (256 + 256 workgroups), block-cyclic', ('par', 0, 0, True)),
                   ('one after the other again', ('seq', 0, 0, False)),
                   ('side by side 128 + 128, block-cyclic, again', ('par', 128, 128, True))):
  print(f'{name:64s} {run(*args):8.1f} us per layer (dX + dW)', flush=True)
for which, cap, il in (('dx', 0, False), ('dx', 128, False), ('dw', 0, False), ('dw', 128, False), ('dw', 128, True)):
  print(f'{which} alone, max_wgs {cap:3d}, block-cyclic {il!s:5s}: {alone(which, cap, il):8.1f} us', flush=True)
