"""Does an activation that the previous layer has just written come out of the 256 MB Infinity Cache?  (tools/panel_probe.py with
-DPN_DBG=7 / 8: a trunk layer whose A operand does not come from HBM takes 880-890 us instead of 1015-1030.)

A stack of L forward trunk layers (1024 -> 1024, bias + ReLU, panel layout, mask bits out) over M = 524288 rows, run
  layer-major: every layer over all rows (what Model does: each launch reads 1 GB the previous one wrote), and
  chunk-major: the L layers over R rows at a time (R * 2 KiB = the bytes a launch reads and writes once).

    python tools/chunk_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import ops  # noqa: E402

dev, bf = 'cuda', torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
M, W, L = 524288, 1024, int(os.environ.get('LAYERS', 6))
PAN = ops.LAYOUT_PANEL
acts = [torch.relu(torch.rand((M * W,), generator=g, device=dev) * 2 - 1).to(bf)] + [torch.zeros((M * W,), dtype=bf, device=dev) for _ in range(L)]
bits = [torch.zeros((M * W // 8,), dtype=torch.uint8, device=dev) for _ in range(L)]
Bts = [((torch.rand((W, W), generator=g, device=dev) * 2 - 1) * (6.0 / W) ** 0.5).to(bf) for _ in range(L)]
biases = [0.05 * torch.randn((W,), generator=g, device=dev) for _ in range(L)]


def layer(l, r0, R):
  a = acts[l][r0 * W:(r0 + R) * W]
  c = acts[l + 1][r0 * W:(r0 + R) * W]
  b = bits[l][r0 * W // 8:(r0 + R) * W // 8]
  ops.gemm_nt(a, Bts[l], M=R, N=W, K1=W, lda1=W, bias=biases[l], n_bias=W, relu=True, Cb=c, ldcb=W, nb=W, bits_out=b,
              a1_layout=PAN, c_layout=PAN)


def run(R, chunk_major=True):
  if chunk_major:
    for r0 in range(0, M, R):
      for l in range(L):
        layer(l, r0, R)
  else:                                   # the same launches, layer-major: a chunk is read long after it was written
    for l in range(L):
      for r0 in range(0, M, R):
        layer(l, r0, R)


def timed(fn, reps=3):
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


ref = None
for R, cm in ((M, True), (131072, True), (131072, False), (65536, True), (65536, False), (32768, True), (32768, False), (16384, True), (16384, False),
              (65536, True), (65536, False), (M, True)):
  t = timed(lambda: run(R, cm))
  out = acts[L].clone()
  if ref is None:
    ref = out
  same = torch.equal(out.view(torch.int16), ref.view(torch.int16))
  print(f'{"chunk-major" if cm else "layer-major"} R = {R:7d} rows ({R * W * 2 / 2**20:6.0f} MiB per activation chunk, {M // R * L:4d} launches): {t / L:8.1f} us per layer of {M} rows '
        f'({2.0 * M * W * W * L / t / 1e6:6.1f} TF/s)  {"bitwise equal" if same else "MISMATCH"}', flush=True)
