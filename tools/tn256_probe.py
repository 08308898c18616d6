"""The proposal MLP's weight-gradient GEMMs (2^21 rows, 256 x 256 and 512 x 256 outputs: ONE or TWO output tiles, every one of the
256 workgroups ends with 65536 fp32 atomics on the same 256 KiB): where the time goes (per-workgroup timeline; the same launch
without its atomics), fewer workgroups per launch, and the four launches of a step side by side on four streams with a quarter of
the chip each against one after the other on the whole chip."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import ops  # noqa: E402

dev, bf = 'cuda', torch.bfloat16
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


def timeline(fn):
  buf = torch.zeros((4096, 16), dtype=torch.int64, device=dev)
  ops.L.check(ops.L.debug().mnr_debug_gemm_timeline(buf.data_ptr()))
  fn()
  torch.cuda.synchronize()
  ops.L.check(ops.L.debug().mnr_debug_gemm_timeline(None))
  t = buf.cpu().numpy()
  t = t[t[:, 3] != 0].astype(np.float64)
  med = lambda x: float(np.median(x))
  return (f'{t.shape[0]} workgroups; cycles: prologue {med(t[:, 1] - t[:, 0]):.0f}  loop {med(t[:, 2] - t[:, 1]):.0f}  atomic epilogue {med(t[:, 3] - t[:, 2]):.0f} '
          f'(max {float((t[:, 3] - t[:, 2]).max()):.0f});  100 MHz ticks start -> end: median {med(t[:, 5] - t[:, 4]):.0f}, last end - first start '
          f'{float(t[:, 5].max() - t[:, 4].min()):.0f}')


M, W, F = 1 << 21, 256, 512
feat = torch.randn((M, F), generator=g, device=dev).to(bf)
acts = [torch.relu(torch.randn((M, W), generator=g, device=dev)).to(bf) for _ in range(3)]
dYs = [(torch.randn((M, W), generator=g, device=dev) * (torch.rand((M, W), generator=g, device=dev) > 0.5)).to(bf) for _ in range(4)]
gh = 0.1 * torch.randn((M,), generator=g, device=dev)
wh = torch.randn((W,), generator=g, device=dev)
bits = torch.randint(0, 256, (M, W // 8), generator=g, device=dev, dtype=torch.uint8)
Cs = [torch.zeros((F if i == 0 else W, W), device=dev) for i in range(4)]
bs = [torch.zeros((W,), device=dev) for i in range(4)]


def dw(i, cap=0, rank1=True, kv=None):
  A, K = (feat, F) if i == 0 else (acts[i - 1], W)
  kw = dict(M=M, K=K, N=W, bias_out=bs[i], bias_n_valid=W, max_wgs=cap)
  if kv is not None:
    kw.update(k_valid=kv, n_valid=kv)
  if i == 3 and rank1:
    ops.gemm_tn(A, None, Cs[i], rank1=(gh, wh, bits), **kw)
  else:
    ops.gemm_tn(A, dYs[i], Cs[i], **kw)


for name, fn in (('K = 256, B stored', lambda: dw(2)), ('K = 256, B stored, no atomics on C', lambda: dw(2, kv=0)),
                 ('K = 256, rank1', lambda: dw(3)), ('K = 256, rank1, no atomics on C', lambda: dw(3, kv=0)),
                 ('K = 512 (features), B stored', lambda: dw(0)), ('K = 512, no atomics on C', lambda: dw(0, kv=0))):
  print(f'{name:40s} {timed(fn):8.1f} us   {timeline(fn)}', flush=True)
for cap in (128, 64, 32):
  print(f'max_wgs = {cap:3d}: K = 256 stored {timed(lambda: dw(2, cap)):8.1f} us, rank1 {timed(lambda: dw(3, cap)):8.1f} us, K = 512 {timed(lambda: dw(0, cap)):8.1f} us', flush=True)

streams = [torch.cuda.Stream(device=dev) for _ in range(4)]


def four(side_by_side, cap):
  if not side_by_side:
    for i in range(4):
      dw(i, cap)
    return
  cur = torch.cuda.current_stream()
  ev = torch.cuda.Event()
  ev.record(cur)
  for i, s in enumerate(streams):
    s.wait_event(ev)
    with torch.cuda.stream(s):
      dw(i, cap)
    e = torch.cuda.Event()
    e.record(s)
    cur.wait_event(e)


for rep in range(2):
  print(f'the four launches of a step: one after the other {timed(lambda: four(False, 0)):8.1f} us;  side by side on four streams: '
        f'64 workgroups each {timed(lambda: four(True, 64)):8.1f} us, 128 each {timed(lambda: four(True, 128)):8.1f} us, uncapped {timed(lambda: four(True, 0)):8.1f} us', flush=True)
