#!/usr/bin/env python
"""Per-workgroup timeline (mnr_debug_gemm_timeline) of selected NT GEMM launches INSIDE a real 360.gin train step:
real (ReLU-sparse) operands, real clocks.  Complements tools/nt_pipe_probe.py (dense random operands, one launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multinerf_amd import configs, ops, train_utils
from tests import helpers

dev = torch.device('cuda')
cfg = configs.load_preset('360')
B = 16384
model, state, _, step_fn, _ = train_utils.setup_model(cfg, 0, device=dev)
batch = helpers.synthetic_rays(B, near=cfg.near, far=cfg.far).map(lambda t: t.to(dev))
gen = torch.Generator(device=dev).manual_seed(1)
for _ in range(3):
  state, _, _ = step_fn(gen, state, batch, None, 0.5, 0.0)
torch.cuda.synchronize()

calls = []
orig = ops.gemm_nt
if os.environ.get('STEP_TIMELINE_NO_BITS_OUT'):        # timing experiment only (the backward then reads stale bits)
  _real = orig

  def orig(*a, **k):
    k['bits_out'] = None
    return _real(*a, **k)


def counting(*a, **k):
  calls.append((k.get('M'), k.get('N'), k.get('K1'), k.get('K2', 0), k.get('bits_out') is not None,
                k.get('bits_in') is not None, k.get('relu', False)))
  return orig(*a, **k)


ops.gemm_nt = counting
state, _, _ = step_fn(gen, state, batch, None, 0.5, 0.0)
torch.cuda.synchronize()
ops.gemm_nt = orig
print(f'{len(calls)} NT GEMM launches per step')
want = {}
for i, c in enumerate(calls):
  key = c
  if c[1] % 256 == 0 and c[0] % 256 == 0:
    want.setdefault(key, i)                       # first launch of every distinct (shape, epilogue) on the big tile
buf = torch.zeros((8192 * 2, 16), dtype=torch.int64, device=dev)
for key, idx in want.items():
  n = [0]

  def hooked(*a, _idx=idx, **k):
    on = n[0] == _idx
    n[0] += 1
    if on:
      buf.zero_()
      ops.L.check(ops.L.debug().mnr_debug_gemm_timeline(buf.data_ptr()))
    r = orig(*a, **k)
    if on:
      ops.L.check(ops.L.debug().mnr_debug_gemm_timeline(None))
    return r

  ops.gemm_nt = hooked
  state, _, _ = step_fn(gen, state, batch, None, 0.5, 0.0)
  torch.cuda.synchronize()
  ops.gemm_nt = orig
  t = buf.cpu().numpy()
  t = t[t[:, 3] != 0]
  if t.shape[0] == 0:
    continue                                      # (a launch that went to the weights-resident kernel: no stamps)
  tm = t[:, :4].astype(np.float64)
  rt = t[:, 4:6].astype(np.float64)
  tot = tm[:, 3] - tm[:, 0]
  wg_rt = rt[:, 1] - rt[:, 0]
  ghz = np.median(tot[wg_rt > 0] / wg_rt[wg_rt > 0]) * 100 / 1e3
  span = (rt[:, 1].max() - rt[:, 0].min()) / 100
  med = lambda x: float(np.median(x))
  M, N, K1, K2, bo, bi, relu = key
  flops = 2.0 * M * N * (K1 + K2)
  print(f'M={M} N={N} K={K1}+{K2} bits_out={int(bo)} bits_in={int(bi)} relu={int(relu)}: {len(t)} WGs, span {span:.0f} us '
        f'({flops / span / 1e6:.0f} TFLOP/s), clock {ghz:.2f} GHz; cycles/WG prologue {med(tm[:, 1] - tm[:, 0]):.0f} '
        f'K-loop {med(tm[:, 2] - tm[:, 1]):.0f} ({med(tm[:, 2] - tm[:, 1]) / ((K1 + K2) / 64):.0f}/K-tile) '
        f'epilogue {med(tm[:, 3] - tm[:, 2]):.0f}', flush=True)


# ---- the same for the weight-gradient (TN) launches
calls = []
orig_tn = ops.gemm_tn


def counting_tn(*a, **k):
  calls.append((k.get('M'), k.get('K'), k.get('N'), k.get('bias_out') is not None))
  return orig_tn(*a, **k)


ops.gemm_tn = counting_tn
state, _, _ = step_fn(gen, state, batch, None, 0.5, 0.0)
torch.cuda.synchronize()
ops.gemm_tn = orig_tn
print(f'{len(calls)} TN GEMM launches per step')
want = {}
for i, c in enumerate(calls):
  if c[1] % 128 == 0 and c[2] % 128 == 0:
    want.setdefault(c, i)
for key, idx in want.items():
  n = [0]

  def hooked_tn(*a, _idx=idx, **k):
    on = n[0] == _idx
    n[0] += 1
    if on:
      buf.zero_()
      ops.L.check(ops.L.debug().mnr_debug_gemm_timeline(buf.data_ptr()))
    r = orig_tn(*a, **k)
    if on:
      ops.L.check(ops.L.debug().mnr_debug_gemm_timeline(None))
    return r

  ops.gemm_tn = hooked_tn
  state, _, _ = step_fn(gen, state, batch, None, 0.5, 0.0)
  torch.cuda.synchronize()
  ops.gemm_tn = orig_tn
  t = buf.cpu().numpy()
  t = t[t[:, 3] != 0]
  tm = t[:, :4].astype(np.float64)
  rt = t[:, 4:6].astype(np.float64)
  steps = (t[:, 7] >> 32).astype(np.float64)
  isb = (t[:, 7] & 1).astype(bool)
  tot = tm[:, 3] - tm[:, 0]
  wg_rt = rt[:, 1] - rt[:, 0]
  ghz = np.median(tot[wg_rt > 0] / wg_rt[wg_rt > 0]) * 100 / 1e3
  span = (rt[:, 1].max() - rt[:, 0].min()) / 100
  loop = (tm[:, 2] - tm[:, 1]) / steps
  M, K, N, hb = key
  flops = 2.0 * M * N * K
  print(f'TN M={M} K={K} N={N} bias={int(hb)}: {len(t)} WGs, span {span:.0f} us ({flops / span / 1e6:.0f} TFLOP/s), clock {ghz:.2f} GHz; '
        f'steps/WG {np.median(steps):.0f}; cycles per 64-row step: plain {np.median(loop[~isb]):.0f}'
        + (f', with bias MFMAs {np.median(loop[isb]):.0f}' if isb.any() else '')
        + f'; prologue {np.median(tm[:, 1] - tm[:, 0]):.0f} epilogue {np.median(tm[:, 3] - tm[:, 2]):.0f}', flush=True)
