"""List the host <-> device synchronisation points of one train step (torch.cuda.set_sync_debug_mode('warn')): every one of
them stops the host from running ahead of the GPU.

    python tools/sync_probe.py [--preset 360]
"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import configs, synthetic, train_utils  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--preset', default='360')
ap.add_argument('--rays', type=int, default=16384)
a = ap.parse_args()
dev = torch.device('cuda', 0)
cfg = configs.load_preset(a.preset, [])
cfg.batch_size = a.rays
model, state, render_eval_pfn, step, _ = train_utils.setup_model(cfg, 0, device=dev)
batch = synthetic.synthetic_rays(a.rays, seed=1, near=cfg.near, far=cfg.far).map(lambda t: t.to(dev))
if cfg.compute_normal_metrics:
  batch.alphas = torch.rand((a.rays,), device=dev)
  batch.normals = torch.randn((a.rays, 3), device=dev)
gen = torch.Generator(device=dev).manual_seed(2)
for _ in range(3):
  state, stats, _ = step(gen, state, batch, None, 0.5, 0.0)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
  warnings.simplefilter('always')
  state, stats, _ = step(gen, state, batch, None, 0.5, 0.0)
  n_train = len(w)
  for x in w:
    print('train step:', str(x.message)[:160], '@', x.filename.split('/')[-1], x.lineno)
with warnings.catch_warnings(record=True) as w:
  warnings.simplefilter('always')
  render_eval_pfn(state.params, 1.0, None, batch.rays)
  for x in w:
    print('render    :', str(x.message)[:160], '@', x.filename.split('/')[-1], x.lineno)
  print(f'{n_train} synchronising calls in the train step, {len(w)} in the render call')
torch.cuda.set_sync_debug_mode('default')
