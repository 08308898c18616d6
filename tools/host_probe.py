"""Host time per train step (Python + launches, no synchronisation inside the loop) next to the GPU time per step: when the first
is not well below the second, the step is launch-bound and the GPU waits for the host.

    python tools/host_probe.py [--preset blender_256]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import configs, synthetic, train_utils  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--preset', default='blender_256')
ap.add_argument('--rays', type=int, default=16384)
a = ap.parse_args()
dev = torch.device('cuda', 0)
cfg = configs.load_preset(a.preset, [])
cfg.batch_size = a.rays
model, state, _, step, _ = train_utils.setup_model(cfg, 0, device=dev)
batch = synthetic.synthetic_rays(a.rays, seed=1, near=cfg.near, far=cfg.far).map(lambda t: t.to(dev))
if cfg.compute_normal_metrics:
  batch.alphas = torch.rand((a.rays,), device=dev)
  batch.normals = torch.randn((a.rays, 3), device=dev)
gen = torch.Generator(device=dev).manual_seed(2)
for _ in range(5):
  state, stats, _ = step(gen, state, batch, None, 0.5, 0.0)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
  state, stats, _ = step(gen, state, batch, None, 0.5, 0.0)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'{a.preset}: host {1e3 * t_host / N:.2f} ms per step to enqueue, {1e3 * t_all / N:.2f} ms per step until the GPU is done')
