"""models._PAIR_DXDW on against off: one 360.gin train step at full width on 2048 rays (gradients must agree up to the order of the
weight-gradient atomics), then bench.py lines of both arms, twice, on one box."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multinerf_amd import configs, models, synthetic, train_utils  # noqa: E402

cfg = configs.load_preset('360')
model = models.Model(config=cfg).build('cuda')
flat = model.init_flat_params(seed=3)
B = 2048
batch = synthetic.synthetic_rays(B, near=cfg.near, far=cfg.far).map(lambda t: t.cuda())
gs = {}
for on in (False, True, False):
  models._PAIR_DXDW = on
  state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone(), 'params': None})
  gen = torch.Generator(device='cuda').manual_seed(5)
  _, stats, _ = train_utils.create_train_step(model, cfg)(gen, state, batch, None, 0.5, 0.0, return_grads=True)
  torch.cuda.synchronize()
  g = stats['_grads'].double()
  if on in gs:
    print(f'off vs off (atomics noise): {((g - gs[on]).norm() / g.norm()).item():.3e}')
  gs.setdefault(on, g)
rel = ((gs[True] - gs[False]).norm() / gs[False].norm()).item()
print(f'pair on vs off: |dg| / |g| = {rel:.3e}, finite {bool(torch.isfinite(gs[True]).all())}')
assert rel < 1e-3
del model, flat, batch, gs
torch.cuda.empty_cache()
for arm in (0, 1, 0, 1):
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no_cpu_baseline', '--no_aux', '--pair_dxdw', str(arm)],
                     capture_output=True, text=True, env=dict(os.environ, MNR_SKIP_PREFLIGHT='1'))
  try:
    b = json.loads(r.stdout.strip().splitlines()[-1])
    print(f'pair_dxdw={arm}: {b["value"]:.0f} rays/s {b["ms_per_step"]:.3f} ms  mfma union {b["roofline"]["gemm_ms_per_step"]:.2f} ms  final_loss {b["config"]["final_loss"]:.7f}', flush=True)
  except Exception as e:
    print('arm', arm, 'failed', e, r.stderr[-500:])
