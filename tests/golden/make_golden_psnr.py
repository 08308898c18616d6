"""Equal-step PSNR reference for the headline configuration: the ORACLE trains configs/360.gin AS IS (1024-wide NeRF MLP,
9.0 M parameters, levels 64/64/32) on the procedural unbounded scene for STEPS steps of RAYS rays, on the CPU, fp32.

    python tests/golden/make_golden_psnr.py [--seed S]  # ~10 min on 8 cores; writes tests/golden/psnr360.json (seed 360) or
                                                        # psnr360_s<S>.json; the committed seeds are 360, 361, 362
    python tests/golden/make_golden_psnr.py [--seed S] --dense_dtype bfloat16 | bf16_fwd_bwd
                                                        # the same run with the Dense operands rounded to bf16 (the oracle's
                                                        # emulation of the MFMA inputs): psnr360_bf16[_s<S>].json; what of the
                                                        # HIP - fp32-oracle difference is the precision the reference's own TPU
                                                        # default pays (internal/math.py:21-23)

tests/test_gpu_convergence.py::test_equal_step_psnr_360_full_width replays the same initialisation, the same batch and
the same jitter at every step through the HIP path and compares the PSNR on the same held-out rays (north_star: "PSNR
within 0.1 dB of reference at equal step count").  Everything random is a function of SEED and the step number only
(protocol(): shared by both sides), so nothing but this json has to travel to the GPU box.
"""

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

STEPS, RAYS, EVAL_RAYS, SEED = 600, 256, 2048, 360
BINDINGS = ['Config.max_steps = 600', 'Config.lr_delay_steps = 100', 'Config.batch_size = 256']


def protocol(model, cfg, step, seed=None):
  """(batch, noise, train_frac) of training step `step` (1-based), identical for the oracle and the HIP run."""
  from multinerf_amd import synthetic
  from oracle import bridge
  seed = SEED if seed is None else seed
  batch = synthetic.unbounded_scene_rays(RAYS, seed=seed * 100000 + step)
  noise = bridge.make_noise(model, RAYS, seed=seed * 100000 + step)
  train_frac = float(np.clip((step - 1) / (cfg.max_steps - 1), 0, 1))                 # train.py:118
  return batch, noise, train_frac


def eval_rays(seed=None):
  from multinerf_amd import synthetic
  seed = SEED if seed is None else seed
  return synthetic.unbounded_scene_rays(EVAL_RAYS, seed=seed * 100000 + 99999)


def golden_path(seed, bf16=False):
  """psnr360.json for the original seed, psnr360_s<seed>.json for the further ones (round 3: three seeds); psnr360_bf16*.json
  for the bf16-emulating oracle (round 4; bf16=True: forward operands, bf16='bf16_fwd_bwd': the backward pass's gradients too)."""
  stem = 'psnr360_bf16fb' if bf16 == 'bf16_fwd_bwd' else 'psnr360_bf16' if bf16 else 'psnr360'
  name = f'{stem}.json' if seed == SEED else f'{stem}_s{seed}.json'
  return os.path.join(ROOT, 'tests', 'golden', name)


def psnr(rgb, gt):
  return float(-10.0 / np.log(10.0) * np.log(np.mean((np.asarray(rgb, np.float64) - np.asarray(gt, np.float64)) ** 2)))


def main():
  seed = int(sys.argv[sys.argv.index('--seed') + 1]) if '--seed' in sys.argv else SEED
  dd_name = sys.argv[sys.argv.index('--dense_dtype') + 1] if '--dense_dtype' in sys.argv else None
  assert dd_name in (None, 'bfloat16', 'bf16_fwd_bwd'), dd_name
  dd = {None: None, 'bfloat16': torch.bfloat16, 'bf16_fwd_bwd': 'bf16_fwd_bwd'}[dd_name]
  from multinerf_amd import configs, models
  from oracle import bridge, models as omodels, train_utils as otrain
  torch.set_num_threads(int(os.environ.get('PSNR_THREADS', os.cpu_count())))
  cfg = configs.load_preset('360', BINDINGS)
  model = models.Model(config=cfg)
  om, on, op = bridge.oracle_hparams(model)
  params = omodels.init_params(om, on, op, seed=seed)
  st = otrain.init_opt_state(params)
  ev = eval_rays(seed)
  curve = []
  t0 = time.time()
  for step in range(1, STEPS + 1):
    batch, noise, tf = protocol(model, cfg, step, seed)
    params, st, stats, _ = otrain.train_step(params, st, om, on, op, cfg, batch, tf, noise=noise, dense_dtype=dd)
    if step % 50 == 0 or step == 1:
      with torch.no_grad():
        rend, _ = omodels.model_apply(om, on, op, params, ev.rays, 1.0, False, dense_dtype=dd)
      e = psnr(rend[-1]['rgb'].numpy(), ev.rgb.numpy())
      curve.append(dict(step=step, train_loss=float(stats['loss']), train_psnr=float(stats['psnr']), eval_psnr=e))
      print(f'step {step}: loss {float(stats["loss"]):.5f} train psnr {float(stats["psnr"]):.3f} eval psnr {e:.3f}  ({time.time() - t0:.0f} s)', flush=True)
  out = dict(steps=STEPS, rays=RAYS, eval_rays=EVAL_RAYS, seed=seed, bindings=BINDINGS, curve=curve,
             dense_dtype=dd_name or 'float32',
             note=('oracle (torch-CPU restatement of the reference' + (', Dense operands and backward gradients rounded to bf16' if dd == 'bf16_fwd_bwd' else ', Dense operands rounded to bf16' if dd else ', fp32') +
                   '), configs/360.gin as is, procedural unbounded scene'))
  with open(golden_path(seed, bf16=('bf16_fwd_bwd' if dd == 'bf16_fwd_bwd' else dd is not None)), 'w') as f:
    json.dump(out, f, indent=1)


if __name__ == '__main__':
  main()
