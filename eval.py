#!/usr/bin/env python
"""Evaluation script (reference eval.py:40-260, minus TensorBoard / SSIM / LPIPS): restore the latest
checkpoint, render the test set, report PSNR per image and on average, optionally save PNGs.

  python eval.py --gin_configs configs/blender_256.gin --gin_bindings "Config.data_dir = '...'" \
      --gin_bindings "Config.checkpoint_dir = '...'"
"""

import argparse
import json
import math
import os
import time

import numpy as np
import torch

from multinerf_amd import checkpoints, configs, datasets, models, train_utils
from multinerf_amd import dist as mdist


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gin_configs', action='append', default=[])
  ap.add_argument('--gin_bindings', action='append', default=[])
  ap.add_argument('--preset', default=None)
  args = ap.parse_args()
  mdist.init_from_env()
  rank, world = mdist.rank(), mdist.world_size()
  dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
  torch.cuda.set_device(dev)
  config = configs.load_preset(args.preset, args.gin_bindings) if args.preset else \
      configs.load_config(args.gin_configs, args.gin_bindings, save_config=False)
  dataset = datasets.load_dataset('test', config.data_dir, config, device=dev)
  model, state, render_eval_pfn, _, _ = train_utils.setup_model(config, 20200823, dataset=dataset, device=dev)
  if not config.checkpoint_dir or not os.path.isdir(config.checkpoint_dir):
    raise SystemExit(f'eval.py: Config.checkpoint_dir = {config.checkpoint_dir!r} is not a directory')
  if checkpoints.latest_checkpoint(config.checkpoint_dir) is None:
    # (the reference polls until train.py writes one, eval.py:92-104; evaluating random-init weights is never wanted)
    raise SystemExit(f'eval.py: no checkpoint in {config.checkpoint_dir}')
  state = checkpoints.restore_checkpoint(config.checkpoint_dir, model, state)
  step = int(state.step)
  if rank == 0:
    print(f'Evaluating checkpoint at step {step}.')
  out_dir = os.path.join(config.checkpoint_dir, 'test_preds') if config.checkpoint_dir else None
  if out_dir and config.eval_save_output and rank == 0:
    os.makedirs(out_dir, exist_ok=True)
  psnrs = []
  n = min(dataset.size, config.eval_dataset_limit)
  for idx in range(n):
    batch = next(dataset)
    t0 = time.time()
    rendering = models.render_image(lambda rng, r: render_eval_pfn(state.params, 1.0, None, r), batch.rays, None,
                                    config, verbose=False, world_size=world, rank=rank)
    torch.cuda.synchronize()
    if rank != 0:
      continue
    mse = float(((rendering['rgb'] - batch.rgb)**2).mean())
    psnr = -10. / math.log(10.) * math.log(max(mse, 1e-30))
    psnrs.append(psnr)
    print(f'Eval image {idx + 1}/{n}: {time.time() - t0:.3f}s, psnr {psnr:.3f}', flush=True)
    if out_dir and config.eval_save_output:
      from PIL import Image
      img = (rendering['rgb'].clamp(0, 1).cpu().numpy() * 255 + 0.5).astype(np.uint8)
      Image.fromarray(img).save(os.path.join(out_dir, f'color_{idx:03d}.png'))
  if rank == 0:
    print(f'Average test psnr over {len(psnrs)} images: {np.mean(psnrs):.3f}')
    if out_dir:
      with open(os.path.join(config.checkpoint_dir, f'metric_psnr_{step}.txt'), 'w') as f:
        f.write(' '.join(str(p) for p in psnrs))
  mdist.barrier()


if __name__ == '__main__':
  main()
