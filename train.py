#!/usr/bin/env python
"""Training script with the reference's flags and loop (train.py:40-290), one process per GPU.

  python train.py --gin_configs configs/blender_256.gin \
      --gin_bindings "Config.data_dir = '/data/nerf_synthetic/lego'" \
      --gin_bindings "Config.checkpoint_dir = '/tmp/lego'"
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...   # data parallel

Offline (no dataset in the image): --gin_bindings "Config.dataset_loader = 'procedural'".
TensorBoard is absent: the same statistics are printed and appended to <checkpoint_dir>/train_log.jsonl.
"""

import argparse
import json
import math
import os
import time

import numpy as np
import torch

from multinerf_amd import checkpoints, configs, datasets, models, train_utils, utils
from multinerf_amd import dist as mdist


def mse_to_psnr(mse):
  return -10. / math.log(10.) * math.log(max(mse, 1e-30))          # image.py:28-30


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gin_configs', action='append', default=[])
  ap.add_argument('--gin_bindings', action='append', default=[])
  ap.add_argument('--preset', default=None, help='a built-in copy of a reference config: ' + ', '.join(configs.PRESETS))
  args = ap.parse_args()

  mdist.init_from_env()
  rank, world = mdist.rank(), mdist.world_size()
  dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
  torch.cuda.set_device(dev)

  if args.preset:
    config = configs.load_preset(args.preset, args.gin_bindings)
  else:
    config = configs.load_config(args.gin_configs, args.gin_bindings, save_config=(rank == 0))
  if config.batch_size % world != 0:
    raise ValueError('Batch size must be divisible by the number of devices.')          # train.py:53
  global_batch = config.batch_size
  config.batch_size = global_batch // world                                              # datasets.py:256 per-process split
  dataset = datasets.load_dataset('train', config.data_dir, config, device=dev)
  dataset._gen.manual_seed(20200823 + rank)                                              # train.py:99-100
  test_dataset = datasets.load_dataset('test', config.data_dir, config, device=dev)
  cameras = dataset.cameras

  model, state, render_eval_pfn, train_pstep, lr_fn = train_utils.setup_model(config, 20200823, dataset=dataset, device=dev)
  if rank == 0:
    print(f'Number of parameters being optimized: {model.num_params}')
  if dataset.size > model.num_glo_embeddings and model.num_glo_features > 0:
    raise ValueError(f'Number of glo embeddings {model.num_glo_embeddings} must be at least equal to number of '
                     f'train images {dataset.size}')                                     # train.py:75-78
  if config.checkpoint_dir:
    os.makedirs(config.checkpoint_dir, exist_ok=True)
    state = checkpoints.restore_checkpoint(config.checkpoint_dir, model, state)
  init_step = state.step + 1
  log = open(os.path.join(config.checkpoint_dir, 'train_log.jsonl'), 'a') if (config.checkpoint_dir and rank == 0) else None

  gen = torch.Generator(device=dev).manual_seed(20200823 + rank)
  num_steps = config.early_exit_steps if config.early_exit_steps is not None else config.max_steps
  stats_buffer, train_start, total_time, total_steps = [], time.time(), 0.0, 0
  for step in range(init_step, num_steps + 1):
    batch = next(dataset)
    train_frac = float(np.clip((step - 1) / (config.max_steps - 1), 0, 1))              # train.py:118
    logging_step = step % config.print_every == 0 or step == num_steps
    # (the per-key statistics of train_utils.py:304,323-335 -- weight_l2s, grad_norms / maxes, opt_update_norms / maxes -- on the
    # steps that are logged: they cost two copies of the parameter vector)
    state, stats, gen = train_pstep(gen, state, batch, cameras, train_frac, 1.0, tree_stats=logging_step)
    stats_buffer.append(stats)
    if step % config.print_every == 0 or step == num_steps:                              # train.py:141-216
      # train.py:150-186: the logged numbers are AVERAGES over the steps since the last print
      mats = [b.materialize() for b in stats_buffer]
      s = dict(mats[-1])
      s['loss'] = float(np.mean([m['loss'] for m in mats]))
      s['psnr'] = float(np.mean([m['psnr'] for m in mats]))
      s['losses'] = {k: float(np.mean([m['losses'][k] for m in mats])) for k in mats[-1]['losses']}
      torch.cuda.synchronize()
      elapsed = time.time() - train_start
      steps_done = len(stats_buffer)
      rays_per_sec = global_batch * steps_done / elapsed
      total_time += elapsed
      total_steps += steps_done
      if rank == 0:
        msg = (f'{step}/{num_steps}: loss={s["loss"]:.5f}, psnr={s["psnr"]:.3f}, lr={lr_fn(step):.2e} | '
               + ', '.join(f'{k}={v:.5f}' for k, v in s['losses'].items()) + f', {rays_per_sec:.0f} r/s')
        print(msg, flush=True)
        if log:
          tree = {k: {kk: float(vv) for kk, vv in s[k].items() if kk.count('/') == 0} for k in
                  ('weight_l2s', 'grad_norms', 'grad_maxes', 'opt_update_norms', 'opt_update_maxes') if isinstance(s.get(k), dict)}
          log.write(json.dumps(dict(step=step, loss=s['loss'], psnr=s['psnr'], lr=lr_fn(step), losses=s['losses'],
                                    train_rays_per_sec=rays_per_sec, **tree)) + '\n')
          log.flush()
      stats_buffer, train_start = [], time.time()
    if config.checkpoint_dir and rank == 0 and (step == 1 or step % config.checkpoint_every == 0):
      checkpoints.save_checkpoint(config.checkpoint_dir, model, state, int(step), keep=100)     # train.py:218-223
    if config.train_render_every > 0 and step % config.train_render_every == 0:         # train.py:225-283
      test_case = next(test_dataset)
      t0 = time.time()
      rendering = models.render_image(lambda rng, r: render_eval_pfn(state.params, train_frac, None, r),
                                      test_case.rays, None, config, verbose=False, world_size=world, rank=rank)
      torch.cuda.synchronize()
      if rank == 0:
        h, w = rendering['rgb'].shape[:2]
        psnr = mse_to_psnr(float(((rendering['rgb'] - test_case.rgb)**2).mean()))
        print(f'Eval {step}: {time.time() - t0:.3f}s, {h * w / (time.time() - t0):.0f} rays/sec, test psnr {psnr:.3f}',
              flush=True)
        if log:
          log.write(json.dumps(dict(step=step, test_psnr=psnr)) + '\n')
          log.flush()
  if config.checkpoint_dir and rank == 0 and config.max_steps % config.checkpoint_every != 0:
    checkpoints.save_checkpoint(config.checkpoint_dir, model, state, int(state.step), keep=100, overwrite=True)   # train.py:284-287
  mdist.barrier()


if __name__ == '__main__':
  main()
