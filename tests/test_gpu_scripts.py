"""train.py / eval.py end to end on the procedural dataset (subprocess, one GPU).  -m gpu."""

import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_then_eval(tmp_path):
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  ck = str(tmp_path / 'ckpt')
  binds = ["Config.dataset_loader = 'procedural'", f"Config.checkpoint_dir = '{ck}'", 'Config.max_steps = 60',
           'Config.batch_size = 2048', 'Config.print_every = 20', 'Config.train_render_every = 60',
           'Config.checkpoint_every = 30', 'Config.lr_delay_steps = 0', 'Config.cast_rays_in_train_step = True',
           'NerfMLP.net_width = 128', 'PropMLP.net_width = 128', 'NerfMLP.bottleneck_width = 128',
           'Config.eval_dataset_limit = 2', 'Config.render_chunk_size = 4096']
  args = ['--preset', 'blender_256']
  for b in binds:
    args += ['--gin_bindings', b]
  env = dict(os.environ, PYTHONPATH=ROOT)
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py')] + args, capture_output=True, text=True, env=env,
                     timeout=600, cwd=ROOT)
  print(r.stdout[-2000:], r.stderr[-2000:])
  assert r.returncode == 0
  assert sorted(f for f in os.listdir(ck) if f.startswith('checkpoint_')) == ['checkpoint_1', 'checkpoint_30', 'checkpoint_60']
  log = [json.loads(l) for l in open(os.path.join(ck, 'train_log.jsonl'))]
  losses = [e['loss'] for e in log if 'loss' in e]
  assert len(losses) == 3 and losses[-1] < losses[0]
  assert any('test_psnr' in e for e in log)
  # resume: nothing left to do, exits cleanly from the restored step
  r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py')] + args, capture_output=True, text=True, env=env,
                      timeout=600, cwd=ROOT)
  assert r2.returncode == 0 and '60/60' not in r2.stdout
  r3 = subprocess.run([sys.executable, os.path.join(ROOT, 'eval.py')] + args, capture_output=True, text=True, env=env,
                      timeout=600, cwd=ROOT)
  print(r3.stdout[-1500:], r3.stderr[-1500:])
  assert r3.returncode == 0 and 'Evaluating checkpoint at step 60' in r3.stdout
  assert 'Average test psnr over 2 images' in r3.stdout
  assert os.path.exists(os.path.join(ck, 'test_preds', 'color_001.png'))


def test_bench_json_contract():
  """bench.py prints ONE JSON line with the driver's keys (+ roofline); cpu_baseline / aux legs are skipped here."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no_cpu_baseline',
                      '--no_aux'], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.strip()]
  assert len(lines) == 1
  d = json.loads(lines[0])
  for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
    assert k in d, k
  assert d['metric'] == 'train_rays_per_sec' and d['unit'] == 'rays/s' and d['n_gpus'] == 1 and d['steps'] == 2
  assert d['scaling'] == 'weak' and d['higher_is_better'] is True and d['vs_baseline'] is None
  assert '360.gin' in d['config']['workload'] and d['config']['global_batch'] == 16384
  rf = d['roofline']
  assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
  assert 0.05 < rf['frac'] < 1.0 and d['value'] > 1e5
  assert abs(d['value'] - 16384 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
