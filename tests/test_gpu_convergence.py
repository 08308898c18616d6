"""Equal-step training on a procedural scene: HIP path vs the CPU oracle (SURVEY.md 8d 'PSNR').

Both start from the same parameters and see the same ray batches and the same sampling jitter
for every step; the HIP path computes its Dense layers in bf16 x bf16 -> fp32, the oracle in fp32.
The north-star asks for test PSNR within 0.1 dB at equal step count (on real scenes, which are
not available offline); this is the procedural stand-in.  -m gpu.
"""

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multinerf_amd import configs, models, train_utils
from oracle import models as omodels
from oracle import train_utils as otrain
from tests import helpers

STEPS = 150
B = 256
BINDINGS = [
    'NerfMLP.net_width = 128', 'PropMLP.net_width = 128', 'NerfMLP.bottleneck_width = 128',
    'Model.num_prop_samples = 64', 'Model.num_nerf_samples = 32',
    f'Config.max_steps = {STEPS}', 'Config.lr_delay_steps = 0',
]


def _psnr(model_rgb, gt):
  mse = float(((model_rgb - gt)**2).mean())
  return -10.0 / math.log(10.0) * math.log(mse)        # image.mse_to_psnr, image.py:28-30


@pytest.mark.parametrize('extra', [[], ['Model.stop_level_grad = False', 'Model.resample_padding = 0.01']],
                         ids=['stop_level_grad', 'through_the_sampling'])
def test_equal_step_psnr_matches_oracle(extra):
  """`through_the_sampling`: the same protocol with Model.stop_level_grad = False (models.py:198-201; resample_padding 0.01 as
  blender_refnerf.gin sets it: with 0 a closed dilated bin makes the REFERENCE's gradient NaN): 150 steps in
  which every step's gradient went through the sampling VJPs must train as the oracle's do."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
  cfg = configs.load_preset('blender_256', BINDINGS + extra)
  model = models.Model(config=cfg)
  model.build('cuda')
  om, on, op = helpers.oracle_hparams(model)
  params = omodels.init_params(om, on, op, seed=7)
  flat = model.flat_from_tree(params)
  test_batch = helpers.procedural_scene_rays(2048, seed=999)

  def eval_psnr_hip(flat_now):
    rend, _ = model.apply({'flat': flat_now}, None, test_batch.rays.map(lambda t: t.cuda()), 1.0, False)
    return _psnr(rend[-1]['rgb'].cpu(), test_batch.rgb)

  def eval_psnr_oracle(p):
    with torch.no_grad():
      rend, _ = omodels.model_apply(om, on, op, p, test_batch.rays, 1.0, False)
    return _psnr(rend[-1]['rgb'], test_batch.rgb)

  psnr0 = eval_psnr_hip(flat)
  assert abs(psnr0 - eval_psnr_oracle(params)) < 0.1

  state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone(), 'params': None})
  step_fn = train_utils.create_train_step(model, cfg)
  ost = otrain.init_opt_state(params)
  p_o = params
  hist = []
  for it in range(STEPS):
    batch = helpers.procedural_scene_rays(B, seed=1000 + it)
    noise = helpers.make_noise(model, B, seed=it)
    tf = it / max(1, STEPS - 1)
    state, stats, _ = step_fn(0, state, batch.map(lambda t: t.cuda()), None, tf, 0.0, noise=noise)
    p_o, ost, stats_o, _ = otrain.train_step(p_o, ost, om, on, op, cfg, batch, tf, noise=noise)
    if it % 25 == 0 or it == STEPS - 1:
      s = stats.materialize()
      hist.append((it, s['loss'], float(stats_o['loss'])))
      print(f'step {it:4d}: loss hip {s["loss"]:.5f} oracle {float(stats_o["loss"]):.5f}')
  psnr_hip = eval_psnr_hip(state.params['flat'])
  psnr_or = eval_psnr_oracle(p_o)
  print(f'test PSNR after {STEPS} steps: hip {psnr_hip:.3f} dB, oracle {psnr_or:.3f} dB, start {psnr0:.3f} dB, '
        f'diff {psnr_hip - psnr_or:+.3f} dB')
  assert psnr_hip > psnr0 + 3.0, 'training did not reduce the test error'
  # dW accumulates with fp32 atomics (order varies run to run) on top of the bf16 / fp32 difference: the
  # trajectories are not bit-reproducible; observed |diff| 0.05 - 0.3 dB at this size.
  assert abs(psnr_hip - psnr_or) <= 0.5


def test_equal_step_psnr_360_full_width():
  """The headline configuration itself: configs/360.gin as is (1024-wide NeRF MLP, 9.0 M parameters, levels 64/64/32) on
  the procedural UNBOUNDED scene (multinerf_amd.synthetic.unbounded_scene_rays: content inside the unit ball, a ground
  plane running through the contracted region to the horizon, sky at infinity behind the opaque last interval), 600
  steps of 256 rays from the oracle's initialisation with the oracle's batches and jitter at every step.  The oracle's
  side ran on the CPU ahead of time (tests/golden/make_golden_psnr.py [--seed S] -> tests/golden/psnr360*.json); here the
  HIP path replays the protocol and must land within 0.1 dB of the oracle's held-out PSNR at equal step count
  (north_star).  Up to five seeds (initialisation, batches and jitter all differ), each replayed MNR_PSNR_REPEATS = 8 times
  (the fp32 atomics of the weight gradients make every replay a different trajectory), against three oracle runs per seed: plain
  fp32, bf16 forward operands, and bf16 operands in the forward AND the backward matmuls (the reference's TPU default precision);
  every run's difference, the seed means and the grand means are printed.  Asserted: the grand mean of the SIGNED differences
  against the reference-precision oracle within 0.1 dB (no error bars subtracted), every seed's mean within 0.3 dB, the grand
  mean against the fp32 oracle within 0.1 dB at two standard errors, no single run further than 0.5 dB off (see the comment at the assertions)."""
  import importlib.util
  import json
  import os
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  here = os.path.dirname(os.path.abspath(__file__))
  spec = importlib.util.spec_from_file_location('make_golden_psnr', os.path.join(here, 'golden', 'make_golden_psnr.py'))
  G = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(G)
  seeds = [sd for sd in range(G.SEED, G.SEED + 5) if os.path.exists(G.golden_path(sd))]
  assert G.SEED in seeds
  cfg = configs.load_preset('360', G.BINDINGS)
  model = models.Model(config=cfg)
  model.build('cuda')
  assert model.nerf_plan.W == 1024 and model.num_params == 9007493
  om, on, op = helpers.oracle_hparams(model)
  out = os.environ.get('MNR_PSNR_LOG')
  repeats = int(os.environ.get('MNR_PSNR_REPEATS', '8'))
  all_rows, finals, tails, finals_bf, tails_bf, finals_fb, tails_fb = [], {}, {}, {}, {}, {}, {}
  for seed, rep in [(sd, r) for sd in seeds for r in range(repeats)]:
    ref = json.load(open(G.golden_path(seed)))
    assert ref['steps'] == G.STEPS and ref['rays'] == G.RAYS and ref['seed'] == seed and ref['bindings'] == G.BINDINGS
    # the same protocol through the oracle with the Dense operands rounded to bf16 (round 4: what of the difference is precision)
    ref_bf = json.load(open(G.golden_path(seed, bf16=True))) if os.path.exists(G.golden_path(seed, bf16=True)) else None
    want_bf = {c['step']: c for c in ref_bf['curve']} if ref_bf else {}
    # ... and with the backward matmuls' incoming gradients rounded as well (oracle.models.BF16_FWD_BWD: every Dense matmul of the
    # reference's TPU default precision, forward and backward; also what the HIP path stores: dY in bf16)
    p_fb = G.golden_path(seed, bf16='bf16_fwd_bwd')
    want_fb = {c['step']: c for c in json.load(open(p_fb))['curve']} if os.path.exists(p_fb) else {}
    flat = model.flat_from_tree(omodels.init_params(om, on, op, seed=seed))
    ev = G.eval_rays(seed)
    ev_rays = ev.rays.map(lambda t: t.cuda())
    state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone(), 'params': None})
    step_fn = train_utils.create_train_step(model, cfg)
    want = {c['step']: c for c in ref['curve']}
    rows = []
    for step in range(1, G.STEPS + 1):
      batch, noise, tf = G.protocol(model, cfg, step, seed)
      state, stats, _ = step_fn(0, state, batch.map(lambda t: t.cuda()), None, tf, 0.0, noise=noise)
      if step in want:
        rend, _ = model.apply({'flat': state.params['flat']}, None, ev_rays, 1.0, False)
        e = G.psnr(rend[-1]['rgb'].cpu().numpy(), ev.rgb.numpy())
        s = stats.materialize()
        rows.append(dict(seed=seed, replay=rep, step=step, hip_eval_psnr=e, oracle_eval_psnr=want[step]['eval_psnr'], hip_train_loss=s['loss'],
                         oracle_train_loss=want[step]['train_loss'],
                         oracle_bf16_eval_psnr=want_bf[step]['eval_psnr'] if step in want_bf else None,
                         oracle_bf16fb_eval_psnr=want_fb[step]['eval_psnr'] if step in want_fb else None))
        print(f'seed {seed} replay {rep} step {step:4d}: eval PSNR hip {e:.3f} oracle {want[step]["eval_psnr"]:.3f} ({e - want[step]["eval_psnr"]:+.3f} dB); '
              f'train loss hip {s["loss"]:.5f} oracle {want[step]["train_loss"]:.5f}')
    first, last = rows[0], rows[-1]
    assert abs(first['hip_eval_psnr'] - first['oracle_eval_psnr']) < 0.05                  # same start
    assert last['oracle_eval_psnr'] > first['oracle_eval_psnr'] + 5.0, 'the reference run did not learn the scene'
    finals[(seed, rep)] = last['hip_eval_psnr'] - last['oracle_eval_psnr']
    if last['oracle_bf16_eval_psnr'] is not None:
      finals_bf[(seed, rep)] = last['hip_eval_psnr'] - last['oracle_bf16_eval_psnr']
      tails_bf[(seed, rep)] = float(np.mean([r['hip_eval_psnr'] - r['oracle_bf16_eval_psnr'] for r in rows[-3:]]))
    if last['oracle_bf16fb_eval_psnr'] is not None:
      finals_fb[(seed, rep)] = last['hip_eval_psnr'] - last['oracle_bf16fb_eval_psnr']
      tails_fb[(seed, rep)] = float(np.mean([r['hip_eval_psnr'] - r['oracle_bf16fb_eval_psnr'] for r in rows[-3:]]))
    # the mean over the last three checkpoints averages out the step-to-step wobble of either trajectory
    tails[(seed, rep)] = float(np.mean([r['hip_eval_psnr'] - r['oracle_eval_psnr'] for r in rows[-3:]]))
    print(f'equal-step PSNR, 360.gin full width, seed {seed} replay {rep}: final diff {finals[(seed, rep)]:+.3f} dB, mean of the last '
          f'three checkpoints {tails[(seed, rep)]:+.3f} dB')
    all_rows += rows
  if out:
    with open(out, 'w') as f:
      for r in all_rows:
        f.write(json.dumps(r) + '\n')
  vals = np.array(list(finals.values()))
  tvals = np.array(list(tails.values()))
  n = len(vals)
  mean, tmean = float(vals.mean()), float(tvals.mean())
  se = float(vals.std(ddof=1) / np.sqrt(n)) if n > 1 else float('inf')
  tse = float(tvals.std(ddof=1) / np.sqrt(n)) if n > 1 else float('inf')
  seed_means = {sd: round(float(np.mean([v for (s_, _), v in finals.items() if s_ == sd])), 3) for sd in seeds}
  print(f'equal-step PSNR over seeds {seeds} x {repeats} replays: final diffs {[round(float(v), 3) for v in vals]} dB; seed means {seed_means}; '
        f'grand mean {mean:+.3f} +- {se:.3f} dB (standard error), mean |diff| {float(np.abs(vals).mean()):.3f} dB; last-three-checkpoint '
        f'means: grand mean {tmean:+.3f} +- {tse:.3f} dB')
  if finals_bf:
    vb, tb = np.array(list(finals_bf.values())), np.array(list(tails_bf.values()))
    seb = float(vb.std(ddof=1) / np.sqrt(len(vb))) if len(vb) > 1 else float('inf')
    # (the oracle is deterministic: ONE trajectory per seed and precision, so oracle_bf16 - oracle_fp32 is a single chaotic draw per seed)
    ob = {sd: round(float(np.mean([finals[k] - finals_bf[k] for k in finals_bf if k[0] == sd])), 3) for sd in seeds}
    print(f'equal-step PSNR against the bf16-EMULATING oracle: final diffs {[round(float(v), 3) for v in vb]} dB; grand mean {float(vb.mean()):+.3f} +- {seb:.3f} dB, '
          f'last-three-checkpoint grand mean {float(tb.mean()):+.3f} dB; oracle_bf16 - oracle_fp32 at step 600 per seed: {ob}')
  assert finals_fb, 'tests/golden/psnr360_bf16fb*.json missing (make_golden_psnr.py --dense_dtype bf16_fwd_bwd)'
  vf, tf_ = np.array(list(finals_fb.values())), np.array(list(tails_fb.values()))
  seeds_fb = sorted({k[0] for k in finals_fb})
  seed_means_fb = {sd: round(float(np.mean([v for (s_, _), v in finals_fb.items() if s_ == sd])), 3) for sd in seeds_fb}
  mean_fb, tmean_fb = float(vf.mean()), float(tf_.mean())
  ofb = {sd: round(float(np.mean([finals[k] - finals_fb[k] for k in finals_fb if k[0] == sd])), 3) for sd in seeds_fb}
  print(f'equal-step PSNR against the oracle at the reference\'s TPU default precision in BOTH passes (bf16 operands of every Dense matmul, '
        f'forward and backward): final diffs {[round(float(v), 3) for v in vf]} dB; seed means {seed_means_fb}; grand mean {mean_fb:+.3f} dB, '
        f'last-three-checkpoint grand mean {tmean_fb:+.3f} dB; that oracle minus the fp32 oracle at step 600 per seed: {ofb}')
  # What is asserted (round 4).  north_star: "PSNR within 0.1 dB of reference at equal step count".  The reference's Dense layers run
  # at jax's default precision (internal/models.py: nn.Dense without a precision argument; internal/math.py:21-23 raises it only for
  # its own matmul helper), which on its TPUs rounds the operands of every matmul, forward AND backward, to bf16.  Round 3 compared
  # with the fp32 oracle only and found a persistent -0.07 dB (13 triples), which it could not attribute.  Round 4 ran the oracle
  # at that precision (oracle.models.BF16_FWD_BWD; tests/golden/psnr360_bf16fb*.json) and the gap is the backward rounding: seed
  # 362, where the HIP path ends 0.18-0.31 dB under the fp32 oracle on every replay, the bf16 forward+backward oracle ends 0.23 dB
  # under it too (19.947 against 20.180; rounding the forward operands alone: 20.170), and on seeds 360 / 361 all four agree to
  # +-0.08 dB (profiles/r4e_psnr_s.log, r4g_psnr_s.log).  So:
  #   * the grand mean of the signed differences against the reference-precision oracle is held to 0.1 dB, PLAINLY (no standard
  #     errors subtracted; measured +0.040 over 15 runs, +0.038 over 25: profiles/r4g_, r4h_psnr360_equal_step.jsonl), and every
  #     seed's mean over its replays (eight since the end of round 4: the grand means' standard error drops from 0.016 to 0.013 dB, and
  #     the last-three-checkpoint mean has read +0.059 / +0.067) to 0.3 dB (seed 362's replays scatter by +-0.1 dB: its mean read +0.06 and +0.15);
  #   * against the plain fp32 oracle the grand mean is held to 0.1 dB at two standard errors (measured -0.053 +- 0.013 over five seeds x
  #     eight replays, -0.08 over the first three seeds: the precision cost of bf16 matmuls on this scene) and reported next to it;
  #   * one run may be 0.5 dB off (a 600-step run is chaotic and the weight gradients are summed with fp32 atomics in arrival
  #     order: one seed's difference moves by +-0.05 dB from replay to replay of the same binary).
  assert abs(mean_fb) <= 0.1 and abs(tmean_fb) <= 0.1, (mean_fb, tmean_fb)
  assert max(abs(v) for v in seed_means_fb.values()) <= 0.3, seed_means_fb
  # (the fp32-oracle gate at its earlier strength: within 0.1 dB at two standard errors; with fewer than five seeds
  # x eight replays the standard error is what it is)
  assert abs(mean) <= 0.1 + 2 * se and abs(tmean) <= 0.1 + 2 * tse, (mean, se, tmean, tse)
  assert float(np.abs(vals).max()) <= 0.5 and float(np.abs(vf).max()) <= 0.5, (finals, finals_fb)


def test_equal_step_psnr_360_full_width_fp32_mode():
  """The same protocol (configs/360.gin AS IS, 600 steps of 256 rays, the oracle's initialisation, batches and jitter) with the
  Dense layers in fp32 (models.Model.dense_precision = 'fp32': the fp32-Dense debug build, the reference's jax-cpu precision)
  against the PLAIN fp32 oracle's curve (tests/golden/psnr360*.json): one run per seed, three seeds by default (30 s each on the
  MI355X; MNR_PSNR_F32_SEEDS=5 for all five goldens: profiles/r6g_psnr_fp32_mode.txt).  The bf16 product's grand mean against
  this oracle is -0.05 dB, carried by one seed at -0.14 (the test above); if that is the precision of its matmuls and nothing
  else, this arm has no bias left: asserted |grand mean| <= 0.05 dB + two standard errors and every seed within 0.3 dB (measured on
  the MI355X: +0.064 / -0.072 / +0.025 / -0.078 / -0.122, grand mean -0.036 +- 0.035; a
  600-step run is chaotic: a ReLU unit that takes the other side of its kink, or the order of the weight gradients' fp32
  atomics, moves one trajectory by a few hundredths of a dB)."""
  import importlib.util
  import json
  import os
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  here = os.path.dirname(os.path.abspath(__file__))
  spec = importlib.util.spec_from_file_location('make_golden_psnr', os.path.join(here, 'golden', 'make_golden_psnr.py'))
  G = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(G)
  seeds = [sd for sd in range(G.SEED, G.SEED + int(os.environ.get('MNR_PSNR_F32_SEEDS', '3'))) if os.path.exists(G.golden_path(sd))]
  cfg = configs.load_preset('360', G.BINDINGS)
  model = models.Model(config=cfg, dense_precision='fp32').build('cuda')
  assert model.nerf_plan.W == 1024 and model.num_params == 9007493
  om, on, op = helpers.oracle_hparams(model)
  diffs = {}
  import time
  for seed in seeds:
    ref = json.load(open(G.golden_path(seed)))
    want = {c['step']: c for c in ref['curve']}
    flat = model.flat_from_tree(omodels.init_params(om, on, op, seed=seed))
    ev = G.eval_rays(seed)
    ev_rays = ev.rays.map(lambda t: t.cuda())
    state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone(), 'params': None})
    step_fn = train_utils.create_train_step(model, cfg)
    t0 = time.perf_counter()
    e = None
    for step in range(1, G.STEPS + 1):
      batch, noise, tf = G.protocol(model, cfg, step, seed)
      state, stats, _ = step_fn(0, state, batch.map(lambda t: t.cuda()), None, tf, 0.0, noise=noise)
      if step in want and (step == 1 or step % 200 == 0 or step == G.STEPS):
        rend, _ = model.apply({'flat': state.params['flat']}, None, ev_rays, 1.0, False)
        e = G.psnr(rend[-1]['rgb'].cpu().numpy(), ev.rgb.numpy())
        print(f'fp32 mode, seed {seed} step {step:4d}: eval PSNR hip {e:.3f} oracle_fp32 {want[step]["eval_psnr"]:.3f} ({e - want[step]["eval_psnr"]:+.3f} dB)')
        if step == 1:
          assert abs(e - want[step]['eval_psnr']) < 0.01
    diffs[seed] = e - want[G.STEPS]['eval_psnr']
    print(f'fp32 mode, seed {seed}: final diff {diffs[seed]:+.3f} dB ({time.perf_counter() - t0:.0f} s)')
  v = np.array(list(diffs.values()))
  se = float(v.std(ddof=1) / np.sqrt(len(v))) if len(v) > 1 else float('inf')
  print(f'equal-step PSNR, fp32-Dense mode against the plain fp32 oracle over seeds {seeds}: {[round(float(x), 3) for x in v]} dB; '
        f'grand mean {float(v.mean()):+.3f} +- {se:.3f} dB')
  assert abs(float(v.mean())) <= 0.05 + 2 * (se if len(v) > 1 else 0.0), (float(v.mean()), se)
  assert float(np.abs(v).max()) <= 0.3, diffs
