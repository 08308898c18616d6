#!/usr/bin/env python
"""Benchmark of the MultiNeRF hot path on MI355X: training rays/s on configs/360.gin.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full train_step (forward of the 3 sampling levels, losses, backward,
per-module clip, Adam) on one batch of synthetic rays already resident in HBM
(SURVEY.md 8d inputs; random-init weights; real jitter).  Data parallel: each rank
owns `batch_size` rays (weak scaling), one RCCL all-reduce of the flat fp32 gradient
per step.  Rank 0 prints ONE JSON line.

roofline: the dominant kernels are the bf16 MFMA kernels (gemm_nt_kernel forward/dX,
gemm_tn_kernel dW, the fused per-level chain kernels of the proposal MLP).  achieved = algorithmic training FLOPs of one step (SURVEY.md 8d:
1815.994 MFLOP/ray at 360.gin x rays per launch-set) / summed duration of those
kernels within the step, measured with HIP events on the launch stream during the
timed region (a second, identical, instrumented pass so the events do not perturb
`value`).  peak = 2500 TFLOP/s dense bf16 (MI355X_MICROARCH.md).

cpu_baseline: the torch-CPU oracle ("port" of the reference's jax-cpu path: JAX is not
installable here) running the same train_step on a bounded sample, host cores stated.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np
import torch


def algorithmic_flops_per_ray(model):
  """Dense MACs only. Returns (forward, training) FLOPs per ray."""
  fwd = 0.0
  train = 0.0
  for (i, is_prop, n, _) in model._level_plan():
    plan = model.prop_plan if is_prop else model.nerf_plan
    macs = 0
    no_dx = 0           # MACs whose input needs no gradient (first layer, skip-concat features, view encoding)
    for li, (d, concat) in enumerate(plan.trunk):
      macs += d.fan_in * d.fan_out
      if li == 0:
        no_dx += d.fan_in * d.fan_out
      elif concat:
        no_dx += plan.F * d.fan_out
    macs += plan.density.fan_in
    if plan.has_rgb and plan.bottleneck is None:       # use_viewdirs = False: rgb straight off the trunk
      macs += plan.rgb.fan_in * plan.rgb.fan_out
    elif plan.has_rgb:
      macs += plan.bottleneck.fan_in * plan.bottleneck.fan_out
      for li, (d, concat) in enumerate(plan.view):
        macs += d.fan_in * d.fan_out
        if li == 0:
          no_dx += (plan.vi_width - plan.hp.bottleneck_width) * d.fan_out
      macs += plan.rgb.fan_in * plan.rgb.fan_out
    extra_fwd = extra_train = 0
    if plan.ref:
      # Ref-NeRF: small heads, the view MLP's skip input needs dX, and the forward-mode tangent network for
      # the density-gradient normals (3 tangent rows per sample through the trunk; its backward = dW + dX).
      for d in (plan.gradpred, plan.diffuse, plan.tint, plan.rough):
        macs += d.fan_in * d.fan_out
      no_dx -= (plan.vi_width - plan.hp.bottleneck_width) * plan.view[0][0].fan_out   # IDE / n.v columns do need dX
      tmacs = sum(d.fan_in * d.fan_out for d, _ in plan.trunk) + plan.density.fan_in
      t_no_dx = plan.trunk[0][0].fan_in * plan.trunk[0][0].fan_out + sum(plan.F * d.fan_out for d, c in plan.trunk if c)
      extra_fwd = 3 * tmacs
      extra_train = 3 * (3 * tmacs - t_no_dx)
    fwd += 2.0 * n * (macs + extra_fwd)
    train += 2.0 * n * (3 * macs - no_dx + extra_train)
  return fwd, train


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--batch_size', type=int, default=16384, help='rays per GPU (Config.batch_size of 360.gin)')
  ap.add_argument('--global_batch', type=int, default=0,
                  help='total rays per step over all GPUs (overrides --batch_size: rays per GPU = global_batch / gpus, "strong" '
                       'scaling); BASELINE config 3 is --gpus 8 --global_batch 65536')
  ap.add_argument('--preset', default='360')
  ap.add_argument('--gin_bindings', action='append', default=[])
  ap.add_argument('--no_cpu_baseline', action='store_true')
  ap.add_argument('--cpu_rays', type=int, default=256, help='rays of the CPU-baseline sample (SURVEY 8d: B = 256)')
  ap.add_argument('--check_collectives', action='store_true',
                  help='before the benchmark: all-reduce / all-gather known patterns over the process group, assert the results '
                       'and time a 36 MB fp32 all-reduce (first contact with RCCL must not be the first bug)')
  ap.add_argument('--no_aux', action='store_true', help='skip the secondary measurements (4096x192 north-star shape, render)')
  ap.add_argument('--pair_dxdw', type=int, default=-1, help='A/B: models._PAIR_DXDW (1 on, 0 off; default: the module\'s own setting)')
  ap.add_argument('--tangent_chain', type=int, default=-1, help='A/B: models._TANGENT_CHAIN (1 on, 0 off; default: the module\'s own setting)')
  ap.add_argument('--head_k32', type=int, default=-1, help='A/B: models._HEAD_K32')
  ap.add_argument('--rank1_last', type=int, default=-1, help='A/B: models._RANK1_LAST (1 on, 0 off; default: the module\'s own setting)')
  args = ap.parse_args()

  # torch-only preflight in a child process before this process creates a HIP context (multinerf_amd/preflight.py): a box
  # whose first host -> device copy aborts is then on record as `BOX_FAULT: ...` instead of a core dump of the benchmark
  pre = None
  # (on EVERY local rank: the HSA_ENABLE_SDMA=0 workaround lives in the environment of the process that found it needed)
  if os.environ.get('MNR_SKIP_PREFLIGHT') != '1':
    from multinerf_amd import preflight
    pre = preflight.check(verbose=False)
    if not pre['ok']:
      raise SystemExit('BOX_FAULT: torch-only preflight failed (lines above); libmnerf_hip.so was never loaded')

  from multinerf_amd import configs, dist as mdist, models, ops, synthetic, train_utils
  from multinerf_amd import streams as mstreams

  if args.pair_dxdw >= 0:
    models._PAIR_DXDW = bool(args.pair_dxdw)
  if args.rank1_last >= 0:
    models._RANK1_LAST = bool(args.rank1_last)
  if args.head_k32 >= 0:
    models._HEAD_K32 = bool(args.head_k32)
  if args.tangent_chain >= 0:
    models._TANGENT_CHAIN = bool(args.tangent_chain)
  mdist.init_from_env()
  rank, world = mdist.rank(), mdist.world_size()
  if world != args.gpus:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run')
  local = int(os.environ.get('LOCAL_RANK', '0'))
  ndev = torch.cuda.device_count()
  if ndev < 1:
    raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
  if local >= ndev:
    raise SystemExit(f'LOCAL_RANK {local} but only {ndev} visible GPU(s): one process per GPU on ONE node')
  torch.cuda.set_device(local)
  dev = torch.device('cuda', local)
  dist_info = mdist.describe(dev)
  if world > 1:
    # every rank must sit on its own device: gather (hostname, device index, PCI bus id) and fail loudly on a clash
    ids = mdist.all_gather_objects((os.uname().nodename, local, dist_info['device_uuid']))
    if len(set(ids)) != world:
      raise SystemExit(f'rank / device mismatch: {ids}')
    dist_info['rank_devices'] = [list(i) for i in ids]
  if args.check_collectives or (world > 1 and os.environ.get('MNR_SKIP_COLLECTIVE_CHECK') != '1'):
    dist_info['collective_check'] = mdist.check_collectives(dev)

  cfg = configs.load_preset(args.preset, args.gin_bindings)
  if args.global_batch:
    if args.global_batch % world:
      raise SystemExit(f'--global_batch {args.global_batch} is not a multiple of --gpus {world}')
    args.batch_size = args.global_batch // world
  cfg.batch_size = args.batch_size
  model, state, render_eval_pfn, train_pstep, lr_fn = train_utils.setup_model(cfg, 0, device=dev)
  B = args.batch_size
  batch = synthetic.synthetic_rays(B, seed=20200823 + rank, near=cfg.near, far=cfg.far).map(lambda t: t.to(dev))
  gen = torch.Generator(device=dev).manual_seed(1234 + rank)
  if cfg.compute_normal_metrics:      # blender ground-truth normals / alphas (synthetic)
    batch.alphas = torch.rand((B,), generator=gen, device=dev)
    batch.normals = torch.randn((B, 3), generator=gen, device=dev)
  if cfg.compute_disp_metrics:
    batch.disps = torch.rand((B,), generator=gen, device=dev)
  if cfg.rawnerf_mode:                # RawNeRF: per-ray exposure index / value, Bayer-mask lossmult
    batch.rays.exposure_idx = torch.randint(0, 5, (B, 1), generator=gen, device=dev).to(torch.int32)
    batch.rays.exposure_values = 0.5 + torch.rand((B, 1), generator=gen, device=dev)
    batch.rays.lossmult = (torch.rand((B, 3), generator=gen, device=dev) > 0.4).float()
  train_frac = 0.5

  def step():
    nonlocal state
    state, stats, _ = train_pstep(gen, state, batch, None, train_frac, 0.0)
    return stats

  for _ in range(args.warmup):
    stats = step()
  torch.cuda.synchronize()
  mdist.barrier()
  torch.cuda.synchronize()
  # per-step HIP events on the launch stream (median of per-step hipEvent times); `value` stays the wall clock
  # over exactly K steps between the two barrier + synchronize brackets
  evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
  t0 = time.perf_counter()
  evs[0].record()
  for i in range(args.steps):
    stats = step()
    evs[i + 1].record()
  torch.cuda.synchronize()
  mdist.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
  median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
  if world > 1:
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
  s = stats.materialize()

  # ---- roofline of the dominant kernels: instrumented pass (HIP events around every GEMM launch)
  fwd_flops, train_flops = algorithmic_flops_per_ray(model)
  ops.PROFILE.enable()
  nprof = max(1, min(3, args.steps))
  for _ in range(nprof):
    step()
  torch.cuda.synchronize()
  by_tag = ops.PROFILE.collect_by_tag()
  ops.PROFILE.disable()
  # (busy_ms: union of the launch intervals -- equal to their sum unless two streams run MFMA kernels side by side)
  gemm_ms, gemm_launches = by_tag.get('gemm', {}).get('busy_ms', 0.0), by_tag.get('gemm', {}).get('launches', 0)
  gemm_ms_sum = by_tag.get('gemm', {}).get('ms', 0.0)
  # secondary report: the bandwidth-bound kernels against the 8 TB/s HBM peak, algorithmic bytes
  hbm_kernels = {tag: {'ms_per_step': d['ms'] / nprof, 'launches_per_step': d['launches'] / nprof,
                       'algorithmic_GBps': d['bytes'] / (d['ms'] * 1e-3) / 1e9 if d['ms'] > 0 else None,
                       'frac_of_8TBps': d['bytes'] / (d['ms'] * 1e-3) / 8e12 if d['ms'] > 0 else None}
                 for tag, d in by_tag.items() if tag != 'gemm'}
  gemm_ms_per_step = gemm_ms / nprof
  achieved_tflops = train_flops * B / (gemm_ms_per_step * 1e-3) / 1e12

  # ---- secondary measurements (reported under "aux", never as `value`)
  aux = {}
  if not args.no_aux and args.preset == '360' and not args.gin_bindings:
    def timed(fn, warm, reps, rounds=3):
      """Seconds per call: `rounds` timed blocks of `reps` calls each (barrier + synchronize on both sides, max over ranks),
      the MEDIAN block (one 15-ms hiccup in a single block of ten 16-ms steps read as -10 % in round 3)."""
      for _ in range(warm):
        fn()
      blocks = []
      for _ in range(rounds):
        torch.cuda.synchronize()
        mdist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
          fn()
        torch.cuda.synchronize()
        mdist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
          t = torch.tensor([dt], device=dev, dtype=torch.float64)
          torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
          dt = float(t.item())
        blocks.append(dt / reps)
      return sorted(blocks)[len(blocks) // 2]
    # (i) render: deterministic forward with extras on one 16384-ray chunk per GPU (train_utils.py:377-396)
    rays_only = batch.rays
    dt = timed(lambda: render_eval_pfn(state.params, 1.0, None, rays_only), 2, 5)
    aux['render_rays_per_sec'] = {'value': B * world / dt, 'ms_per_chunk': 1e3 * dt, 'chunk_rays_per_gpu': B,
                                  'algorithmic_tflops_per_gpu': fwd_flops * B / dt / 1e12}
    # (ii) the north-star's synthetic shape: 4096 rays x 192 samples (360.gin with num_nerf_samples = 64)
    cfg_b = configs.load_preset('360', ['Model.num_nerf_samples = 64'])
    Bb = 4096
    cfg_b.batch_size = Bb
    model_b, state_b, _, step_b, _ = train_utils.setup_model(cfg_b, 0, device=dev)
    batch_b = synthetic.synthetic_rays(Bb, seed=20200823 + rank, near=cfg_b.near, far=cfg_b.far).map(lambda t: t.to(dev))
    def run_b():
      nonlocal state_b
      state_b, _, _ = step_b(gen, state_b, batch_b, None, train_frac, 0.0)
    dt = timed(run_b, 3, 10)
    fwd_b, train_b = algorithmic_flops_per_ray(model_b)
    aux['train_4096x192'] = {'value': Bb * world / dt, 'unit': 'rays/s', 'ms_per_step': 1e3 * dt,
                             'algorithmic_train_mflop_per_ray': train_b / 1e6,
                             'whole_step_frac_of_mfma_peak': train_b * Bb / dt / 2.5e15}
    del model_b, state_b, step_b, batch_b
    # (iii) the per-rank shards of BASELINE configs 3 and 5 on ONE GPU: configs/360.gin at 65536 / 8 = 8192 rays, configs/llff_raw.gin
    # at 16384 / 4 = 4096 rays.  No N > 1 node is available to the builder, so these bound strong-scaling efficiency before any
    # xGMI cost: a rank that runs its shard at x of the 16384-ray rate cannot scale better than x.
    def shard_rate(preset, rays, unit_rays):
      cfg_s = configs.load_preset(preset, [])
      cfg_s.batch_size = rays
      model_s, state_s, _, step_s, _ = train_utils.setup_model(cfg_s, 0, device=dev)
      b_s = synthetic.synthetic_rays(rays, seed=20200823 + rank, near=cfg_s.near, far=cfg_s.far).map(lambda t: t.to(dev))
      if cfg_s.rawnerf_mode:
        b_s.rays.exposure_idx = torch.randint(0, 5, (rays, 1), generator=gen, device=dev).to(torch.int32)
        b_s.rays.exposure_values = 0.5 + torch.rand((rays, 1), generator=gen, device=dev)
        b_s.rays.lossmult = (torch.rand((rays, 3), generator=gen, device=dev) > 0.4).float()
      st = [state_s]
      def run_s():
        st[0], _, _ = step_s(gen, st[0], b_s, None, train_frac, 0.0)
      dt_s = timed(run_s, 3, 10)
      return {'value': rays * world / dt_s, 'unit': 'rays/s', 'ms_per_step': 1e3 * dt_s, 'rays_per_gpu': rays, 'of_ranks': unit_rays // rays}
    aux['shard_360_8192'] = shard_rate('360', 8192, 65536)
    aux['shard_360_8192']['fraction_of_the_16384_ray_rate'] = aux['shard_360_8192']['value'] / (B * world / (elapsed / args.steps)) if B == 16384 else None
    aux['shard_llff_raw_4096'] = shard_rate('llff_raw', 4096, 16384)

  # HBM bytes of the GEMM kernels per train step from the PMC passes of the last profiled build (same command,
  # same workload); null for any other workload.  tools/profile_round.sh regenerates the inputs.
  traffic, traffic_note = None, 'no PMC pass for this workload'
  tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'traffic.json')
  from multinerf_amd import _lib as mlib
  lib_info = mlib.describe()
  if args.preset == '360' and not args.gin_bindings and B == 16384 and os.path.exists(tpath):
    with open(tpath) as f:
      tj = json.load(f)
    if tj.get('lib_digest') and tj.get('lib_digest') == lib_info.get('source_digest') and not lib_info['overridden']:
      traffic, traffic_note = tj.get('gemm_hbm_bytes_per_step'), tj.get('source')
    else:
      traffic_note = (f"profiles/traffic.json was measured on build {str(tj.get('lib_digest'))[:12]}, the loaded library is "
                      f"{str(lib_info.get('source_digest'))[:12]}: not reported")

  out = None
  if rank == 0:
    rays_per_sec = B * world * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    out = {
        'metric': 'train_rays_per_sec',
        'value': rays_per_sec,
        'unit': 'rays/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'ms_per_step_median_hip_event': median_ms,
        'ms_per_step_min_max_hip_event': [step_ms[0], step_ms[-1]],
        'higher_is_better': True,
        'scaling': 'strong' if args.global_batch else 'weak',
        'vs_baseline': None,
        'dtype': 'bf16 MFMA inputs / fp32 accumulate, fp32 everywhere else',
        'data': 'synthetic rays (SURVEY 8d, seed 20200823+rank), he_uniform random-init weights, random jitter',
        'config': {
            'workload': f'configs/360.gin native levels (64,64,32), train_step, batch_size={B} rays per GPU'
                        if args.preset == '360' and not args.gin_bindings else
                        f'preset {args.preset} {args.gin_bindings}, train_step, batch_size={B} rays per GPU',
            'global_batch': B * world,
            'parallelism': f'dp{world}',
            'train_frac': train_frac,
            'params': model.num_params, 'workspace_GiB': round(model.workspace_bytes() / 2 ** 30, 2),
            'backward_streams': mstreams.describe_env() if not model.single_mlp else {'side_stream': False},
            'pair_dxdw': bool(models._PAIR_DXDW), 'rank1_last': bool(models._RANK1_LAST), 'tangent_chain': bool(models._TANGENT_CHAIN), 'head_k32': bool(models._HEAD_K32),
            'algorithmic_train_mflop_per_ray': train_flops / 1e6,
            'algorithmic_fwd_mflop_per_ray': fwd_flops / 1e6,
            'whole_step_tflops_per_gpu': train_flops * B / (ms_per_step * 1e-3) / 1e12,
            'final_loss': s['loss'], 'final_psnr': s['psnr'],
            'reference_anchor': 'derived 182,555 rays/s aggregate on unknown hardware (BASELINE.md); not a published per-device number',
        },
        'roofline': {
            'bound': 'mfma',
            'kernel': 'gemm_nt_kernel + gemm_nt_panel_kernel + gemm_nt_wres_kernel + gemm_tn_kernel + gemm_tn_gcol_kernel + gemm_tn_rank1_kernel + mlp_chain_fwd/bwd_kernel (bf16 MFMA 32x32x16)',
            'achieved': achieved_tflops,
            'peak': 2500.0,
            'unit': 'TFLOP/s',
            'frac': achieved_tflops / 2500.0,
            # two fractions of the same 2.5 PFLOP/s: `frac` / `gemm_frac` = algorithmic FLOPs / time spent INSIDE the MFMA
            # kernels (HIP events); `whole_step_frac` = SURVEY 8(d)'s definition, rays/s x FLOPs/ray / peak per GPU
            'gemm_frac': achieved_tflops / 2500.0,
            # With the proposal levels' backward on its own stream (multinerf_amd/streams.py, the default) MFMA kernels of
            # two streams overlap: `gemm_ms_per_step` is the UNION of their launch intervals (time during which at least one
            # MFMA kernel runs), `gemm_ms_per_step_sum_of_launches` the plain sum (what a rocprofv3 --stats table adds up
            # to: a launch that shares the chip takes longer).  MNR_SIDE_STREAM=0 makes the two equal.
            'frac_definition': 'algorithmic training FLOPs per step / union of the MFMA kernels\' launch intervals / 2.5 PFLOP/s',
            'whole_step_achieved': train_flops * B / (ms_per_step * 1e-3) / 1e12,
            'whole_step_frac': train_flops * B / (ms_per_step * 1e-3) / 2.5e15,
            'traffic': traffic,
            'traffic_source': traffic_note,
            'gemm_ms_per_step': gemm_ms_per_step,
            'gemm_ms_per_step_sum_of_launches': gemm_ms_sum / nprof,
            'gemm_launches_per_step': gemm_launches / nprof,
            'gemm_share_of_step': gemm_ms_per_step / ms_per_step,
            'hbm_bound_kernels': hbm_kernels,
        },
    }
    if aux:
      out['aux'] = aux
    out['library'] = lib_info
    out['preflight'] = None if pre is None else {'ok': pre['ok'], 'workaround': pre['workaround'], 'versions': pre['versions']}
    out['distributed'] = dist_info

  # ---- CPU baseline (rank 0, N=1 only): the oracle on a bounded sample of the same workload (SURVEY 8d protocol:
  # B = 256 rays, forward and train_step separately, 1 warm-up + 3 timed runs each, medians)
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    from oracle import bridge as obridge
    from oracle import models as omodels
    from oracle import train_utils as otrain
    om, on, op = obridge.oracle_hparams(model)
    params = omodels.init_params(om, on, op, seed=0)
    nb = args.cpu_rays
    cb = synthetic.synthetic_rays(nb, near=cfg.near, far=cfg.far)
    if cfg.rawnerf_mode:
      g0 = torch.Generator().manual_seed(5)
      cb.rays.exposure_idx = torch.randint(0, 5, (nb, 1), generator=g0).to(torch.int32)
      cb.rays.exposure_values = 0.5 + torch.rand((nb, 1), generator=g0)
      cb.rays.lossmult = (torch.rand((nb, 3), generator=g0) > 0.4).float()
    if cfg.compute_normal_metrics:
      g0 = torch.Generator().manual_seed(6)
      cb.alphas = torch.rand((nb,), generator=g0)
      cb.normals = torch.randn((nb, 3), generator=g0)
    noise = obridge.make_noise(model, nb)
    st = otrain.init_opt_state(params)
    cores = torch.get_num_threads()

    def med3(fn):
      fn()                                                             # warm-up (allocator, thread pool)
      ts = []
      for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
      return sorted(ts)[1]

    with torch.no_grad():
      dt_fwd = med3(lambda: omodels.model_apply(om, on, op, params, cb.rays, train_frac, False))
    dt = med3(lambda: otrain.train_step(params, st, om, on, op, cfg, cb, train_frac, noise=noise))
    out['cpu_baseline'] = {
        'value': nb / dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
        'forward_rays_per_sec': nb / dt_fwd,
        'sample': f'oracle train_step (value) and deterministic Model forward (forward_rays_per_sec) on {nb} rays of the same '
                  f'workload, 1 warm-up + 3 runs each, medians; fp32 torch-CPU restatement of the reference (NOT the '
                  f'reference\'s JAX, which cannot be installed here); host has {os.cpu_count()} logical CPUs, torch used '
                  f'{cores} threads',
    }
  if rank == 0:
    print(json.dumps(out))


if __name__ == '__main__':
  main()
