"""Same-process A/B of the render path (deterministic forward with extras, one 16384-ray chunk of configs/360.gin) with the
proposal levels' IPE features produced inside the chain kernel (MNR_FUSED_IPE, default) or by mnr_cast_rays_ipe, plus the
leaf timings of the two ways at the proposal-level shape and the fused kernel's per-phase timeline.

    python tools/render_probe.py [--rays 16384] [--reps 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import configs, models, ops, synthetic, train_utils  # noqa: E402
from multinerf_amd import _lib as L  # noqa: E402


def timed(fn, reps):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  ts = []
  for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
  ts.sort()
  return ts[len(ts) // 2], ts[0]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rays', type=int, default=16384)
  ap.add_argument('--reps', type=int, default=10)
  ap.add_argument('--only', choices=['on', 'off'], default=None, help='render only, with the switch fixed (for rocprofv3 --kernel-trace)')
  a = ap.parse_args()
  dev = torch.device('cuda', 0)
  cfg = configs.load_preset('360', [])
  cfg.batch_size = a.rays
  model, state, render_eval_pfn, _, _ = train_utils.setup_model(cfg, 0, device=dev)
  batch = synthetic.synthetic_rays(a.rays, seed=20200823, near=cfg.near, far=cfg.far).map(lambda t: t.to(dev))
  rays = batch.rays
  if a.only:
    models._FUSED_IPE = a.only == 'on'
    med, mn = timed(lambda: render_eval_pfn(state.params, 1.0, None, rays), a.reps)
    print(f'render, in-kernel IPE {a.only}: median {med:7.3f} ms  min {mn:7.3f} ms')
    return
  for rnd in range(2):
    for on in (True, False):
      models._FUSED_IPE = on
      med, mn = timed(lambda: render_eval_pfn(state.params, 1.0, None, rays), a.reps)
      print(f'render, in-kernel IPE {"on " if on else "off"}: median {med:7.3f} ms  min {mn:7.3f} ms  -> {a.rays / med * 1e3:10.0f} rays/s', flush=True)
  # leaves at the proposal-level shape
  plan = model.prop_plan
  hp, W = plan.hp, plan.W
  n = model.num_prop_samples
  B = a.rays
  M = B * n
  flat = state.params['flat'] if isinstance(state.params, dict) else state.params
  model.pack_weights(flat, ipe=True)
  g = torch.Generator(device=dev).manual_seed(1)
  tdist = torch.cumsum(0.02 + torch.rand((B, n + 1), generator=g, device=dev) * 0.3, dim=-1).contiguous()
  R = rays.map(lambda r: r.reshape(-1, r.shape[-1]).contiguous())
  radii = R.radii.reshape(-1).contiguous()
  kw = dict(ray_shape=model.ray_shape, warp_contract=(hp.warp_fn == 'contract'), min_deg=hp.min_deg_point, max_deg=hp.max_deg_point)
  bias = lambda d: flat[d.bias_off:d.bias_off + d.fan_out]
  lay = lambda key: model._w(plan, plan.packed[key]['f_off'], plan.packed[key]['n_pad'], plan.packed[key]['f_ld'])
  ref_layers = [(lay(('trunk', i)), bias(d)) for i, (d, _) in enumerate(plan.trunk)]
  ipe_layers = [(lay('trunk0_ipe'), bias(plan.trunk[0][0]))] + ref_layers[1:]
  w_head, b_head = lay('density')[0], flat[plan.density.bias_off:plan.density.bias_off + 1]
  feat = torch.empty((M, plan.ldF), dtype=torch.bfloat16, device=dev)
  out = torch.empty((M,), dtype=torch.float32, device=dev)
  t_ipe = timed(lambda: ops.cast_rays_ipe(tdist, R.origins, R.directions, radii, plan.basis_dev, ld_feat=plan.ldF, out=feat, **kw), a.reps)[0]
  t_chain = timed(lambda: ops.mlp_chain_fwd(feat, plan.ldF, ref_layers, M=M, W=W, w_head=w_head, b_head=b_head, head_out=out), a.reps)[0]
  t_fused = timed(lambda: ops.mlp_chain_fwd_ipe(tdist, R.origins, R.directions, radii, plan.basis_dev, ipe_layers, M=M, W=W,
                                                w_head=w_head, b_head=b_head, head_out=out, **kw), a.reps)[0]
  print(f'proposal level (M = {M}): cast_rays_ipe {t_ipe * 1e3:7.1f} us + chain (inference) {t_chain * 1e3:7.1f} us = {(t_ipe + t_chain) * 1e3:7.1f} us;'
        f'  chain with in-kernel IPE {t_fused * 1e3:7.1f} us')
  # timeline of the fused kernel (second tile of every workgroup)
  tl = torch.zeros((256 * 32,), dtype=torch.int64, device=dev)
  L.check(ops.L.debug().mnr_debug_chain_timeline(tl.data_ptr()))
  ops.mlp_chain_fwd_ipe(tdist, R.origins, R.directions, radii, plan.basis_dev, ipe_layers, M=M, W=W, w_head=w_head, b_head=b_head,
                        head_out=out, **kw)
  torch.cuda.synchronize()
  L.check(ops.L.debug().mnr_debug_chain_timeline(None))
  t = tl.cpu().view(256, 32).double()
  t = t[t[:, 0] > 0]
  D = len(plan.trunk)
  names = ['layer 0 (Gaussians + 3 x (encode, MFMA))'] + sum([[f'layer {li} MFMAs', f'layer {li} epilogue', f'layer {li} copy-out / head'] for li in range(D)], [])
  slots = [1] + sum([[2 + 3 * li, 3 + 3 * li, 4 + 3 * li] for li in range(D)], [])
  prev = t[:, 0]
  print(f'fused kernel, cycles per phase of a tile (median over {t.shape[0]} workgroups):')
  for nm, sl in zip(names, slots):
    cur = t[:, sl]
    print(f'  {nm:44s} {torch.median(cur - prev).item():9.0f}')
    prev = cur
  print(f'  tile total {torch.median(t[:, 4 + 3 * (D - 1)] - t[:, 0]).item():9.0f} cycles, {torch.median(t[:, 31] - t[:, 30]).item() * 10:9.0f} ns')


if __name__ == '__main__':
  main()
