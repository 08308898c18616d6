"""Shared test helpers: seeded synthetic rays and oracle<->product config bridging."""

import dataclasses

import numpy as np
import torch

from multinerf_amd import configs, gin, models, utils


from multinerf_amd.synthetic import synthetic_rays, procedural_scene_rays  # noqa: F401  (moved into the package)


from oracle.bridge import oracle_hparams, make_noise  # noqa: F401,E402


def assert_adam_matches_oracle(model, cfg, flat0, raw_grad, opt0, state2, what=''):
  """The optimiser arithmetic, numerically (train_utils.py:326-330 + optax.adam): the oracle's clip_gradients ->
  nan_to_num -> adam_update applied to the KERNEL's own raw gradient `raw_grad`, from parameters `flat0` and
  moments `opt0` = (mu, nu, count) or None for zeros, must reproduce the kernel's new parameters and moments
  (`state2`).  Feeding the same gradient to both sides separates optimiser errors (eps, bias correction, clip
  scale, learning-rate schedule, step count) from the bf16 noise of the gradient itself.  Returns (mu, nu, count)."""
  from oracle import train_utils as otrain
  flat0 = flat0.detach().float().cpu()
  g = raw_grad.detach().float().cpu()
  params = model.params_tree(flat0.clone())
  grads = model.params_tree(g.clone())
  if opt0 is None:
    opt = otrain.init_opt_state(params)
  else:
    opt = {'count': opt0[2], 'mu': model.params_tree(opt0[0].detach().float().cpu().clone()),
           'nu': model.params_tree(opt0[1].detach().float().cpu().clone())}
  grads = otrain.clip_gradients(grads, cfg)
  grads = otrain.tree_map(lambda z: torch.nan_to_num(z), grads)
  new_p, new_opt = otrain.adam_update(params, grads, opt, cfg)
  lr = float(otrain.lr_fn(cfg, opt['count']))
  ref_p = model.flat_from_tree(new_p, device='cpu').double()
  ref_mu = model.flat_from_tree(new_opt['mu'], device='cpu').double()
  ref_nu = model.flat_from_tree(new_opt['nu'], device='cpu').double()
  got_p, got_mu, got_nu = (t.detach().double().cpu() for t in (state2.params['flat'], state2.mu, state2.nu))
  assert state2.step == new_opt['count'], (state2.step, new_opt['count'])
  # mu = b1 mu + (1 - b1) g is a sum of terms of either sign: where they cancel, one fp32 rounding of the larger terms
  # (6e-8 relative to THEM) is a large relative error of the small result.  So mu is held to a few ulps of the largest
  # element in absolute terms, and per element only to 2e-3 with the 1e-5 max|mu| floor (the worst element of 9 M sat at
  # 0.99e-4 on one box and 1.1e-4 on another in round 3: the old 1e-4 bound was a coin flip, not a check).
  e_mu = ((got_mu - ref_mu).abs() / (ref_mu.abs() + 1e-5 * ref_mu.abs().max() + 1e-30)).max().item()
  e_mu_abs = ((got_mu - ref_mu).abs().max() / (ref_mu.abs().max() + 1e-30)).item()
  e_nu = ((got_nu - ref_nu).abs() / (ref_nu.abs() + 1e-5 * ref_nu.abs().max() + 1e-30)).max().item()
  # the update itself, relative to the learning rate (|update| <= ~lr): 1e-3 lr absolute + fp32 rounding of the parameter
  upd_err = ((got_p - ref_p).abs() - 2.0 ** -23 * ref_p.abs()).clamp_min(0).max().item() / lr
  print(f'{what}Adam vs oracle on the kernel gradient: mu rel {e_mu:.2e} (abs / max {e_mu_abs:.2e}), nu rel {e_nu:.2e}, |update err| / lr {upd_err:.2e} (lr {lr:.3e})')
  assert e_mu < 2e-3 and e_mu_abs < 1e-6 and e_nu < 2e-4 and upd_err < 1e-3, (e_mu, e_mu_abs, e_nu, upd_err)
  return state2.mu.detach().clone(), state2.nu.detach().clone(), state2.step


def to_float64(x):
  """A pytree (dict / list / tuple / the package's dataclasses) with every floating tensor upcast to float64."""
  if torch.is_tensor(x):
    return x.detach().double() if x.is_floating_point() else x
  if isinstance(x, dict):
    return {k: to_float64(v) for k, v in x.items()}
  if dataclasses.is_dataclass(x) and not isinstance(x, type):
    return dataclasses.replace(x, **{f.name: to_float64(getattr(x, f.name)) for f in dataclasses.fields(x)})
  if isinstance(x, (list, tuple)):
    return type(x)(to_float64(v) for v in x)
  return x


def flat_from_tree_f64(model, tree):
  """models.Model.flat_from_tree without its cast to float32 (the float64 oracle's gradient as one vector)."""
  flat = torch.zeros(model.num_params, dtype=torch.float64)
  views = model.params_tree(flat)
  for mname, mod in views.items():
    for dname, d in mod.items():
      if isinstance(d, dict):
        for k, v in d.items():
          v.copy_(tree[mname][dname][k].detach().double().reshape(v.shape))
      else:
        d.copy_(tree[mname][dname].detach().double().reshape(d.shape))
  return flat


def kernel_relu_sides(model, B):
  """Which side of its kink every ReLU unit took in the kernels' last training forward pass (models.Model._saved): per level
  {'masks': [bool [B, n, width] per activated Dense layer in the reference's call order: trunk layers, then view-MLP layers]},
  or None for a level whose MLP is not a ReLU network.  Input of the oracle's `relu_sides` test hook (oracle.models.mlp_apply)."""
  true_plan = {p.module_name: p for p in model._tplans}
  out = {}
  for lv in model._saved['levels']:
    plan, n = lv['plan'], lv['n']
    if plan.hp.net_activation != 'relu':
      out[lv['level']] = None
      continue
    tp, rows, mlp = true_plan[plan.module_name], B * n, lv['mlp']
    masks = []
    for i in range(len(plan.trunk)):
      bits = mlp['bits'][i] if mlp.get('bits') else None
      if bits is not None:
        m = ((bits[:rows].cpu()[:, :, None] >> torch.arange(8, dtype=torch.uint8)) & 1).reshape(rows, -1).bool()
      else:
        m = mlp['acts'][i][:rows].cpu() > 0
      masks.append(m[:, :tp.W].reshape(B, n, tp.W))
    wv = tp.hp.net_width_viewdirs
    for v in mlp.get('vacts', []):
      masks.append((v[:rows].cpu() > 0)[:, :wv].reshape(B, n, wv))
    out[lv['level']] = {'masks': masks}
  return out


def oracle_train_step_f64(params, om, on, op, cfg, batch, train_frac, noise, relu_sides=None):
  """oracle.train_utils.train_step evaluated in float64 on the SAME float32 inputs (parameters, rays, noise upcast): the
  reference arithmetic without rounding, what the fp32-Dense debug mode (models.Model.dense_precision = 'fp32') is held
  against.  (The oracle is pinned to the reference's own code in float64: tests/test_oracle_models_golden.py.)
  `relu_sides` (kernel_relu_sides): every ReLU unit takes the side of its kink the kernels took; where float64's own sign(z)
  disagrees, |z| must be evaluation noise (asserted: at most 5e-5 of the units + 10, all with |z| < 5e-3 where typical |z| is of
  order one: the noise of a pre-activation is not its own 1e-7 rounding but the fp32 error of the sample positions, ~1e-6, times the
  encoding's highest frequency, 2^12 .. 2^16).  -> (stats, gradient tree)."""
  from oracle import train_utils as otrain
  p64 = to_float64(params)
  _, _, stats, grads = otrain.train_step(p64, otrain.init_opt_state(p64), om, on, op, cfg, to_float64(batch), train_frac,
                                         noise=to_float64(noise), relu_sides=relu_sides)
  if relu_sides is not None:
    units = dis = 0
    zmax = 0.0
    for lv in relu_sides.values():
      st = (lv or {}).get('stats', {})
      units, dis, zmax = units + st.get('units', 0), dis + st.get('disagree', 0), max(zmax, st.get('max_abs_z', 0.0))
    print(f'RELU_SIDES: {dis} of {units} units took the other side of the kink in fp32 (largest float64 |z| among them {zmax:.2e})')
    assert dis <= 5e-5 * units + 10 and zmax < 5e-3, (dis, units, zmax)
  return stats, grads


def check_fp32_mode_gradient(model, g, g_64, g_32, tag, grad_tol=2e-4, cost_factor=2.0, kink_tol=5e-2):
  """The fp32-Dense debug mode's gradient `g` (models.Model.dense_precision = 'fp32') against the float64 oracle's `g_64`, per
  top-level module: relative L2 <= max(grad_tol, cost_factor x |oracle_fp32 - oracle_fp64|) (`g_32`: the plain fp32 oracle; where
  fp32 arithmetic itself costs more than grad_tol the kernels must be about as close to float64 as the fp32 oracle is).

  (Normally both oracles were evaluated with the kernels' side of every ReLU kink, `kernel_relu_sides`; what follows is the
  fallback for a caller that did not.)  One thing fp32 cannot promise is the SIDE of a ReLU kink: a unit whose pre-activation is within fp32 rounding of 0 for one
  sample gets mask 1 in one evaluation and 0 in the other.  Such a flip is not an arithmetic error of ~1e-7 but a different
  (equally valid) subgradient; it puts a rank-one error into ONE column of that layer's weight gradient (x_{l-1}[s] (x) e_j) and a
  smooth error into every layer upstream of it.  A module that misses the bound is therefore examined per Dense layer: if one
  of its layers carries >= 90 % of its error energy in <= 3 output columns (units), the case is reported as KINK_FLIP and
  held to `kink_tol` instead; an error spread over the columns of every layer (a missing term, a wrong factor) still fails.
  Returns the largest relative error over the modules that met the plain bound."""
  g, g_64, g_32 = g.double().cpu(), g_64.double().cpu(), g_32.double().cpu()
  assert torch.isfinite(g).all()
  ranges = model.param_ranges()
  worst = 0.0
  for mod, b, e in model.modules:
    a, r, r32 = g[b:e], g_64[b:e], g_32[b:e]
    if r.norm() < 1e-12:
      assert a.norm() < 1e-6, mod
      continue
    rel = ((a - r).norm() / r.norm()).item()
    cost32 = ((r32 - r).norm() / r.norm()).item()
    bound = max(grad_tol, cost_factor * cost32)
    print(f'F32MODE {tag} {mod}: gradient |kernel_fp32 - oracle_fp64| {rel:.3e} (|oracle_fp32 - oracle_fp64| {cost32:.3e})')
    if rel <= bound:
      worst = max(worst, rel)
      continue
    flipped = None
    for p in model._tplans:
      if p.module_name != mod:
        continue
      for d in [d for d, _ in p.trunk] + [d for d, _ in p.view]:          # (the Dense layers behind an activation)
        kb, ke = ranges[f'{mod}/{d.name}/kernel']
        err = (g[kb:ke] - g_64[kb:ke]).view(d.fan_in, d.fan_out)
        col = err.pow(2).sum(0)
        if col.sum().sqrt() <= (grad_tol / 3) * g_64[kb:ke].norm():
          continue                                         # nothing to explain in this layer
        top = col.topk(3)
        if top.values.sum() >= 0.9 * col.sum():
          flipped = (d.name, top.indices.tolist(), (top.values.sum() / col.sum()).item())
    if flipped is not None and rel <= kink_tol:
      print(f'F32MODE {tag} {mod}: KINK_FLIP {flipped[0]} units {flipped[1]} carry {flipped[2]:.3f} of that layer\'s error energy: a ReLU '
            f'pre-activation within fp32 rounding of 0 took the other side in the two evaluations (held to {kink_tol})')
      continue
    raise AssertionError((tag, mod, rel, cost32, bound, flipped))
  return worst
