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


def oracle_train_step_f64(params, om, on, op, cfg, batch, train_frac, noise):
  """oracle.train_utils.train_step evaluated in float64 on the SAME float32 inputs (parameters, rays, noise upcast): the
  reference arithmetic without rounding, what the fp32-Dense debug mode (models.Model.dense_precision = 'fp32') is held
  against.  (The oracle is pinned to the reference's own code in float64: tests/test_oracle_models_golden.py.)
  -> (stats, gradient tree)."""
  from oracle import train_utils as otrain
  p64 = to_float64(params)
  _, _, stats, grads = otrain.train_step(p64, otrain.init_opt_state(p64), om, on, op, cfg, to_float64(batch), train_frac,
                                         noise=to_float64(noise))
  return stats, grads
