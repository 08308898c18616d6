"""Oracle restatement of reference internal/train_utils.py (TEST INFRASTRUCTURE ONLY).

The loss terms (compute_data_loss, interlevel / distortion / orientation /
predicted-normal losses) and clip_gradients are pinned against the reference's own
internal/train_utils.py executed on stand-ins (tests/golden/make_golden_models.py,
tests/test_oracle_models_golden.py, rtol 1e-9 in float64); so is the gradient of loss_fn, against the
reference's forward + losses differentiated by complex step along seeded directions.
PARITY UNPINNED: optax.adam (train_utils.py:372) is restated from its published algorithm
(scale_by_adam with eps outside the sqrt, eps_root=0, bias correction by
1-b^t with t = count+1; scale_by_schedule evaluates lr at the PRE-increment
count).  The reference does not pin an optax version (requirements.txt:1-12).
"""

import collections
import math as _pm

import numpy as np
import torch

from oracle import image
from oracle import math as rmath
from oracle import models
from oracle import ref_utils
from oracle import stepfun

F32_EPS = float(np.finfo(np.float32).eps)


def tree_leaves(tree, prefix=()):
  """Flatten a nested dict to [(name_tuple, tensor)] in creation order."""
  out = []
  for k, v in tree.items():
    if isinstance(v, dict):
      out += tree_leaves(v, prefix + (k,))
    else:
      out.append((prefix + (k,), v))
  return out


def tree_map(fn, *trees):
  t0 = trees[0]
  if isinstance(t0, dict):
    return {k: tree_map(fn, *[t[k] for t in trees]) for k in t0}
  return fn(*trees)


def tree_norm_sq(tree):
  """train_utils.py:43-44."""
  if not isinstance(tree, dict):
    return torch.sum(tree**2)
  return sum(torch.sum(v**2) for _, v in tree_leaves(tree))


def tree_norm(tree):
  """train_utils.py:47-48."""
  return torch.sqrt(tree_norm_sq(tree))


def tree_abs_max(tree):
  """train_utils.py:51-53."""
  if not isinstance(tree, dict):
    return tree.abs().max()
  return max(v.abs().max() for _, v in tree_leaves(tree))


def summarize_tree(tree, fn, ancestry=(), max_depth=3):
  """train_utils.py:61-69 -- {'A/B/C': fn(subtree)} down to max_depth."""
  stats = {}
  for k, v in tree.items():
    name = ancestry + (k,)
    stats['/'.join(name)] = fn(v)
    if hasattr(v, 'items') and len(ancestry) < (max_depth - 1):
      stats.update(summarize_tree(v, fn, ancestry=name, max_depth=max_depth))
  return stats


def compute_data_loss(batch, renderings, rays, config):
  """train_utils.py:72-136 (robustnerf branch out of scope)."""
  data_losses = []
  stats = collections.defaultdict(list)
  gt = batch.rgb[..., :3]
  lossmult = rays.lossmult.expand(gt.shape)
  if config.disable_multiscale_loss:
    lossmult = torch.ones_like(lossmult)

  for rendering in renderings:
    resid_sq = (rendering['rgb'] - gt)**2
    denom = lossmult.sum()
    stats['mses'].append((lossmult * resid_sq).sum() / denom)

    if config.data_loss_type == 'mse':
      data_loss = resid_sq
    elif config.data_loss_type == 'charb':
      data_loss = torch.sqrt(resid_sq + config.charb_padding**2)
    elif config.data_loss_type == 'rawnerf':
      rgb_render_clip = torch.clamp(rendering['rgb'], max=1.)
      resid_sq_clip = (rgb_render_clip - gt)**2
      scaling_grad = 1. / (1e-3 + rgb_render_clip.detach())
      data_loss = resid_sq_clip * scaling_grad**2
    else:
      raise ValueError(config.data_loss_type)
    data_losses.append((lossmult * data_loss).sum() / denom)

    if config.compute_disp_metrics:
      disp = 1 / (1 + rendering['distance_mean'])
      stats['disparity_mses'].append(((disp - batch.disps)**2).mean())

    if config.compute_normal_metrics:
      if 'normals' in rendering:
        weights = rendering['acc'] * batch.alphas
        normalized_normals_gt = ref_utils.l2_normalize(batch.normals)
        normalized_normals = ref_utils.l2_normalize(rendering['normals'])
        normal_mae = ref_utils.compute_weighted_mae(weights, normalized_normals,
                                                    normalized_normals_gt)
      else:
        normal_mae = torch.tensor(float('nan'))
      stats['normal_maes'].append(normal_mae)

  data_losses = torch.stack(data_losses)
  loss = (config.data_coarse_loss_mult * torch.sum(data_losses[:-1]) +
          config.data_loss_mult * data_losses[-1])
  stats = {k: torch.stack([torch.as_tensor(x) for x in stats[k]]) for k in stats}
  return loss, stats


def interlevel_loss(ray_history, config):
  """train_utils.py:139-150."""
  last = ray_history[-1]
  c = last['sdist'].detach()
  w = last['weights'].detach()
  loss_interlevel = 0.
  for ray_results in ray_history[:-1]:
    cp = ray_results['sdist']
    wp = ray_results['weights']
    loss_interlevel = loss_interlevel + torch.mean(stepfun.lossfun_outer(c, w, cp, wp))
  return config.interlevel_loss_mult * loss_interlevel


def distortion_loss(ray_history, config):
  """train_utils.py:153-159."""
  last = ray_history[-1]
  loss = torch.mean(stepfun.lossfun_distortion(last['sdist'], last['weights']))
  return config.distortion_loss_mult * loss


def orientation_loss(rays, model, ray_history, config):
  """train_utils.py:162-178."""
  total_loss = 0.
  for i, ray_results in enumerate(ray_history):
    w = ray_results['weights']
    n = ray_results[config.orientation_loss_target]
    if n is None:
      raise ValueError('Normals cannot be None if orientation loss is on.')
    v = -1. * rays.viewdirs
    n_dot_v = (n * v[..., None, :]).sum(dim=-1)
    loss = torch.mean((w * torch.clamp(n_dot_v, max=0.0)**2).sum(dim=-1))
    if i < model.num_levels - 1:
      total_loss = total_loss + config.orientation_coarse_loss_mult * loss
    else:
      total_loss = total_loss + config.orientation_loss_mult * loss
  return total_loss


def predicted_normal_loss(model, ray_history, config):
  """train_utils.py:181-197."""
  total_loss = 0.
  for i, ray_results in enumerate(ray_history):
    w = ray_results['weights']
    n = ray_results['normals']
    n_pred = ray_results['normals_pred']
    if n is None or n_pred is None:
      raise ValueError('Predicted normals and gradient normals cannot be None if '
                       'predicted normal loss is on.')
    loss = torch.mean((w * (1.0 - torch.sum(n * n_pred, dim=-1))).sum(dim=-1))
    if i < model.num_levels - 1:
      total_loss = total_loss + config.predicted_normal_coarse_loss_mult * loss
    else:
      total_loss = total_loss + config.predicted_normal_loss_mult * loss
  return total_loss


def clip_gradients(grad, config):
  """train_utils.py:200-218 -- per top-level module: clip by value, then norm."""
  out = {}
  for k, g in grad.items():
    if config.grad_max_val > 0:
      g = tree_map(lambda z: torch.clamp(z, -config.grad_max_val, config.grad_max_val), g)
    if config.grad_max_norm > 0:
      mult = torch.clamp(config.grad_max_norm / (F32_EPS + tree_norm(g)), max=1)
      g = tree_map(lambda z: mult * z, g)
    out[k] = g
  return out


def loss_fn(params, model, nerf_mlp, prop_mlp, config, batch, train_frac, noise, dense_dtype=None, relu_sides=None):
  """The closure of train_utils.py:265-314."""
  rays = batch.rays
  compute_extras = config.compute_disp_metrics or config.compute_normal_metrics
  renderings, ray_history = models.model_apply(
      model, nerf_mlp, prop_mlp, params, rays, train_frac=train_frac,
      compute_extras=compute_extras, zero_glo=False,
      noise=noise if config.randomized else None, dense_dtype=dense_dtype, relu_sides=relu_sides)

  losses = {}
  data_loss, stats = compute_data_loss(batch, renderings, rays, config)
  losses['data'] = data_loss
  if config.interlevel_loss_mult > 0:
    losses['interlevel'] = interlevel_loss(ray_history, config)
  if config.distortion_loss_mult > 0:
    losses['distortion'] = distortion_loss(ray_history, config)
  if config.orientation_coarse_loss_mult > 0 or config.orientation_loss_mult > 0:
    losses['orientation'] = orientation_loss(rays, model, ray_history, config)
  if config.predicted_normal_coarse_loss_mult > 0 or config.predicted_normal_loss_mult > 0:
    losses['predicted_normals'] = predicted_normal_loss(model, ray_history, config)

  stats['weight_l2s'] = summarize_tree(params, tree_norm_sq)
  if config.weight_decay_mults:
    losses['weight'] = sum(m * stats['weight_l2s'][k]
                           for k, m in config.weight_decay_mults.items())
  stats['loss'] = sum(losses.values())
  stats['losses'] = losses
  return stats['loss'], stats, (renderings, ray_history)


def lr_fn(config, step):
  """train_utils.py:364-371."""
  return rmath.learning_rate_decay(step, config.lr_init, config.lr_final,
                                   config.max_steps, config.lr_delay_steps,
                                   config.lr_delay_mult)


def init_opt_state(params):
  """optax.adam init: count 0, mu = nu = 0."""
  return {'count': 0,
          'mu': tree_map(torch.zeros_like, params),
          'nu': tree_map(torch.zeros_like, params)}


def adam_update(params, grads, opt_state, config):
  """optax.adam(lr_fn, b1, b2, eps) + apply_updates (train_utils.py:330, 372)."""
  b1, b2, eps = config.adam_beta1, config.adam_beta2, config.adam_eps
  count = opt_state['count']
  t = count + 1
  mu = tree_map(lambda m, g: b1 * m + (1 - b1) * g, opt_state['mu'], grads)
  nu = tree_map(lambda v, g: b2 * v + (1 - b2) * g * g, opt_state['nu'], grads)
  bc1 = 1 - b1**t
  bc2 = 1 - b2**t
  lr = float(lr_fn(config, count))
  new_params = tree_map(
      lambda p, m, v: p - lr * (m / bc1) / (torch.sqrt(v / bc2) + eps), params, mu, nu)
  return new_params, {'count': t, 'mu': mu, 'nu': nu}


def train_step(params, opt_state, model, nerf_mlp, prop_mlp, config, batch, train_frac,
               noise=None, grad_allreduce=None, dense_dtype=None, relu_sides=None):
  """train_utils.py:239-339.  Returns (new_params, new_opt_state, stats, grads).
  (`relu_sides`: test hook, oracle.models.mlp_apply.)

  `grad_allreduce(tree)` stands in for jax.lax.pmean (train_utils.py:319-321).
  """
  leaves = tree_map(lambda p: p.detach().clone().requires_grad_(True), params)
  loss, stats, _ = loss_fn(leaves, model, nerf_mlp, prop_mlp, config, batch, train_frac,
                           noise, dense_dtype=dense_dtype, relu_sides=relu_sides)
  flat = [v for _, v in tree_leaves(leaves)]
  gflat = torch.autograd.grad(loss, flat, allow_unused=True)
  gflat = [torch.zeros_like(p) if g is None else g for g, p in zip(gflat, flat)]
  it = iter(gflat)
  grad = tree_map(lambda p: next(it), leaves)
  if grad_allreduce is not None:
    grad = grad_allreduce(grad)

  stats = dict(stats)
  stats['grad_norms'] = summarize_tree(grad, tree_norm)
  stats['grad_maxes'] = summarize_tree(grad, tree_abs_max)
  raw_grad = grad
  grad = clip_gradients(grad, config)
  grad = tree_map(lambda g: torch.nan_to_num(g), grad)
  with torch.no_grad():
    new_params, new_opt = adam_update(params, grad, opt_state, config)
    # train_utils.py:332-335
    it2 = iter([v for _, v in tree_leaves(new_params)])
    opt_delta = tree_map(lambda p: next(it2) - p, params)
    stats['opt_update_norms'] = summarize_tree(opt_delta, tree_norm)
    stats['opt_update_maxes'] = summarize_tree(opt_delta, tree_abs_max)
  stats['psnrs'] = image.mse_to_psnr(stats['mses'])
  stats['psnr'] = stats['psnrs'][-1]
  return new_params, new_opt, stats, raw_grad
