"""Training step and model creation (reference internal/train_utils.py).

Keeps the reference's call surface -- `setup_model` (train_utils.py:399-419),
`create_train_step` -> `train_pstep(rng, state, batch, cameras, train_frac,
loss_threshold)` (:221-346), `create_optimizer` (:349-374), `create_render_fn`
-> `render_eval_pfn(variables, train_frac, _, rays)` (:377-396) -- while the
bodies enqueue HIP kernels.  `jax.pmap` + `pmean`/`all_gather` become one
process per GPU with RCCL collectives over torch.distributed (multinerf_amd/dist.py).
"""

import dataclasses
import math
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch

from multinerf_amd import camera_utils, dist as mdist
from multinerf_amd import streams as mstreams
from multinerf_amd import models
from multinerf_amd import ops
from multinerf_amd import utils

f32 = torch.float32


def log_lerp(t, v0, v1):
  """math.py:57-63."""
  if v0 <= 0 or v1 <= 0:
    raise ValueError(f'Interpolants {v0} and {v1} must be positive.')
  lv0, lv1 = math.log(v0), math.log(v1)
  return math.exp(min(max(t, 0.), 1.) * (lv1 - lv0) + lv0)


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
  """math.py:66-98 (host side: one scalar per step)."""
  if lr_delay_steps > 0:
    delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(
        0.5 * math.pi * min(max(step / lr_delay_steps, 0.), 1.))
  else:
    delay_rate = 1.
  return delay_rate * log_lerp(step / max_steps, lr_init, lr_final)


@dataclasses.dataclass
class TrainState:
  """flax.training.train_state.TrainState restated: step, params, optimiser moments."""
  step: int
  params: Dict[str, Any]          # {'flat': fp32 vector, 'params': nested views}
  mu: torch.Tensor
  nu: torch.Tensor


def create_optimizer(config, variables):
  """train_utils.py:349-374: Adam state (zeros) + the learning-rate schedule."""
  flat = variables['flat']
  lr_fn = lambda step: learning_rate_decay(step, config.lr_init, config.lr_final, config.max_steps,
                                           config.lr_delay_steps, config.lr_delay_mult)
  state = TrainState(step=0, params=variables, mu=torch.zeros_like(flat), nu=torch.zeros_like(flat))
  return state, lr_fn


def create_train_step(model: models.Model, config, dataset=None):
  """train_utils.py:221-346.  Returns train_pstep(rng, state, batch, cameras, train_frac,
  loss_threshold) -> (new_state, stats, rng).  `batch.rays` holds THIS rank's shard."""
  # train_utils.py:234-237: the dataset names the projection; perspective without one.
  camtype = getattr(dataset, 'camtype', camera_utils.ProjectionType.PERSPECTIVE) if dataset is not None \
      else camera_utils.ProjectionType.PERSPECTIVE
  if config.data_loss_type not in ('mse', 'charb', 'rawnerf'):
    raise NotImplementedError(f'data_loss_type {config.data_loss_type!r} is out of scope')
  use_orient = config.orientation_loss_mult > 0 or config.orientation_coarse_loss_mult > 0
  use_prednorm = config.predicted_normal_loss_mult > 0 or config.predicted_normal_coarse_loss_mult > 0
  if config.orientation_loss_target not in ('normals', 'normals_pred'):
    raise ValueError(f'orientation_loss_target {config.orientation_loss_target!r} is not a ray_history field')
  lr_fn = lambda step: learning_rate_decay(step, config.lr_init, config.lr_final, config.max_steps,
                                           config.lr_delay_steps, config.lr_delay_mult)
  side_cache = {}
  ranges = model.param_ranges()

  def wd_range(key):
    if key not in ranges:
      raise KeyError(f'Config.weight_decay_mults: {key!r} is not a key of the parameter tree (module, module/Dense_k or '
                     f'module/Dense_k/kernel|bias); modules: {[m for m, _, _ in model.modules]}')
    return ranges[key]

  for _k in (config.weight_decay_mults or {}):
    wd_range(_k)

  def backward_streams(dev):
    """The proposal levels' backward on its own stream next to the NeRF level's (multinerf_amd/streams.py); None = off."""
    if dev.type != 'cuda' or model.single_mlp or model.num_levels < 2:
      return None
    if 'v' not in side_cache:
      side_cache['v'] = mstreams.BackwardStreams.from_env(dev)
    return side_cache['v']

  def train_step(rng, state: TrainState, batch, cameras, train_frac, loss_threshold, noise=None,
                 return_grads=False, tree_stats=False):
    with model.library():          # (a no-op for the product; Model.dense_precision = 'fp32': the fp32-Dense debug build)
      return train_step_impl(rng, state, batch, cameras, train_frac, loss_threshold, noise, return_grads, tree_stats)

  def train_step_impl(rng, state: TrainState, batch, cameras, train_frac, loss_threshold, noise, return_grads, tree_stats):
    flat = state.params['flat']
    dev = flat.device
    rays = batch.rays
    if config.cast_rays_in_train_step:                                 # train_utils.py:267-268
      rays = camera_utils.cast_ray_batch(cameras, rays, camtype)
    compute_extras = config.compute_disp_metrics or config.compute_normal_metrics
    use_rng = rng if config.randomized else None
    renderings, ray_history = model._forward(flat, use_rng, rays, train_frac, compute_extras, zero_glo=False,
                                             noise=noise if config.randomized else None,
                                             keep_for_backward=True)
    saved = model._saved
    levels, Bp, B0 = saved['levels'], saved['Bp'], saved['B0']
    R = saved['rays']

    nlev = len(levels)
    # stats layout: [mse_l, data_l]*nlev | interlevel | distortion | denom | orientation | predicted_normals
    #               | disparity_mse_l * nlev | normal_mae_l * nlev | weight loss
    stats = model._buf(('train', 'stats'), (4 * nlev + 6,), f32)
    stats.zero_()
    denom = stats[2 * nlev + 2:2 * nlev + 3]
    lossmult = R.lossmult
    if config.disable_multiscale_loss:
      lossmult = torch.ones_like(lossmult)
    gt = batch.rgb[..., :3].reshape(-1, 3).contiguous()
    if gt.shape[0] != Bp:
      gt = torch.cat([gt, gt[-1:].expand(Bp - gt.shape[0], 3)], 0).contiguous()
    ops.lossmult_sum(lossmult, B0, denom)

    # (the kernels' layout of the parameters: `flat` itself unless a trunk width is padded to the GEMM tile, models.Model.build)
    flat_true, flat = flat, model._flat_exec
    grads = model._buf(('train', 'grads'), (model.num_params_exec,), f32)
    grads.zero_()

    # Data, interlevel and distortion losses (train_utils.py:85-111, 131-159) are evaluated AND differentiated inside
    # each level's backward launch (ops.composite_bwd(losses=...) -> mnr_level_bwd): only their specs are built here.
    g_w = [None] * nlev                                               # upstream d loss / d weights (Ref-NeRF normal losses only)
    data_spec, w_spec = [None] * nlev, [None] * nlev
    last = levels[-1]
    for li, lv in enumerate(levels):
      mult = config.data_loss_mult if li == nlev - 1 else config.data_coarse_loss_mult
      data_spec[li] = dict(type=config.data_loss_type, charb_padding=config.charb_padding, mult=mult, rgb_out=lv['rgb_out'],
                           gt=gt, lossmult=lossmult, denom=denom, stats=stats[2 * li:2 * li + 2])
      if li < nlev - 1 and config.interlevel_loss_mult > 0:           # train_utils.py:139-150
        w_spec[li] = dict(mode='interlevel', mult=config.interlevel_loss_mult, sdist=lv['sdist'], t_ref=last['sdist'],
                          w_ref=last['weights'], stat=stats[2 * nlev:2 * nlev + 1])
      elif li == nlev - 1 and config.distortion_loss_mult > 0:        # train_utils.py:153-159
        w_spec[li] = dict(mode='distortion', mult=config.distortion_loss_mult, sdist=lv['sdist'],
                          stat=stats[2 * nlev + 1:2 * nlev + 2])
      rd = saved['renderings'][li]
      k0 = 2 * nlev + 5
      if config.compute_disp_metrics:                                  # train_utils.py:113-115
        if batch.disps is None:
          raise ValueError('compute_disp_metrics needs batch.disps')
        ops.render_metrics(B0, distance_mean=rd['distance_mean'].reshape(-1).contiguous(),
                           disps=batch.disps.reshape(-1).contiguous().float(), out_disp=stats[k0 + li:k0 + li + 1])
      if config.compute_normal_metrics:                                # train_utils.py:117-128
        if 'normals' in rd:
          if batch.alphas is None or batch.normals is None:
            raise ValueError('compute_normal_metrics needs batch.alphas and batch.normals')
          ops.render_metrics(B0, acc=rd['acc'].reshape(-1).contiguous(),
                             alphas=batch.alphas.reshape(-1).contiguous(),
                             normals=rd['normals'].reshape(-1, 3).contiguous(),
                             normals_gt=batch.normals.reshape(-1, 3).contiguous().float(),
                             out_normal=stats[k0 + nlev + li:k0 + nlev + li + 1])
        else:
          stats[k0 + nlev + li] = float('nan')

    g_nrm, g_npr = [None] * nlev, [None] * nlev
    if use_orient or use_prednorm:                                     # train_utils.py:162-197
      for li, lv in enumerate(levels):
        # train_utils.py:162-197: the orientation loss needs its target field, the predicted-normal loss both fields
        tgt = 'npred' if config.orientation_loss_target == 'normals_pred' else 'normals'
        if use_orient and lv['mlp'].get(tgt) is None:
          raise ValueError('Normals cannot be None if orientation loss is on.')
        if use_prednorm and (lv['mlp'].get('normals') is None or lv['mlp'].get('npred') is None):
          raise ValueError('Predicted normals and gradient normals cannot be None if predicted normal loss is on.')
        fine = li == nlev - 1
        mo = config.orientation_loss_mult if fine else config.orientation_coarse_loss_mult
        mp = config.predicted_normal_loss_mult if fine else config.predicted_normal_coarse_loss_mult
        if g_w[li] is None:
          g_w[li] = model._buf(('train', 'g_w', li), (Bp, lv['n']), f32)
          g_w[li].zero_()
        g_nrm[li], g_npr[li] = ops.ref_losses(mo, mp, config.orientation_loss_target == 'normals_pred',
                                              lv['weights'], lv['mlp'].get('normals'), lv['mlp'].get('npred'), R.viewdirs,
                                              stats[2 * nlev + 3:2 * nlev + 5], g_w[li], True, B_valid=B0)

    g_expo = None
    if model.expo_off is not None and R.exposure_idx is not None:
      g_expo = model._buf(('train', 'g_expo'), (Bp, 3), f32)
      g_expo.zero_()
    # Levels are independent in the backward pass (stop_level_grad).  With more than one rank the NeRF level goes
    # first: after it the gradients of NerfMLP_0 (95 % of the parameters at 360.gin) and of the GLO table are final, and
    # their all-reduce runs under the proposal levels' backward (~1/5 of the step) instead of after it.
    order = list(range(nlev))
    early = []                                                         # [(begin, end, handle)]
    overlap = mdist.world_size() > 1 and nlev > 1 and not model.single_mlp and model.stop_level_grad and model._pad_index is None
    if overlap:
      order = [nlev - 1] + order[:-1]

    def level_backward(li):
      lv = levels[li]
      if data_spec[li]['mult'] > 0 or w_spec[li] is not None or g_w[li] is not None:
        model.backward_level(lv, flat, grads, None, g_w[li], g_expo, g_nrm[li], g_npr[li],
                             losses=dict(B_valid=B0, data=data_spec[li], weights=w_spec[li]))
      else:                                                            # no gradient reaches this level: its mse for the log
        d = data_spec[li]
        ops.data_loss(d['type'], d['charb_padding'], d['mult'], d['rgb_out'], gt, lossmult, denom, d['stats'], B_valid=B0,
                      want_grad=False)
      if overlap and li == nlev - 1:
        for name, b, e in model.modules:
          if name in ('NerfMLP_0', 'Embed_0'):
            # (weight decay of every key inside this module goes in before its reduce: train_utils.py:300-305)
            for key, mult in (config.weight_decay_mults or {}).items():
              kb, ke = wd_range(key)
              if b <= kb and ke <= e:
                ops.weight_decay(flat, kb, ke, mult, grads, stats[4 * nlev + 5:4 * nlev + 6])
            early.append((b, e, mdist.all_reduce_sum_async(grads[b:e])))

    def needs_grad(li):
      return data_spec[li]['mult'] > 0 or w_spec[li] is not None or g_w[li] is not None

    def props_backward(lis):
      """The proposal levels `lis` (ascending): as ONE pass when their buffers are grouped (models.Model._props_group) and
      every one of them receives gradient, else level by level."""
      lvs = [levels[li] for li in lis]
      if len(lis) > 1 and all(lv.get('group') == (k, len(lis)) for k, lv in enumerate(lvs)) and all(map(needs_grad, lis)):
        model.backward_prop_levels(lvs, flat, grads, [g_w[li] for li in lis],
                                   [dict(B_valid=B0, data=data_spec[li], weights=w_spec[li]) for li in lis])
      else:
        for li in lis:
          level_backward(li)

    def levels_backward_through_the_sampling():
      """Model.stop_level_grad = False (models.py:198-201): the levels are no longer independent.  Last level first; each
      level's loss gradient with respect to its own sample distances (the compositing's optical-depth increments, the
      Gaussians' interval ends behind the features, the distortion loss, and what the next level sent back) goes through
      the VJP of the level's resampling into the previous level's distances and weights (oracle/models.py:487-496; the
      reference differentiates stepfun.sample_intervals, max_dilate_weights and the logits of models.py:183-185)."""
      g_s_next = None
      for li in reversed(range(nlev)):
        lv = levels[li]
        plan = lv['plan']
        Bp_, n_ = lv['sdist'].shape[0], lv['n']
        through = li >= 1                                # level 0 resamples a constant histogram: nothing upstream of it
        g_x = model._buf(('train', 'g_x', li), (Bp_, n_), f32) if through else None
        g_feat = [] if through else None
        g_tfeat = [] if (through and plan.tangent) else None     # (density-gradient normals: a function of the sample positions too)
        model.backward_level(lv, flat, grads, None, g_w[li], g_expo, g_nrm[li], g_npr[li],
                             losses=dict(B_valid=B0, data=data_spec[li], weights=w_spec[li]), g_x_out=g_x, g_feat_out=g_feat,
                             g_tfeat_out=g_tfeat)
        if not through:
          break
        while len(g_feat) > 2:                           # more than one skip layer: fold the extra matrices into the first
          extra = g_feat.pop()
          ops.add_cols_bf16(g_feat[0], extra, g_feat[0], plan.ldF)
        hp = plan.hp
        g_t0, g_t1 = ops.cast_rays_ipe_bwd(
            lv['tdist'], R.origins, R.directions, lv['radii'], plan.basis_dev, g_feat[0], g_feat[1] if len(g_feat) > 1 else None,
            ray_shape=model.ray_shape, warp_contract=(hp.warp_fn == 'contract'), min_deg=hp.min_deg_point,
            max_deg=hp.max_deg_point, disable_integration=model.disable_integration,
            g_t0=model._buf(('train', 'g_t0', li), (Bp_ * n_,), f32), g_t1=model._buf(('train', 'g_t1', li), (Bp_ * n_,), f32))
        if g_tfeat:
          # ... and through the tangent rows of the density-gradient normals (mnr_cast_rays_ipe_tangent_bwd accumulates)
          while len(g_tfeat) > 2:
            extra = g_tfeat.pop()
            ops.add_cols_bf16(g_tfeat[0], extra, g_tfeat[0], plan.ldF)
          ops.cast_rays_ipe_tangent_bwd(
              lv['tdist'], R.origins, R.directions, lv['radii'], plan.basis_dev, g_tfeat[0], g_tfeat[1] if len(g_tfeat) > 1 else None,
              g_t0, g_t1, ray_shape=model.ray_shape, warp_contract=(hp.warp_fn == 'contract'), min_deg=hp.min_deg_point,
              max_deg=hp.max_deg_point, disable_integration=model.disable_integration)
        ccfg = lv['ccfg']
        fine = li == nlev - 1
        g_s = ops.sdist_bwd(
            lv['sdist'], lv['near'], lv['far'], model.raydist_fn, B_valid=B0, g_x=g_x, raw_density=lv['raw_density'],
            density_noise=lv['dnoise'], density_noise_std=ccfg.density_noise_std, density_bias=hp.density_bias,
            density_act=hp.density_activation, dirs=R.directions, g_t0=g_t0, g_t1=g_t1,
            distortion_mult=(config.distortion_loss_mult if fine else 0.0), weights=lv['weights'], g_sdist_in=g_s_next,
            out=model._buf(('train', 'g_sdist', li), (Bp_, n_ + 1), f32))
        sd_prev, w_prev, u_base, jitter = lv['rs_in']
        g_s_next, g_w_prev = ops.resample_level_bwd(
            sd_prev, w_prev, u_base, jitter, g_s, **lv['rs_kw'],
            g_sdist_prev=model._buf(('train', 'g_sdist_prev', li), tuple(sd_prev.shape), f32),
            g_w_prev=model._buf(('train', 'g_w_prev', li), tuple(w_prev.shape), f32))
        if g_w[li - 1] is None:
          g_w[li - 1] = g_w_prev
        else:
          g_w[li - 1].add_(g_w_prev)

    bs = backward_streams(dev)
    prop_lis = list(range(nlev - 1))
    if not model.stop_level_grad:
      levels_backward_through_the_sampling()
    elif bs is None:
      if order[0] == nlev - 1:
        level_backward(nlev - 1)
        props_backward(prop_lis)
      else:
        props_backward(prop_lis)
        level_backward(nlev - 1)
    else:
      # Two streams side by side: the proposal levels (HBM-bound) on `bs.prop`, the NeRF level (MFMA-bound) on the caller's
      # stream, which then waits for the side stream.
      cur = torch.cuda.current_stream(dev)
      ready = torch.cuda.Event()
      ready.record(cur)
      bs.prop.wait_event(ready)
      with torch.cuda.stream(bs.prop):
        props_backward(prop_lis)
        done_props = torch.cuda.Event()
        done_props.record(bs.prop)
      level_backward(nlev - 1)
      cur.wait_event(done_props)
    if g_expo is not None:
      n_off = model.num_glo_embeddings * 3
      ops.exposure_scale_bwd(R.exposure_values.reshape(-1).contiguous().float(),
                             R.exposure_idx.reshape(-1).to(torch.int32).contiguous(), g_expo,
                             grads[model.expo_off:model.expo_off + n_off], B0)

    flat, grads = flat_true, model.true_grads(grads)                  # back in the callers' layout (the same tensors unless padded)
    # losses['weight'] = sum_k mult_k * |params[k]|^2 over the summarize_tree keys k of the parameter tree (train_utils.py:60-68,
    # 300-305: a module, a Dense inside it, or one kernel / bias); its gradient 2 * mult_k * params[k] goes into the same ranges
    for key, mult in (config.weight_decay_mults or {}).items():
      kb, ke = wd_range(key)
      if any(b <= kb and ke <= e for b, e, _ in early):
        continue                                                       # added before that module's early reduce
      ops.weight_decay(flat, kb, ke, mult, grads, stats[4 * nlev + 5:4 * nlev + 6])

    # pmean over the 'batch' axis (train_utils.py:319-321): RCCL all-reduce of the flat buffers.
    if early:
      done = sorted((b, e) for b, e, _ in early)
      pos = 0
      for b, e in done + [(model.num_params, model.num_params)]:      # the ranges the early reduces did not cover
        if b > pos:
          mdist.all_reduce_mean_(grads[pos:b])
        pos = max(pos, e)
      for b, e, h in early:
        mdist.finish_mean_(grads[b:e], h)
    else:
      mdist.all_reduce_mean_(grads)
    mdist.all_reduce_mean_(stats)

    raw_grads = grads.clone() if (return_grads or tree_stats) else None
    old_flat = flat.clone() if tree_stats else None                   # (logging steps only: stats['opt_update_*'], 'weight_l2s')
    # clip per top-level module, nan_to_num, Adam (train_utils.py:326-330)
    sq = model._buf(('train', 'sqnorm'), (len(model.modules),), f32)
    sq.zero_()
    step_count = state.step
    lr = lr_fn(step_count)            # optax evaluates the schedule at the pre-increment count
    for mi, (_, b, e) in enumerate(model.modules):
      ops.grad_sqnorm(grads, b, e, config.grad_max_val, sq[mi:mi + 1])
      ops.clip_adam(grads, flat, state.mu, state.nu, b, e, sq[mi:mi + 1], lr=lr, b1=config.adam_beta1,
                    b2=config.adam_beta2, eps=config.adam_eps, step=step_count + 1,
                    grad_max_val=config.grad_max_val, grad_max_norm=config.grad_max_norm)
    new_state = TrainState(step=step_count + 1, params=state.params, mu=state.mu, nu=state.nu)

    # (copies: `stats` and `sq` are workspace tensors the next step overwrites, and callers keep TrainStats in buffers)
    out_stats = {'_raw': stats.clone(), '_nlev': nlev, 'grad_sqnorms': sq.clone(), '_disp': config.compute_disp_metrics,
                 '_normal': config.compute_normal_metrics}
    if return_grads:
      out_stats['_grads'] = raw_grads
    if tree_stats:
      # what train_utils.py:304,323-324,334-335 log per summarize_tree key: evaluated lazily in TrainStats.materialize
      out_stats['_tree'] = dict(ranges=ranges, grads=raw_grads, old=old_flat, new=flat.clone())   # (a copy: `flat` is updated in place by the next step)
    return new_state, TrainStats(out_stats), rng

  return train_step


class TrainStats(dict):
  """Lazy view of the device-side stats vector (no host sync until a value is read)."""

  def materialize(self):
    raw = self['_raw'].detach().cpu().numpy()
    n = self['_nlev']
    mses = raw[0:2 * n:2]
    data = raw[1:2 * n:2]
    out = {
        'mses': mses,
        'psnrs': -10. / math.log(10.) * np.log(mses),
        'losses': {'data': float(data.sum()), 'interlevel': float(raw[2 * n]), 'distortion': float(raw[2 * n + 1])},
    }
    # (each term under its own multipliers, train_utils.py:281-290)
    if raw[2 * n + 3] != 0:
      out['losses']['orientation'] = float(raw[2 * n + 3])
    if raw[2 * n + 4] != 0:
      out['losses']['predicted_normals'] = float(raw[2 * n + 4])
    if raw[4 * n + 5] != 0:
      out['losses']['weight'] = float(raw[4 * n + 5])
    if self.get('_disp'):
      out['disparity_mses'] = raw[2 * n + 5:3 * n + 5]
    if self.get('_normal'):
      out['normal_maes'] = raw[3 * n + 5:4 * n + 5]
    out['loss'] = sum(out['losses'].values())
    out['psnr'] = float(out['psnrs'][-1])
    # per top-level module, AFTER the clip by value (what clip_adam's norm clip sees); the reference logs the
    # pre-clip norm (train_utils.py:323-324): identical whenever grad_max_val = 0, as in every shipped config
    out['grad_norms'] = np.sqrt(self['grad_sqnorms'].detach().cpu().numpy())
    t = self.get('_tree')
    if t is not None:
      # train_utils.py:304 weight_l2s (squared norms of the parameters the step started from), :323-324 grad_norms / grad_maxes of
      # the pmean'ed gradient before clipping, :332-335 opt_update_norms / opt_update_maxes of new - old parameters; one entry
      # per summarize_tree key (module, module/Dense_k, module/Dense_k/kernel|bias)
      delta = t['new'] - t['old']
      l2, gn, gm, un, um = {}, {}, {}, {}, {}
      for key, (b, e) in t['ranges'].items():
        l2[key] = float((t['old'][b:e].double() ** 2).sum())
        gn[key] = float(t['grads'][b:e].double().norm())
        gm[key] = float(t['grads'][b:e].abs().max()) if e > b else 0.0
        un[key] = float(delta[b:e].double().norm())
        um[key] = float(delta[b:e].abs().max()) if e > b else 0.0
      out.update(weight_l2s=l2, grad_norms=gn, grad_maxes=gm, opt_update_norms=un, opt_update_maxes=um)
    return out


def create_render_fn(model: models.Model):
  """train_utils.py:377-396: deterministic forward with extras; pixels all-gathered across ranks."""

  def render_eval_fn(variables, train_frac, _, rays):
    renderings, ray_history = model._forward(variables['flat'], None, rays, train_frac, True)
    if mdist.world_size() > 1:
      # the pixel buffers of every level travel in ONE all-gather per chunk (the ray_* visualisation samples stay local)
      keys = [(li, k) for li, r in enumerate(renderings) for k in r if not k.startswith('ray_')]
      gathered = mdist.all_gather_packed([renderings[li][k].float() for li, k in keys])
      renderings = [dict(r) for r in renderings]
      for (li, k), g in zip(keys, gathered):
        renderings[li][k] = g
    return renderings, ray_history

  return render_eval_fn


def setup_model(config, rng, dataset=None, device='cuda'):
  """train_utils.py:399-419 -> (model, state, render_eval_pfn, train_pstep, lr_fn)."""
  dummy_rays = utils.dummy_rays(include_exposure_idx=config.rawnerf_mode, include_exposure_values=True)
  model, variables = models.construct_model(rng, dummy_rays, config, device=device)
  state, lr_fn = create_optimizer(config, variables)
  render_eval_pfn = create_render_fn(model)
  train_pstep = create_train_step(model, config, dataset=dataset)
  return model, state, render_eval_pfn, train_pstep, lr_fn
