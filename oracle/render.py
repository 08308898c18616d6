"""Oracle restatement of reference internal/render.py (TEST INFRASTRUCTURE ONLY)."""

import numpy as np
import torch

from oracle import stepfun

F32_EPS = float(np.finfo(np.float32).eps)


def lift_gaussian(d, t_mean, t_var, r_var, diag):
  """render.py:21-41 -- lift a 1-D Gaussian along a ray into 3-D."""
  mean = d[..., None, :] * t_mean[..., None]
  d_mag_sq = torch.clamp(torch.sum(d**2, dim=-1, keepdim=True), min=1e-10)
  if diag:
    d_outer_diag = d**2
    null_outer_diag = 1 - d_outer_diag / d_mag_sq
    t_cov_diag = t_var[..., None] * d_outer_diag[..., None, :]
    xy_cov_diag = r_var[..., None] * null_outer_diag[..., None, :]
    return mean, t_cov_diag + xy_cov_diag
  d_outer = d[..., :, None] * d[..., None, :]
  eye = torch.eye(d.shape[-1], dtype=d.dtype)
  null_outer = eye - d[..., :, None] * (d / d_mag_sq)[..., None, :]
  t_cov = t_var[..., None, None] * d_outer[..., None, :, :]
  xy_cov = r_var[..., None, None] * null_outer[..., None, :, :]
  return mean, t_cov + xy_cov


def conical_frustum_to_gaussian(d, t0, t1, base_radius, diag, stable=True):
  """render.py:44-78 -- mip-NeRF eq. 7 (stable) / eqs. 37-39."""
  if stable:
    mu = (t0 + t1) / 2
    hw = (t1 - t0) / 2
    eps = F32_EPS
    t_mean = mu + (2 * mu * hw**2) / torch.clamp(3 * mu**2 + hw**2, min=eps)
    denom = torch.clamp(3 * mu**2 + hw**2, min=eps)
    t_var = (hw**2) / 3 - (4 / 15) * hw**4 * (12 * mu**2 - hw**2) / denom**2
    r_var = (mu**2) / 4 + (5 / 12) * hw**2 - (4 / 15) * (hw**4) / denom
  else:
    t_mean = (3 * (t1**4 - t0**4)) / (4 * (t1**3 - t0**3))
    r_var = 3 / 20 * (t1**5 - t0**5) / (t1**3 - t0**3)
    t_mosq = 3 / 5 * (t1**5 - t0**5) / (t1**3 - t0**3)
    t_var = t_mosq - t_mean**2
  r_var = r_var * base_radius**2
  return lift_gaussian(d, t_mean, t_var, r_var, diag)


def cylinder_to_gaussian(d, t0, t1, radius, diag):
  """render.py:81-100."""
  t_mean = (t0 + t1) / 2
  r_var = radius**2 / 4
  t_var = (t1 - t0)**2 / 12
  return lift_gaussian(d, t_mean, t_var, r_var, diag)


def cast_rays(tdist, origins, directions, radii, ray_shape, diag=True):
  """render.py:103-127."""
  t0 = tdist[..., :-1]
  t1 = tdist[..., 1:]
  if ray_shape == 'cone':
    gaussian_fn = conical_frustum_to_gaussian
  elif ray_shape == 'cylinder':
    gaussian_fn = cylinder_to_gaussian
  else:
    raise ValueError('ray_shape must be \'cone\' or \'cylinder\'')
  means, covs = gaussian_fn(directions, t0, t1, radii, diag)
  means = means + origins[..., None, :]
  return means, covs


def compute_alpha_weights(density, tdist, dirs, opaque_background=False):
  """render.py:130-151 -- (weights, alpha, trans)."""
  t_delta = tdist[..., 1:] - tdist[..., :-1]
  delta = t_delta * torch.linalg.norm(dirs[..., None, :], dim=-1)
  density_delta = density * delta
  if opaque_background:
    density_delta = torch.cat([
        density_delta[..., :-1],
        torch.full_like(density_delta[..., -1:], float('inf'))
    ], dim=-1)
  alpha = 1 - torch.exp(-density_delta)
  trans = torch.exp(-torch.cat([
      torch.zeros_like(density_delta[..., :1]),
      torch.cumsum(density_delta[..., :-1], dim=-1)
  ], dim=-1))
  weights = alpha * trans
  return weights, alpha, trans


def volumetric_rendering(rgbs, weights, tdist, bg_rgbs, t_far, compute_extras,
                         extras=None):
  """render.py:154-213."""
  eps = F32_EPS
  rendering = {}
  acc = weights.sum(dim=-1)
  bg_w = torch.clamp(1 - acc[..., None], min=0)
  rgb = (weights[..., None] * rgbs).sum(dim=-2) + bg_w * bg_rgbs
  rendering['rgb'] = rgb

  if compute_extras:
    rendering['acc'] = acc
    if extras is not None:
      for k, v in extras.items():
        if v is not None:
          rendering[k] = (weights[..., None] * v).sum(dim=-2)

    expectation = lambda x: (weights * x).sum(dim=-1) / torch.clamp(acc, min=eps)
    t_mids = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
    dm = torch.exp(expectation(torch.log(t_mids)))
    dm = torch.nan_to_num(dm, nan=float('inf'))
    rendering['distance_mean'] = torch.minimum(
        torch.maximum(dm, tdist[..., 0]), tdist[..., -1])

    t_aug = torch.cat([tdist, t_far], dim=-1)
    weights_aug = torch.cat([weights, bg_w], dim=-1)
    ps = [5, 50, 95]
    distance_percentiles = stepfun.weighted_percentile(t_aug, weights_aug, ps)
    for i, p in enumerate(ps):
      s = 'median' if p == 50 else 'percentile_' + str(p)
      rendering['distance_' + s] = distance_percentiles[..., i]
  return rendering
