"""Oracle restatement of reference internal/coord.py (TEST INFRASTRUCTURE ONLY)."""

import numpy as np
import torch

from oracle import math as rmath

F32_EPS = float(np.finfo(np.float32).eps)


def contract(x):
  """coord.py:21-27 -- mip-NeRF 360 eq. 10 scene contraction."""
  eps = F32_EPS
  x_mag_sq = torch.clamp(torch.sum(x**2, dim=-1, keepdim=True), min=eps)
  z = torch.where(x_mag_sq <= 1, x, ((2 * torch.sqrt(x_mag_sq) - 1) / x_mag_sq) * x)
  return z


def inv_contract(z):
  """coord.py:30-36."""
  eps = F32_EPS
  z_mag_sq = torch.clamp(torch.sum(z**2, dim=-1, keepdim=True), min=eps)
  return torch.where(z_mag_sq <= 1, z, z / (2 * torch.sqrt(z_mag_sq) - z_mag_sq))


def contract_jacobian(x):
  """Closed-form Jacobian of `contract` (what jax.linearize yields, coord.py:58).

  J = I inside the unit ball; outside J = s*I + (2(1-sqrt(m))/m^2) x x^T with
  s = (2 sqrt(m) - 1)/m, m = max(eps, |x|^2).
  """
  eps = F32_EPS
  m = torch.clamp(torch.sum(x**2, dim=-1, keepdim=True), min=eps)
  sq = torch.sqrt(m)
  s = (2 * sq - 1) / m
  c = 2 * (1 - sq) / (m * m)
  eye = torch.eye(x.shape[-1], dtype=x.dtype)
  j_out = s[..., None] * eye + c[..., None] * (x[..., :, None] * x[..., None, :])
  return torch.where((m <= 1)[..., None], eye.expand(j_out.shape), j_out)


def track_linearize(fn, mean, cov):
  """coord.py:39-60 -- Kalman-style push of (mean, cov) through fn.

  The reference uses jax.linearize + two vmaps; we evaluate the same JVPs with
  torch.func.jvp on the three basis directions (differentiable, any fn).
  """
  if (mean.dim() + 1) != cov.dim():
    raise ValueError('cov must be non-diagonal')
  if fn is contract:
    fn_mean = contract(mean)
    jac = contract_jacobian(mean)
  else:
    fn_mean = fn(mean)
    cols = []
    for k in range(mean.shape[-1]):
      e = torch.zeros_like(mean)
      e[..., k] = 1
      cols.append(torch.func.jvp(fn, (mean,), (e,))[1])
    jac = torch.stack(cols, dim=-1)  # jac[..., i, k] = d fn_i / d x_k
  fn_cov = jac @ cov @ jac.transpose(-1, -2)
  return fn_mean, fn_cov


def construct_ray_warps(fn, t_near, t_far):
  """coord.py:63-99 -- fn is None, 'piecewise', or one of the names below."""
  if fn is None:
    fn_fwd = lambda x: x
    fn_inv = lambda x: x
  elif fn == 'piecewise':
    fn_fwd = lambda x: torch.where(x < 1, .5 * x, 1 - .5 / x)
    fn_inv = lambda x: torch.where(x < .5, 2 * x, .5 / (1 - x))
  else:
    table = {
        'reciprocal': (torch.reciprocal, torch.reciprocal),
        'log': (torch.log, torch.exp),
        'exp': (torch.exp, torch.log),
        'sqrt': (torch.sqrt, torch.square),
        'square': (torch.square, torch.sqrt),
    }
    fn_fwd, fn_inv = table[fn]
  s_near, s_far = [fn_fwd(x) for x in (t_near, t_far)]
  t_to_s = lambda t: (fn_fwd(t) - s_near) / (s_far - s_near)
  s_to_t = lambda s: fn_inv(s * s_far + (1 - s) * s_near)
  return t_to_s, s_to_t


def expected_sin(mean, var):
  """coord.py:102-104."""
  return torch.exp(-0.5 * var) * rmath.safe_sin(mean)


def integrated_pos_enc(mean, var, min_deg, max_deg):
  """coord.py:107-126 -- layout [sin block (degree-major, basis-minor), cos block]."""
  scales = (2.0**torch.arange(min_deg, max_deg)).to(mean.dtype)
  shape = mean.shape[:-1] + (-1,)
  scaled_mean = torch.reshape(mean[..., None, :] * scales[:, None], shape)
  scaled_var = torch.reshape(var[..., None, :] * scales[:, None]**2, shape)
  return expected_sin(
      torch.cat([scaled_mean, scaled_mean + 0.5 * rmath.PI], dim=-1),
      torch.cat([scaled_var] * 2, dim=-1))


def lift_and_diagonalize(mean, cov, basis):
  """coord.py:129-133 -- basis is [3, K]."""
  fn_mean = rmath.matmul(mean, basis)
  fn_cov_diag = torch.sum(basis * rmath.matmul(cov, basis), dim=-2)
  return fn_mean, fn_cov_diag


def pos_enc(x, min_deg, max_deg, append_identity=True):
  """coord.py:136-147 -- plain sin (no safe_sin)."""
  scales = (2.0**torch.arange(min_deg, max_deg)).to(x.dtype)
  shape = x.shape[:-1] + (-1,)
  scaled_x = torch.reshape(x[..., None, :] * scales[:, None], shape)
  four_feat = torch.sin(torch.cat([scaled_x, scaled_x + 0.5 * rmath.PI], dim=-1))
  if append_identity:
    return torch.cat([x, four_feat], dim=-1)
  return four_feat
