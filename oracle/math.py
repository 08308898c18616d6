"""Oracle restatement of reference internal/math.py (TEST INFRASTRUCTURE ONLY)."""

import math as _pm

import numpy as np
import torch

PI = _pm.pi


def matmul(a, b):
  """math.py:21-23 -- full-precision matmul."""
  return torch.matmul(a, b)


def safe_trig_helper(x, fn, t=100 * PI):
  """math.py:26-28 -- mod x by 100*pi (sign of divisor) only when |x| >= t."""
  return fn(torch.where(torch.abs(x) < t, x, torch.remainder(x, t)))


def safe_cos(x):
  """math.py:31-33."""
  return safe_trig_helper(x, torch.cos)


def safe_sin(x):
  """math.py:36-38."""
  return safe_trig_helper(x, torch.sin)


class _SafeExp(torch.autograd.Function):
  """math.py:41-54 -- exp(min(x, 88)) whose gradient is its own output."""

  @staticmethod
  def forward(ctx, x):
    y = torch.exp(torch.clamp(x, max=88.))
    ctx.save_for_backward(y)
    return y

  @staticmethod
  def backward(ctx, g):
    (y,) = ctx.saved_tensors
    return g * y


def safe_exp(x):
  return _SafeExp.apply(x)


def log_lerp(t, v0, v1):
  """math.py:57-63."""
  if v0 <= 0 or v1 <= 0:
    raise ValueError(f'Interpolants {v0} and {v1} must be positive.')
  lv0 = np.log(v0)
  lv1 = np.log(v1)
  return np.exp(np.clip(t, 0, 1) * (lv1 - lv0) + lv0)


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0,
                        lr_delay_mult=1):
  """math.py:66-98 -- log-linear decay with a sine warm-up."""
  if lr_delay_steps > 0:
    delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(
        0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
  else:
    delay_rate = 1.
  return delay_rate * log_lerp(step / max_steps, lr_init, lr_final)


def interp(x, xp, fp):
  """math.py:101-105 -- row-wise np.interp (the 'gpu resampling' variant)."""
  xs, xps, fps = [a.reshape(-1, a.shape[-1]) for a in (x, xp, fp)]
  out = np.stack([
      np.interp(xs[i].numpy(), xps[i].numpy(), fps[i].numpy())
      for i in range(xs.shape[0])
  ])
  return torch.as_tensor(out, dtype=x.dtype).reshape(x.shape)


def sorted_interp(x, xp, fp, return_index=False):
  """math.py:108-127 -- brute-force-mask interpolation, xp/fp sorted.

  Returns fp0 + clip(nan_to_num((x-xp0)/(xp1-xp0)),0,1)*(fp1-fp0) where
  (xp0,fp0) come from the last i with xp[i] <= x and (xp1,fp1) from the first
  i with xp[i] > x.  With return_index the integer `count(xp <= x) - 1` is
  returned too: this is the bit-exact "sample index" artefact.
  """
  mask = x[..., None, :] >= xp[..., :, None]

  def find_interval(v):
    neg = torch.where(mask, v[..., None], v[..., :1, None])
    pos = torch.where(~mask, v[..., None], v[..., -1:, None])
    return neg.max(dim=-2).values, pos.min(dim=-2).values

  fp0, fp1 = find_interval(fp)
  xp0, xp1 = find_interval(xp)
  offset = torch.clamp(torch.nan_to_num((x - xp0) / (xp1 - xp0), nan=0.0), 0, 1)
  ret = fp0 + offset * (fp1 - fp0)
  if return_index:
    idx = mask.sum(dim=-2).to(torch.int32) - 1
    return ret, idx
  return ret
