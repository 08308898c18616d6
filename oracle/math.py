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


# ----------------------------------------------------------------------------- the sampling path's exp / log
#
# csrc/resample.hip does not call the device library's expf / logf (their last bit differs from the host library's):
# it spells both out (Cephes' single-precision algorithms), every operation a separately rounded fp32 +, -, *.  These
# are the same operations in the same order in NumPy float32, so logits, softmax and CDF are the same bits on both
# sides: the precondition of bit-exact sample indices.  (float32 only; ~1 ulp, like the libraries'.)

_F = np.float32


def kexp(x):
  """rs_exp of csrc/resample.hip on a float32 array / tensor (returns the same type)."""
  is_t = isinstance(x, torch.Tensor)
  a = (x.detach().numpy() if is_t else np.asarray(x)).astype(np.float32)
  with np.errstate(all='ignore'):
    xs = np.where(np.isfinite(a), a, _F(0))
    fn = np.floor(xs * _F(1.44269504088896341) + _F(0.5)).astype(np.float32)
    r = xs - fn * _F(0.693359375)
    r = r - fn * _F(-2.12194440e-4)
    z = r * r
    p = np.full_like(r, _F(1.9875691500E-4))
    for c in (1.3981999507E-3, 8.3334519073E-3, 4.1665795894E-2, 1.6666665459E-1, 5.0000001201E-1):
      p = p * r + _F(c)
    p = p * z + r
    p = p + _F(1.0)
    n = np.clip(fn, -190, 127).astype(np.int32)
    pow2 = lambda k: ((k + 127).astype(np.int32) << 23).view(np.float32)
    normal = p * pow2(np.maximum(n, -126))
    sub = (p * pow2(np.minimum(n, -127) + 64)) * _F(2.0 ** -64)
    out = np.where(n >= -126, normal, sub).astype(np.float32)
    out = np.where(a < _F(-103.9720840454), _F(0), out)
    out = np.where(a > _F(88.7228317261), _F(np.inf), out)
    out = np.where(np.isnan(a), a, out).astype(np.float32)
  return torch.from_numpy(out) if is_t else out


def klog(x):
  """rs_log of csrc/resample.hip on a float32 array / tensor (returns the same type)."""
  is_t = isinstance(x, torch.Tensor)
  a = (x.detach().numpy() if is_t else np.asarray(x)).astype(np.float32)
  with np.errstate(all='ignore'):
    ok = np.isfinite(a) & (a > 0)
    xs = np.where(ok, a, _F(1))
    tiny = xs < _F(1.17549435e-38)
    xs = np.where(tiny, xs * _F(8388608.0), xs).astype(np.float32)
    e = np.where(tiny, -23, 0).astype(np.int32)
    bits = xs.view(np.uint32)
    e = e + ((bits >> 23) & 0xff).astype(np.int32) - 126
    m = ((bits & np.uint32(0x007fffff)) | np.uint32(0x3f000000)).view(np.float32)
    small = m < _F(0.707106781186547524)
    e = np.where(small, e - 1, e)
    m = np.where(small, m + m - _F(1.0), m - _F(1.0)).astype(np.float32)
    z = m * m
    y = np.full_like(m, _F(7.0376836292E-2))
    for sign, c in ((-1, 1.1514610310E-1), (1, 1.1676998740E-1), (-1, 1.2420140846E-1), (1, 1.4249322787E-1),
                    (-1, 1.6668057665E-1), (1, 2.0000714765E-1), (-1, 2.4999993993E-1), (1, 3.3333331174E-1)):
      y = y * m + _F(c) if sign > 0 else y * m - _F(c)
    y = y * m
    y = y * z
    fe = e.astype(np.float32)
    y = y + _F(-2.12194440e-4) * fe
    y = y + _F(-0.5) * z
    r = m + y
    r = r + _F(0.693359375) * fe
    out = r.astype(np.float32)
    out = np.where(a == 0, _F(-np.inf), out)
    out = np.where(a < 0, _F(np.nan), out)
    out = np.where(a == np.inf, a, out)
    out = np.where(np.isnan(a), a, out).astype(np.float32)
  return torch.from_numpy(out) if is_t else out
