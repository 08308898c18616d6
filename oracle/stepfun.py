"""Oracle restatement of reference internal/stepfun.py (TEST INFRASTRUCTURE ONLY).

Conventions as the reference (stepfun.py:15-23): last axis runs along the ray,
`t` has n+1 fence-posts, `w`/`y`/`p` have n bins.
"""

import numpy as np
import torch

from oracle import math as rmath


def _eps(x):
  return torch.finfo(x.dtype if x.dtype.is_floating_point else torch.float32).eps


F32_EPS = float(np.finfo(np.float32).eps)


def seq_cumsum(x):
  """Strict left-to-right cumulative sum in x's own dtype (no grad).

  torch.cumsum on CPU accumulates fp32 rows in double; numpy accumulates in the
  array dtype, element by element -- the order the HIP kernel uses.
  """
  return torch.from_numpy(np.cumsum(x.detach().numpy(), axis=-1, dtype=x.detach().numpy().dtype))


LANES = 16   # lanes that share one ray in csrc/resample.hip (RSP_LPR; tests/test_oracle_leaves.py holds the two together)

# Association order and transcendental functions of the fp32 sampling path.
#   'kernel'    : the HIP level kernel's documented order (blocked sums, its own exp / log): the oracle the kernel's
#                 sample INDICES are bit-exact against, by construction;
#   'reference' : the reference's (stepfun.py:146,156 as NumPy executes them): strict left-to-right sums and
#                 cumulative sums, libm exp / log.  The kernel's index mismatch rate against THIS order is measured and
#                 reported (tests/test_gpu_kernels.py::test_resample_level, tests/test_oracle_leaves.py), not assumed zero.
_ORDER = 'kernel'


class reference_order:
  """Context manager: evaluate the fp32 sampling path in the reference's association order (see _ORDER)."""

  def __enter__(self):
    global _ORDER
    self._prev, _ORDER = _ORDER, 'reference'
    return self

  def __exit__(self, *exc):
    global _ORDER
    _ORDER = self._prev
    return False


def blocked_cumsum(x, chunk=None):
  """Cumulative sum in the association order of the HIP level kernel (csrc/resample.hip), in x's own dtype (no grad).

  The 16 lanes of a ray each own a contiguous chunk of `chunk` = ceil(len / 16) elements: a lane sums its chunk left to
  right (starting from 0), the 16 chunk sums are added left to right (starting from 0) into exclusive chunk offsets, and
  the running sum at element k is offset[chunk of k] + (prefix of k inside its chunk).  This, not a strict left-to-right
  sum, is the documented order behind bit-exact sample indices; `chunk` overrides the chunk length when the summed array
  is a slice of the array the chunking is defined on (the CDF sums w[:-1] with the chunking of w)."""
  a = x.detach().numpy()
  n = a.shape[-1]
  if _ORDER == 'reference':                              # strict left to right
    cs = np.cumsum(a, axis=-1, dtype=a.dtype)
    tot = cs[..., -1] if n else np.zeros(a.shape[:-1], dtype=a.dtype)
    return torch.from_numpy(np.ascontiguousarray(cs)), torch.from_numpy(np.ascontiguousarray(tot))
  ch = int(chunk) if chunk is not None else -(-n // LANES)
  pad = LANES * ch - n
  assert pad >= 0
  ap = np.concatenate([a, np.zeros(a.shape[:-1] + (pad,), dtype=a.dtype)], axis=-1).reshape(a.shape[:-1] + (LANES, ch))
  local = np.zeros_like(ap)
  run = np.zeros(ap.shape[:-1], dtype=a.dtype)
  for i in range(ch):                                   # inside a chunk: left to right
    run = run + ap[..., i]
    local[..., i] = run
  sums = local[..., -1]                                 # (adding the zero padding changes nothing)
  offs = np.zeros_like(sums)
  tot = np.zeros(sums.shape[:-1], dtype=a.dtype)
  for l in range(LANES):                                # chunk sums: left to right
    offs[..., l] = tot
    tot = tot + sums[..., l]
  out = (offs[..., None] + local).reshape(a.shape[:-1] + (LANES * ch,))[..., :n]
  return (torch.from_numpy(np.ascontiguousarray(out).reshape(a.shape)),
          torch.from_numpy(np.ascontiguousarray(tot).reshape(a.shape[:-1])))


def blocked_sum(x, chunk=None):
  """Sum over the last axis in the kernel's blocked order (see blocked_cumsum); keeps the axis."""
  return blocked_cumsum(x, chunk)[1][..., None]


def searchsorted(a, v):
  """stepfun.py:30-53 -- (idx_lo, idx_hi) with a[idx_lo] <= v < a[idx_hi]."""
  i = torch.arange(a.shape[-1])
  v_ge_a = v[..., None, :] >= a[..., :, None]
  idx_lo = torch.where(v_ge_a, i[:, None], i[:1, None]).max(dim=-2).values
  idx_hi = torch.where(~v_ge_a, i[:, None], i[-1:, None]).min(dim=-2).values
  return idx_lo, idx_hi


def query(tq, t, y, outside_value=0):
  """stepfun.py:56-61."""
  idx_lo, idx_hi = searchsorted(t, tq)
  idx = torch.clamp(idx_lo, max=y.shape[-1] - 1)
  y_b = y.expand(idx.shape[:-1] + y.shape[-1:])
  yq = torch.where(idx_lo == idx_hi, torch.as_tensor(outside_value, dtype=y.dtype),
                   torch.gather(y_b, -1, idx))
  return yq


def inner_outer(t0, t1, y1):
  """stepfun.py:64-77 -- inner and outer measures of (t1, y1) on t0."""
  cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
  idx_lo, idx_hi = searchsorted(t1, t0)
  cy1_lo = torch.gather(cy1, -1, idx_lo)
  cy1_hi = torch.gather(cy1, -1, idx_hi)
  y0_outer = cy1_hi[..., 1:] - cy1_lo[..., :-1]
  y0_inner = torch.where(idx_hi[..., :-1] <= idx_lo[..., 1:],
                         cy1_lo[..., 1:] - cy1_hi[..., :-1],
                         torch.zeros_like(y0_outer))
  return y0_inner, y0_outer


def lossfun_outer(t, w, t_env, w_env, eps=F32_EPS):
  """stepfun.py:80-86 -- proposal weights must upper-bound the nerf weights."""
  _, w_outer = inner_outer(t, t_env, w_env)
  return torch.clamp(w - w_outer, min=0)**2 / (w + eps)


def weight_to_pdf(t, w, eps=F32_EPS**2):
  """stepfun.py:89-91."""
  return w / torch.clamp(t[..., 1:] - t[..., :-1], min=eps)


def pdf_to_weight(t, p):
  """stepfun.py:94-96."""
  return p * (t[..., 1:] - t[..., :-1])


def max_dilate(t, w, dilation, domain=(-np.inf, np.inf)):
  """stepfun.py:99-113 -- max-pool dilation of a non-negative step function."""
  t0 = t[..., :-1] - dilation
  t1 = t[..., 1:] + dilation
  t_dilate = torch.sort(torch.cat([t, t0, t1], dim=-1), dim=-1).values
  t_dilate = torch.clamp(t_dilate, domain[0], domain[1])
  inside = ((t0[..., None, :] <= t_dilate[..., None]) &
            (t1[..., None, :] > t_dilate[..., None]))
  w_dilate = torch.where(inside, w[..., None, :],
                         torch.zeros_like(w[..., None, :])).max(dim=-1).values[..., :-1]
  return t_dilate, w_dilate


def max_dilate_weights(t, w, dilation, domain=(-np.inf, np.inf),
                       renormalize=False, eps=F32_EPS**2):
  """stepfun.py:116-128."""
  p = weight_to_pdf(t, w)
  t_dilate, p_dilate = max_dilate(t, p, dilation, domain=domain)
  w_dilate = pdf_to_weight(t_dilate, p_dilate)
  if renormalize:
    # (float32 without grad: the kernel's blocked order, so that the sampled indices downstream are reproducible bit
    # for bit; anything else, e.g. the float64 goldens or a differentiated call: torch's sum)
    if w_dilate.dtype == torch.float32 and not w_dilate.requires_grad:
      total = blocked_sum(w_dilate)
    else:
      total = w_dilate.sum(dim=-1, keepdim=True)
    w_dilate = w_dilate / torch.clamp(total, min=eps)
  return t_dilate, w_dilate


def integrate_weights(w, sequential=True, blocked=False):
  """stepfun.py:131-150 -- [0, min(1, cumsum(w[:-1])), 1].

  `blocked` picks the level kernel's documented accumulation order (blocked_cumsum with the chunking of the
  whole of w; no grad), `sequential` a strict left-to-right one (the order of the render_extras kernel's
  percentiles; no grad); otherwise torch.cumsum (differentiable).
  """
  if blocked:
    cs = blocked_cumsum(w[..., :-1], chunk=-(-w.shape[-1] // LANES))[0]
  else:
    cs = seq_cumsum(w[..., :-1]) if sequential else torch.cumsum(w[..., :-1], dim=-1)
  cw = torch.clamp(cs, max=1)
  shape = cw.shape[:-1] + (1,)
  return torch.cat([torch.zeros(shape, dtype=w.dtype), cw,
                    torch.ones(shape, dtype=w.dtype)], dim=-1)


def resample_logits(sdist, weights, anneal, resample_padding, differentiable=False):
  """models.py:183-185: where(sdist[1:] > sdist[:-1], anneal * log(weights + padding), -inf), with the sampling path's
  own log in float32 (math.klog: the kernel's, bit for bit).  `differentiable` (Model.stop_level_grad = False, models.py:198-201):
  torch's log on the weights as they are, so that autograd reaches the previous level."""
  if differentiable:
    # (closed bins are constants: their weight is replaced by 1 INSIDE the log so that autograd's 0 * d log(0) is 0, not NaN.
    # A dilated histogram has such bins whenever two fence-posts are clipped to the same domain end, their weight is p * 0 = 0,
    # and with resample_padding = 0 the reference's own autodiff yields 0 * inf = NaN there; the NaN propagates to every
    # parameter-gradient element that weight depends on and the reference's train_step replaces those elements by 0
    # (nan_to_num, element-wise, train_utils.py:326-328); the complex-step golden through the reference's code, like this
    # form and like the HIP kernel, gives the derivative of the function that is actually evaluated.)
    open_ = sdist[..., 1:] > sdist[..., :-1]
    w = torch.where(open_, weights + resample_padding, torch.ones_like(weights))
    return torch.where(open_, anneal * torch.log(w), torch.full_like(w, -float('inf')))
  w = weights.detach() + resample_padding
  lg = rmath.klog(w) if (w.dtype == torch.float32 and _ORDER == 'kernel') else torch.log(w)
  return torch.where(sdist[..., 1:] > sdist[..., :-1], anneal * lg, torch.full_like(w, -float('inf')))


def softmax_seq(logits):
  """jax.nn.softmax (stepfun.py:156) with the level kernel's blocked denominator (blocked_cumsum) and, in float32, its
  exp (math.kexp)."""
  m = logits.max(dim=-1, keepdim=True).values
  e = rmath.kexp(logits - m) if (logits.dtype == torch.float32 and _ORDER == 'kernel') else torch.exp(logits - m)
  denom = blocked_sum(e)
  return e / denom


def invert_cdf(u, t, w_logits, use_gpu_resampling=False, return_index=False, differentiable=False):
  """stepfun.py:153-161.  `differentiable`: torch's softmax and cumulative sum (autograd through the CDF and, in
  math.sorted_interp, through the interval end points) instead of the level kernel's association order."""
  if differentiable:
    w = torch.softmax(w_logits, dim=-1)
    cw = integrate_weights(w, sequential=False)
  else:
    w = softmax_seq(w_logits)
    cw = integrate_weights(w, blocked=True)
  if use_gpu_resampling:
    return rmath.interp(u, cw, t)
  return rmath.sorted_interp(u, cw, t, return_index=return_index)


def sample_u(u_jitter, batch_shape, num_samples, single_jitter=False,
             deterministic_center=False, dtype=torch.float32):
  """The `u` construction of stepfun.py:191-209 with the jitter as an INPUT.

  `u_jitter` stands in for jax.random.uniform(rng, ...)/max_jitter, i.e. it is
  uniform in [0,1) with shape [..., 1] (single_jitter) or [..., num_samples];
  None means rng=None (deterministic).
  """
  eps = F32_EPS
  if u_jitter is None:
    if deterministic_center:
      pad = 1 / (2 * num_samples)
      u = torch.linspace(pad, 1. - pad - eps, num_samples, dtype=dtype)
    else:
      u = torch.linspace(0, 1. - eps, num_samples, dtype=dtype)
    u = u.expand(tuple(batch_shape) + (num_samples,))
  else:
    u_max = eps + (1 - eps) / num_samples
    max_jitter = (1 - u_max) / (num_samples - 1) - eps
    d = 1 if single_jitter else num_samples
    assert u_jitter.shape[-1] == d
    u = (torch.linspace(0, 1 - u_max, num_samples, dtype=dtype) +
         u_jitter.to(dtype) * max_jitter)
  return u


def sample(u_jitter, t, w_logits, num_samples, single_jitter=False,
           deterministic_center=False, use_gpu_resampling=False,
           return_index=False, differentiable=False):
  """stepfun.py:164-211 (rng replaced by the explicit `u_jitter` in [0,1))."""
  u = sample_u(u_jitter, t.shape[:-1], num_samples, single_jitter,
               deterministic_center, dtype=t.dtype)
  return invert_cdf(u, t, w_logits, use_gpu_resampling=use_gpu_resampling,
                    return_index=return_index, differentiable=differentiable)


def sample_intervals(u_jitter, t, w_logits, num_samples, single_jitter=False,
                     domain=(-np.inf, np.inf), use_gpu_resampling=False,
                     return_index=False, differentiable=False):
  """stepfun.py:214-263 -- sample intervals (fence-posts between sampled centers)."""
  if num_samples <= 1:
    raise ValueError(f'num_samples must be > 1, is {num_samples}.')
  out = sample(u_jitter, t, w_logits, num_samples, single_jitter,
               deterministic_center=True, use_gpu_resampling=use_gpu_resampling,
               return_index=return_index, differentiable=differentiable)
  centers, idx = out if return_index else (out, None)
  mid = (centers[..., 1:] + centers[..., :-1]) / 2
  minval, maxval = domain
  first = torch.clamp(2 * centers[..., :1] - mid[..., :1], min=minval)
  last = torch.clamp(2 * centers[..., -1:] - mid[..., -1:], max=maxval)
  t_samples = torch.cat([first, mid, last], dim=-1)
  if return_index:
    return t_samples, idx
  return t_samples


def lossfun_distortion(t, w):
  """stepfun.py:266-276 -- iint w_i w_j |t_i - t_j| + intra-interval term."""
  ut = (t[..., 1:] + t[..., :-1]) / 2
  dut = torch.abs(ut[..., :, None] - ut[..., None, :])
  loss_inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
  loss_intra = torch.sum(w**2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
  return loss_inter + loss_intra


def weighted_percentile(t, w, ps):
  """stepfun.py:298-308 -- np.interp of ps/100 into integrate_weights(w)."""
  cw = integrate_weights(w)
  cw_mat = cw.reshape(-1, cw.shape[-1]).detach().numpy()
  t_mat = t.reshape(-1, t.shape[-1]).detach().numpy()
  q = np.array(ps, dtype=cw_mat.dtype) / 100
  out = np.stack([np.interp(q, cw_mat[i], t_mat[i]) for i in range(cw_mat.shape[0])])
  return torch.as_tensor(out, dtype=t.dtype).reshape(cw.shape[:-1] + (len(ps),))
