"""Model.stop_level_grad = False (reference models.py:56,198-201): the VJPs of the sample positions, kernel by kernel and
composed, against the oracle's autograd (whose differentiated sampling path is pinned by the reference's own code through the
complex-step golden `blender_sampling_grad`, tests/golden/make_golden_models.py).  -m gpu; runs unchanged on the kernel-source
simulator (MNR_TESTS_ON_SIMULATOR=1, tests/test_sim_gpu_suite.py).

The oracle differentiates torch's own softmax / cumulative sum (float64 here), the kernels re-run their fp32 forward pass: a
sample that sits within an ulp of a CDF fence-post can land in neighbouring bins on the two sides, and its gradient then
differs by a finite amount.  Such rays are COUNTED (bounded, and reported), everything else is held to a float32 tolerance.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multinerf_amd import ops
from oracle import stepfun as ostep

dev = lambda t: None if t is None else t.cuda()


def _rays_close(got, want, rtol, atol, what, max_bad_rays=0):
  """Per-ray comparison: a ray is bad if any of its elements is outside atol + rtol * |want| (scaled by the ray's largest
  |want|); returns the number of bad rays and asserts it is at most `max_bad_rays`."""
  got, want = got.double().cpu(), want.double().cpu()
  scale = want.abs().amax(dim=-1, keepdim=True)
  bad = ((got - want).abs() > atol + rtol * (want.abs() + scale)).any(dim=-1)
  nbad = int(bad.sum())
  worst = ((got - want).abs() / (atol / max(rtol, 1e-30) + want.abs() + scale))[~bad].max().item() if (~bad).any() else 0.0
  print(f'{what}: {nbad} of {bad.numel()} rays outside rtol {rtol:g}; worst relative error of the others {worst:.2e}')
  assert nbad <= max_bad_rays, (what, nbad, max_bad_rays)
  return nbad


def _level_case(B, n_prev, n, use_dilation, dilation, anneal, padding, single_jitter, seed, peaked=False):
  g = torch.Generator().manual_seed(seed)
  # an incoming step function the way a level hands it on: sorted fence-posts in [0, 1] with clamped ends, weights that sum to <= 1
  c = torch.sort(torch.rand((B, n_prev), generator=g, dtype=torch.float64), dim=-1).values
  mid = (c[:, 1:] + c[:, :-1]) / 2
  first = torch.clamp(2 * c[:, :1] - mid[:, :1], min=0.0)
  last = torch.clamp(2 * c[:, -1:] - mid[:, -1:], max=1.0)
  sdist = torch.cat([first, mid, last], dim=-1)
  w = torch.rand((B, n_prev), generator=g, dtype=torch.float64) ** (6.0 if peaked else 1.5)
  w = w / w.sum(-1, keepdim=True) * (0.3 + 0.7 * torch.rand((B, 1), generator=g, dtype=torch.float64))
  jit = torch.rand((B, 1 if single_jitter else n), generator=g, dtype=torch.float64)
  g_out = torch.randn((B, n + 1), generator=g, dtype=torch.float64)
  # (the float32 values both sides see, carried in float64 on the oracle's side: a fence-post spacing of 1e-5 changes by
  # 0.6 % when its end points are rounded, and the pdf's gradient with it)
  return tuple(x.float().double() for x in (sdist, w, jit, g_out))


def _oracle_level_vjp(sdist, w, jit, g_out, n, use_dilation, dilation, anneal, padding, single_jitter, domain=(0.0, 1.0)):
  """The level's sampling as oracle/models.py runs it with stop_level_grad = False, in float64, and its VJP by autograd."""
  s = sdist.clone().requires_grad_(True)
  ww = w.clone().requires_grad_(True)
  t, wt = s, ww
  if use_dilation:
    t, wt = ostep.max_dilate_weights(s, ww, dilation, domain=domain, renormalize=True)
    t, wt = t[..., 1:-1], wt[..., 1:-1]
  logits = ostep.resample_logits(t, wt, anneal, padding, differentiable=True)
  out = ostep.sample_intervals(jit, t, logits, n, single_jitter=single_jitter, domain=domain, differentiable=True)
  (out * g_out).sum().backward()
  return out.detach(), s.grad, ww.grad


CASES = [
    # (B, n_prev, n, dilation on, dilation, anneal, padding, single_jitter)
    (96, 64, 64, True, 0.0103125, 0.909, 0.0, True),          # 360.gin level 1 (dilation 0.0025 + 0.5 / 64)
    (96, 64, 32, True, 0.00262207, 0.909, 0.0, True),         # 360.gin level 2 (0.0025 + 0.5 / 4096)
    (64, 128, 32, True, 0.00640625, 0.5, 0.0, True),          # blender_256 level 1 (0.0025 + 0.5 / 128)
    (64, 128, 128, False, 0.0, 1.0, 0.01, False),             # blender_refnerf / llff_raw: no dilation, padding, per-sample jitter
    (40, 8, 12, True, 0.08, 1.0, 0.0, False),                 # short rays, wide dilation (many clipped fence-posts)
    (33, 40, 24, True, 0.004, 0.3, 0.0, True),                # sample counts that are no multiples of 16 (chunk tails)
]


@pytest.mark.parametrize('B,n_prev,n,use_dil,dil,anneal,pad,single', CASES)
@pytest.mark.parametrize('peaked', [False, True])
def test_resample_level_bwd_is_the_oracles_autograd(B, n_prev, n, use_dil, dil, anneal, pad, single, peaked):
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  sdist, w, jit, g_out = _level_case(B, n_prev, n, use_dil, dil, anneal, pad, single, seed=7 + n_prev + n, peaked=peaked)
  out_o, gs_o, gw_o = _oracle_level_vjp(sdist, w, jit, g_out, n, use_dil, dil, anneal, pad, single)
  eps = float(np.finfo(np.float32).eps)
  u_max = eps + (1 - eps) / n
  max_jitter = (1 - u_max) / (n - 1) - eps
  u_base = torch.linspace(0, 1 - u_max, n, dtype=torch.float32)
  near, far = torch.full((B,), 2.0), torch.full((B,), 6.0)
  kw = dict(n_samples=n, use_dilation=use_dil, dilation=dil, domain=(0.0, 1.0), anneal=anneal, resample_padding=pad,
            single_jitter=single, max_jitter=max_jitter, raydist_fn=None)
  s32, w32, j32 = dev(sdist.float().contiguous()), dev(w.float().contiguous()), dev(jit.float().contiguous())
  sd, _ = ops.resample_level(s32, w32, dev(u_base), j32, dev(near), dev(far), **kw)
  gs, gw = ops.resample_level_bwd(s32, w32, dev(u_base), j32, dev(g_out.float().contiguous()), **kw)
  torch.cuda.synchronize()
  # the forward pass the VJP belongs to (fp32 kernel order against the float64 oracle)
  fwd_bad = _rays_close(sd, out_o, 1e-4, 2e-6, 'sdist', max_bad_rays=max(1, B // 24))
  # a ray whose sample changed bins (fwd_bad) has a different gradient by construction: allow those plus the same margin
  lim = fwd_bad + max(1, B // 24)
  # Fence-posts that sit exactly ON a domain end (a previous level's clamped first / last fence-post) tie with the dilation's
  # clipped copies of it; which of the tied entries math.sorted_interp's max / min hands the gradient to is a convention (jax
  # splits it evenly, torch picks one, the kernel takes the bracketing index), and the composed model never sees it: the
  # previous level's clamp blocks exactly that component (stepfun.py:258-259).  They are left out of the comparison.
  inner = ((sdist > 0.0) & (sdist < 1.0)).double()
  gs, gs_o = gs.double().cpu() * inner, gs_o * inner
  _rays_close(gs, gs_o, 2e-3, 1e-7, 'g_sdist_prev', max_bad_rays=lim)
  _rays_close(gw, gw_o, 2e-3, 1e-7, 'g_w_prev', max_bad_rays=lim)
  assert torch.isfinite(gs).all() and torch.isfinite(gw).all()
