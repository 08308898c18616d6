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


# ----------------------------------------------------------------------------- features -> interval ends


def _ray_case(B, n, seed, far_samples=False, ndc=False):
  g = torch.Generator().manual_seed(seed)
  o = (torch.rand((B, 3), generator=g, dtype=torch.float64) * 2 - 1) * (0.3 if ndc else 1.0)
  tgt = 0.3 * torch.randn((B, 3), generator=g, dtype=torch.float64)
  d = tgt - o
  d = d / d.norm(dim=-1, keepdim=True) * (1.0 + 0.2 * torch.rand((B, 1), generator=g, dtype=torch.float64))
  radii = (3e-4 + 7e-4 * torch.rand((B, 1), generator=g, dtype=torch.float64)) * (30.0 if ndc else 1.0)
  c = torch.sort(torch.rand((B, n + 1), generator=g, dtype=torch.float64), dim=-1).values
  if far_samples:        # reciprocal spacing between 0.2 and 1e3: the contracted region of 360.gin
    tdist = 1.0 / (c * (1.0 / 1e3) + (1 - c) * (1.0 / 0.2))
    tdist = torch.flip(tdist, dims=(-1,))
    tdist = torch.sort(tdist, dim=-1).values
  else:
    tdist = (0.0 if ndc else 2.0) + c * (1.0 if ndc else 4.0)
  return tuple(x.float().double() for x in (o, d, radii, tdist))


IPE_CASES = [
    # (name, ray_shape, contract, basis, min_deg, max_deg, far samples, ndc, disable_integration)
    ('360', 'cone', True, ('icosahedron', 2), 0, 12, True, False, False),
    ('360-near', 'cone', True, ('icosahedron', 2), 0, 12, False, False, False),        # samples on both sides of the unit sphere
    ('blender', 'cone', False, ('octahedron', 1), 0, 16, False, False, False),
    ('llff_raw', 'cylinder', False, ('octahedron', 1), 0, 16, False, True, False),
    ('no-integration', 'cone', True, ('icosahedron', 2), 0, 8, True, False, True),
]


@pytest.mark.parametrize('name,ray_shape,contract,basis,min_deg,max_deg,far,ndc,no_int', IPE_CASES)
def test_cast_rays_ipe_bwd_is_the_oracles_autograd(name, ray_shape, contract, basis, min_deg, max_deg, far, ndc, no_int):
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from multinerf_amd import geopoly
  from oracle import coord as ocoord
  from oracle import render as orender
  B, n = 24, 16
  o, d, radii, tdist = _ray_case(B, n, seed=31 + len(name), far_samples=far, ndc=ndc)
  if name == '360-near':
    tdist = tdist * 0.5 - 0.6        # t in [0.4, 2.4]: with |o| <= 1.7 the means cross |x| = 1
    tdist = tdist.float().double()
  P = torch.as_tensor(geopoly.generate_basis(*basis), dtype=torch.float64)      # [K, 3]
  K, L = P.shape[0], max_deg - min_deg
  F = 2 * K * L
  ld = (F + 127) // 128 * 128
  gen = torch.Generator().manual_seed(5)
  gA = torch.randn((B * n, ld), generator=gen).to(torch.bfloat16)
  gB = torch.randn((B * n, ld), generator=gen).to(torch.bfloat16)
  # oracle: features(tdist) in float64, loss = <gA + gB, features>
  t = tdist.clone().requires_grad_(True)
  means, covs = orender.cast_rays(t, o, d, radii, ray_shape, diag=False)
  if no_int:
    covs = torch.zeros_like(covs)
  if contract:
    means, covs = ocoord.track_linearize(ocoord.contract, means, covs)
  lm, lv = ocoord.lift_and_diagonalize(means, covs, P.t())
  feat = ocoord.integrated_pos_enc(lm, lv, min_deg, max_deg).reshape(B * n, F)
  gsum = (gA.double() + gB.double())[:, :F]
  (feat * gsum).sum().backward()
  want = t.grad
  f32c = lambda x: dev(x.float().contiguous())
  g_t0, g_t1 = ops.cast_rays_ipe_bwd(f32c(tdist), f32c(o), f32c(d), f32c(radii.reshape(-1)), f32c(P), dev(gA), dev(gB),
                                     ray_shape=ray_shape, warp_contract=contract, min_deg=min_deg, max_deg=max_deg,
                                     disable_integration=no_int)
  torch.cuda.synchronize()
  g_t0, g_t1 = g_t0.view(B, n).double().cpu(), g_t1.view(B, n).double().cpu()
  got = torch.zeros((B, n + 1), dtype=torch.float64)
  got[:, :n] += g_t0
  got[:, 1:] += g_t1
  # fp32 evaluation of sin(mean 2^l): an argument of 2^11 |x| carries 2^11 eps of absolute error, and the gradient adds 504
  # such terms with random signs: a few 1e-3 of the gradient's scale at degree 12, more at degree 16
  tol = 2e-3 if max_deg <= 12 else 1.5e-2                                     # (measured 2e-4 / 3e-3)
  _rays_close(got, want, tol, 1e-6, f'g_tdist [{name}]')
  # one source only
  g0, g1 = ops.cast_rays_ipe_bwd(f32c(tdist), f32c(o), f32c(d), f32c(radii.reshape(-1)), f32c(P), dev(gA), None,
                                 ray_shape=ray_shape, warp_contract=contract, min_deg=min_deg, max_deg=max_deg,
                                 disable_integration=no_int)
  t2 = tdist.clone().requires_grad_(True)
  means, covs = orender.cast_rays(t2, o, d, radii, ray_shape, diag=False)
  if no_int:
    covs = torch.zeros_like(covs)
  if contract:
    means, covs = ocoord.track_linearize(ocoord.contract, means, covs)
  lm, lv = ocoord.lift_and_diagonalize(means, covs, P.t())
  feat = ocoord.integrated_pos_enc(lm, lv, min_deg, max_deg).reshape(B * n, F)
  (feat * gA.double()[:, :F]).sum().backward()
  got = torch.zeros((B, n + 1), dtype=torch.float64)
  got[:, :n] += g0.view(B, n).double().cpu()
  got[:, 1:] += g1.view(B, n).double().cpu()
  _rays_close(got, t2.grad, tol, 1e-6, f'g_tdist, one source [{name}]')


@pytest.mark.parametrize('name,ray_shape,contract,basis,min_deg,max_deg,far,ndc,no_int', IPE_CASES)
def test_cast_rays_ipe_tangent_bwd_is_the_oracles_autograd(name, ray_shape, contract, basis, min_deg, max_deg, far, ndc, no_int):
  """mnr_cast_rays_ipe_tangent_bwd: the tangent rows T_c = d features / d mean_c (the input of the density-gradient normals' forward-mode
  network, models.py:478-492; the covariance an input held fixed, the contraction inside, :445-446) differentiated with respect to the
  interval ends, against reverse-over-forward autodiff of the oracle's featurisation in float64."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from multinerf_amd import geopoly
  from oracle import coord as ocoord
  from oracle import render as orender
  B, n = 12, 8
  o, d, radii, tdist = _ray_case(B, n, seed=41 + len(name), far_samples=far, ndc=ndc)
  if name == '360-near':
    tdist = (tdist * 0.5 - 0.6).float().double()
  P = torch.as_tensor(geopoly.generate_basis(*basis), dtype=torch.float64)
  K, L = P.shape[0], max_deg - min_deg
  F = 2 * K * L
  ld = (F + 127) // 128 * 128
  M = B * n
  gen = torch.Generator().manual_seed(7)
  # upstream gradients shrinking with the degree, as a trained network's first layer does to them: T's entries grow like 4^l
  scale = torch.cat([2.0 ** -(2.0 * torch.arange(L, dtype=torch.float64).repeat_interleave(K))] * 2)
  mk = lambda: torch.cat([(torch.randn((3 * M, F), generator=gen, dtype=torch.float64) * scale), torch.zeros((3 * M, ld - F), dtype=torch.float64)], 1).to(torch.bfloat16)
  gA, gB = mk(), mk()

  def loss_of(t, G):
    means, covs = orender.cast_rays(t, o, d, radii, ray_shape, diag=False)
    if no_int:
      covs = torch.zeros_like(covs)

    def feat_of(mu):
      m2, c2 = (ocoord.track_linearize(ocoord.contract, mu, covs) if contract else (mu, covs))
      lm, lv = ocoord.lift_and_diagonalize(m2, c2, P.t())
      return ocoord.integrated_pos_enc(lm, lv, min_deg, max_deg).reshape(M, F)

    total = 0.0
    for c in range(3):
      e = torch.zeros_like(means)
      e[..., c] = 1.0
      _, Tc = torch.autograd.functional.jvp(feat_of, means, e, create_graph=True)
      total = total + (Tc * G[c * M:(c + 1) * M, :F]).sum()
    return total

  f32c = lambda x: dev(x.float().contiguous())
  for G, gb in (((gA.double() + gB.double()), gB), (gA.double(), None)):
    t = tdist.clone().requires_grad_(True)
    loss_of(t, G).backward()
    want = t.grad
    g_t0 = dev(torch.zeros(M, dtype=torch.float32))
    g_t1 = dev(torch.zeros(M, dtype=torch.float32))
    ops.cast_rays_ipe_tangent_bwd(f32c(tdist), f32c(o), f32c(d), f32c(radii.reshape(-1)), f32c(P), dev(gA), None if gb is None else dev(gb),
                                  g_t0, g_t1, ray_shape=ray_shape, warp_contract=contract, min_deg=min_deg, max_deg=max_deg,
                                  disable_integration=no_int)
    torch.cuda.synchronize()
    got = torch.zeros((B, n + 1), dtype=torch.float64)
    got[:, :n] += g_t0.view(B, n).double().cpu()
    got[:, 1:] += g_t1.view(B, n).double().cpu()
    tol = 1e-3 if max_deg <= 12 else 5e-3                                     # (measured on the MI355X: <= 5e-5)
    _rays_close(got, want, tol, 1e-6, f'g_tdist through the tangent rows [{name}]')


# ----------------------------------------------------------------------------- compositing / distortion -> sample distances


@pytest.mark.parametrize('quad', [1, 0])
@pytest.mark.parametrize('raydist,opaque,n', [('reciprocal', True, 32), (None, False, 128), ('reciprocal', True, 21), ('piecewise', False, 16)])
def test_sdist_bwd_with_the_compositing_and_distortion_terms(raydist, opaque, n, quad):
  """mnr_level_bwd's g_x output + mnr_sdist_bwd against autograd of the oracle's compositing (render.py:130-213) and distortion
  loss (stepfun.py:266-276) with respect to sdist, through s_to_t."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from oracle import coord as ocoord
  from oracle import render as orender
  B = 40
  g = torch.Generator().manual_seed(n)
  r64 = lambda *s: torch.rand(s, generator=g, dtype=torch.float64).float().double()
  sdist = torch.sort(r64(B, n + 1), dim=-1).values
  near = torch.full((B, 1), 0.2 if raydist else 2.0, dtype=torch.float64)
  far = torch.full((B, 1), 1e3 if raydist == 'reciprocal' else 6.0, dtype=torch.float64)
  dirs = (torch.randn((B, 3), generator=g, dtype=torch.float64) * 0.7).float().double()
  raw_den = (torch.randn((B, n), generator=g, dtype=torch.float64) * 2).float().double()
  raw_rgb = torch.randn((B, n, 3), generator=g, dtype=torch.float64).float().double()
  g_rgb_out = torch.randn((B, 3), generator=g, dtype=torch.float64).float().double()
  g_w_up = (0.1 * torch.randn((B, n), generator=g, dtype=torch.float64)).float().double()
  g_in = torch.randn((B, n + 1), generator=g, dtype=torch.float64).float().double()
  gt0 = torch.randn((B, n), generator=g, dtype=torch.float64).float().double()
  gt1 = torch.randn((B, n), generator=g, dtype=torch.float64).float().double()
  mult = 0.01
  B_valid = B - 3
  # ---- oracle
  s = sdist.clone().requires_grad_(True)
  _, s_to_t = ocoord.construct_ray_warps(raydist, near, far)
  tdist = s_to_t(s)
  density = torch.nn.functional.softplus(raw_den - 1.0)
  weights = orender.compute_alpha_weights(density, tdist, dirs, opaque_background=opaque)[0]
  rgb = torch.sigmoid(raw_rgb) * (1 + 2 * 0.001) - 0.001
  acc = weights.sum(-1)
  rgb_out = (weights[..., None] * rgb).sum(-2) + torch.clamp(1 - acc, min=0)[..., None] * 1.0
  loss = (rgb_out * g_rgb_out).sum() + (weights * g_w_up).sum()
  loss = loss + mult * ostep.lossfun_distortion(s[:B_valid], weights[:B_valid]).mean()
  loss = loss + (tdist[:, :-1] * gt0).sum() + (tdist[:, 1:] * gt1).sum() + (s * g_in).sum()
  loss.backward()
  want = s.grad
  # ---- kernels
  f32c = lambda x: dev(x.float().contiguous())
  ccfg = ops.composite_cfg(n, opaque_background=opaque, density_act='softplus', density_bias=-1.0, density_noise_std=0.0,
                           has_rgb=True, rgb_act='sigmoid', rgb_premultiplier=1.0, rgb_bias=0.0, rgb_padding=0.001, bg_mode=0,
                           bg_value=1.0)
  td32, w32 = f32c(tdist.detach()), f32c(weights.detach())
  stat = dev(torch.zeros(1))
  g_x = dev(torch.empty((B, n)))
  from multinerf_amd import _lib
  _lib.debug().mnr_level_bwd_set_quad(quad)                # both forms of the level kernel write g_x: four lanes per ray / lane per ray
  try:
    _run_composite_bwd = lambda: ops.composite_bwd(ccfg, f32c(raw_den), td32, f32c(dirs), w32, raw_rgb=f32c(raw_rgb), g_rgb_out=f32c(g_rgb_out),
                    g_weights=f32c(g_w_up), g_x_out=g_x,
                    losses=dict(B_valid=B_valid, data=None, weights=dict(mode='distortion', mult=mult, sdist=f32c(sdist), stat=stat)))
    _run_composite_bwd()
  finally:
    _lib.debug().mnr_level_bwd_set_quad(1)
  got = ops.sdist_bwd(f32c(sdist), f32c(near.reshape(-1)), f32c(far.reshape(-1)), raydist, B_valid=B_valid, g_x=g_x,
                      raw_density=f32c(raw_den), density_bias=-1.0, density_act='softplus', dirs=f32c(dirs),
                      g_t0=f32c(gt0.reshape(-1)), g_t1=f32c(gt1.reshape(-1)), distortion_mult=mult, weights=w32,
                      g_sdist_in=f32c(g_in))
  torch.cuda.synchronize()
  _rays_close(got, want, 2e-4, 1e-6, f'g_sdist [{raydist}, n={n}]')


# ----------------------------------------------------------------------------- composed: Model(stop_level_grad=False) train step


COMPOSED = [
    # (name, preset, bindings, rays, strict)
    # blender_256.gin AS IS apart from the switch (256-wide MLPs, 128 + 32 samples, 16 degrees): dilation, annealing, two MLPs
    ('blender_256', 'blender_256', [], 16, True),
    # configs/360.gin at its own widths with the encoding cut to two degrees (three levels, contraction incl. its second
    # derivative, distortion loss): the well-conditioned form, held tightly (see tests/test_sim_model.py)
    ('360-2deg', '360', ['NerfMLP.max_deg_point = 2', 'PropMLP.max_deg_point = 2'], 16, True),
    # configs/360.gin AS IS apart from the switch: at twelve degrees the oracle's own bf16-vs-fp32 distance of this gradient is
    # of order one; held to that distance
    ('360', '360', [], 16, False),
    # llff_raw.gin: one shared 256-wide MLP on the fused chain (skip concat), cylinders, no dilation
    ('llff_raw', 'llff_raw', ['Model.num_prop_samples = 64', 'Model.num_nerf_samples = 64'], 8, True),
    # blender_refnerf.gin AS IS apart from the switch: next to density-gradient normals (the tangent rows' VJP, round 6)
    ('blender_refnerf', 'blender_refnerf', ['Model.resample_padding = 0.01'], 8, True),
]


@pytest.mark.parametrize('name,preset,bindings,B,strict', COMPOSED)
def test_train_step_through_the_sampling_matches_the_oracle(name, preset, bindings, B, strict):
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  import os
  from multinerf_amd import configs, models, train_utils
  from oracle import models as omodels
  from oracle import train_utils as otrain
  from tests import helpers
  if os.environ.get('MNR_TESTS_ON_SIMULATOR') == '1':
    bindings = bindings + ['NerfMLP.net_width = 128', 'PropMLP.net_width = 128', 'Model.num_prop_samples = 32', 'Model.num_nerf_samples = 16']
    B = 4
  cfg = configs.load_preset(preset, bindings + ['Model.stop_level_grad = False'])
  model = models.Model(config=cfg).build('cuda')
  om, on, op = helpers.oracle_hparams(model)
  params = omodels.init_params(om, on, op, seed=5)
  g = torch.Generator().manual_seed(6)
  for mname, mod in params.items():
    if mname in ('exposure_scaling_offsets', 'Embed_0'):
      continue
    for dd in mod.values():
      dd['bias'] = 0.05 * torch.randn(dd['bias'].shape, generator=g)
  flat = model.flat_from_tree(params)
  batch = helpers.synthetic_rays(B, near=cfg.near, far=cfg.far)
  if preset == 'llff_raw':
    batch.rays.exposure_idx = torch.randint(0, 5, (B, 1), generator=g).to(torch.int32)
    batch.rays.exposure_values = 0.5 + torch.rand((B, 1), generator=g)
    batch.rays.lossmult = (torch.rand((B, 3), generator=g) > 0.4).float()
    batch.rgb = batch.rgb * 0.3
  if cfg.compute_normal_metrics:
    batch.alphas = torch.rand((B,), generator=g)
    batch.normals = torch.randn((B, 3), generator=g)
  noise = helpers.make_noise(model, B)
  tf = 0.4
  st = otrain.init_opt_state(params)
  _, _, stats_o, grads_o = otrain.train_step(params, st, om, on, op, cfg, batch, tf, noise=noise, dense_dtype=torch.bfloat16)
  _, _, _, grads_32 = otrain.train_step(params, st, om, on, op, cfg, batch, tf, noise=noise)
  g_ref = model.flat_from_tree(grads_o, device='cpu').double()
  g_32 = model.flat_from_tree(grads_32, device='cpu').double()
  state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone().cuda(), 'params': None})
  step = train_utils.create_train_step(model, cfg)
  noise_d = {k: {lv: t.cuda() for lv, t in d.items()} for k, d in noise.items()}
  _, stats, _ = step(0, state, batch.map(lambda t: t.cuda()), None, tf, 0.0, noise=noise_d, return_grads=True)
  torch.cuda.synchronize()
  s = stats.materialize()
  assert abs(s['loss'] - float(stats_o['loss'])) <= 0.02 * abs(float(stats_o['loss'])) + 1e-5
  gg = stats['_grads'].double().cpu()
  assert torch.isfinite(gg).all()
  for mod, b, e in model.modules:
    a, r, r32 = gg[b:e], g_ref[b:e], g_32[b:e]
    if r.norm() < 1e-12:
      continue
    cos = (a @ r / (a.norm() * r.norm() + 1e-30)).item()
    rel = ((a - r).norm() / (r.norm() + 1e-30)).item()
    cost = ((r - r32).norm() / (r32.norm() + 1e-30)).item()
    print(f'SAMPLING_GRAD {name} {mod}: grad cos {cos:.6f} rel err {rel:.3e} (the oracle\'s own bf16 cost {cost:.3e})')
    if strict:
      assert cos > 0.985 and rel < max(0.05, 1.5 * cost), (mod, cos, rel, cost)        # (measured on the MI355X: >= 0.9925 / <= 0.66 cost)
    else:
      assert rel < max(0.1, cost), (mod, cos, rel, cost)
