"""Known-answer and property tests that the reference's own unit tests hold for the hot-path leaves, restated against
the oracle (CPU).  Each test cites the reference test it restates (tests/<file>:<lines> of the reference repository);
the random inputs are our own (seeded), the literal tables are the reference's golden vectors.
"""

import math

import numpy as np
import pytest
import torch

from multinerf_amd import geopoly
from oracle import camera_utils as ocam
from oracle import coord, image, render, stepfun
from oracle import math as omath

F64 = torch.float64


def _gen(seed):
  return torch.Generator().manual_seed(seed)


# ----------------------------------------------------------------------------- coord


def test_contract_of_reciprocal_warp_is_uniform():
  """coord_test.py:61-69 (Figure 2 of mip-NeRF 360): contracting the reciprocal-spaced distances gives equal steps."""
  n = 10
  eps = float(np.finfo(np.float32).eps)
  _, s_to_t = coord.construct_ray_warps('reciprocal', torch.tensor(1.0, dtype=F64), torch.tensor(float('inf'), dtype=F64))
  s = torch.linspace(0, 1 - eps, n + 1, dtype=F64)
  tc = coord.contract(s_to_t(s)[:, None])[:, 0]
  np.testing.assert_allclose((tc[1:] - tc[:-1]).numpy(), np.full(n, 1 / n), atol=1e-5, rtol=1e-5)


def test_contract_bounded_noop_inside_and_invertible():
  """coord_test.py:71-120: |contract(x)| < 2; identity for |x| <= 1; inv_contract inverts it."""
  g = _gen(1)
  x = torch.where(torch.rand((4000, 3), generator=g) < 0.5, 1.0, -1.0).to(F64) * torch.exp(
      torch.empty((4000, 3), dtype=F64).uniform_(-3, 8, generator=g))
  z = coord.contract(x)
  assert (z.norm(dim=-1) < 2).all()
  small = x / x.norm(dim=-1, keepdim=True) * torch.rand((4000, 1), generator=g, dtype=F64)
  np.testing.assert_allclose(coord.contract(small).numpy(), small.numpy(), atol=1e-12)
  keep = x.norm(dim=-1) < 1e3                       # far points lose digits in 2 - 1/|x|
  np.testing.assert_allclose(coord.inv_contract(coord.contract(x[keep])).numpy(), x[keep].numpy(), rtol=1e-6)


def test_reciprocal_ray_warp_closed_form():
  """coord_test.py:199-221."""
  g = _gen(2)
  n = 100
  t_near = torch.exp(torch.randn(n, generator=g, dtype=F64))
  t_far = t_near + torch.exp(torch.randn(n, generator=g, dtype=F64))
  u = torch.rand(n, generator=g, dtype=F64)
  t = t_near * (1 - u) + t_far * u
  s = torch.rand(n, generator=g, dtype=F64)
  t_to_s, s_to_t = coord.construct_ray_warps('reciprocal', t_near, t_far)
  np.testing.assert_allclose(s_to_t(s).numpy(), (1 / (s / t_far + (1 - s) / t_near)).numpy(), rtol=1e-10)
  np.testing.assert_allclose(t_to_s(t).numpy(), ((t_far * (t - t_near)) / (t * (t_far - t_near))).numpy(), rtol=1e-9, atol=1e-12)
  # extents, coord_test.py:185-197
  for fn in (None, 'reciprocal', 'log', 'sqrt'):
    t2s, s2t = coord.construct_ray_warps(fn, t_near, t_far)
    np.testing.assert_allclose(s2t(torch.zeros_like(t_near)).numpy(), t_near.numpy(), rtol=1e-9)
    np.testing.assert_allclose(s2t(torch.ones_like(t_near)).numpy(), t_far.numpy(), rtol=1e-9)


def test_ipe_with_zero_variance_is_pos_enc():
  """coord_test.py:129-140."""
  x = torch.linspace(-math.pi, math.pi, 10000, dtype=F64)[:, None]
  z_ipe = coord.integrated_pos_enc(x, torch.zeros_like(x), 0, 10)
  z_pe = coord.pos_enc(x, 0, 10, append_identity=False)
  np.testing.assert_allclose(z_pe.numpy(), z_ipe.numpy(), atol=1e-4)


def test_track_linearize_matches_autograd_jacobian():
  """coord_test.py:142-183: the linearised covariance is J cov J^T."""
  g = _gen(3)
  mean = torch.randn((20, 3), generator=g, dtype=F64) * 3
  a = torch.randn((20, 3, 3), generator=g, dtype=F64)
  cov = a @ a.transpose(-1, -2)
  fn_mean, fn_cov = coord.track_linearize(coord.contract, mean, cov)
  for i in range(20):
    J = torch.autograd.functional.jacobian(lambda v: coord.contract(v[None])[0], mean[i])
    np.testing.assert_allclose(fn_cov[i].numpy(), (J @ cov[i] @ J.T).numpy(), rtol=1e-6, atol=1e-10)
  np.testing.assert_allclose(fn_mean.numpy(), coord.contract(mean).numpy(), rtol=1e-12)


# ----------------------------------------------------------------------------- render


def test_alpha_weights_of_a_delta_density_are_one_hot():
  """render_test.py:443-463."""
  g = _gen(4)
  n, d = 100, 128
  r = torch.randn((n, d), generator=g, dtype=F64)
  mask = (r == r.max(dim=-1, keepdim=True).values)
  density = 1e10 * mask.to(F64)
  tvals = torch.sort(2 * torch.rand((n, d + 1), generator=g, dtype=F64) - 1, -1).values
  dirs = torch.randn((n, 3), generator=g, dtype=F64)
  weights, alpha, _ = render.compute_alpha_weights(density, tvals, dirs)
  np.testing.assert_allclose(weights.numpy(), mask.to(F64).numpy(), atol=1e-5)
  np.testing.assert_allclose(alpha.numpy(), mask.to(F64).numpy(), atol=1e-5)


def test_stable_conical_frustum_matches_direct_moments():
  """render_test.py:320-331: the stable parameterisation equals eqs. 37-39 evaluated directly."""
  g = _gen(5)
  n = 200
  d = torch.randn((n, 3), generator=g, dtype=F64)
  t0 = torch.exp(torch.randn(n, generator=g, dtype=F64))
  t1 = t0 + torch.exp(torch.randn(n, generator=g, dtype=F64))
  r = torch.exp(torch.randn(n, generator=g, dtype=F64))
  for diag in (False, True):
    m_s, c_s = render.conical_frustum_to_gaussian(d, t0, t1, r, diag, stable=True)
    m_u, c_u = render.conical_frustum_to_gaussian(d, t0, t1, r, diag, stable=False)
    np.testing.assert_allclose(m_s.numpy(), m_u.numpy(), rtol=1e-8)
    np.testing.assert_allclose(c_s.numpy(), c_u.numpy(), rtol=1e-6, atol=1e-10)


def test_conical_frustum_gaussian_matches_sampled_moments():
  """render_test.py:66-93,180-230: mean / covariance of points sampled uniformly inside the frustum."""
  g = _gen(6)
  d = torch.tensor([0.3, -0.5, 0.8], dtype=F64)
  t0, t1, radius = 1.5, 2.5, 0.2
  n = 400000
  # uniform in volume: t with density ~ t^2, uniform in the disc of radius t * radius
  u = torch.rand(n, generator=g, dtype=F64)
  t = (t0**3 + u * (t1**3 - t0**3))**(1 / 3)
  rr = torch.sqrt(torch.rand(n, generator=g, dtype=F64)) * radius * t
  th = torch.rand(n, generator=g, dtype=F64) * 2 * math.pi
  dn = d / d.norm()
  e1 = torch.linalg.cross(dn, torch.tensor([1.0, 0, 0], dtype=F64))
  e1 = e1 / e1.norm()
  e2 = torch.linalg.cross(dn, e1)
  pts = t[:, None] * d + (rr * torch.cos(th))[:, None] * e1 * d.norm() + (rr * torch.sin(th))[:, None] * e2 * d.norm()
  mean, cov = render.conical_frustum_to_gaussian(d[None], torch.tensor([t0], dtype=F64), torch.tensor([t1], dtype=F64),
                                                 torch.tensor([radius], dtype=F64), diag=False)
  np.testing.assert_allclose(mean.reshape(3).numpy(), pts.mean(0).numpy(), atol=3e-3)
  np.testing.assert_allclose(cov.reshape(3, 3).numpy(), torch.cov(pts.T).numpy(), atol=3e-3)


# ----------------------------------------------------------------------------- image (the reference's golden tables)

_SRGB_TO_LINEAR = [
    0.00000000, 0.00122856, 0.00245712, 0.00372513, 0.00526076, 0.00711347, 0.00929964, 0.01183453, 0.01473243,
    0.01800687, 0.02167065, 0.02573599, 0.03021459, 0.03511761, 0.04045585, 0.04623971, 0.05247922, 0.05918410,
    0.06636375, 0.07402734, 0.08218378, 0.09084171, 0.10000957, 0.10969563, 0.11990791, 0.13065430, 0.14194246,
    0.15377994, 0.16617411, 0.17913227, 0.19266140, 0.20676863, 0.22146071, 0.23674440, 0.25262633, 0.26911288,
    0.28621066, 0.30392596, 0.32226467, 0.34123330, 0.36083785, 0.38108405, 0.40197787, 0.42352500, 0.44573134,
    0.46860245, 0.49214387, 0.51636110, 0.54125960, 0.56684470, 0.59312177, 0.62009590, 0.64777250, 0.67615650,
    0.70525320, 0.73506740, 0.76560410, 0.79686830, 0.82886493, 0.86159873, 0.89507430, 0.92929670, 0.96427040,
    1.00000000]


def test_srgb_to_linear_golden_and_round_trip():
  """image_test.py:91-110 (golden table) and :62-75 (round trip)."""
  srgb = torch.linspace(0, 1, 64, dtype=F64)
  np.testing.assert_allclose(image.srgb_to_linear(srgb).numpy(), np.array(_SRGB_TO_LINEAR), atol=1e-5, rtol=1e-5)
  np.testing.assert_allclose(image.linear_to_srgb(image.srgb_to_linear(srgb)).numpy(), srgb.numpy(), atol=1e-6)


def test_mse_to_psnr_golden_and_round_trip():
  """image_test.py:112-127: the golden table is 43.429447 ... 0 in equal steps; :52-60 round trip."""
  mse = torch.exp(torch.linspace(-10, 0, 64, dtype=F64))
  want = np.linspace(43.429447, 0.0, 64)
  np.testing.assert_allclose(image.mse_to_psnr(mse).numpy(), want, atol=1e-5, rtol=1e-5)
  np.testing.assert_allclose(image.psnr_to_mse(image.mse_to_psnr(mse)).numpy(), mse.numpy(), rtol=1e-10)


# ----------------------------------------------------------------------------- math


@pytest.mark.parametrize('fn', ['sorted_interp', 'interp'])
def test_interp_matches_numpy(fn):
  """math_test.py:156-178."""
  g = _gen(7)
  n, d0, d1 = 100, 10, 20
  x = torch.randn((n, d0), generator=g, dtype=F64)
  xp = torch.randn((n, d1), generator=g, dtype=F64)
  fp = torch.randn((n, d1), generator=g, dtype=F64)
  if fn == 'sorted_interp':
    xp, fp = torch.sort(xp, -1).values, torch.sort(fp, -1).values
    z = omath.sorted_interp(x, xp, fp)
  else:
    xp, order = torch.sort(xp, -1)          # np.interp needs increasing xp; math.interp is jnp.interp vmapped
    fp = torch.gather(fp, -1, order)
    z = omath.interp(x, xp, fp)
  want = np.stack([np.interp(x[i].numpy(), xp[i].numpy(), fp[i].numpy()) for i in range(n)])
  np.testing.assert_allclose(z.numpy(), want, atol=1e-9)


def test_safe_sin_and_learning_rate_endpoints():
  """math_test.py:40-60 (safe_sin == sin below 100 pi) and :100-154 (schedule endpoints, delay)."""
  x = torch.linspace(-300, 300, 20001, dtype=F64)
  np.testing.assert_allclose(omath.safe_sin(x).numpy(), np.sin(x.numpy()), atol=1e-9)
  lr = lambda s, **k: omath.learning_rate_decay(s, 1e-2, 1e-4, 1000, **k)
  assert abs(lr(0) - 1e-2) < 1e-12 and abs(lr(1000) - 1e-4) < 1e-12 and abs(lr(1100) - 1e-4) < 1e-12
  assert abs(lr(500) - 1e-3) < 1e-9                                   # log-linear midpoint
  assert abs(lr(0, lr_delay_steps=100, lr_delay_mult=0.1) - 1e-3) < 1e-12
  assert abs(lr(100, lr_delay_steps=100, lr_delay_mult=0.1) - lr(100)) < 1e-12


# ----------------------------------------------------------------------------- stepfun


def _rand_step(g, n, d, lo=-3.0, hi=3.0):
  t = torch.sort(torch.empty((n, d + 1), dtype=F64).uniform_(lo, hi, generator=g), -1).values
  w = torch.softmax(2 * torch.randn((n, d), generator=g, dtype=F64), -1)
  return t, w


def test_searchsorted_brackets_and_out_of_bounds():
  """stepfun_test.py:53-114."""
  g = _gen(8)
  a = torch.sort(torch.randn((10, 30), generator=g, dtype=F64), -1).values
  v = torch.empty((10, 17), dtype=F64).uniform_(float(a.min()) + 1e-7, float(a.max()) - 1e-7, generator=g)
  v = torch.max(torch.min(v, a[:, -1:] - 1e-7), a[:, :1] + 1e-7)
  lo, hi = stepfun.searchsorted(a, v)
  assert (torch.gather(a, -1, lo) <= v).all() and (v < torch.gather(a, -1, hi)).all() and (hi == lo + 1).all()
  lo, hi = stepfun.searchsorted(a, a[:, :1] - 1.0)
  assert (lo == 0).all() and (hi == 0).all()
  lo, hi = stepfun.searchsorted(a, a[:, -1:] + 1.0)
  assert (lo == 29).all() and (hi == 29).all()


def test_inner_outer_against_brute_force():
  """stepfun_test.py:27-50 (the reference's pure-Python inner / outer) and :340-400."""
  g = _gen(9)
  for _ in range(5):
    t0, _ = _rand_step(g, 1, 9)
    t1, w1 = _rand_step(g, 1, 23)
    t0, t1, w1 = t0[0], t1[0], w1[0]
    inner, outer = stepfun.inner_outer(t0, t1, w1)
    want_in, want_out = [], []
    for i in range(len(t0) - 1):
      want_in.append(sum(float(w1[j]) for j in range(len(t1) - 1) if t1[j] >= t0[i] and t1[j + 1] < t0[i + 1]))
      want_out.append(sum(float(w1[j]) for j in range(len(t1) - 1) if t1[j + 1] >= t0[i] and t1[j] <= t0[i + 1]))
    np.testing.assert_allclose(inner.numpy(), want_in, atol=1e-12)
    np.testing.assert_allclose(outer.numpy(), want_out, atol=1e-12)


def test_lossfun_outer_is_zero_on_itself_and_on_coarsenings():
  """stepfun_test.py:300-338: a histogram never exceeds its own (or a coarser) envelope."""
  g = _gen(10)
  t, w = _rand_step(g, 6, 16)
  assert float(stepfun.lossfun_outer(t, w, t, w).abs().max()) < 1e-12
  t_c, w_c = t[:, ::2], w.reshape(6, 8, 2).sum(-1)
  assert float(stepfun.lossfun_outer(t, w, t_c, w_c).abs().max()) < 1e-12
  assert float(stepfun.lossfun_outer(t, w * 2, t_c, w_c).min()) >= 0.0       # non-negative otherwise


def test_distortion_loss_against_brute_force():
  """stepfun_test.py:227-298: lossfun_distortion == sum_ij w_i w_j E|x_i - x_j| for x uniform in the intervals."""
  g = _gen(11)
  t, w = _rand_step(g, 3, 7)
  got = stepfun.lossfun_distortion(t, w)
  m = 1201
  for i in range(3):
    tot = 0.0
    for a in range(7):
      xa = torch.linspace(float(t[i, a]), float(t[i, a + 1]), m, dtype=F64)
      for b in range(7):
        xb = torch.linspace(float(t[i, b]), float(t[i, b + 1]), m, dtype=F64)
        tot += float(w[i, a] * w[i, b]) * float((xa[:, None] - xb[None, :]).abs().mean())
    assert abs(float(got[i]) - tot) < 2e-3 * max(1.0, tot)


def test_weighted_percentile_matches_cdf_inversion():
  """stepfun_test.py:402-440."""
  g = _gen(12)
  t, w = _rand_step(g, 5, 40, 0.0, 1.0)
  ps = [5, 50, 95]
  got = stepfun.weighted_percentile(t, w, ps)
  cw = torch.cat([torch.zeros((5, 1), dtype=F64), torch.cumsum(w, -1)], -1)
  for i in range(5):
    want = np.interp(np.array(ps) / 100, cw[i].numpy(), t[i].numpy())
    np.testing.assert_allclose(got[i].numpy(), want, atol=1e-9)


def test_sample_single_bin_and_intervals_are_exact():
  """stepfun_test.py:520-586: one bin -> deterministic samples are its uniform grid; intervals of a flat histogram."""
  t = torch.tensor([[3.0, 4.0]], dtype=F64)
  logits = torch.zeros((1, 1), dtype=F64)
  s = stepfun.sample(None, t, logits, 10, deterministic_center=True)
  np.testing.assert_allclose(s[0].numpy(), np.linspace(3.05, 3.95, 10), atol=1e-6)
  ti = stepfun.sample_intervals(None, t, logits, 10, single_jitter=True, domain=(3.0, 4.0))
  np.testing.assert_allclose(ti[0].numpy(), np.linspace(3, 4, 11), atol=1e-5)


def test_max_dilate_bounds_the_original():
  """stepfun_test.py:150-200: the dilated step function is an upper envelope of the original."""
  g = _gen(13)
  t, w = _rand_step(g, 4, 20, 0.0, 1.0)
  p = stepfun.weight_to_pdf(t, w)
  td, pd_ = stepfun.max_dilate(t, p, 0.02, domain=(0.0, 1.0))
  tq = torch.rand((4, 500), generator=g, dtype=F64)
  assert (stepfun.query(tq, td, pd_) >= stepfun.query(tq, t, p) - 1e-12).all()


# ----------------------------------------------------------------------------- geopoly, camera_utils


def _same_point_set(a, b, tol=1e-5):
  """Every row of a is within tol of some row of b and vice versa (order-free, as geopoly_test.py:27-34)."""
  d = np.sqrt(((a[:, None, :] - b[None, :, :])**2).sum(-1))
  return a.shape == b.shape and (d.min(1) < tol).all() and (d.min(0) < tol).all()


# geopoly_test.py:76-136: the two golden bases, written with the constants they are made of.
_A, _B, _C = 0.85065081, 0.52573111, 0.80901699          # icosahedron: (phi, 1) / sqrt(phi + 2) and phi / 2
_D = 0.30901699                                           # 1 / (2 phi)
_ICO2_GOLDEN = [(_A, 0, _B), (_C, .5, _D), (_B, _A, 0), (1, 0, 0), (_C, .5, -_D), (_A, 0, -_B), (_D, _C, -.5),
                (0, _B, -_A), (.5, _D, -_C), (0, 1, 0), (-_B, _A, 0), (-_D, _C, -.5), (0, _B, _A), (-_D, _C, .5),
                (_D, _C, .5), (.5, _D, _C), (.5, -_D, _C), (0, 0, 1), (-.5, _D, _C), (-_C, .5, _D), (-_C, .5, -_D)]
_P, _Q, _R, _S, _T = 0.31622777, 0.94868330, 0.70710678, 0.40824829, 0.81649658
_OCT4_GOLDEN = [(0, 0, -1), (0, -_P, -_Q), (0, -_R, -_R), (0, -_Q, -_P), (0, -1, 0), (-_P, 0, -_Q), (-_S, -_S, -_T),
                (-_S, -_T, -_S), (-_P, -_Q, 0), (-_R, 0, -_R), (-_T, -_S, -_S), (-_R, -_R, 0), (-_Q, 0, -_P),
                (-_Q, -_P, 0), (-1, 0, 0), (0, -_P, _Q), (0, -_R, _R), (0, -_Q, _P), (_S, -_S, _T), (_S, -_T, _S),
                (_P, -_Q, 0), (_T, -_S, _S), (_R, -_R, 0), (_Q, -_P, 0), (_P, 0, -_Q), (_S, _S, -_T), (_S, _T, -_S),
                (_R, 0, -_R), (_T, _S, -_S), (_Q, 0, -_P), (_S, -_S, -_T), (_S, -_T, -_S), (_T, -_S, -_S)]


def test_generate_basis_golden():
  """geopoly_test.py:76-136: the reference's golden icosahedron-2 and octahedron-4 bases (as point sets)."""
  assert _same_point_set(geopoly.generate_basis('icosahedron', 2), np.array(_ICO2_GOLDEN, dtype=np.float64))
  assert _same_point_set(geopoly.generate_basis('octahedron', 4), np.array(_OCT4_GOLDEN, dtype=np.float64))


def test_generate_basis_counts_and_symmetry():
  """geopoly.py:78-124: unit vectors, no duplicate and no antipodal pair after symmetry removal; 10 v^2 + 2 vertices
  on the full tesselated icosahedron, 4 v^2 + 2 on the octahedron (Euler), half of them kept."""
  for shape, per in (('icosahedron', 10), ('octahedron', 4)):
    for v in (1, 2, 3):
      full = geopoly.generate_basis(shape, v, remove_symmetries=False)
      half = geopoly.generate_basis(shape, v)
      assert full.shape == (per * v * v + 2, 3) and half.shape == (full.shape[0] // 2, 3)
      np.testing.assert_allclose(np.linalg.norm(half, axis=-1), 1.0, atol=1e-12)
      g = half @ half.T
      np.fill_diagonal(g, 0.0)
      assert np.abs(g).max() < 1 - 1e-6


def test_convert_to_ndc_maps_points_on_rays_consistently():
  """camera_utils_test.py:test_convert_to_ndc: projecting points of a ray == walking along the NDC ray."""
  g = _gen(14)
  n = 50
  focal, w, h = 2.0, 3.0, 2.0
  pixtocam = torch.linalg.inv(torch.tensor([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1.0]], dtype=F64))
  near = 1.0
  origins = torch.randn((n, 3), generator=g, dtype=F64) * 0.2
  directions = torch.randn((n, 3), generator=g, dtype=F64) * 0.3 + torch.tensor([0.0, 0.0, -1.0], dtype=F64)
  o_ndc, d_ndc = ocam.convert_to_ndc(origins, directions, pixtocam, near)
  # a point at depth z (z < -near) projects to (xmult x/z... ) and lies on o_ndc + s d_ndc with s = 1 + near / z'
  for z in (-1.5, -4.0, -50.0):
    t = (z - origins[:, 2]) / directions[:, 2]
    p = origins + t[:, None] * directions
    xm, ym = 1.0 / pixtocam[0, 2], 1.0 / pixtocam[1, 2]
    proj = torch.stack([xm * p[:, 0] / p[:, 2], ym * p[:, 1] / p[:, 2], 1 + 2 * near / p[:, 2]], -1)
    s = (proj[:, 2] - o_ndc[:, 2]) / d_ndc[:, 2]
    np.testing.assert_allclose((o_ndc + s[:, None] * d_ndc).numpy(), proj.numpy(), atol=1e-9)


# ----------------------------------------------------------------------------- pos_enc against a stable recursion


def _pos_enc_by_angle_doubling(x, n):
  """sin / cos of 2^k x for k < n by repeated squaring of the rotation by x (coord_test.py:34-43 uses the same idea as
  its high-degree reference): no large arguments ever reach sin / cos.  Output layout of coord.pos_enc without the
  identity: [sin(2^0 x) .. sin(2^(n-1) x), cos(2^0 x) .. cos(2^(n-1) x)] per input."""
  s, c = np.sin(x), np.cos(x)
  sins, coss = [], []
  for _ in range(n):
    sins.append(s)
    coss.append(c)
    s, c = 2 * s * c, c * c - s * s
  return np.stack(sins + coss, -1)


def test_angle_doubling_reference_on_multiples_of_half_pi():
  """coord_test.py:48-59: the recursion itself on x = -pi .. pi in steps of pi/2."""
  z = _pos_enc_by_angle_doubling(np.linspace(-np.pi, np.pi, 5), 10)
  want_sin = np.zeros((5, 10))
  want_cos = np.ones((5, 10))
  want_sin[:, 0] = [0, -1, 0, 1, 0]
  want_cos[:, 0] = [-1, 0, 1, 0, -1]
  want_cos[:, 1] = [1, -1, 1, -1, 1]
  np.testing.assert_allclose(z, np.concatenate([want_sin, want_cos], -1), atol=1e-10)


@pytest.mark.parametrize('n,tol', [(5, 1e-5), (10, 1e-4), (15, 0.005), (20, 0.2)])
def test_pos_enc_against_the_stable_recursion(n, tol):
  """coord_test.py:112-127: fp32 pos_enc of degree n stays within the reference's tolerances of the stable recursion
  (the direct form loses ~2^n ulp of phase; the tolerances are the reference test's own)."""
  x = np.linspace(-np.pi, np.pi, 10001)
  z = coord.pos_enc(torch.as_tensor(x, dtype=torch.float32)[:, None], 0, n, append_identity=False).double().numpy()
  assert np.abs(z - _pos_enc_by_angle_doubling(x, n)).max() < tol
