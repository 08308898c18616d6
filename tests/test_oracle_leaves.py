"""Pin the oracle's leaves against outputs of the reference's own source files.

tests/golden/leaves.npz was produced by tests/golden/make_golden.py, which
executes /root/reference/internal/{math,stepfun,render,coord,ref_utils,image}.py
with a NumPy(float64) stand-in for jax.numpy.  Here the oracle is evaluated in
float64 (must agree to round-off: it is the same arithmetic) and in float32 (the
working precision of the parity tests; tolerance stated per case).
"""

import numpy as np
import pytest
import torch

from oracle import coord, image, ref_utils, render, stepfun
from oracle import math as rmath

DT = [torch.float64, torch.float32]


def T(x, dt):
  x = np.asarray(x)
  if x.dtype.kind == 'f':
    return torch.as_tensor(x, dtype=dt)
  return torch.as_tensor(x)


def close(a, b, dt, tol64=1e-10, tol32=2e-5, rtol32=2e-5):
  a = a.detach().numpy().astype(np.float64)
  b = np.asarray(b, dtype=np.float64)
  if dt == torch.float64:
    np.testing.assert_allclose(a, b, atol=tol64, rtol=tol64)
  else:
    np.testing.assert_allclose(a, b, atol=tol32, rtol=rtol32)


@pytest.mark.parametrize('dt', DT)
def test_math(golden, dt):
  g = golden
  x = T(g['safe_sin_x'], dt)
  if dt == torch.float64:
    close(rmath.safe_sin(x), g['safe_sin_y'], dt)
    close(rmath.safe_cos(x), g['safe_cos_y'], dt)
  else:
    # fp32: |x| up to 5000 -> argument spacing 4.9e-4 dominates (tests/math_test.py:38-47 uses 1e-4 on the wrapped value).
    x64 = x.double()
    close(rmath.safe_sin(x), rmath.safe_sin(x64).numpy(), dt, tol32=2e-3)
  for s, v in zip(g['lr_steps'], g['lr_vals']):
    assert abs(rmath.learning_rate_decay(float(s), 2e-3, 2e-5, 250000, 512, 0.01) - v) < 1e-12
  xq, xp, fp = T(g['si_x'], dt), T(g['si_xp'], dt), T(g['si_fp'], dt)
  close(rmath.sorted_interp(xq, xp, fp), g['si_sorted'], dt)
  close(rmath.interp(xq, xp, fp), g['si_interp'], dt)
  # tests/math_test.py:156-178: sorted_interp == np.interp inside the range.
  inside = (g['si_x'] >= g['si_xp'][:, :1]) & (g['si_x'] <= g['si_xp'][:, -1:])
  np.testing.assert_allclose(g['si_sorted'][inside], g['si_interp'][inside], atol=1e-12)


@pytest.mark.parametrize('dt', DT)
def test_stepfun(golden, dt):
  g = golden
  t_env, w_env = T(g['sf_t_env'], dt), T(g['sf_w_env'], dt)
  t, w = T(g['sf_t'], dt), T(g['sf_w'], dt)
  lo, hi = stepfun.searchsorted(t_env, t)
  if dt == torch.float64:
    assert np.array_equal(lo.numpy(), g['sf_search_lo'])
    assert np.array_equal(hi.numpy(), g['sf_search_hi'])
  inner, outer = stepfun.inner_outer(t, t_env, w_env)
  close(inner, g['sf_inner'], dt)
  close(outer, g['sf_outer'], dt)
  close(stepfun.lossfun_outer(t, w, t_env, w_env), g['sf_lossfun_outer'], dt)
  close(stepfun.lossfun_distortion(t, w), g['sf_lossfun_distortion'], dt)
  close(stepfun.query(T(g['sf_query_tq'], dt), t_env, w_env), g['sf_query'], dt)
  close(stepfun.weight_to_pdf(t_env, w_env), g['sf_pdf'], dt, rtol32=1e-4, tol32=1e-4)
  for name in 'abc':
    dil, d0, d1 = g[f'sf_dilate_{name}_args']
    td, wd = stepfun.max_dilate_weights(t_env, w_env, float(dil), domain=(float(d0), float(d1)),
                                        renormalize=True)
    close(td, g[f'sf_dilate_{name}_t'], dt)
    # fp32: dilated fence-posts can tie/swap at 1e-7; weights are compared as a measure.
    close(wd.sum(-1), g[f'sf_dilate_{name}_w'].sum(-1), dt)
    if dt == torch.float64:
      close(wd, g[f'sf_dilate_{name}_w'], dt)
  close(stepfun.integrate_weights(w_env), g['sf_integrate'], dt)
  logits, u = T(g['sf_logits'], dt), T(g['sf_u'], dt)
  close(stepfun.invert_cdf(u, t_env, logits), g['sf_invert_cdf'], dt)
  close(stepfun.invert_cdf(u, t_env, logits, use_gpu_resampling=True), g['sf_invert_cdf_gpu'], dt)
  for ns in (8, 32):
    close(stepfun.sample(None, t_env, logits, ns), g[f'sf_sample_det_{ns}'], dt)
    close(stepfun.sample(None, t_env, logits, ns, deterministic_center=True),
          g[f'sf_sample_detc_{ns}'], dt)
    close(stepfun.sample_intervals(None, t_env, logits, ns, single_jitter=True, domain=(0., 1.)),
          g[f'sf_sample_intervals_det_{ns}'], dt)
    close(stepfun.sample_intervals(T(g[f'sf_jit1_{ns}'], dt), t_env, logits, ns,
                                   single_jitter=True, domain=(0., 1.)),
          g[f'sf_sample_intervals_jit1_{ns}'], dt)
    close(stepfun.sample_intervals(T(g[f'sf_jitn_{ns}'], dt), t_env, logits, ns,
                                   single_jitter=False, domain=(0., 1.)),
          g[f'sf_sample_intervals_jitn_{ns}'], dt)
  close(stepfun.weighted_percentile(t_env, w_env, [5, 50, 95]), g['sf_percentile'], dt)


@pytest.mark.parametrize('dt', DT)
def test_sample_single_interval_known_answer(golden, dt):
  """reference tests/stepfun_test.py:579-586."""
  t = torch.tensor([1., 2, 3, 4, 5, 6], dtype=dt)
  logits = torch.tensor([0., 0, 100, 0, 0], dtype=dt)
  out = stepfun.sample_intervals(None, t, logits, 10, single_jitter=True)
  np.testing.assert_allclose(out.numpy(), np.linspace(3, 4, 11), atol=1e-5, rtol=1e-5)
  np.testing.assert_allclose(golden['sf_single_interval'], np.linspace(3, 4, 11), atol=1e-5)
  with pytest.raises(ValueError):
    stepfun.sample_intervals(None, t, logits, 1)


@pytest.mark.parametrize('dt', DT)
def test_render(golden, dt):
  g = golden
  d, o, radii, tdist = [T(g[k], dt) for k in ('rd_d', 'rd_o', 'rd_radii', 'rd_tdist')]
  for shape in ('cone', 'cylinder'):
    for diag in (False, True):
      m, c = render.cast_rays(tdist, o, d, radii, shape, diag=diag)
      gm, gc = g[f'rd_cast_{shape}_{int(diag)}_mean'], g[f'rd_cast_{shape}_{int(diag)}_cov']
      if dt == torch.float64:
        close(m, gm, dt)
        close(c, gc, dt)
      else:
        # Row 0 reaches t ~ 1e6 (360.gin far plane): compare relative to scale.
        np.testing.assert_allclose(m.numpy(), gm, rtol=2e-5, atol=2e-5 * np.abs(gm).max(-1, keepdims=True).max(-2, keepdims=True).max())
        sc = np.abs(gc).reshape(gc.shape[0], -1).max(-1)
        err = np.abs(c.numpy() - gc).reshape(gc.shape[0], -1).max(-1)
        assert (err <= 1e-4 * sc).all(), (err / sc)
  with pytest.raises(ValueError):
    render.cast_rays(tdist, o, d, radii, 'sphere')
  density, rgbs = T(g['rd_density'], dt), T(g['rd_rgbs'], dt)
  for opaque in (False, True):
    w, a, tr = render.compute_alpha_weights(density, tdist, d, opaque_background=opaque)
    close(w, g[f'rd_alpha_{int(opaque)}_w'], dt)
    close(a, g[f'rd_alpha_{int(opaque)}_a'], dt)
    close(tr, g[f'rd_alpha_{int(opaque)}_t'], dt)
    extras = {'normals': T(g[f'rd_vr_{int(opaque)}_normals_in'], dt),
              'roughness': T(g[f'rd_vr_{int(opaque)}_roughness_in'], dt)}
    t_far = torch.full((tdist.shape[0], 1), 1e6, dtype=dt)
    out = render.volumetric_rendering(rgbs, w, tdist, 0.5, t_far, True, extras=extras)
    for k, v in out.items():
      ref = g[f'rd_vr_{int(opaque)}_{k}']
      if dt == torch.float64:
        close(v, ref, dt)
      else:
        np.testing.assert_allclose(v.numpy(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('dt', DT)
def test_coord(golden, dt):
  g = golden
  x, cov = T(g['cd_x'], dt), T(g['cd_cov'], dt)
  close(coord.contract(x), g['cd_contract'], dt)
  fm, fc = coord.track_linearize(coord.contract, x, cov)
  close(fm, g['cd_tl_mean'], dt)
  # golden Jacobian is a central difference (rel 1e-9) -> 1e-6 abs on |cov| ~ 0.1.
  close(fc, g['cd_tl_cov'], dt, tol64=2e-6, tol32=2e-5)
  # generic (non-closed-form) path == closed form.
  fm2, fc2 = coord.track_linearize(lambda z: coord.contract(z), x.double(), cov.double())
  np.testing.assert_allclose(fc2.numpy(), coord.track_linearize(coord.contract, x.double(), cov.double())[1].numpy(), atol=1e-12)
  with pytest.raises(ValueError):
    coord.track_linearize(coord.contract, x, cov[..., 0])
  s = T(g['cd_s'], dt)
  near, far = torch.full((4, 1), 0.2, dtype=dt), torch.full((4, 1), 1e6, dtype=dt)
  for name, fn in [('none', None), ('piecewise', 'piecewise'), ('reciprocal', 'reciprocal')]:
    t_to_s, s_to_t = coord.construct_ray_warps(fn, near, far)
    t = s_to_t(s[:, :-1] if name == 'piecewise' else s)
    np.testing.assert_allclose(t.numpy(), g[f'cd_warp_{name}_t'], rtol=1e-10 if dt == torch.float64 else 1e-5)
    np.testing.assert_allclose(t_to_s(T(g[f'cd_warp_{name}_t'], dt)).numpy(), g[f'cd_warp_{name}_s'],
                               atol=1e-10 if dt == torch.float64 else 1e-5)
  _, s_to_t = coord.construct_ray_warps(None, torch.full((4, 1), 2., dtype=dt), torch.full((4, 1), 6., dtype=dt))
  close(s_to_t(s), g['cd_warp_lin26_t'], dt)
  basis = T(g['cd_basis'], dt)
  lm, lv = coord.lift_and_diagonalize(T(g['cd_tl_mean'], dt), T(g['cd_tl_cov'], dt), basis)
  close(lm, g['cd_lift_mean'], dt)
  close(lv, g['cd_lift_var'], dt, tol32=5e-5)
  m, v = T(g['cd_ipe_mean'], dt), T(g['cd_ipe_var'], dt)
  e = coord.integrated_pos_enc(m, v, 0, 12)
  assert e.shape[-1] == 2 * 21 * 12
  if dt == torch.float64:
    close(e, g['cd_ipe_0_12'], dt)
    close(coord.integrated_pos_enc(m[:8, :3], v[:8, :3], 0, 16), g['cd_ipe_0_16'], dt)
  else:
    # fp32 argument spacing at 2*2^11 is 4.9e-4 (tests/coord_test.py:112-127 uses per-degree tolerances).
    e64 = coord.integrated_pos_enc(m.double(), v.double(), 0, 12)
    close(e, e64.numpy(), dt, tol32=2e-3)
  close(coord.pos_enc(T(g['cd_viewdirs'], dt), 0, 4), g['cd_pos_enc_0_4'], dt, tol32=1e-5)


@pytest.mark.parametrize('dt', DT)
def test_ref_utils_image(golden, dt):
  g = golden
  vd, nrm = T(g['cd_viewdirs'], dt), T(g['ru_normals'], dt)
  close(ref_utils.reflect(vd, nrm), g['ru_reflect'], dt)
  close(ref_utils.l2_normalize(T(g['ru_l2n_in'], dt)), g['ru_l2n'], dt, tol32=1e-5)
  kinv = T(g['ru_kappa_inv'], dt)
  for deg in (1, 3, 5):
    close(ref_utils.generate_ide_fn(deg)(vd, kinv), g[f'ru_ide_{deg}'], dt, tol32=2e-4, rtol32=2e-4)
  assert np.array_equal(ref_utils.get_ml_array(5), g['ru_ml_array_5'])
  with pytest.raises(ValueError):
    ref_utils.generate_ide_fn(6)
  close(image.linear_to_srgb(T(g['im_linear'], dt)), g['im_srgb'], dt)
  close(image.mse_to_psnr(T(g['im_mse'], dt)), g['im_psnr'], dt, tol32=1e-4)


def test_ide_vs_scipy():
  """reference tests/ref_utils_test.py:61-83: IDE(kappa_inv=0) == sph_harm, atol 0.02."""
  import scipy.special
  rs = np.random.RandomState(0)
  xyz = rs.normal(size=(50, 3))
  xyz /= np.linalg.norm(xyz, axis=-1, keepdims=True)
  deg = 5
  de = ref_utils.generate_dir_enc_fn(deg)(torch.as_tensor(xyz)).numpy()
  ml = ref_utils.get_ml_array(deg)
  theta = np.arccos(xyz[:, 2])
  phi = np.arctan2(xyz[:, 1], xyz[:, 0])
  sh = np.stack([scipy.special.sph_harm(m, l, phi, theta) for m, l in ml.T], -1)
  np.testing.assert_allclose(de, np.concatenate([sh.real, sh.imag], -1), atol=0.02)


# ----------------------------------------------------------------------------- summation-order contract of the sampling path


def test_oracle_lanes_match_the_kernel_source():
  """oracle.stepfun.blocked_cumsum restates the level kernel's chunking by hand: the chunk count must be the kernel's
  lanes per ray (csrc/resample.hip RSP_LPR), or 'bit-exact by construction' silently stops being about the kernel."""
  import os
  import re
  src = open(os.path.join(os.path.dirname(__file__), '..', 'multinerf_amd', 'csrc', 'resample.hip')).read()
  m = re.search(r'^#define\s+RSP_LPR\s+(\d+)', src, re.M)
  assert m and int(m.group(1)) == stepfun.LANES


def _rand_stepfun(gen, B, n):
  d = torch.rand((B, n + 1), generator=gen) + 0.02
  t = torch.cumsum(d, -1)
  t = (t - t[:, :1]) / (t[:, -1:] - t[:, :1])
  w = torch.rand((B, n), generator=gen) ** 3
  return t.float(), (w / w.sum(-1, keepdim=True)).float()


@pytest.mark.parametrize('n_prev,n,dil', [(64, 64, 0.0103125), (64, 32, 0.00262207), (128, 128, 0.0)])
def test_kernel_order_vs_reference_order(n_prev, n, dil):
  """The fp32 oracle evaluates the sampling path in the HIP kernel's association order (blocked sums, the kernel's own
  exp / log); the reference sums left to right with the host library's exp / log (stepfun.py:146,156).  Both are held to
  the float64 goldens by tolerance above; here the two are compared with each other on the resample inputs of the GPU
  tests: the sample positions agree to a few ulps of the CDF and the INDEX mismatch rate between the two orders is
  reported and bounded (it is what 'bit-exact sample indices' costs when the order is not pinned)."""
  gen = torch.Generator().manual_seed(3)
  B = 400
  sd, w = _rand_stepfun(gen, B, n_prev)
  w = w * 0.98
  u_jit = torch.rand((B, 1), generator=gen)

  def run():
    s_, w_ = sd, w
    if dil > 0:
      s_, w_ = stepfun.max_dilate_weights(s_, w_, dil, domain=(0., 1.), renormalize=True)
      s_, w_ = s_[..., 1:-1], w_[..., 1:-1]
    logits = stepfun.resample_logits(s_, w_, 0.9090909, 0.0)
    return stepfun.sample_intervals(u_jit, s_, logits, n, single_jitter=True, domain=(0., 1.), return_index=True)

  s_k, i_k = run()
  with stepfun.reference_order():
    s_r, i_r = run()
  mism = int((i_k != i_r).sum())
  rate = mism / i_k.numel()
  print(f'n_prev={n_prev} n={n} dilation={dil}: kernel-order vs reference-order index mismatches {mism} of {i_k.numel()} '
        f'({rate:.2e}); max |s| difference {(s_k - s_r).abs().max().item():.2e}')
  assert rate < 5e-3
  # a flipped index moves a sample across a CDF knot it was within an ulp of: the position itself barely moves
  np.testing.assert_allclose(s_k.numpy(), s_r.numpy(), atol=5e-5, rtol=0)
