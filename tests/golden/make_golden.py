#!/usr/bin/env python
"""Generate golden vectors by EXECUTING the reference's own leaf modules.

JAX is not installable in the build container, but the reference's hot-path
leaves (internal/{math,stepfun,render,coord,ref_utils,image}.py) only use a
NumPy-shaped subset of `jax.numpy` plus a handful of jax transforms.  This
script installs a tiny stand-in (`jax.numpy` -> NumPy in float64; `jax.vmap`,
`jax.linearize`, `jax.custom_jvp`, `jax.nn.softmax`, `jax.random.uniform`
restated), imports the reference files FROM WHERE THEY LIE (/root/reference,
read-only; nothing is copied) and records their outputs on seeded inputs.

Because the stand-in computes in float64, the goldens are "the reference's
arithmetic at higher precision"; the fp32 oracle is compared with a tolerance
(tests/test_oracle_leaves.py), the integer outputs (searchsorted indices,
sample indices) exactly.

Run (in the build container only):  python tests/golden/make_golden.py
Writes tests/golden/leaves.npz.  /root/reference does not exist on the GPU box;
tests read only the committed .npz.
"""

import math
import os
import sys
import types

import numpy as np

REF = os.environ.get('MULTINERF_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'leaves.npz')


class JitterKey:
  """Stands in for a jax PRNG key: carries the uniform[0,1) draws to hand out."""

  def __init__(self, u01):
    self.u01 = np.asarray(u01, dtype=np.float64)


def install_jax_standin():
  jnp = types.ModuleType('jax.numpy')
  for name in dir(np):
    if not name.startswith('__'):
      setattr(jnp, name, getattr(np, name))

  def matmul(a, b, precision=None):
    return np.matmul(a, b)

  jnp.matmul = matmul
  jnp.zeros = lambda shape, dtype=np.float64: np.zeros(shape, dtype)
  jnp.ones = lambda shape, dtype=np.float64: np.ones(shape, dtype)
  jnp.array = lambda x, dtype=None: np.array(x, dtype=dtype)
  jnp.copy = np.copy
  # NEP-50: np.float32 scalars would drag float64 expressions down to float32;
  # hand out Python floats so the stand-in stays float64 throughout.
  jnp.finfo = lambda dt: types.SimpleNamespace(
      eps=float(np.finfo(dt).eps), max=float(np.finfo(dt).max), min=float(np.finfo(dt).min))
  # jax gathers never raise on out-of-range indices (they clamp / fill).
  jnp.take_along_axis = lambda a, idx, axis: np.take_along_axis(
      a, np.clip(idx, 0, a.shape[axis] - 1), axis)

  jax = types.ModuleType('jax')
  jax.numpy = jnp

  nn = types.ModuleType('jax.nn')

  def softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)

  nn.softmax = softmax
  nn.relu = lambda x: np.maximum(x, 0)
  nn.softplus = lambda x: np.logaddexp(x, 0)
  nn.sigmoid = lambda x: 1 / (1 + np.exp(-x))
  jax.nn = nn

  lax = types.ModuleType('jax.lax')
  lax.Precision = types.SimpleNamespace(HIGHEST='highest')
  lax.stop_gradient = lambda x: x
  jax.lax = lax

  random = types.ModuleType('jax.random')

  def uniform(key, shape=(), minval=0., maxval=1.):
    u = np.broadcast_to(key.u01, shape)
    return u * (maxval - minval) + minval

  random.uniform = uniform
  jax.random = random

  def vmap(fn, in_axes=0, out_axes=0):
    def wrapped(*args):
      axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
      n = args[0].shape[axes[0]]
      outs = [fn(*[np.take(a, i, axis=ax) for a, ax in zip(args, axes)]) for i in range(n)]
      return np.stack(outs, axis=out_axes)
    return wrapped

  jax.vmap = vmap

  def linearize(fn, x):
    """(fn(x), v -> J v) by central differences in float64 (rel. err ~1e-9)."""
    y = fn(x)
    h = 1e-6

    def lin(v):
      return (fn(x + h * v) - fn(x - h * v)) / (2 * h)

    return y, lin

  jax.linearize = linearize

  class custom_jvp:
    def __init__(self, fn):
      self.fn = fn

    def __call__(self, *a, **k):
      return self.fn(*a, **k)

    def defjvp(self, f):
      self.jvp = f
      return f

  jax.custom_jvp = custom_jvp

  sys.modules['jax'] = jax
  sys.modules['jax.numpy'] = jnp
  sys.modules['jax.nn'] = nn
  sys.modules['jax.lax'] = lax
  sys.modules['jax.random'] = random
  sys.modules['dm_pix'] = types.ModuleType('dm_pix')
  if not hasattr(np, 'math'):
    np.math = math  # reference ref_utils.py:55 uses np.math.factorial (NumPy < 2)
  return jax


def main():
  install_jax_standin()
  sys.path.insert(0, REF)
  from internal import coord, image, ref_utils, render, stepfun
  from internal import math as rmath

  g = {}
  rs = np.random.RandomState(20200823)

  # ---------------- math ----------------
  x = np.concatenate([rs.uniform(-400, 400, 64), rs.uniform(-5000, 5000, 64),
                      np.array([0., 314.15, 314.16, -314.16, 1e4])])
  g['safe_sin_x'] = x
  g['safe_sin_y'] = rmath.safe_sin(x)
  g['safe_cos_y'] = rmath.safe_cos(x)
  steps = np.array([0, 1, 100, 256, 512, 1000, 125000, 250000], dtype=np.float64)
  g['lr_steps'] = steps
  g['lr_vals'] = np.array([rmath.learning_rate_decay(s, 2e-3, 2e-5, 250000, 512, 0.01)
                           for s in steps])
  xp = np.sort(rs.uniform(0, 1, (7, 33)), -1)
  fp = np.sort(rs.uniform(-2, 3, (7, 33)), -1)
  xq = np.sort(rs.uniform(-0.1, 1.1, (7, 19)), -1)
  g['si_x'], g['si_xp'], g['si_fp'] = xq, xp, fp
  g['si_sorted'] = rmath.sorted_interp(xq, xp, fp)
  g['si_interp'] = rmath.interp(xq, xp, fp)

  # ---------------- stepfun ----------------
  B, n_env, n_in = 9, 24, 11
  t_env = np.cumsum(rs.uniform(0.0, 1.0, (B, n_env + 1)), -1)
  t_env = (t_env - t_env[:, :1]) / (t_env[:, -1:] - t_env[:, :1])
  w_env = rs.dirichlet(np.ones(n_env), B)
  t_in = np.sort(rs.uniform(-0.05, 1.05, (B, n_in + 1)), -1)
  w_in = rs.dirichlet(np.ones(n_in), B) * 0.9
  g['sf_t_env'], g['sf_w_env'], g['sf_t'], g['sf_w'] = t_env, w_env, t_in, w_in
  lo, hi = stepfun.searchsorted(t_env, t_in)
  g['sf_search_lo'], g['sf_search_hi'] = lo.astype(np.int64), hi.astype(np.int64)
  inner, outer = stepfun.inner_outer(t_in, t_env, w_env)
  g['sf_inner'], g['sf_outer'] = inner, outer
  g['sf_lossfun_outer'] = stepfun.lossfun_outer(t_in, w_in, t_env, w_env)
  g['sf_lossfun_distortion'] = stepfun.lossfun_distortion(t_in, w_in)
  tq = rs.uniform(-0.1, 1.1, (B, 17))
  g['sf_query_tq'] = tq
  g['sf_query'] = stepfun.query(tq, t_env, w_env)
  g['sf_pdf'] = stepfun.weight_to_pdf(t_env, w_env)
  for name, dil, dom in [('a', 0.0103125, (0., 1.)), ('b', 0.05, (-np.inf, np.inf)),
                         ('c', 0.00262207, (0., 1.))]:
    td, wd = stepfun.max_dilate_weights(t_env, w_env, dil, domain=dom, renormalize=True)
    g[f'sf_dilate_{name}_t'], g[f'sf_dilate_{name}_w'] = td, wd
    g[f'sf_dilate_{name}_args'] = np.array([dil, dom[0], dom[1]])
  g['sf_integrate'] = stepfun.integrate_weights(w_env)
  logits = np.log(w_env) * 0.9
  logits[0, 3] = -np.inf          # a zero-width / masked bin
  logits[1, :] = 0.0              # flat
  g['sf_logits'] = logits
  u = np.sort(rs.uniform(0, 1 - 1e-7, (B, 13)), -1)
  g['sf_u'] = u
  g['sf_invert_cdf'] = stepfun.invert_cdf(u, t_env, logits)
  g['sf_invert_cdf_gpu'] = stepfun.invert_cdf(u, t_env, logits, use_gpu_resampling=True)
  for ns in (8, 32):
    g[f'sf_sample_det_{ns}'] = stepfun.sample(None, t_env, logits, ns)
    g[f'sf_sample_detc_{ns}'] = stepfun.sample(None, t_env, logits, ns, deterministic_center=True)
    g[f'sf_sample_intervals_det_{ns}'] = stepfun.sample_intervals(
        None, t_env, logits, ns, single_jitter=True, domain=(0., 1.))
    u1 = rs.uniform(0, 1, (B, 1))
    un = rs.uniform(0, 1, (B, ns))
    g[f'sf_jit1_{ns}'], g[f'sf_jitn_{ns}'] = u1, un
    g[f'sf_sample_intervals_jit1_{ns}'] = stepfun.sample_intervals(
        JitterKey(u1), t_env, logits, ns, single_jitter=True, domain=(0., 1.))
    g[f'sf_sample_intervals_jitn_{ns}'] = stepfun.sample_intervals(
        JitterKey(un), t_env, logits, ns, single_jitter=False, domain=(0., 1.))
  g['sf_percentile'] = stepfun.weighted_percentile(t_env, w_env, [5, 50, 95])
  # The reference's own known-answer test (tests/stepfun_test.py:579-586).
  g['sf_single_interval'] = stepfun.sample_intervals(
      None, np.array([1., 2, 3, 4, 5, 6]), np.array([0., 0, 100, 0, 0]), 10, single_jitter=True)

  # ---------------- render ----------------
  R, n = 6, 10
  d = rs.normal(size=(R, 3)) * np.array([1.0, 1.1, 0.9])
  o = rs.uniform(-1, 1, (R, 3))
  radii = rs.uniform(3e-4, 1e-3, (R, 1))
  tdist = np.cumsum(rs.uniform(0.05, 2.0, (R, n + 1)), -1)
  tdist[0] = 1.0 / np.linspace(1 / 0.2, 1e-6, n + 1)   # reciprocal-spaced, huge far
  g['rd_d'], g['rd_o'], g['rd_radii'], g['rd_tdist'] = d, o, radii, tdist
  for shape in ('cone', 'cylinder'):
    for diag in (False, True):
      m, c = render.cast_rays(tdist, o, d, radii, shape, diag=diag)
      g[f'rd_cast_{shape}_{int(diag)}_mean'] = m
      g[f'rd_cast_{shape}_{int(diag)}_cov'] = c
  density = np.exp(rs.normal(size=(R, n)))
  density[2] = 0.0
  rgbs = rs.uniform(0, 1, (R, n, 3))
  g['rd_density'], g['rd_rgbs'] = density, rgbs
  for opaque in (False, True):
    w, a, tr = render.compute_alpha_weights(density, tdist, d, opaque_background=opaque)
    g[f'rd_alpha_{int(opaque)}_w'], g[f'rd_alpha_{int(opaque)}_a'] = w, a
    g[f'rd_alpha_{int(opaque)}_t'] = tr
    t_far = np.full((R, 1), 1e6)
    extras = {'normals': rs.normal(size=(R, n, 3)), 'roughness': rs.uniform(size=(R, n, 1))}
    g[f'rd_vr_{int(opaque)}_normals_in'] = extras['normals']
    g[f'rd_vr_{int(opaque)}_roughness_in'] = extras['roughness']
    out = render.volumetric_rendering(rgbs, w, tdist, 0.5, t_far, True, extras=extras)
    for k, v in out.items():
      g[f'rd_vr_{int(opaque)}_{k}'] = v

  # ---------------- coord ----------------
  xs = np.concatenate([rs.normal(size=(20, 3)) * 0.3, rs.normal(size=(20, 3)) * 5,
                       rs.normal(size=(5, 3)) * 1e3, np.zeros((1, 3))])
  g['cd_x'] = xs
  g['cd_contract'] = coord.contract(xs)
  A = rs.normal(size=(xs.shape[0], 3, 3))
  cov = A @ np.swapaxes(A, -1, -2) * 0.01
  g['cd_cov'] = cov
  fm, fc = coord.track_linearize(coord.contract, xs, cov)
  g['cd_tl_mean'], g['cd_tl_cov'] = fm, fc
  near, far = np.full((4, 1), 0.2), np.full((4, 1), 1e6)
  s = np.linspace(0, 1, 9)[None].repeat(4, 0)
  g['cd_s'] = s
  import jax.numpy as jnp
  jnp.reciprocal.__name__  # noqa (np ufunc has __name__)
  for name, fn in [('none', None), ('piecewise', 'piecewise'), ('reciprocal', jnp.reciprocal)]:
    t_to_s, s_to_t = coord.construct_ray_warps(fn, near, far)
    t = s_to_t(s[:, :-1] if name == 'piecewise' else s)
    g[f'cd_warp_{name}_t'] = t
    g[f'cd_warp_{name}_s'] = t_to_s(t)
  near2, far2 = np.full((4, 1), 2.), np.full((4, 1), 6.)
  _, s_to_t = coord.construct_ray_warps(None, near2, far2)
  g['cd_warp_lin26_t'] = s_to_t(s)
  basis = rs.normal(size=(3, 21))
  g['cd_basis'] = basis
  lm, lv = coord.lift_and_diagonalize(fm, fc, basis)
  g['cd_lift_mean'], g['cd_lift_var'] = lm, lv
  lmc = np.clip(lm, -2, 2)
  g['cd_ipe_mean'], g['cd_ipe_var'] = lmc, np.abs(lv)
  g['cd_ipe_0_12'] = coord.integrated_pos_enc(lmc, np.abs(lv), 0, 12)
  g['cd_ipe_0_16'] = coord.integrated_pos_enc(lmc[:8, :3], np.abs(lv)[:8, :3], 0, 16)
  vd = rs.normal(size=(11, 3))
  vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
  g['cd_viewdirs'] = vd
  g['cd_pos_enc_0_4'] = coord.pos_enc(vd, 0, 4, append_identity=True)

  # ---------------- ref_utils ----------------
  nrm = rs.normal(size=(11, 3))
  nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
  g['ru_normals'] = nrm
  g['ru_reflect'] = ref_utils.reflect(vd, nrm)
  g['ru_l2n'] = ref_utils.l2_normalize(rs.normal(size=(5, 3)) * 1e-3)
  g['ru_l2n_in'] = None
  kinv = rs.uniform(0, 1.5, (11, 1))
  g['ru_kappa_inv'] = kinv
  for deg in (1, 3, 5):
    g[f'ru_ide_{deg}'] = ref_utils.generate_ide_fn(deg)(vd, kinv)
  g['ru_ml_array_5'] = ref_utils.get_ml_array(5)

  # ---------------- image ----------------
  lin = np.concatenate([np.linspace(0, 1, 33), np.array([0.0031308, 0.003, 1e-9])])
  g['im_linear'] = lin
  g['im_srgb'] = image.linear_to_srgb(lin, xnp=np)
  g['im_mse'] = np.array([1e-4, 1e-3, 0.01, 0.1, 1.0])
  g['im_psnr'] = image.mse_to_psnr(g['im_mse'])

  # l2_normalize input (recorded after the call above consumed the RandomState).
  rs2 = np.random.RandomState(7)
  x_l2 = rs2.normal(size=(5, 3)) * 1e-3
  x_l2[0] = 0
  g['ru_l2n_in'] = x_l2
  g['ru_l2n'] = ref_utils.l2_normalize(x_l2)

  # ---------------- camera_utils (ray generation leaves; executed with xnp = numpy) ----------------
  # internal/camera_utils.py imports internal.configs / internal.utils (gin, flax): stub what it touches.
  import dataclasses
  cfg_stub = types.ModuleType('internal.configs')
  utils_stub = types.ModuleType('internal.utils')

  @dataclasses.dataclass
  class Pixels:
    pix_x_int: object
    pix_y_int: object
    lossmult: object
    near: object
    far: object
    cam_idx: object
    exposure_idx: object = None
    exposure_values: object = None

  @dataclasses.dataclass
  class Rays:
    origins: object
    directions: object
    viewdirs: object
    radii: object
    imageplane: object
    lossmult: object
    near: object
    far: object
    cam_idx: object
    exposure_idx: object = None
    exposure_values: object = None

  utils_stub.Pixels, utils_stub.Rays = Pixels, Rays
  cfg_stub.Config = object          # only used in a type annotation (camera_utils.py:344)
  sys.modules['internal.configs'] = cfg_stub
  sys.modules['internal.utils'] = utils_stub
  from internal import camera_utils
  rc = np.random.RandomState(424242)
  ncam, B = 5, 48
  W, H, focal = 64, 48, 70.0
  pixtocam = np.linalg.inv(np.array([[focal, 0, W / 2.], [0, focal, H / 2.], [0, 0, 1.]]))
  pixtocams = np.stack([pixtocam * (1 + 0.02 * i) for i in range(ncam)], 0)
  c2w = []
  for i in range(ncam):
    q, _ = np.linalg.qr(rc.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
      q[:, 0] = -q[:, 0]
    c2w.append(np.concatenate([q, rc.uniform(-1, 1, (3, 1))], 1))
  c2w = np.stack(c2w, 0)
  px = rc.randint(0, W, (B,))
  py = rc.randint(0, H, (B,))
  ci = rc.randint(0, ncam, (B, 1))
  g['cam_pixtocams'], g['cam_camtoworlds'] = pixtocams, c2w
  g['cam_pix_x'], g['cam_pix_y'], g['cam_idx'] = px.astype(np.int64), py.astype(np.int64), ci.astype(np.int64)
  dist = dict(k1=0.05, k2=-0.02, k3=0.003, k4=0.0, p1=0.001, p2=-0.002)
  g['cam_dist'] = np.array([dist[k] for k in ('k1', 'k2', 'k3', 'k4', 'p1', 'p2')])
  # forward-facing variant for NDC (dz < 0 in OpenGL coordinates): small rotations about the identity pose
  c2w_ff = np.stack([np.concatenate([np.eye(3) + 0.05 * rc.normal(size=(3, 3)), rc.uniform(-0.2, 0.2, (3, 1))], 1)
                     for _ in range(ncam)], 0)
  g['cam_camtoworlds_ff'] = c2w_ff
  pixels = Pixels(pix_x_int=px, pix_y_int=py, lossmult=np.ones((B, 1)), near=np.full((B, 1), 0.2),
                  far=np.full((B, 1), 100.), cam_idx=ci)
  cases = {
      'persp': ((pixtocams, c2w, None, None), camera_utils.ProjectionType.PERSPECTIVE),
      'single': ((pixtocams[0], c2w[0], None, None), camera_utils.ProjectionType.PERSPECTIVE),
      'dist': ((pixtocams, c2w, dist, None), camera_utils.ProjectionType.PERSPECTIVE),
      'fisheye': ((pixtocams, c2w, dist, None), camera_utils.ProjectionType.FISHEYE),
      'ndc': ((pixtocams, c2w_ff, None, pixtocam), camera_utils.ProjectionType.PERSPECTIVE),
  }
  for name, (cams, ct) in cases.items():
    r = camera_utils.cast_ray_batch(cams, pixels, ct, xnp=np)
    for f in ('origins', 'directions', 'viewdirs', 'radii', 'imageplane'):
      g[f'cam_{name}_{f}'] = getattr(r, f)
  xu, yu = camera_utils._radial_and_tangential_undistort(rc.uniform(-0.6, 0.6, 40), rc.uniform(-0.5, 0.5, 40), **dist)
  g['cam_undistort_in'] = np.stack([xu * 0, yu * 0])  # placeholder overwritten below
  xd_in, yd_in = rc.uniform(-0.6, 0.6, 40), rc.uniform(-0.5, 0.5, 40)
  xu, yu = camera_utils._radial_and_tangential_undistort(xd_in, yd_in, **dist)
  g['cam_undistort_in'] = np.stack([xd_in, yd_in])
  g['cam_undistort_out'] = np.stack([xu, yu])

  # pose utilities (pure NumPy in the reference)
  rp = np.random.RandomState(99)
  poses = []
  for i in range(9):
    q, _ = np.linalg.qr(rp.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
      q[:, 0] = -q[:, 0]
    ang = 2 * np.pi * i / 9
    pos = np.array([3 * np.cos(ang), 2 * np.sin(ang), 0.3 * rp.normal()]) + np.array([5., -2., 1.])
    poses.append(np.concatenate([0.7 * q + 0.3 * np.eye(3), pos[:, None]], 1))
  poses = np.stack(poses, 0)
  g['pose_in'] = poses
  g['pose_pad'] = camera_utils.pad_poses(poses)
  g['pose_average'] = camera_utils.average_pose(poses)
  rc_p, rc_t = camera_utils.recenter_poses(poses)
  g['pose_recenter_poses'], g['pose_recenter_transform'] = rc_p, rc_t
  g['pose_focus_point'] = camera_utils.focus_point_fn(poses)
  pca_p, pca_t = camera_utils.transform_poses_pca(poses)
  g['pose_pca_poses'], g['pose_pca_transform'] = pca_p, pca_t
  g['pose_viewmatrix'] = camera_utils.viewmatrix(np.array([0.2, -0.3, 0.9]), np.array([0., 1., 0.1]), np.array([1., 2., 3.]))

  g = {k: np.asarray(v) for k, v in g.items() if v is not None}
  np.savez_compressed(OUT, **g)
  print(f'wrote {OUT}: {len(g)} arrays, {os.path.getsize(OUT)} bytes')


if __name__ == '__main__':
  main()
