"""Oracle restatement of reference internal/models.py (TEST INFRASTRUCTURE ONLY).

Pinned: the reference has no test of Model.__call__ / MLP.__call__ and flax cannot be
imported here, so tests/golden/make_golden_models.py executes the reference's own
internal/models.py on stand-ins for jax.numpy / flax.linen / gin (complex-step
derivatives for the Ref-NeRF normals and the contraction's Jacobian) and
tests/test_oracle_models_golden.py holds model_apply() to every recorded entry of
`renderings` / `ray_history` at rtol 1e-9 in float64, for the four BASELINE
configurations, rendering and randomized training.  Also: the published parameter
counts (tests/test_oracle_models.py).

Differences of FORM (not of arithmetic) from the reference:
  * flax modules -> plain dataclasses + an explicit nested dict of torch
    tensors with flax's names ('NerfMLP_0'/'Dense_3'/'kernel' [in,out], 'bias').
  * every use of `rng` -> an explicit `noise` dict (None = rng None):
      noise['u_jitter'][level]        uniform [0,1), [B,1] or [B,n]
      noise['density_noise'][level]   standard normal [B,n]
      noise['bottleneck_noise'][level] standard normal [B,n,bottleneck]
      noise['bg_rgbs'][level]         uniform [0,1) [B,3]
  * callables configured through gin (@jnp.reciprocal, @coord.contract,
    @math.safe_exp, nn.relu, ...) -> their names as strings.
"""

import dataclasses
import math as _pm
from typing import Any, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from oracle import coord
from oracle import geopoly
from oracle import image
from oracle import math as rmath
from oracle import ref_utils
from oracle import render
from oracle import stepfun

_ACT = {
    'relu': torch.relu,
    'softplus': F.softplus,
    'silu': F.silu,
    'sigmoid': torch.sigmoid,
    'safe_exp': rmath.safe_exp,
    'exp': torch.exp,
}


@dataclasses.dataclass
class MLP:
  """Hyper-parameters of models.py:341-379 (same names, same defaults)."""
  net_depth: int = 8
  net_width: int = 256
  bottleneck_width: int = 256
  net_depth_viewdirs: int = 1
  net_width_viewdirs: int = 128
  net_activation: str = 'relu'
  min_deg_point: int = 0
  max_deg_point: int = 12
  weight_init: str = 'he_uniform'
  skip_layer: int = 4
  skip_layer_dir: int = 4
  num_rgb_channels: int = 3
  deg_view: int = 4
  use_reflections: bool = False
  use_directional_enc: bool = False
  enable_pred_roughness: bool = False
  roughness_activation: str = 'softplus'
  roughness_bias: float = -1.
  use_diffuse_color: bool = False
  use_specular_tint: bool = False
  use_n_dot_v: bool = False
  bottleneck_noise: float = 0.0
  density_activation: str = 'softplus'
  density_bias: float = -1.
  density_noise: float = 0.
  rgb_premultiplier: float = 1.
  rgb_activation: str = 'sigmoid'
  rgb_bias: float = 0.
  rgb_padding: float = 0.001
  enable_pred_normals: bool = False
  disable_density_normals: bool = False
  disable_rgb: bool = False
  warp_fn: Optional[str] = None
  basis_shape: str = 'icosahedron'
  basis_subdivisions: int = 2

  def check(self):
    """models.py:381-385."""
    if self.use_reflections and not (self.enable_pred_normals or
                                     not self.disable_density_normals):
      raise ValueError('Normals must be computed for reflection directions.')

  def pos_basis_t(self, dtype=torch.float32):
    """models.py:388-389 -- [3, K]."""
    return torch.as_tensor(
        np.ascontiguousarray(
            geopoly.generate_basis(self.basis_shape, self.basis_subdivisions).T),
        dtype=dtype)

  def dir_enc_fn(self):
    """models.py:391-400."""
    if self.use_directional_enc:
      return ref_utils.generate_ide_fn(self.deg_view)
    return lambda direction, _: coord.pos_enc(
        direction, min_deg=0, max_deg=self.deg_view, append_identity=True)


@dataclasses.dataclass
class Model:
  """Hyper-parameters of models.py:47-72 (same names, same defaults)."""
  num_prop_samples: int = 64
  num_nerf_samples: int = 32
  num_levels: int = 3
  bg_intensity_range: Tuple[float, float] = (1., 1.)
  anneal_slope: float = 10
  stop_level_grad: bool = True
  use_viewdirs: bool = True
  raydist_fn: Optional[str] = None
  ray_shape: str = 'cone'
  disable_integration: bool = False
  single_jitter: bool = True
  dilation_multiplier: float = 0.5
  dilation_bias: float = 0.0025
  num_glo_features: int = 0
  num_glo_embeddings: int = 1000
  learned_exposure_scaling: bool = False
  near_anneal_rate: Optional[float] = None
  near_anneal_init: float = 0.95
  single_mlp: bool = False
  resample_padding: float = 0.0
  use_gpu_resampling: bool = False
  opaque_background: bool = False
  vis_num_rays: int = 16  # Config.vis_num_rays (configs.py:76), read at models.py:287.


# ---------------------------------------------------------------------------
# Parameter construction (flax semantics restated).


def mlp_dense_shapes(mlp: MLP, use_viewdirs=True, num_glo_features=0):
  """[(in, out)] for Dense_0.. in flax creation order (models.py:456-585)."""
  k = geopoly.generate_basis(mlp.basis_shape, mlp.basis_subdivisions).shape[0]
  feat = 2 * k * (mlp.max_deg_point - mlp.min_deg_point)
  shapes = []
  width = feat
  for i in range(mlp.net_depth):
    shapes.append((width, mlp.net_width))
    width = mlp.net_width
    if i % mlp.skip_layer == 0 and i > 0:
      width += feat
  x_width = width
  shapes.append((x_width, 1))  # density
  if mlp.enable_pred_normals:
    shapes.append((x_width, 3))
  if not mlp.disable_rgb:
    if use_viewdirs:
      if mlp.use_diffuse_color:
        shapes.append((x_width, mlp.num_rgb_channels))
      if mlp.use_specular_tint:
        shapes.append((x_width, 3))
      if mlp.enable_pred_roughness:
        shapes.append((x_width, 1))
      width = 0
      if mlp.bottleneck_width > 0:
        shapes.append((x_width, mlp.bottleneck_width))
        width += mlp.bottleneck_width
      if mlp.use_directional_enc:
        width += 2 * get_num_ide(mlp.deg_view)
      else:
        width += 3 + 2 * 3 * mlp.deg_view
      if mlp.use_n_dot_v:
        width += 1
      width += num_glo_features
      in_w = width
      for i in range(mlp.net_depth_viewdirs):
        shapes.append((width, mlp.net_width_viewdirs))
        width = mlp.net_width_viewdirs
        if i % mlp.skip_layer_dir == 0 and i > 0:
          width += in_w
      x_width = width
    shapes.append((x_width, mlp.num_rgb_channels))
  return shapes


def get_num_ide(deg_view):
  return ref_utils.get_ml_array(deg_view).shape[1]


def _init_kernel(shape, kind, gen, dtype):
  """jax.nn.initializers.{he,glorot}_{uniform,normal} restated (fan_in = in)."""
  fan_in, fan_out = shape
  if kind == 'he_uniform':
    lim = _pm.sqrt(6.0 / fan_in)
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1).mul(lim).to(dtype)
  if kind == 'glorot_uniform':
    lim = _pm.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1).mul(lim).to(dtype)
  if kind == 'he_normal':
    return (torch.randn(shape, generator=gen, dtype=torch.float64) * _pm.sqrt(2.0 / fan_in)).to(dtype)
  if kind == 'glorot_normal':
    return (torch.randn(shape, generator=gen, dtype=torch.float64) *
            _pm.sqrt(2.0 / (fan_in + fan_out))).to(dtype)
  raise ValueError(kind)


def init_params(model: Model, nerf_mlp: MLP, prop_mlp: Optional[MLP], seed=0,
                dtype=torch.float32):
  """Random-init parameters with flax's tree layout (models.py:98-121, 436-437).

  Top-level creation order: NerfMLP_0, PropMLP_0 (unless single_mlp), Embed_0
  (GLO), exposure_scaling_offsets.  Biases zero; Embed_0 ~ N(0,1)/sqrt(features)
  (flax default_embed_init = variance_scaling(1.0,'fan_in','normal',out_axis=0));
  exposure offsets zero (models.py:116).
  """
  gen = torch.Generator().manual_seed(seed)
  params = {}

  def make(mlp, is_nerf):
    d = {}
    glo = model.num_glo_features if is_nerf else 0
    for k, shp in enumerate(mlp_dense_shapes(mlp, model.use_viewdirs, glo)):
      d[f'Dense_{k}'] = {
          'kernel': _init_kernel(shp, mlp.weight_init, gen, dtype),
          'bias': torch.zeros(shp[1], dtype=dtype),
      }
    return d

  params['NerfMLP_0'] = make(nerf_mlp, True)
  if not model.single_mlp:
    params['PropMLP_0'] = make(prop_mlp, False)
  if model.num_glo_features > 0:
    params['Embed_0'] = {
        'embedding': (torch.randn((model.num_glo_embeddings, model.num_glo_features),
                                  generator=gen, dtype=torch.float64) /
                      _pm.sqrt(model.num_glo_features)).to(dtype)
    }
  if model.learned_exposure_scaling:
    params['exposure_scaling_offsets'] = {
        'embedding': torch.zeros((model.num_glo_embeddings, 3), dtype=dtype)
    }
  return params


def param_count(params):
  n = 0
  for v in params.values():
    n += param_count(v) if isinstance(v, dict) else v.numel()
  return n


# ---------------------------------------------------------------------------
# MLP.__call__ (models.py:402-612).


class _DenseBf16FwdBwd(torch.autograd.Function):
  """x @ kern with BOTH passes at the reference's TPU default precision (flax nn.Dense without a precision argument: every
  matmul operand is rounded to bf16, the products accumulate in fp32): the forward rounds x and the kernel, the backward
  rounds the incoming gradient as well (dX = g W^T, dW = x^T g with g in bf16).  That is also what the HIP path stores: `dY` in bf16."""

  @staticmethod
  def forward(ctx, x, kern):
    bf = torch.bfloat16
    xr, kr = x.to(bf).to(x.dtype), kern.to(bf).to(kern.dtype)
    ctx.save_for_backward(xr, kr)
    return rmath.matmul(xr, kr)

  @staticmethod
  def backward(ctx, g):
    xr, kr = ctx.saved_tensors
    gr = g.to(torch.bfloat16).to(g.dtype)
    dx = rmath.matmul(gr, kr.t())
    dk = rmath.matmul(xr.reshape(-1, xr.shape[-1]).t(), gr.reshape(-1, gr.shape[-1]))
    return dx, dk


BF16_FWD_BWD = 'bf16_fwd_bwd'      # dense_dtype value: bf16 operands in the forward AND the backward matmuls


class _DenseCursor:
  """Hands out Dense_k in call order, like flax's auto-naming.

  `dense_dtype=torch.bfloat16` emulates the MFMA path (and the reference's own
  TPU default precision, math.py:21-23): both Dense operands are rounded to bf16,
  products accumulate in the working dtype, bias is added in the working dtype.
  `dense_dtype=BF16_FWD_BWD` rounds the backward pass's incoming gradient too (`_DenseBf16FwdBwd`).
  """

  def __init__(self, p, dense_dtype=None):
    self.p, self.k, self.dd = p, 0, dense_dtype

  def __call__(self, x):
    layer = self.p[f'Dense_{self.k}']
    self.k += 1
    kern = layer['kernel']
    if isinstance(self.dd, str):
      assert self.dd == BF16_FWD_BWD, self.dd
      return _DenseBf16FwdBwd.apply(x, kern) + layer['bias']
    if self.dd is not None:
      x = x.to(self.dd).to(kern.dtype)
      kern = kern.to(self.dd).to(kern.dtype)
    return rmath.matmul(x, kern) + layer['bias']


def mlp_apply(mlp: MLP, p, gaussians, viewdirs=None, imageplane=None, glo_vec=None,
              exposure=None, density_noise=None, bottleneck_noise=None, dense_dtype=None, relu_sides=None):
  """MLP.__call__ -- returns the same dict as models.py:604-612.

  `relu_sides` (test hook, not a reference argument; ReLU networks only): {'masks': [bool tensor per activated Dense layer, in
  call order], 'stats': dict}.  Layer k then computes z * masks[k] instead of relu(z): the SIDE of every ReLU kink is taken
  from another evaluation of the same network (the fp32-Dense kernels', tests/helpers.py) instead of from sign(z).  The two
  differ only where |z| is within that evaluation's rounding of 0, where both sides are valid subgradients; `stats` receives
  the number of units, the number of disagreements with sign(z) and the largest |z| among them, which the caller bounds."""
  mlp.check()
  dense = _DenseCursor(p, dense_dtype)
  basis = mlp.pos_basis_t(gaussians[0].dtype)
  act = _ACT[mlp.net_activation]
  if relu_sides is not None and mlp.net_activation == 'relu':
    sides = iter(relu_sides['masks'])
    st = relu_sides.setdefault('stats', {})

    def act(z):                                            # noqa: F811
      m = next(sides).reshape(z.shape)
      own = z.detach() > 0
      dis = own != m
      st['units'] = st.get('units', 0) + z.numel()
      st['disagree'] = st.get('disagree', 0) + int(dis.sum())
      if dis.any():
        st['max_abs_z'] = max(st.get('max_abs_z', 0.0), float(z.detach().abs()[dis].max()))
      return z * m.to(z.dtype)

  def predict_density(means, covs):
    """models.py:441-465."""
    if mlp.warp_fn is not None:
      assert mlp.warp_fn == 'contract'
      means, covs = coord.track_linearize(coord.contract, means, covs)
    lifted_means, lifted_vars = coord.lift_and_diagonalize(means, covs, basis)
    x = coord.integrated_pos_enc(lifted_means, lifted_vars, mlp.min_deg_point,
                                 mlp.max_deg_point)
    inputs = x
    for i in range(mlp.net_depth):
      x = act(dense(x))
      if i % mlp.skip_layer == 0 and i > 0:
        x = torch.cat([x, inputs], dim=-1)
    raw_density = dense(x)[..., 0]
    if (density_noise is not None) and (mlp.density_noise > 0):
      raw_density = raw_density + mlp.density_noise * density_noise
    return raw_density, x

  means, covs = gaussians
  if mlp.disable_density_normals:
    raw_density, x = predict_density(means, covs)
    raw_grad_density = None
    normals = None
  else:
    # models.py:473-492: per-sample value_and_grad w.r.t. means.  Samples are
    # independent, so grad of the SUM w.r.t. means is the per-sample gradient.
    means_g = means if means.requires_grad else means.detach().requires_grad_(True)
    with torch.enable_grad():
      raw_density, x = predict_density(means_g, covs)
      (raw_grad_density,) = torch.autograd.grad(
          raw_density.sum(), means_g, create_graph=True)
    normals = -ref_utils.l2_normalize(raw_grad_density)

  if mlp.enable_pred_normals:
    grad_pred = dense(x)
    normals_pred = -ref_utils.l2_normalize(grad_pred)
    normals_to_use = normals_pred
  else:
    grad_pred = None
    normals_pred = None
    normals_to_use = normals

  density = _ACT[mlp.density_activation](raw_density + mlp.density_bias)

  roughness = None
  if mlp.disable_rgb:
    rgb = torch.zeros_like(means)
  else:
    if viewdirs is not None:
      if mlp.use_diffuse_color:
        raw_rgb_diffuse = dense(x)
      if mlp.use_specular_tint:
        tint = torch.sigmoid(dense(x))
      if mlp.enable_pred_roughness:
        raw_roughness = dense(x)
        roughness = _ACT[mlp.roughness_activation](raw_roughness + mlp.roughness_bias)

      if mlp.bottleneck_width > 0:
        bottleneck = dense(x)
        if (bottleneck_noise is not None) and (mlp.bottleneck_noise > 0):
          bottleneck = bottleneck + mlp.bottleneck_noise * bottleneck_noise
        x = [bottleneck]
      else:
        x = []

      dir_enc_fn = mlp.dir_enc_fn()
      if mlp.use_reflections:
        refdirs = ref_utils.reflect(-viewdirs[..., None, :], normals_to_use)
        dir_enc = dir_enc_fn(refdirs, roughness)
      else:
        dir_enc = dir_enc_fn(viewdirs, roughness)
        dir_enc = dir_enc[..., None, :].expand(bottleneck.shape[:-1] + (dir_enc.shape[-1],))
      x.append(dir_enc)

      if mlp.use_n_dot_v:
        dotprod = torch.sum(normals_to_use * viewdirs[..., None, :], dim=-1, keepdim=True)
        x.append(dotprod)

      if glo_vec is not None:
        x.append(glo_vec[..., None, :].expand(bottleneck.shape[:-1] + glo_vec.shape[-1:]))

      x = torch.cat(x, dim=-1)
      inputs = x
      for i in range(mlp.net_depth_viewdirs):
        x = act(dense(x))
        if i % mlp.skip_layer_dir == 0 and i > 0:
          x = torch.cat([x, inputs], dim=-1)

    rgb = _ACT[mlp.rgb_activation](mlp.rgb_premultiplier * dense(x) + mlp.rgb_bias)

    if mlp.use_diffuse_color:
      diffuse_linear = torch.sigmoid(raw_rgb_diffuse - _pm.log(3.0))
      if mlp.use_specular_tint:
        specular_linear = tint * rgb
      else:
        specular_linear = 0.5 * rgb
      rgb = torch.clamp(image.linear_to_srgb(specular_linear + diffuse_linear), 0.0, 1.0)

    rgb = rgb * (1 + 2 * mlp.rgb_padding) - mlp.rgb_padding

  return dict(density=density, rgb=rgb, raw_grad_density=raw_grad_density,
              grad_pred=grad_pred, normals=normals, normals_pred=normals_pred,
              roughness=roughness)


# ---------------------------------------------------------------------------
# Model.__call__ (models.py:75-312).


def model_apply(model: Model, nerf_mlp: MLP, prop_mlp: Optional[MLP], params, rays,
                train_frac, compute_extras, zero_glo=True, noise=None, dense_dtype=None, relu_sides=None):
  """Returns (renderings, ray_history) exactly as models.py:312.

  `rays` is any object with the utils.Rays fields (utils.py:44-57) as torch
  tensors.  `noise` None == rng None.
  """
  dtype = rays.origins.dtype
  nerf_p = params['NerfMLP_0']
  prop_p = nerf_p if model.single_mlp else params['PropMLP_0']
  prop_mlp_ = nerf_mlp if model.single_mlp else prop_mlp

  if model.num_glo_features > 0:
    if not zero_glo:
      glo_vec = params['Embed_0']['embedding'][rays.cam_idx[..., 0]]
    else:
      glo_vec = torch.zeros(rays.origins.shape[:-1] + (model.num_glo_features,), dtype=dtype)
  else:
    glo_vec = None

  _, s_to_t = coord.construct_ray_warps(model.raydist_fn, rays.near, rays.far)

  if model.near_anneal_rate is None:
    init_s_near = 0.
  else:
    init_s_near = float(np.clip(1 - train_frac / model.near_anneal_rate, 0,
                                model.near_anneal_init))
  init_s_far = 1.
  sdist = torch.cat([torch.full_like(rays.near, init_s_near),
                     torch.full_like(rays.far, init_s_far)], dim=-1)
  weights = torch.ones_like(rays.near)
  prod_num_samples = 1

  ray_history = []
  renderings = []
  for i_level in range(model.num_levels):
    is_prop = i_level < (model.num_levels - 1)
    num_samples = model.num_prop_samples if is_prop else model.num_nerf_samples

    dilation = model.dilation_bias + model.dilation_multiplier * (
        init_s_far - init_s_near) / prod_num_samples
    prod_num_samples *= num_samples

    use_dilation = model.dilation_bias > 0 or model.dilation_multiplier > 0
    if i_level > 0 and use_dilation:
      sdist, weights = stepfun.max_dilate_weights(
          sdist, weights, dilation, domain=(init_s_near, init_s_far), renormalize=True)
      sdist = sdist[..., 1:-1]
      weights = weights[..., 1:-1]

    if model.anneal_slope > 0:
      bias = lambda x, s: (s * x) / ((s - 1) * x + 1)
      anneal = bias(train_frac, model.anneal_slope)
    else:
      anneal = 1.

    u_jit = None if noise is None else noise['u_jitter'][i_level]
    if model.stop_level_grad:
      logits_resample = stepfun.resample_logits(sdist, weights, anneal, model.resample_padding)
      sdist = stepfun.sample_intervals(
          u_jit, sdist.detach(), logits_resample.detach(), num_samples,
          single_jitter=model.single_jitter, domain=(init_s_near, init_s_far),
          use_gpu_resampling=model.use_gpu_resampling)
      sdist = sdist.detach()                                    # models.py:200-201
    else:
      # gradients THROUGH the sampling (no BASELINE config; pinned by the golden `blender_sampling_grad`): torch's own log,
      # softmax and cumulative sum in place of the level kernel's association order, autograd through dilation, the CDF
      # and the interval end points of math.sorted_interp
      logits_resample = stepfun.resample_logits(sdist, weights, anneal, model.resample_padding, differentiable=True)
      sdist = stepfun.sample_intervals(
          u_jit, sdist, logits_resample, num_samples,
          single_jitter=model.single_jitter, domain=(init_s_near, init_s_far),
          use_gpu_resampling=model.use_gpu_resampling, differentiable=True)

    tdist = s_to_t(sdist)

    gaussians = render.cast_rays(tdist, rays.origins, rays.directions, rays.radii,
                                 model.ray_shape, diag=False)
    if model.disable_integration:
      gaussians = (gaussians[0], torch.zeros_like(gaussians[1]))

    mlp = prop_mlp_ if is_prop else nerf_mlp
    p = prop_p if is_prop else nerf_p
    dn = bn = None
    if noise is not None:
      dn = noise.get('density_noise', {}).get(i_level)
      bn = noise.get('bottleneck_noise', {}).get(i_level)
    ray_results = mlp_apply(
        mlp, p, gaussians,
        viewdirs=rays.viewdirs if model.use_viewdirs else None,
        imageplane=rays.imageplane,
        glo_vec=None if is_prop else glo_vec,
        exposure=rays.exposure_values,
        density_noise=dn, bottleneck_noise=bn, dense_dtype=dense_dtype,
        relu_sides=None if relu_sides is None else relu_sides[i_level])     # (test hook, see mlp_apply)

    weights = render.compute_alpha_weights(
        ray_results['density'], tdist, rays.directions,
        opaque_background=model.opaque_background)[0]

    lo, hi = model.bg_intensity_range
    if lo == hi:
      bg_rgbs = lo
    elif noise is None:
      bg_rgbs = (lo + hi) / 2
    else:
      bg_rgbs = lo + (hi - lo) * noise['bg_rgbs'][i_level]

    if rays.exposure_idx is not None:
      ray_results['rgb'] = ray_results['rgb'] * rays.exposure_values[..., None, :]
      if model.learned_exposure_scaling:
        exposure_idx = rays.exposure_idx[..., 0]
        mask = exposure_idx > 0
        scaling = 1 + mask[..., None] * params['exposure_scaling_offsets']['embedding'][exposure_idx]
        ray_results['rgb'] = ray_results['rgb'] * scaling[..., None, :]

    rendering = render.volumetric_rendering(
        ray_results['rgb'], weights, tdist, bg_rgbs, rays.far, compute_extras,
        extras={k: v for k, v in ray_results.items()
                if k.startswith('normals') or k in ['roughness']})

    if compute_extras:
      n = model.vis_num_rays
      rendering['ray_sdist'] = sdist.reshape([-1, sdist.shape[-1]])[:n, :]
      rendering['ray_weights'] = weights.reshape([-1, weights.shape[-1]])[:n, :]
      rgb = ray_results['rgb']
      rendering['ray_rgbs'] = (rgb.reshape((-1,) + rgb.shape[-2:]))[:n, :, :]

    renderings.append(rendering)
    ray_results['sdist'] = sdist.clone()
    ray_results['tdist'] = tdist.clone()  # oracle extra (not in the reference dict)
    ray_results['weights'] = weights.clone()
    ray_history.append(ray_results)

  if compute_extras:
    ws = [r['ray_weights'] for r in renderings]
    rgbs = [r['ray_rgbs'] for r in renderings]
    final_rgb = torch.sum(rgbs[-1] * ws[-1][..., None], dim=-2)
    for i in range(len(rgbs) - 1):
      renderings[i]['ray_rgbs'] = final_rgb[:, None, :].expand(rgbs[i].shape)

  return renderings, ray_history
