"""Model / MLP host layer: the reference's Python surface over the HIP kernels.

Mirrors reference internal/models.py: `Model` (models.py:47-312), `MLP` /
`NerfMLP` / `PropMLP` (models.py:341-622) with the same gin-configurable
hyper-parameter names and defaults, `construct_model` (models.py:315-338) and
`render_image` (models.py:625-706).  The bodies do not contain arithmetic: the
level loop enqueues kernels of libmnerf_hip.so (include/mnerf.h) on the current
HIP stream and keeps every intermediate in device memory.

Parameters live in ONE flat fp32 device vector (the buffer RCCL all-reduces and
the fused clip+Adam kernel streams); `variables['params']` is a nested dict of
views into it with flax's names ('NerfMLP_0'/'Dense_3'/'kernel' [in,out]).
"""

import ctypes as C
import dataclasses
import math
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from multinerf_amd import _lib as L
from multinerf_amd import geopoly
from multinerf_amd import gin
from multinerf_amd import ops
from multinerf_amd import utils

F32_EPS = float(np.finfo(np.float32).eps)
bf16 = torch.bfloat16
f32 = torch.float32


def _rup(x, m):
  return (x + m - 1) // m * m


def _in_library(fn):
  """Run a Model method inside the model's library context (`Model.library`: a no-op for the product)."""
  import functools

  @functools.wraps(fn)
  def wrapped(self, *a, **kw):
    with self.library():
      return fn(self, *a, **kw)
  return wrapped


def _rank1_last(plan, D, W, feature_gradient):
  """Whether the last hidden layer's dY = relu'(z) * (g (x) w_density) of a Dense(1)-headed chain (the proposal MLP, reference
  models.py:457-460) can stay unstored: its weight-gradient GEMM then builds it from the factors (`mnr_gemm_tn_args.rank1_*`).  Not
  when another reader needs the matrix (a skip concat's second GEMM, the feature gradient of a one-layer trunk)."""
  concat_last = bool(plan.trunk[D - 1][1])
  k_in = plan.ldF if D == 1 else W
  return bool(_RANK1_LAST and W % 256 == 0 and k_in % 256 == 0 and not concat_last and not (feature_gradient and D == 1))


# Module-level switches of the host orchestration.  Every one of them was a same-box A/B in rounds 2-4 (profiles/HISTORY.md,
# profiles/r4_ab.md) and is settled; they stay as plain constants because the bitwise / parity tests compare the two forms of
# each (tests/test_gpu_chain.py, tests/test_sim_model.py set them on the module).  The environment is not read.
_USE_BITS = True      # 1-bit ReLU masks (False: the dX GEMMs re-read the saved activations)
_USE_CHAIN = True     # fused per-level Dense chain for 128 / 256-wide trunks (False: one GEMM per layer)
_FUSED_IPE = True     # rendering builds the proposal levels' IPE features inside the chain kernel
_HEAD_VCOL = True     # the density head's forward column as a vector next to the N = 256 bottleneck GEMM (panel trunks)
_HEAD_GCOL = True     # the density head's weight gradient as an extra column of the bottleneck's dW GEMM
_PANEL = True         # wide (>= 512) per-layer trunks keep activations, gradients and masks in the panel layout (csrc/gemm_blk.hip)
_MERGE_PROPS = True   # the backward pass of all proposal levels as one pass (they share PropMLP_0 and the sample count)
_RANK1_LAST = True    # the proposal MLP's last dY built inside its weight-gradient GEMM from (g, w_density, mask bits) instead of stored
# A panel-storage trunk layer's dW and dX GEMMs read the same dY matrix: side by side on two streams with half the chip each, the
# dW kernel's M-splits block-cyclic (mnr_gemm_tn_args.m_interleave), so that both walk M from top to bottom and part of the second
# reads of dY is served by the Infinity Cache instead of HBM: 532.4 / 533.8 k -> 538.1 / 538.1 k rays/s on one box, gradients equal
# up to the atomics' order (profiles/r5_ab.md (d); pacing the two launches against each other returned nothing and is gone)
_PAIR_DXDW = True
# The density-gradient normals' tangent network (3 M rows per level) on the fused masked-linear chain: T_l = bits_l * (T_{l-1} W_l)
# forward and G_{l-1} = bits_{l-1} * (G_l W_l^T) backward are both what mlp_chain_bwd_kernel computes (a Dense chain without bias
# whose ReLU is a given 1-bit mask), so every run of trunk layers that does not read the features is ONE launch per direction
# instead of one GEMM per layer: the rows are written once (the weight-gradient GEMMs need them) and never read back in between
_TANGENT_CHAIN = True
_HEAD_K32 = True      # the merged head's dX GEMM into a panel trunk at K = the head's columns rounded to 32 (288 at 360.gin), not 64 (320)


# =============================================================================
# Hyper-parameters (gin surface).


@dataclasses.dataclass
class MLP:
  """A PosEnc MLP (fields/defaults: reference models.py:343-379)."""
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

  def setup_checks(self):
    """models.py:381-385."""
    if self.use_reflections and not (self.enable_pred_normals or not self.disable_density_normals):
      raise ValueError('Normals must be computed for reflection directions.')

  def is_ref(self):
    """The Ref-NeRF head (models.py:512-563,588-599: reflections, IDE, n.v, roughness, diffuse colour, tint; every flag on its
    own, blender_refnerf.gin sets all of them): anything beyond the two normal fields, which have lighter paths of their own."""
    return (not self.disable_rgb) and bool(self.use_reflections or self.use_directional_enc or self.enable_pred_roughness or
                                           self.use_diffuse_color or self.use_specular_tint or self.use_n_dot_v)

  def ref_features(self):
    """MNR_REF_* bits of mnr_ref_head_fwd / _bwd (include/mnerf.h)."""
    return ((ops.REF_PRED_NORMALS if self.enable_pred_normals else 0) |
            (0 if self.disable_density_normals else ops.REF_DENSITY_NORMALS) |
            (ops.REF_REFLECT if self.use_reflections else 0) | (ops.REF_IDE if self.use_directional_enc else 0) |
            (ops.REF_N_DOT_V if self.use_n_dot_v else 0) | (ops.REF_ROUGHNESS if self.enable_pred_roughness else 0))

  def hip_supported(self):
    """Which reference features the HIP path implements in this round (scope)."""
    bad = []
    ref_flags = ('enable_pred_normals', 'use_reflections', 'use_directional_enc', 'enable_pred_roughness',
                 'use_diffuse_color', 'use_specular_tint', 'use_n_dot_v')
    on = [f for f in ref_flags if getattr(self, f)]
    if not self.disable_density_normals:
      on.append('density normals')
    if self.use_directional_enc and not self.enable_pred_roughness:
      # not a HIP-path restriction: the reference's IDE multiplies by the roughness (ref_utils.py:147: exp(-sigma * kappa_inv)
      # with kappa_inv = None is a TypeError there)
      bad.append('use_directional_enc without enable_pred_roughness (undefined in the reference: ref_utils.py:147)')
    if self.use_directional_enc and not self.use_reflections and not self.disable_rgb:
      # the IDE of the per-ray view direction is [B, 2T], the roughness it is attenuated by [B, n, 1] (ref_utils.py:154):
      # the reference's own broadcast fails (tests/golden/make_golden_models.py)
      bad.append('use_directional_enc without use_reflections (undefined in the reference: ref_utils.py:154)')
    if self.use_n_dot_v and not self.enable_pred_normals and self.disable_density_normals:
      bad.append('use_n_dot_v without normals (undefined in the reference: models.py:560-563)')
    # Every other set of the Ref-NeRF flags runs: the normal fields on their own or together (a Dense(3) head, models.py:494-503;
    # the tangent network, what configs/llff_raw.gin asks for with the orientation loss), and the Ref-NeRF head with any of its
    # parts switched off (mnr_ref_head_fwd's feature bits; tests/golden/models.npz holds the reference's outputs for eleven such sets).
    if on and self.disable_rgb:
      # (the normal fields and the Ref-NeRF head live in the merged head of an MLP with a colour branch)
      bad.append('normals ' + str(on) + ' on a density-only MLP (disable_rgb)')
    if self.enable_pred_roughness and self.roughness_activation != 'softplus':
      bad.append('roughness_activation != softplus')
    if self.is_ref() and not self.use_directional_enc and self.deg_view > 11:
      bad.append('deg_view > 11 in the positional encoding of a per-sample direction')
    if self.net_activation not in ('relu', 'softplus', 'silu'):   # what the reference registers (configs.py:29-31)
      bad.append(f'net_activation={self.net_activation}')
    if self.warp_fn not in (None, 'contract'):
      bad.append(f'warp_fn={self.warp_fn}')
    if self.net_width <= 0:
      bad.append('net_width must be positive')
    # (a net_width that is not a multiple of 128, e.g. configs/debug.gin's PropMLP.net_width = 64, runs on kernels of the next
    # multiple of 128 with structurally zero padding: Model.build / Model._to_exec)
    # (bottleneck_width / net_width_viewdirs off the kernels' tile, models.py:345-347: zero-padded execution layout as well)
    if not self.disable_rgb and self.bottleneck_width <= 0:
      bad.append('bottleneck_width must be positive')
    if not self.disable_rgb and self.net_width_viewdirs <= 0:
      bad.append('net_width_viewdirs must be positive')
    if self.num_rgb_channels != 3:
      bad.append('num_rgb_channels != 3')
    return bad


@gin.configurable
@dataclasses.dataclass
class NerfMLP(MLP):
  pass


@gin.configurable
@dataclasses.dataclass
class PropMLP(MLP):
  pass


# =============================================================================
# Static plan of one MLP: layer shapes, parameter offsets, packed bf16 operand layout.


@dataclasses.dataclass
class DenseSpec:
  name: str            # 'Dense_k'
  fan_in: int
  fan_out: int
  kernel_off: int = 0  # offsets into the flat fp32 parameter vector
  bias_off: int = 0
  in_segs: Tuple[int, ...] = ()   # widths of the concatenated blocks its input rows are made of (trunk | features | ...): what
                                  # Model.build maps between the callers' layout and a zero-padded execution layout


class MLPPlan:
  """Everything about one MLP that does not depend on the batch."""

  def __init__(self, hp: MLP, module_name: str, use_viewdirs: bool, num_glo_features: int, param_base: int):
    hp.setup_checks()
    self.hp = hp
    self.module_name = module_name
    self.basis = geopoly.generate_basis(hp.basis_shape, hp.basis_subdivisions)   # [K,3]
    self.K = self.basis.shape[0]
    self.L = hp.max_deg_point - hp.min_deg_point
    self.F = 2 * self.K * self.L
    self.ldF = _rup(self.F, 128)
    self.W = hp.net_width
    self.has_rgb = (not hp.disable_rgb)
    self.use_viewdirs = use_viewdirs and self.has_rgb
    self.dense: List[DenseSpec] = []
    k = 0

    def add(segs, fo):
      """A Dense whose input is the concatenation of blocks of the widths `segs`."""
      nonlocal k
      segs = tuple(int(w) for w in segs if w > 0)
      d = DenseSpec(f'Dense_{k}', sum(segs), fo, in_segs=segs)
      k += 1
      self.dense.append(d)
      return d

    # trunk (models.py:455-459)
    self.trunk: List[Tuple[DenseSpec, bool]] = []   # (spec, input_is_concat_with_features)
    concat = False
    first = True
    for i in range(hp.net_depth):
      self.trunk.append((add((self.F,) if first else ((self.W, self.F) if concat else (self.W,)), self.W), (not first) and concat))
      first = False
      concat = (i % hp.skip_layer == 0 and i > 0)
    self.x_concat = concat                     # trunk output carries the features too (depth ending on a skip)
    self.x_width = self.W + (self.F if concat else 0)
    x_segs = (self.W, self.F) if concat else (self.W,)
    self.density = add(x_segs, 1)              # models.py:460
    self.ref = hp.is_ref() and self.use_viewdirs           # (without view directions the reference skips the branch: models.py:512)
    self.features = hp.ref_features() if self.ref else 0   # MNR_REF_* bits
    self.pn = hp.enable_pred_normals and not self.ref      # predicted normals without the rest of the Ref-NeRF head
    self.dn = (not hp.disable_density_normals) and not self.ref   # density-gradient normals without it
    self.tangent = not hp.disable_density_normals          # the forward-mode tangent network runs next to the trunk
    self.diffuse_on = self.ref and hp.use_diffuse_color    # colour = tone-mapped tinted specular + diffuse (models.py:588-599)
    self.view: List[Tuple[DenseSpec, bool]] = []
    self.bottleneck = None
    self.rgb = None
    self.gradpred = self.diffuse = self.tint = self.rough = None
    self.glo = 0
    if hp.enable_pred_normals:
      self.gradpred = add(x_segs, 3)                                 # models.py:495
    if self.has_rgb:
      if self.use_viewdirs:
        if hp.use_diffuse_color:
          self.diffuse = add(x_segs, hp.num_rgb_channels)            # models.py:515
        if hp.use_specular_tint:
          self.tint = add(x_segs, 3)                                 # models.py:518
        if hp.enable_pred_roughness:
          self.rough = add(x_segs, 1)                                # models.py:521
        self.bottleneck = add(x_segs, hp.bottleneck_width)           # models.py:527
        if hp.use_directional_enc:
          from multinerf_amd import ref_utils
          self.dir_enc_dim = 2 * len(ref_utils.ide_tables(hp.deg_view)[0])   # IDE: real + imaginary parts
        else:
          self.dir_enc_dim = 3 + 2 * 3 * hp.deg_view                 # coord.pos_enc, append_identity
        self.glo = num_glo_features
        self.glo_col = hp.bottleneck_width + self.dir_enc_dim + (1 if hp.use_n_dot_v else 0)   # models.py:565-568
        self.vi_width = self.glo_col + num_glo_features
        self.ldVI = _rup(self.vi_width, 128)
        vi_segs = (hp.bottleneck_width, self.glo_col - hp.bottleneck_width, num_glo_features)
        WV = hp.net_width_viewdirs
        concat, first = False, True
        for i in range(hp.net_depth_viewdirs):                       # models.py:576-580
          self.view.append((add(vi_segs if first else (((WV,) + vi_segs) if concat else (WV,)), WV), (not first) and concat))
          first = False
          concat = (i % hp.skip_layer_dir == 0 and i > 0)
        self.v_concat = concat
        rgb_segs = ((WV,) + vi_segs) if concat else (WV,)
        if hp.net_depth_viewdirs == 0:
          rgb_segs = vi_segs
      else:
        rgb_segs = x_segs
      self.rgb = add(rgb_segs, hp.num_rgb_channels)                  # models.py:585
      if self.use_viewdirs:
        # merged head: (Dense, first column) -- bottleneck first, then the scalar / 3-vector heads
        bw = hp.bottleneck_width
        self.head_segs = [(self.bottleneck, 0), (self.density, bw)]
        if self.ref:
          # (the same 11 fp32 side columns for every feature set: a head that is switched off is a zero column nobody reads)
          self.head_segs += [(d, c) for d, c in ((self.gradpred, bw + 1), (self.diffuse, bw + 4), (self.tint, bw + 7),
                                                 (self.rough, bw + 10)) if d is not None]
        elif self.pn:
          self.head_segs += [(self.gradpred, bw + 1)]
        self.head_cols = bw + (11 if self.ref else 4 if self.pn else 1)
      else:
        # use_viewdirs = False (models.py:57,226,585): rgb = Dense(3)(trunk output); density and rgb share one 4-column head
        self.head_segs = [(self.density, 0), (self.rgb, 1)]
        self.head_cols = 4
    # flat parameter offsets: kernel then bias, Dense_k in creation order
    off = param_base
    for d in self.dense:
      d.kernel_off = off
      off += d.fan_in * d.fan_out
      d.bias_off = off
      off += d.fan_out
    self.param_begin, self.param_end = param_base, off

  @property
  def num_params(self):
    return self.param_end - self.param_begin


# =============================================================================
# Model.


@gin.configurable
@dataclasses.dataclass
class Model:
  """A mip-NeRF 360 model containing all MLPs (fields/defaults: reference models.py:50-72)."""
  config: Any = None
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
  # Not a reference field.  'bf16' (the product): Dense layers bf16 x bf16 -> fp32 on the MFMA units, activations and their
  # gradients stored in bf16 (the reference's TPU default precision).  'fp32' (DEBUG, parity only): the reference's jax-cpu
  # precision (flax Dense in fp32, models.py:436-437, math.py:21-23): the same host code and the same kernel sources from the
  # fp32-Dense build libmnerf_hip_f32.so (float storage, plain-FMA Dense layers, csrc/dense_f32.inc); per-layer GEMMs only (no
  # fused chain, no panel storage, no vector columns: those are MFMA layouts).  Never benchmarked.
  dense_precision: str = 'bf16'

  # ------------------------------------------------------------------ construction

  def __post_init__(self):
    self.nerf_hp = NerfMLP()
    self.prop_hp = self.nerf_hp if self.single_mlp else PropMLP()
    self._built = False
    if self.dense_precision not in ('bf16', 'fp32'):
      raise ValueError("dense_precision must be 'bf16' or 'fp32'")
    self._f32 = self.dense_precision == 'fp32'

  def _adt_is_f32(self):
    return self._f32

  def library(self):
    """Context in which this model's C-ABI calls run: a no-op for the product, the fp32-Dense debug build for
    dense_precision = 'fp32' (multinerf_amd/_lib.py: dense_f32)."""
    return L.dense_f32(self._f32)

  def hip_supported(self):
    bad = []
    for name, hp in (('NerfMLP', self.nerf_hp), ('PropMLP', self.prop_hp)):
      bad += [f'{name}: {b}' for b in hp.hip_supported()]
    # (stop_level_grad = False next to density-gradient normals, which are a function of the sample positions as well, models.py:
    # 198-201 with :478-492: the tangent rows' VJP with respect to the interval ends is mnr_cast_rays_ipe_tangent_bwd, round 6)
    if not self.use_viewdirs and any(hp.enable_pred_normals or not hp.disable_density_normals or hp.is_ref()
                                     for hp in (self.nerf_hp, self.prop_hp) if not hp.disable_rgb):
      bad.append('normals (the Ref-NeRF head or one of its fields) without view directions')
    if self.ray_shape not in ('cone', 'cylinder'):
      raise ValueError('ray_shape must be \'cone\' or \'cylinder\'')
    if self.num_glo_features > 0 and self.single_mlp:
      # the reference feeds glo_vec to the last level only (models.py:228): one shared MLP cannot take both widths
      raise ValueError('num_glo_features > 0 is incompatible with single_mlp')
    if self.raydist_fn not in L.RAYDIST:
      bad.append(f'raydist_fn={self.raydist_fn}')
    return bad

  def _ray_pad_unit(self):
    unit = 8
    for n in (self.num_prop_samples, self.num_nerf_samples):
      u = 256 // math.gcd(int(n), 256)
      unit = unit * u // math.gcd(unit, u)
    return unit

  def build(self, device='cuda'):
    """Lay out parameters and device-side constant tables (flax `init` without the RNG part)."""
    bad = self.hip_supported()
    if bad:
      raise NotImplementedError('not yet implemented on the HIP path: ' + '; '.join(bad))
    if not self.stop_level_grad and self.resample_padding == 0 and (self.dilation_bias > 0 or self.dilation_multiplier > 0):
      import warnings
      warnings.warn('Model.stop_level_grad = False with resample_padding = 0: where two dilated fence-posts are clipped to the same '
                    'domain end a bin has weight 0 and the reference\'s own autodiff yields NaN gradient elements there (which its '
                    'train_step zeroes element-wise, train_utils.py:326-328); this path differentiates the function that is evaluated '
                    '(that bin is a constant), so the two trajectories differ.', stacklevel=2)
    self.device = torch.device(device)

    def layout(nerf_hp, prop_hp):
      """(plans, modules, glo_off, expo_off, num_params) of the flat parameter vector for these hyper-parameters."""
      nerf = MLPPlan(nerf_hp, 'NerfMLP_0', self.use_viewdirs, self.num_glo_features, 0)
      plans, end = [nerf], nerf.param_end
      if not self.single_mlp:
        prop = MLPPlan(prop_hp, 'PropMLP_0', self.use_viewdirs, 0, nerf.param_end)
        plans.append(prop)
        end = prop.param_end
      modules = [(p.module_name, p.param_begin, p.param_end) for p in plans]
      glo_off = expo_off = None
      if self.num_glo_features > 0:
        # nn.Embed(num_glo_embeddings, num_glo_features) auto-named Embed_0 (models.py:101-106)
        glo_off = end
        end += self.num_glo_embeddings * self.num_glo_features
        modules.append(('Embed_0', glo_off, end))
      if self.learned_exposure_scaling:
        # nn.Embed(num_glo_embeddings, 3, zeros init, name='exposure_scaling_offsets') (models.py:112-121)
        expo_off = end
        end += self.num_glo_embeddings * 3
        modules.append(('exposure_scaling_offsets', expo_off, end))
      return plans, modules, glo_off, expo_off, end

    # The parameter vector the callers see (`num_params`, `modules`, params_tree / flat_from_tree / param_ranges) is flax's,
    # at the widths the configuration names.  The kernels want trunk widths that are multiples of 128 (the GEMM tile): a width
    # such as configs/debug.gin's PropMLP.net_width = 64 runs as the next multiple of 128 on an EXECUTION layout whose extra
    # rows / columns are structurally zero (`_to_exec` scatters the parameters into it before a pass, `true_grads` gathers the
    # gradient back: zero weights and biases give zero activations behind the ReLU and zero gradients; behind another
    # activation the padded units are non-zero but feed zero rows of the next kernel, and their gradients are dropped).
    def pad(hp):
      """The hyper-parameters the kernels run: trunk and view-MLP widths on the 128-column GEMM tile, the bottleneck on its
      64-column K granule (models.py:345-347,526-527,577 take any width)."""
      w = dict(net_width=_rup(hp.net_width, 128))
      if not hp.disable_rgb and self.use_viewdirs:
        w.update(bottleneck_width=_rup(hp.bottleneck_width, 64), net_width_viewdirs=_rup(hp.net_width_viewdirs, 128))
      return hp if all(getattr(hp, k) == v for k, v in w.items()) else dataclasses.replace(hp, **w)

    self._tplans, self.modules, self.glo_off_true, self.expo_off_true, self.num_params = layout(self.nerf_hp, self.prop_hp)
    nerf_x = pad(self.nerf_hp)
    prop_x = nerf_x if self.single_mlp else pad(self.prop_hp)
    self._pad_index = None
    if nerf_x is self.nerf_hp and prop_x is self.prop_hp:
      self._plans, self.glo_off, self.expo_off, self.num_params_exec = self._tplans, self.glo_off_true, self.expo_off_true, self.num_params
    else:
      self._plans, _, self.glo_off, self.expo_off, self.num_params_exec = layout(nerf_x, prop_x)
      idx = torch.empty(self.num_params, dtype=torch.int64)
      for pt, px in zip(self._tplans, self._plans):
        for dt, dx in zip(pt.dense, px.dense):
          # input rows: block by block of the concatenated input (a padded block's extra rows sit behind its own rows, and what
          # follows it moves up); output columns keep their index
          assert len(dt.in_segs) == len(dx.in_segs)
          rows, ot, ox = [], 0, 0
          for wt, wx in zip(dt.in_segs, dx.in_segs):
            assert wx >= wt
            rows.append(ox + torch.arange(wt))
            ot, ox = ot + wt, ox + wx
          rows = torch.cat(rows)
          cols = torch.arange(dt.fan_out)
          idx[dt.kernel_off:dt.kernel_off + dt.fan_in * dt.fan_out] = (dx.kernel_off + rows[:, None] * dx.fan_out + cols[None, :]).reshape(-1)
          idx[dt.bias_off:dt.bias_off + dt.fan_out] = dx.bias_off + cols
      for o_t, o_x, cnt in ((self.glo_off_true, self.glo_off, self.num_glo_embeddings * self.num_glo_features),
                            (self.expo_off_true, self.expo_off, self.num_glo_embeddings * 3)):
        if o_t is not None:
          idx[o_t:o_t + cnt] = o_x + torch.arange(cnt)
      assert idx.unique().numel() == self.num_params
      self._pad_index = idx.to(self.device)
    self.nerf_plan = self._plans[0]
    self.prop_plan = self._plans[0] if self.single_mlp else self._plans[1]
    for p in self._plans:
      p.basis_dev = torch.as_tensor(p.basis, dtype=f32, device=self.device).contiguous()
      self._layout_packed(p)
    self._ws: Dict[Any, torch.Tensor] = {}
    self._built = True
    return self

  # Packed bf16 operands ------------------------------------------------------------
  #
  # For every Dense two bf16 images are kept, both zero-padded to the GEMM tile grid:
  #   'f' (forward):  Bt[N_pad][K_pad] = kernel^T, K segments padded separately ([W | ldF])
  #   'b' (backward): kernel[in_rows_pad][out_pad] as stored by flax (rows = dX outputs)
  # The NeRF head (bottleneck + density) is merged into one operand pair so that x7 is read once.

  def _layout_packed(self, p: MLPPlan):
    descs: List[L.PackDesc] = []
    off = 0
    p.packed = {}

    def alloc(rows, cols):
      nonlocal off
      o = off
      off += rows * cols
      return o

    def seg_cols(concat, base_w, base_ld, feat=True):
      """Column layout of an input that is [x (base_w, padded to base_ld) | features (F -> ldF)]."""
      return base_ld + (p.ldF if concat else 0)

    def pack_layer(key, d: DenseSpec, in_segments, n_pad):
      """in_segments: list of (first kernel row, rows, first padded K column, padded width): where each
      block of kernel rows lands among the padded K columns of the operand."""
      kpad = max(s[2] + s[3] for s in in_segments)
      fo = alloc(n_pad, kpad)                 # forward operand [n_pad][kpad]
      for (r0, rows, c0, _) in in_segments:
        descs.append(L.PackDesc(d.kernel_off + r0 * d.fan_out, rows, d.fan_out, fo, kpad, 0, c0, 1))
      p.packed[key] = dict(f_off=fo, f_ld=kpad, n_pad=n_pad, kpad=kpad)
      return p.packed[key]

    # trunk
    for i, (d, concat) in enumerate(p.trunk):
      if i == 0:
        segs = [(0, p.F, 0, p.ldF)]
      elif concat:
        segs = [(0, p.W, 0, p.W), (p.W, p.F, p.W, p.ldF)]
      else:
        segs = [(0, p.W, 0, p.W)]
      e = pack_layer(('trunk', i), d, segs, _rup(d.fan_out, 128))
      if i > 0:
        # backward operand for dX_{i-1}: kernel rows [0, W) as [W][out_pad]
        bo = alloc(_rup(p.W, 128), _rup(d.fan_out, 64))
        descs.append(L.PackDesc(d.kernel_off, p.W, d.fan_out, bo, _rup(d.fan_out, 64), 0, 0, 0))
        e.update(b_off=bo, b_ld=_rup(d.fan_out, 64))
      if not self.stop_level_grad and (i == 0 or concat):
        # stop_level_grad = False: the FEATURES receive gradient (models.py:198-201): backward operand of the feature rows,
        # kernel rows [0, F) of layer 0 / [W, W + F) of a skip layer, as [ldF][out_pad]
        bf = alloc(p.ldF, _rup(d.fan_out, 64))
        descs.append(L.PackDesc(d.kernel_off + (0 if i == 0 else p.W) * d.fan_out, p.F, d.fan_out, bf, _rup(d.fan_out, 64), 0, 0, 0))
        e.update(bf_off=bf, bf_ld=_rup(d.fan_out, 64))
    assert not p.x_concat, 'net_depth ending on a skip layer is not supported on the HIP path'
    # heads
    if p.has_rgb and p.use_viewdirs:
      bw = p.hp.bottleneck_width
      nh = _rup(p.head_cols, 128)                               # backward (dHB) width
      nh_f = _rup(p.head_cols, 256)                             # forward rows: a multiple of the 256-wide GEMM tile
      fo = alloc(nh_f, p.W)                                     # merged head fwd: row c0.. of each Dense
      bo = alloc(_rup(p.W, 128), nh)                            # merged head bwd: [W][nh]
      for (d, c0) in p.head_segs:
        descs.append(L.PackDesc(d.kernel_off, p.W, d.fan_out, fo, p.W, c0, 0, 1))
        descs.append(L.PackDesc(d.kernel_off, p.W, d.fan_out, bo, nh, 0, c0, 0))
      p.packed['head'] = dict(f_off=fo, f_ld=p.W, n_pad=nh_f, nb_pad=nh, b_off=bo, b_ld=nh)
      if p.tangent:
        pack_layer('density', p.density, [(0, p.W, 0, p.W)], 128)   # tangent rows only need the density column
      WV = p.hp.net_width_viewdirs
      for i, (d, concat) in enumerate(p.view):
        if i == 0:
          segs = [(0, p.vi_width, 0, p.ldVI)]
        elif concat:
          segs = [(0, WV, 0, WV), (WV, p.vi_width, WV, p.ldVI)]
        else:
          segs = [(0, WV, 0, WV)]
        e = pack_layer(('view', i), d, segs, _rup(d.fan_out, 128))
        # dX target: the bottleneck columns (all view-input columns for Ref-NeRF, whose IDE / n.v
        # columns carry gradient) for layer 0, the previous hidden layer otherwise.
        full = p.ref or p.glo > 0               # gradient also needed w.r.t. the non-bottleneck view-input columns
        rows = (p.vi_width if full else bw) if i == 0 else WV
        bo = alloc(_rup(rows, 128), _rup(d.fan_out, 64))
        descs.append(L.PackDesc(d.kernel_off, rows, d.fan_out, bo, _rup(d.fan_out, 64), 0, 0, 0))
        e.update(b_off=bo, b_ld=_rup(d.fan_out, 64), b_rows=_rup(rows, 128))
        if concat:
          # second backward image: the skip-concat rows [WV, WV+vi_width) -> gradient w.r.t. the view input
          b2 = alloc(p.ldVI, _rup(d.fan_out, 64))
          descs.append(L.PackDesc(d.kernel_off + WV * d.fan_out, p.vi_width, d.fan_out, b2, _rup(d.fan_out, 64), 0, 0, 0))
          e.update(b2_off=b2)
      assert not getattr(p, 'v_concat', False), 'view MLP ending on a skip layer is not supported on the HIP path'
      pack_layer('rgb', p.rgb, [(0, p.rgb.fan_in, 0, _rup(p.rgb.fan_in, 64))], 128)
    elif p.has_rgb:
      # use_viewdirs = False: [density | rgb] as one 4-column forward operand (rows 0..3 of a 128-row tile)
      fo = alloc(128, p.W)
      for (d, c0) in p.head_segs:
        descs.append(L.PackDesc(d.kernel_off, p.W, d.fan_out, fo, p.W, c0, 0, 1))
      p.packed['head4'] = dict(f_off=fo, f_ld=p.W, n_pad=128)
    else:
      pack_layer('density', p.density, [(0, p.W, 0, p.W)], 128)
    # Inference image of trunk layer 0 for the chain with the in-kernel IPE producer (mnr_mlp_chain_fwd_ipe): kernel^T with
    # its K columns group-major (four degrees per group of MNR_CHAIN_IPE_GROUP_COLS columns, include/mnerf.h); packed by a
    # table of its own, only when a forward pass without a backward pass asks for it.
    p.pack_descs_ipe = []
    if (not p.has_rgb) and p.L % 4 == 0 and p.K <= 24 and not any(c for _, c in p.trunk):
      G = L.CHAIN_IPE_GROUP_COLS
      d0 = p.trunk[0][0]
      n_pad = _rup(d0.fan_out, 128)
      ld = (p.L // 4) * G
      fo = alloc(n_pad, ld)
      for l in range(p.L):
        for sc in range(2):
          col = (l // 4) * G + (l % 4) * 2 * p.K + sc * p.K
          p.pack_descs_ipe.append(L.PackDesc(d0.kernel_off + (sc * p.K * p.L + l * p.K) * d0.fan_out, p.K, d0.fan_out, fo, ld, 0, col, 1))
      p.packed['trunk0_ipe'] = dict(f_off=fo, f_ld=ld, n_pad=n_pad)
      arr = (L.PackDesc * len(p.pack_descs_ipe))(*p.pack_descs_ipe)
      p.pack_descs_ipe_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
    p.packed_elems = off
    p.pack_descs = descs
    arr = (L.PackDesc * len(descs))(*descs)
    p.pack_descs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
    p.pack_max_elems = max(d.rows_in * d.cols_out for d in descs)
    p.wbf = torch.zeros(off, dtype=f32 if self._f32 else bf16, device=self.device)     # padding stays zero forever
    if p.has_rgb:
      p.head_bias = torch.zeros(_rup(p.head_cols, 128), dtype=f32, device=self.device)
    if p.ref:
      p.ide = ops.IdeTablesDev(p.hp.deg_view, self.device) if p.hp.use_directional_enc else None

  @_in_library
  def pack_weights(self, flat_params, ipe=False):
    """fp32 master parameters -> bf16 GEMM operands (one launch per MLP; `ipe`: also the group-major layer-0 image of the
    chain with the in-kernel IPE producer)."""
    for p in self._plans:
      ops.pack_weights(flat_params, p.pack_descs_dev, len(p.pack_descs), p.pack_max_elems, p.wbf)
      if ipe and p.pack_descs_ipe:
        ops.pack_weights(flat_params, p.pack_descs_ipe_dev, len(p.pack_descs_ipe), p.K * p.trunk[0][0].fan_out, p.wbf)
      if p.has_rgb:
        for (d, c0) in p.head_segs:
          p.head_bias[c0:c0 + d.fan_out].copy_(flat_params[d.bias_off:d.bias_off + d.fan_out])

  # Parameters ------------------------------------------------------------------------

  def init_flat_params(self, seed=0):
    """Random init with flax semantics (he/glorot *_uniform kernels, zero biases; models.py:436-437)."""
    assert self._built
    gen = torch.Generator().manual_seed(seed)
    flat = torch.zeros(self.num_params, dtype=f32)
    for p in self._tplans:
      for d in p.dense:
        kind = p.hp.weight_init
        if kind == 'he_uniform':
          lim = math.sqrt(6.0 / d.fan_in)
        elif kind == 'glorot_uniform':
          lim = math.sqrt(6.0 / (d.fan_in + d.fan_out))
        else:
          raise NotImplementedError(f'weight_init {kind}')
        w = (torch.rand((d.fan_in, d.fan_out), generator=gen, dtype=torch.float64) * 2 - 1) * lim
        flat[d.kernel_off:d.kernel_off + d.fan_in * d.fan_out] = w.reshape(-1).float()
    if self.glo_off_true is not None:
      # flax nn.Embed default init: variance_scaling(1.0, 'fan_in', 'normal', out_axis=0) -> std 1/sqrt(features)
      G = self.num_glo_features
      e = torch.randn((self.num_glo_embeddings, G), generator=gen, dtype=torch.float64) / math.sqrt(G)
      flat[self.glo_off_true:self.glo_off_true + e.numel()] = e.reshape(-1).float()
    return flat.to(self.device)

  def params_tree(self, flat):
    """Nested dict of views into `flat` with flax's names (train.py:194-195 layout)."""
    tree = {}
    for p in self._tplans:
      m = {}
      for d in p.dense:
        m[d.name] = {
            'kernel': flat[d.kernel_off:d.kernel_off + d.fan_in * d.fan_out].view(d.fan_in, d.fan_out),
            'bias': flat[d.bias_off:d.bias_off + d.fan_out],
        }
      tree[p.module_name] = m
    if self.glo_off_true is not None:
      n = self.num_glo_embeddings * self.num_glo_features
      tree['Embed_0'] = {'embedding': flat[self.glo_off_true:self.glo_off_true + n].view(-1, self.num_glo_features)}
    if self.expo_off_true is not None:
      n = self.num_glo_embeddings * 3
      tree['exposure_scaling_offsets'] = {'embedding': flat[self.expo_off_true:self.expo_off_true + n].view(-1, 3)}
    return tree

  def param_ranges(self):
    """{key: (begin, end)} of the flat parameter vector for every key `train_utils.summarize_tree` (train_utils.py:60-68,
    max_depth 3) gives the reference's `params` tree: 'NerfMLP_0', 'NerfMLP_0/Dense_3', 'NerfMLP_0/Dense_3/kernel', ...,
    'Embed_0', 'Embed_0/embedding'.  A Dense's kernel and bias are adjacent in the flat vector, so every key is one range."""
    out = {}
    for name, b, e in self.modules:
      out[name] = (b, e)
    for p in self._tplans:
      for d in p.dense:
        kb, ke = d.kernel_off, d.kernel_off + d.fan_in * d.fan_out
        bb, be = d.bias_off, d.bias_off + d.fan_out
        assert bb == ke, 'kernel and bias of a Dense are adjacent (Model._plan_params)'
        out[f'{p.module_name}/{d.name}'] = (kb, be)
        out[f'{p.module_name}/{d.name}/kernel'] = (kb, ke)
        out[f'{p.module_name}/{d.name}/bias'] = (bb, be)
    if self.glo_off_true is not None:
      out['Embed_0/embedding'] = out['Embed_0']
    if self.expo_off_true is not None:
      out['exposure_scaling_offsets/embedding'] = out['exposure_scaling_offsets']
    return out

  def flat_from_tree(self, tree, device=None):
    """Inverse of params_tree for externally supplied parameters (e.g. the oracle's)."""
    flat = torch.zeros(self.num_params, dtype=f32)
    for p in self._tplans:
      for d in p.dense:
        flat[d.kernel_off:d.kernel_off + d.fan_in * d.fan_out] = tree[p.module_name][d.name]['kernel'].detach().reshape(-1).float().cpu()
        flat[d.bias_off:d.bias_off + d.fan_out] = tree[p.module_name][d.name]['bias'].detach().float().cpu()
    if self.glo_off_true is not None:
      n = self.num_glo_embeddings * self.num_glo_features
      flat[self.glo_off_true:self.glo_off_true + n] = tree['Embed_0']['embedding'].detach().reshape(-1).float().cpu()
    if self.expo_off_true is not None:
      n = self.num_glo_embeddings * 3
      flat[self.expo_off_true:self.expo_off_true + n] = tree['exposure_scaling_offsets']['embedding'].detach().reshape(-1).float().cpu()
    return flat.to(device or self.device)

  def _to_exec(self, flat):
    """The parameter vector the kernels read: `flat` itself, or (padded trunk widths, `build`) its scatter into the
    zero-padded execution layout."""
    if self._pad_index is None:
      return flat
    if flat.numel() == self.num_params_exec and self.num_params_exec != self.num_params:
      return flat                                        # (already in the execution layout)
    buf = self._buf(('exec', 'flat'), (self.num_params_exec,), f32, zero=True)
    buf.index_copy_(0, self._pad_index, flat)
    return buf

  def true_grads(self, grads_exec):
    """Gradient in the callers' layout from the execution layout's (the padded entries' gradients are dropped)."""
    return grads_exec if self._pad_index is None else grads_exec.index_select(0, self._pad_index)

  # Workspace -------------------------------------------------------------------------

  def _buf(self, key, shape, dtype, zero=False):
    if dtype is bf16 and self._f32:
      dtype = f32                                   # (dense_precision = 'fp32': the Dense layers' matrices are stored in fp32)
    k = (key, tuple(shape), dtype)
    t = self._ws.get(k)
    if t is None:
      t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
      self._ws[k] = t
    return t

  def _const(self, key, sig, make):
    """A read-only device tensor kept across calls in ONE slot per `key` and rebuilt when its signature `sig` (the values
    it was made from) changes: e.g. the initial sample distances under near-plane annealing change every step, and are then
    simply rebuilt.  Nothing may write into what this returns."""
    k = ('const', key)
    hit = self._ws.get(k)
    if hit is None or hit[0] != sig:
      hit = (sig, make())
      self._ws[k] = hit
    return hit[1]

  def workspace_bytes(self):
    """Bytes of the per-level / backward workspace allocated so far (features, activations, masks, gradients: everything
    `_buf` hands out; bench.py reports it).  The fused chain's backward keeps one dY matrix per layer (the weight-gradient
    GEMMs read them after the one dX launch) where the per-layer path ping-pongs two: depth * M * W * 2 bytes per level,
    2 GiB instead of 1 GiB per proposal level of 360.gin at 16384 rays, 8 GiB for the 8 x 256 trunk of llff_raw."""
    return sum(t.numel() * t.element_size() for t in self._ws.values() if torch.is_tensor(t))

  def _lvl_buf(self, tag, name, rows, cols, dtype, group=None):
    """A per-level [rows, *cols] buffer.  group = (i, L): rows [i*rows, (i+1)*rows) of ONE buffer the L proposal levels
    share (training), so that their backward pass can run over all L*rows rows at once (`backward_prop_levels`)."""
    if group is None:
      return self._buf((tag, name), (rows,) + tuple(cols), dtype)
    i, Lg = group
    whole = self._buf((('lvl', 'props', Lg), name), (Lg * rows,) + tuple(cols), dtype)
    return whole[i * rows:(i + 1) * rows]

  def _props_group(self, keep_for_backward):
    """Number of proposal levels whose training buffers are grouped (0 = not grouped): more than one proposal level on
    the fused chain with the shared density-only PropMLP_0."""
    Lg = self.num_levels - 1
    if not (keep_for_backward and _MERGE_PROPS and Lg >= 2 and not self.single_mlp and self.stop_level_grad):
      return 0                                         # (stop_level_grad = False: the levels' backward passes run one after the other)
    plan = self.prop_plan
    return Lg if (self._chain_ok(plan) and not plan.has_rgb) else 0

  # ------------------------------------------------------------------ forward

  def apply(self, variables, rng, rays, train_frac, compute_extras, zero_glo=True, **kw):
    """flax-style entry: model.apply(variables, rng, rays, train_frac=..., compute_extras=...)."""
    flat = variables['flat'] if isinstance(variables, dict) and 'flat' in variables else variables
    return self._forward(flat, rng, rays, train_frac, compute_extras, zero_glo, **kw)

  def __call__(self, rng, rays, train_frac, compute_extras, zero_glo=True, **kw):
    """models.py:75-312 on the bound variables (set by construct_model / bind)."""
    return self._forward(self._bound_flat, rng, rays, train_frac, compute_extras, zero_glo, **kw)

  def bind(self, variables):
    self._bound_flat = variables['flat'] if isinstance(variables, dict) else variables
    return self

  def _level_plan(self):
    out = []
    prod = 1
    for i in range(self.num_levels):
      is_prop = i < self.num_levels - 1
      n = self.num_prop_samples if is_prop else self.num_nerf_samples
      out.append((i, is_prop, n, prod))
      prod *= n
    return out

  @_in_library
  def _forward(self, flat, rng, rays, train_frac, compute_extras, zero_glo=True, noise=None,
               keep_for_backward=False, repack=True):
    """The level loop of models.py:147-297.  `noise` (parity tests) overrides `rng`:
    noise['u_jitter'][level] uniform [0,1) of shape [B,1] / [B,n]."""
    if not self._built:
      self.build(flat.device)
    flat = self._flat_exec = self._to_exec(flat)
    fused_ipe = _FUSED_IPE and not keep_for_backward
    if repack:
      self.pack_weights(flat, ipe=fused_ipe)
    dev = self.device
    lead = rays.origins.shape[:-1]
    flat_rays = rays.map(lambda r: r.reshape(-1, r.shape[-1]).contiguous())
    B0 = flat_rays.origins.shape[0]
    # Pad the rays so that every level's row count B * n is a multiple of 256 (GEMM tiles): a multiple of 8 rays for sample
    # counts that are multiples of 32 (every BASELINE config), of 256 / gcd(n, 256) in general.
    Bp = _rup(B0, self._ray_pad_unit())
    if Bp != B0:
      pad = Bp - B0
      flat_rays = flat_rays.map(lambda r: torch.cat([r, r[-1:].expand(pad, r.shape[-1])], 0).contiguous())
    R = flat_rays
    expo = None
    if R.exposure_idx is not None:
      # models.py:257-267: rgb *= exposure_values; rgb *= 1 + [idx > 0] * exposure_scaling_offsets[idx]
      ev = R.exposure_values.reshape(-1).contiguous().float()
      eidx = R.exposure_idx.reshape(-1).to(torch.int32).contiguous()
      offs = None
      if self.learned_exposure_scaling:
        offs = flat[self.expo_off:self.expo_off + self.num_glo_embeddings * 3]
      expo = ops.exposure_scale(ev, eidx, offs)
    near = R.near.reshape(-1).contiguous()
    far = R.far.reshape(-1).contiguous()
    radii = R.radii.reshape(-1).contiguous()
    # GLO (models.py:101-110): the NeRF level looks Embed_0 up by cam_idx; zero_glo feeds zeros instead.
    self._glo_cam = None
    if self.num_glo_features > 0 and not zero_glo:
      self._glo_cam = R.cam_idx.reshape(-1).to(torch.int32).contiguous()

    init_s_near = 0. if self.near_anneal_rate is None else float(
        np.clip(1 - train_frac / self.near_anneal_rate, 0, self.near_anneal_init))
    init_s_far = 1.
    # Constant inputs of the level loop live on the device across calls (`_const`): built per call they are host -> device
    # copies from pageable memory, and torch synchronises the stream after each of those, i.e. the host stops running ahead of
    # the GPU four times per step and every level starts with the GPU waiting for the next launches.
    sdist = self._const('sdist0', (Bp, init_s_near, init_s_far),
                        lambda: torch.tensor([init_s_near, init_s_far], dtype=f32, device=dev).repeat(Bp, 1))
    weights = self._const('weights0', (Bp,), lambda: torch.ones((Bp, 1), dtype=f32, device=dev))

    randomized = (rng is not None) or (noise is not None)
    gen = None
    if rng is not None and noise is None:
      if isinstance(rng, torch.Generator):
        gen = rng
      elif keep_for_backward:
        # a bare seed would reproduce the same jitter / noise / background on every training step (the reference splits
        # its key each step, train_utils.py:263): training takes a torch.Generator, which advances
        raise TypeError('training needs rng to be a torch.Generator (a reused integer seed repeats the same randomness)')
      else:
        gen = torch.Generator(device=dev).manual_seed(int(rng))

    renderings, ray_history, saved = [], [], []
    n_group = self._props_group(keep_for_backward)
    for (i_level, is_prop, n, prod_prev) in self._level_plan():
      plan = self.prop_plan if is_prop else self.nerf_plan
      hp = plan.hp
      dilation = self.dilation_bias + self.dilation_multiplier * (init_s_far - init_s_near) / prod_prev
      use_dilation = (self.dilation_bias > 0 or self.dilation_multiplier > 0) and i_level > 0
      if self.anneal_slope > 0:
        s = self.anneal_slope
        anneal = (s * train_frac) / ((s - 1) * train_frac + 1)      # Schlick bias, models.py:176
      else:
        anneal = 1.

      # --- sampling (stepfun.py:191-209 for u; the rest on device)
      eps = F32_EPS
      jitter = None
      max_jitter = 0.0
      if randomized:
        u_max = eps + (1 - eps) / n
        max_jitter = (1 - u_max) / (n - 1) - eps
        u_base = self._const(('u_rand', i_level), (n, u_max), lambda: torch.linspace(0, 1 - u_max, n, dtype=f32).to(dev))
        d = 1 if self.single_jitter else n
        if noise is not None:
          jitter = noise['u_jitter'][i_level].to(dev).reshape(-1, d)
          if jitter.shape[0] != Bp:
            jitter = torch.cat([jitter, jitter[-1:].expand(Bp - jitter.shape[0], d)], 0)
          jitter = jitter.contiguous().float()
        else:
          jitter = torch.rand((Bp, d), generator=gen, device=dev, dtype=f32)
      else:
        pad = 1 / (2 * n)
        u_base = self._const(('u_det', i_level), (n, pad, eps), lambda: torch.linspace(pad, 1. - pad - eps, n, dtype=f32).to(dev))
      rs_kw = dict(n_samples=n, use_dilation=use_dilation, dilation=dilation, domain=(init_s_near, init_s_far), anneal=anneal,
                   resample_padding=self.resample_padding, single_jitter=self.single_jitter, max_jitter=max_jitter,
                   raydist_fn=self.raydist_fn)
      rs_in = (sdist, weights, u_base, jitter)           # (kept for mnr_resample_level_bwd when stop_level_grad is off)
      sdist, tdist = ops.resample_level(sdist, weights, u_base, jitter, near, far, **rs_kw)

      # --- featurise + MLP
      M = Bp * n
      tag = ('lvl', i_level) if keep_for_backward else ('lvl', 'shared', is_prop)
      group = (i_level, n_group) if (is_prop and n_group) else None
      ipe_in_chain = fused_ipe and self._ipe_chain_ok(plan)
      feat = None
      if not ipe_in_chain:
        feat = self._lvl_buf(tag, 'feat', M, (plan.ldF,), bf16, group)
        ops.cast_rays_ipe(tdist, R.origins, R.directions, radii, plan.basis_dev, ray_shape=self.ray_shape,
                          warp_contract=(hp.warp_fn == 'contract'), min_deg=hp.min_deg_point,
                          max_deg=hp.max_deg_point, ld_feat=plan.ldF, disable_integration=self.disable_integration,
                          out=feat)
      bnoise = None
      if randomized and plan.has_rgb and plan.use_viewdirs and hp.bottleneck_noise > 0:      # models.py:530-533
        bw_ = hp.bottleneck_width                                  # (the execution layout's; bw_t: the configuration's)
        bw_t = self._tplans[self._plans.index(plan)].hp.bottleneck_width
        if noise is not None and 'bottleneck_noise' in noise:
          bnoise = noise['bottleneck_noise'][i_level].to(dev).reshape(-1, n, bw_t).float()
          if bnoise.shape[0] != Bp:
            bnoise = torch.cat([bnoise, bnoise[-1:].expand(Bp - bnoise.shape[0], n, bw_t)], 0)
          bnoise = bnoise.reshape(M, bw_t)
        else:
          bnoise = torch.randn((M, bw_t), generator=gen, device=dev, dtype=f32)
        if bw_ != bw_t:                                             # padded bottleneck columns (they feed zero kernel rows): no noise
          bnoise = torch.cat([bnoise, bnoise.new_zeros((M, bw_ - bw_t))], 1)
        bnoise = bnoise.contiguous()
      if ipe_in_chain:
        mlp_out = self._chain_forward_ipe(plan, flat, tdist, R, radii, M, tag)
      else:
        mlp_out = self._mlp_forward(plan, flat, feat, M, n, R, tag, keep_for_backward, tdist=tdist, bnoise=bnoise,
                                    group=group)

      # --- density noise (models.py:462-464), background colour (:241-254)
      dnoise = None
      if randomized and hp.density_noise > 0:
        if noise is not None and 'density_noise' in noise:
          dnoise = noise['density_noise'][i_level].to(dev).float()
          dnoise = dnoise.reshape(1, 1).expand(Bp, n) if dnoise.numel() == 1 else dnoise.reshape(-1, n)
          if dnoise.shape[0] != Bp:
            dnoise = torch.cat([dnoise, dnoise[-1:].expand(Bp - dnoise.shape[0], n)], 0)
          dnoise = dnoise.contiguous()
        else:
          # with density-gradient normals the reference draws inside vmap(value_and_grad(predict_density)) (models.py:462-464 under
          # :478-481) from a key that is closed over, not mapped: every sample of the level gets the SAME value
          # (tests/golden/make_golden_models.py, case llff_raw_dn)
          dnoise = (torch.randn((1, 1), generator=gen, device=dev, dtype=f32).expand(Bp, n).contiguous() if plan.tangent
                    else torch.randn((Bp, n), generator=gen, device=dev, dtype=f32))
      lo, hi = self.bg_intensity_range
      bg = None
      if lo == hi:
        bg_mode, bg_value = 0, lo
      elif not randomized:
        bg_mode, bg_value = 0, (lo + hi) / 2
      else:
        bg_mode, bg_value = 1, 0.0
        if noise is not None and 'bg_rgbs' in noise:
          u = noise['bg_rgbs'][i_level].to(dev).reshape(-1, 3).float()
          if u.shape[0] != Bp:
            u = torch.cat([u, u[-1:].expand(Bp - u.shape[0], 3)], 0)
        else:
          u = torch.rand((Bp, 3), generator=gen, device=dev, dtype=f32)
        bg = (lo + (hi - lo) * u).contiguous()
      ccfg = ops.composite_cfg(n, opaque_background=self.opaque_background, density_act=hp.density_activation,
                               density_bias=hp.density_bias, density_noise_std=hp.density_noise if dnoise is not None else 0.0,
                               has_rgb=plan.has_rgb, rgb_act='identity' if plan.diffuse_on else hp.rgb_activation,
                               rgb_premultiplier=1.0 if plan.diffuse_on else hp.rgb_premultiplier,
                               rgb_bias=0.0 if plan.diffuse_on else hp.rgb_bias,
                               rgb_padding=0.0 if plan.diffuse_on else hp.rgb_padding, bg_mode=bg_mode, bg_value=bg_value)
      raw_density = mlp_out['raw_density'].view(Bp, n)
      raw_rgb = mlp_out['raw_rgb'].view(Bp, n, 3) if plan.has_rgb else None
      density, rgb, weights, rgb_out, acc = ops.composite_fwd(
          ccfg, raw_density, tdist, R.directions, raw_rgb=raw_rgb, density_noise=dnoise, bg=bg,
          exposure_scale=expo if plan.has_rgb else None)

      rendering = {'rgb': rgb_out[:B0].reshape(lead + (3,))}
      if compute_extras:
        rendering['acc'] = acc[:B0].reshape(lead)
        ex = ops.render_extras(weights, tdist, far)
        for j, k in enumerate(['distance_mean', 'distance_percentile_5', 'distance_median',
                               'distance_percentile_95']):
          rendering[k] = ex[:B0, j].reshape(lead)
        nvis = self.config.vis_num_rays if self.config is not None else 16
        rendering['ray_sdist'] = sdist[:B0][:nvis]
        rendering['ray_weights'] = weights[:B0][:nvis]
        rendering['ray_rgbs'] = (rgb[:B0][:nvis] if rgb is not None
                                 else torch.zeros((min(nvis, B0), n, 3), dtype=f32, device=dev))
        # render.py:187-190: extras composited with the same weights
        if mlp_out.get('normals') is not None:
          rendering['normals'] = ops.weighted_sum(weights, mlp_out['normals'])[:B0].reshape(lead + (3,))
        if mlp_out.get('npred') is not None:
          rendering['normals_pred'] = ops.weighted_sum(weights, mlp_out['npred'])[:B0].reshape(lead + (3,))
        if mlp_out.get('rough') is not None:
          rendering['roughness'] = ops.weighted_sum(weights, mlp_out['rough'])[:B0].reshape(lead + (1,))
      renderings.append(rendering)
      rgb_hist = rgb[:B0] if rgb is not None else torch.zeros((B0, n, 3), dtype=f32, device=dev)
      ray_history.append(dict(
          density=density[:B0].reshape(lead + (n,)), rgb=rgb_hist.reshape(lead + (n, 3)),
          raw_grad_density=(mlp_out['raw_grad'].t().reshape(Bp, n, 3)[:B0].reshape(lead + (n, 3)) if plan.tangent else None),
          grad_pred=(mlp_out['small'][:, 1:4].reshape(Bp, n, 3)[:B0].reshape(lead + (n, 3)) if hp.enable_pred_normals else None),
          normals=(mlp_out['normals'].view(Bp, n, 3)[:B0].reshape(lead + (n, 3)) if plan.tangent else None),
          normals_pred=(mlp_out['npred'].view(Bp, n, 3)[:B0].reshape(lead + (n, 3)) if hp.enable_pred_normals else None),
          roughness=(mlp_out['rough'].view(Bp, n, 1)[:B0].reshape(lead + (n, 1)) if mlp_out.get('rough') is not None else None),
          sdist=sdist[:B0].reshape(lead + (n + 1,)), weights=weights[:B0].reshape(lead + (n,)),
          tdist=tdist[:B0].reshape(lead + (n + 1,))))
      if keep_for_backward:
        saved.append(dict(level=i_level, is_prop=is_prop, n=n, plan=plan, M=M, tag=tag, feat=feat, mlp=mlp_out, group=group,
                          ccfg=ccfg, raw_density=raw_density, raw_rgb=raw_rgb, dnoise=dnoise, bg=bg,
                          tdist=tdist, sdist=sdist, weights=weights, rgb_out=rgb_out, expo=expo, rs_kw=rs_kw, rs_in=rs_in,
                          near=near, far=far, radii=radii))

    if compute_extras:
      # models.py:299-310: proposal levels show the final level's average colour.
      final_rgb = torch.sum(renderings[-1]['ray_rgbs'] * renderings[-1]['ray_weights'][..., None], dim=-2)
      for r in renderings[:-1]:
        r['ray_rgbs'] = final_rgb[:, None, :].expand(r['ray_rgbs'].shape)

    if keep_for_backward:
      self._saved = dict(levels=saved, rays=R, B0=B0, Bp=Bp, renderings=renderings)
    return renderings, ray_history

  # ------------------------------------------------------------------ MLP forward / backward

  def _w(self, plan, off, rows, ld):
    """A [rows, ld] view into the packed bf16 operand buffer."""
    return plan.wbf[off:off + rows * ld].view(rows, ld)

  def _chain_ok(self, plan: MLPPlan):
    """The fused per-level kernels (csrc/fused_mlp.hip) cover a trunk of width 128 / 256, depth <= 8, with at most one
    skip concat: the proposal MLP of every config (with its Dense(1) head inside the kernel) and the 256-wide NeRF
    trunks of blender_256 / llff_raw / blender_refnerf (heads stay per-layer GEMMs on the trunk's output)."""
    skips = [i for i, (_, c) in enumerate(plan.trunk) if c]
    if not (_USE_CHAIN and not self._f32 and _USE_BITS and plan.hp.net_activation == 'relu' and plan.W in (128, 256) and
            1 <= len(plan.trunk) <= L.CHAIN_MAX_DEPTH and
            len(skips) <= 1 and not plan.x_concat):
      return False
    # the kernels read every bias row (and, for a density-only MLP, the fp32 head kernel) as 16-byte vectors straight out
    # of the flat parameter vector: a module base that is not a multiple of 4 floats (e.g. PropMLP_0 behind a Ref-NeRF
    # NerfMLP_0 of 713,230 parameters) takes the per-layer path, which has no such requirement
    offs = [d.bias_off for d, _ in plan.trunk] + ([] if plan.has_rgb else [plan.density.kernel_off])
    return all(o % 4 == 0 for o in offs)

  def _chain_trunk(self, plan: MLPPlan, flat, feat, M, tag, keep, need_bits):
    """models.py:455-459 (the Dense + ReLU trunk incl. its skip concat) as ONE launch; -> (acts, bits) like the per-layer
    loop: every layer's activation when `keep` (training: dW inputs), else only the last one (the heads' input)."""
    W, D = plan.W, len(plan.trunk)
    acts = [self._buf((tag, 'act', i if keep else i % 2), (M, W), bf16) if (keep or i == D - 1) else None for i in range(D)]
    bits = [self._buf((tag, 'bits', i), (M, W // 8), torch.uint8) if need_bits else None for i in range(D)]
    layers, skip = [], 0
    for i, (d, concat) in enumerate(plan.trunk):
      e = plan.packed[('trunk', i)]
      layers.append((self._w(plan, e['f_off'], e['n_pad'], e['f_ld']), flat[d.bias_off:d.bias_off + d.fan_out]))
      if concat:
        skip = i
    ops.mlp_chain_fwd(feat, plan.ldF, layers, M=M, W=W, acts=acts, bits=bits if need_bits else None, skip_layer=skip)
    return acts, bits

  def _chain_forward(self, plan: MLPPlan, flat, feat, M, tag, keep, group=None):
    """models.py:441-465 for a density-only MLP as ONE launch: every Dense + ReLU layer and the density head."""
    W, D = plan.W, len(plan.trunk)
    acts = [self._lvl_buf(tag, ('act', i), M, (W,), bf16, group) for i in range(D)] if keep else None
    bits = [self._lvl_buf(tag, ('bits', i), M, (W // 8,), torch.uint8, group) for i in range(D)] if keep else None
    layers, skip = [], 0
    for i, (d, concat) in enumerate(plan.trunk):
      e = plan.packed[('trunk', i)]
      layers.append((self._w(plan, e['f_off'], e['n_pad'], e['f_ld']), flat[d.bias_off:d.bias_off + d.fan_out]))
      if concat:
        skip = i
    e = plan.packed['density']
    d = plan.density
    raw_density = self._buf((tag, 'raw_density'), (M,), f32)
    ops.mlp_chain_fwd(feat, plan.ldF, layers, M=M, W=W, w_head=self._w(plan, e['f_off'], e['n_pad'], e['f_ld'])[0],
                      b_head=flat[d.bias_off:d.bias_off + 1], head_out=raw_density, acts=acts, bits=bits,
                      skip_layer=skip)
    return dict(acts=acts or [], bits=bits or [], raw_density=raw_density, chain=True)

  def _ipe_chain_ok(self, plan: MLPPlan):
    """A density-only MLP on the fused chain without a skip concat (the proposal MLP of every BASELINE config), with an
    encoding the in-kernel producer covers (groups of four degrees, at most 24 basis directions)."""
    return self._chain_ok(plan) and 'trunk0_ipe' in plan.packed

  def _chain_forward_ipe(self, plan: MLPPlan, flat, tdist, R, radii, M, tag):
    """models.py:441-465 for a density-only MLP AND its featurisation (render.cast_rays + integrated_pos_enc,
    models.py:413-431) as ONE launch; inference only (nothing is kept for a backward pass)."""
    hp, W = plan.hp, plan.W
    layers = []
    for i, (d, _) in enumerate(plan.trunk):
      e = plan.packed['trunk0_ipe' if i == 0 else ('trunk', i)]
      layers.append((self._w(plan, e['f_off'], e['n_pad'], e['f_ld']), flat[d.bias_off:d.bias_off + d.fan_out]))
    e = plan.packed['density']
    d = plan.density
    raw_density = self._buf((tag, 'raw_density'), (M,), f32)
    ops.mlp_chain_fwd_ipe(tdist, R.origins, R.directions, radii, plan.basis_dev, layers, M=M, W=W, ray_shape=self.ray_shape,
                          warp_contract=(hp.warp_fn == 'contract'), min_deg=hp.min_deg_point, max_deg=hp.max_deg_point,
                          disable_integration=self.disable_integration,
                          w_head=self._w(plan, e['f_off'], e['n_pad'], e['f_ld'])[0], b_head=flat[d.bias_off:d.bias_off + 1],
                          head_out=raw_density)
    return dict(acts=[], bits=[], raw_density=raw_density, chain=True)

  def _tangent_forward(self, plan: MLPPlan, tdist, R, M, bits, keep, tag, zs=None):
    """Density-gradient normals by forward mode (models.py:473-492 without a second autodiff pass): the three
    tangent feature rows d features / d mean_c of every sample run through the trunk as 3 * M extra GEMM rows whose ReLU is the
    primal layer's 1-bit mask; the density column of the last layer gives raw_grad [3, M]."""
    hp = plan.hp
    T_feat = self._buf((tag, 'T_feat'), (3 * M, plan.ldF), bf16)
    ops.cast_rays_ipe_tangent(tdist, R.origins, R.directions, R.radii.reshape(-1).contiguous(), plan.basis_dev,
                              ray_shape=self.ray_shape, min_deg=hp.min_deg_point, max_deg=hp.max_deg_point,
                              ld_feat=plan.ldF, out=T_feat, warp_contract=(hp.warp_fn == 'contract'),
                              disable_integration=self.disable_integration)
    T_acts, T_pre = [], []
    t = None
    relu = hp.net_activation == 'relu'
    D = len(plan.trunk)
    # (training only: the chain writes every layer's rows, which the weight-gradient GEMMs want anyway; inference ping-pongs two buffers)
    chain = bool(_TANGENT_CHAIN and keep and relu and self._chain_ok(plan) and all(b is not None for b in bits))
    i = 0
    while i < D:
      d, concat = plan.trunk[i]
      e2 = plan.packed[('trunk', i)]
      run = 0
      if chain and i >= 1 and not concat:
        while i + run < D and not plan.trunk[i + run][1]:
          run += 1
      if run >= 1:
        # layers i .. i + run - 1 as one masked-linear chain per direction c (rows c M .. (c + 1) M share the primal masks):
        # mnr_mlp_chain_bwd with the FORWARD images as its operands walks "down" from its input; chain position j holds layer
        # i + run - j: dY_in = T_{i-1}, Bw[j] = Bt of that layer, bits[j - 1] its mask, dY[j - 1] its result
        outs = [self._buf((tag, 'T_act', i + k), (3 * M, plan.W), bf16) for k in range(run)]
        Bws, cbits = [None] * (run + 1), [None] * (run + 1)
        for j in range(1, run + 1):
          l = i + run - j
          el = plan.packed[('trunk', l)]
          Bws[j] = self._w(plan, el['f_off'], el['n_pad'], el['f_ld'])[:plan.W]
          cbits[j - 1] = bits[l]
        cbits[run] = bits[i]                                  # (unused with dY_in; the launcher wants a pointer)
        for c in range(3):
          rows = slice(c * M, (c + 1) * M)
          dYs = [outs[run - 1 - jj][rows] for jj in range(run)] + [None]
          ops.mlp_chain_bwd(None, None, cbits, Bws, dYs, M=M, W=plan.W, dY_in=t[rows])
        T_acts += outs
        t = outs[-1]
        i += run
        continue
      tout = self._buf((tag, 'T_act', i if keep else i % 2), (3 * M, plan.W), bf16)
      Bt2 = self._w(plan, e2['f_off'], e2['n_pad'], e2['f_ld'])
      # ReLU: T_i = mask_i * (T_{i-1} W_i), the mask applied in the GEMM's epilogue.  Any other activation: the GEMM leaves the
      # tangent pre-activation U_i (kept for the backward pass's act'' term), then T_i = act'(z_i) * U_i
      mk = dict(bits_in=bits[i], bits_row_mod=M) if relu else {}
      dst = tout if relu else self._buf((tag, 'T_pre', i if keep else i % 2), (3 * M, plan.W), bf16)
      if i == 0:
        ops.gemm_nt(T_feat, Bt2, M=3 * M, N=e2['n_pad'], K1=plan.ldF, Cb=dst, ldcb=plan.W, nb=plan.W, **mk)
      elif concat:
        ops.gemm_nt(t, Bt2, M=3 * M, N=e2['n_pad'], K1=plan.W, A2=T_feat, K2=plan.ldF, Cb=dst, ldcb=plan.W, nb=plan.W, **mk)
      else:
        ops.gemm_nt(t, Bt2, M=3 * M, N=e2['n_pad'], K1=plan.W, Cb=dst, ldcb=plan.W, nb=plan.W, **mk)
      if not relu:
        ops.act_tangent_fwd(hp.net_activation, zs[i], dst, tout)
        T_pre.append(dst)
      T_acts.append(tout)
      t = tout
      i += 1
    raw_grad = self._buf((tag, 'raw_grad'), (3, M), f32)
    ed = plan.packed['density']
    ops.gemm_nt(t, self._w(plan, ed['f_off'], ed['n_pad'], ed['f_ld']), M=3 * M, N=ed['n_pad'], K1=plan.W,
                Cf=raw_grad, ldcf=1, f0=0, nf=1)
    self._T_pre = T_pre                                   # (non-ReLU activations: the tangent pre-activations, for _tangent_backward)
    return T_feat, T_acts, raw_grad

  def _mlp_forward(self, plan: MLPPlan, flat, feat, M, n, R, tag, keep, tdist=None, bnoise=None, group=None):
    """MLP.__call__ (models.py:402-612) for the M = B*n samples of one level."""
    hp = plan.hp
    chain = self._chain_ok(plan)
    if chain and not plan.has_rgb:
      return self._chain_forward(plan, flat, feat, M, tag, keep, group)
    assert group is None
    relu = hp.net_activation == 'relu'
    need_bits = ((keep and _USE_BITS) or plan.tangent) and relu
    acts, bits, zs, vzs = [], [], [], []
    x = None
    # panel layout for the trunk of this level (the layout of acts / bits / every trunk dY of the backward pass): every
    # producer and consumer is then a panel-aware GEMM, which the plain merged head of a wide ReLU trunk guarantees
    panel = self._panel_ok(plan, M, keep)
    PAN = ops.LAYOUT_PANEL
    lay_c = dict(c_layout=PAN) if panel else {}
    lay_a = dict(a1_layout=PAN) if panel else {}

    def activate(z_key, out, n_cols, zlist):
      """Non-ReLU net_activation: the GEMM wrote the pre-activation into `out`'s twin buffer; apply act (models.py:457,578)."""
      z = self._buf(z_key, (M, n_cols), bf16)
      zlist.append(z)
      return z

    if chain:
      acts, bits = self._chain_trunk(plan, flat, feat, M, tag, keep, need_bits)
      x = acts[-1]
    for i, (d, concat) in enumerate([] if chain else plan.trunk):
      e = plan.packed[('trunk', i)]
      out = self._buf((tag, 'act', i if keep else i % 2), (M, plan.W), bf16)
      # 1-bit ReLU mask: backward pass (training) and the tangent pass of the density-gradient normals
      bo = self._buf((tag, 'bits', i if (keep or plan.tangent) else i % 2), (M, plan.W // 8), torch.uint8) if need_bits else None
      bits.append(bo)
      Bt = self._w(plan, e['f_off'], e['n_pad'], e['f_ld'])
      bias = flat[d.bias_off:d.bias_off + d.fan_out]
      # non-ReLU activations: the GEMM stores the pre-activation, a second kernel applies softplus / silu
      dst = out if relu else activate((tag, 'z', i if (keep or plan.tangent) else i % 2), out, plan.W, zs)
      if i == 0:
        ops.gemm_nt(feat, Bt, M=M, N=e['n_pad'], K1=plan.ldF, bias=bias, n_bias=d.fan_out, relu=relu,
                    Cb=dst, ldcb=plan.W, nb=plan.W, bits_out=bo, walk_descending=bool(i & 1), **lay_c)
      elif concat:
        ops.gemm_nt(x, Bt, M=M, N=e['n_pad'], K1=plan.W, A2=feat, K2=plan.ldF, bias=bias, n_bias=d.fan_out,
                    relu=relu, Cb=dst, ldcb=plan.W, nb=plan.W, bits_out=bo, walk_descending=bool(i & 1), **lay_a, **lay_c)
      else:
        ops.gemm_nt(x, Bt, M=M, N=e['n_pad'], K1=plan.W, bias=bias, n_bias=d.fan_out, relu=relu,
                    Cb=dst, ldcb=plan.W, nb=plan.W, bits_out=bo, walk_descending=bool(i & 1), **lay_a, **lay_c)
      if not relu:
        ops.act_fwd(hp.net_activation, dst, out)
      acts.append(out)
      x = out
    res = dict(acts=acts, bits=bits, chain_trunk=chain, zs=zs, vzs=vzs, panel=panel)
    raw_density = self._buf((tag, 'raw_density'), (M,), f32)
    if plan.has_rgb and not plan.use_viewdirs:
      # models.py:585 with x = the trunk output: one 4-column head [raw_density | raw_rgb] as an fp32 side output
      e = plan.packed['head4']
      small4 = self._buf((tag, 'small4'), (M, 4), f32)
      ops.gemm_nt(x, self._w(plan, e['f_off'], e['n_pad'], e['f_ld']), M=M, N=e['n_pad'], K1=plan.W, bias=plan.head_bias,
                  n_bias=4, relu=False, Cf=small4, ldcf=4, f0=0, nf=4)
      raw_density.copy_(small4[:, 0])
      raw_rgb = self._buf((tag, 'raw_rgb'), (M, 3), f32)
      raw_rgb.copy_(small4[:, 1:4])
      res.update(raw_rgb=raw_rgb, raw_density=raw_density)
      return res
    if plan.has_rgb:
      bw = hp.bottleneck_width
      e = plan.packed['head']
      VI = self._buf((tag, 'VI'), (M, plan.ldVI), bf16)
      Bt = self._w(plan, e['f_off'], e['n_pad'], e['f_ld'])
      if plan.ref:
        small = self._buf((tag, 'small'), (M, 11), f32)
        ops.gemm_nt(x, Bt, M=M, N=e['n_pad'], K1=plan.W, bias=plan.head_bias, n_bias=plan.head_cols, relu=False,
                    Cb=VI, ldcb=plan.ldVI, nb=bw, Cf=small, ldcf=11, f0=bw, nf=11)
        raw_density.copy_(small[:, 0])
        raw_grad = None
        if plan.tangent:
          T_feat, T_acts, raw_grad = self._tangent_forward(plan, tdist, R, M, bits, keep, tag, zs=zs)
          res.update(T_feat=T_feat, T_acts=T_acts, raw_grad=raw_grad, T_pre=self._T_pre)
        normals, npred, rough = ops.ref_head_fwd(small, raw_grad, R.viewdirs, n, plan.ide, hp.roughness_bias, VI,
                                                 bw, plan.ldVI, features=plan.features, deg_view=hp.deg_view)
        res.update(small=small, normals=normals, npred=npred, rough=rough)
      elif plan.pn:
        # [raw_density | grad_pred] as the fp32 side output; normals_pred = -l2_normalize(grad_pred) (models.py:494-503)
        small = self._buf((tag, 'small_pn'), (M, 4), f32)
        ops.gemm_nt(x, Bt, M=M, N=e['n_pad'], K1=plan.W, bias=plan.head_bias, n_bias=plan.head_cols, relu=False,
                    Cb=VI, ldcb=plan.ldVI, nb=bw, Cf=small, ldcf=4, f0=bw, nf=4, **lay_a)
        raw_density.copy_(small[:, 0])
        ops.viewdir_enc_fill(R.viewdirs, n, hp.deg_view, VI, bw, plan.ldVI)
        res.update(small=small, npred=ops.pred_normals_fwd(small, 1))
      elif panel and _HEAD_VCOL and bw == 256 and plan.W <= 1536:
        # the plain merged head [bottleneck | density] behind a panel-storage trunk: N = 256 and the density column as a VECTOR
        # (row bw of the merged forward image) instead of a second 256-column tile for one column (mnr_gemm_nt_args.vcol:
        # one extra MFMA per wave and k-step; bitwise the merged operand's result)
        ops.gemm_nt(x, Bt[:bw], M=M, N=bw, K1=plan.W, bias=plan.head_bias, n_bias=bw, relu=False, Cb=VI, ldcb=plan.ldVI, nb=bw,
                    vcol=Bt[bw], vcol_out=raw_density, vcol_bias=plan.head_bias[bw:bw + 1], **lay_a)
        ops.viewdir_enc_fill(R.viewdirs, n, hp.deg_view, VI, bw, plan.ldVI)
      else:
        ops.gemm_nt(x, Bt, M=M, N=e['n_pad'], K1=plan.W, bias=plan.head_bias, n_bias=bw + 1, relu=False,
                    Cb=VI, ldcb=plan.ldVI, nb=bw, Cf=raw_density, ldcf=1, f0=bw, nf=1, **lay_a)
        ops.viewdir_enc_fill(R.viewdirs, n, hp.deg_view, VI, bw, plan.ldVI)
      if plan.dn:
        # density-gradient normals without the rest of the Ref-NeRF head (models.py:478-492): for the renderings and the
        # orientation loss; they do not enter the colour
        T_feat, T_acts, raw_grad = self._tangent_forward(plan, tdist, R, M, bits, keep, tag, zs=zs)
        res.update(T_feat=T_feat, T_acts=T_acts, raw_grad=raw_grad, normals=ops.density_normals_fwd(raw_grad), T_pre=self._T_pre)
      if plan.glo > 0:
        ops.glo_fill(self._glo_table(flat), self._glo_cam, M // n, n, VI, plan.glo_col)
      if bnoise is not None:
        # bottleneck += bottleneck_noise * N(0, 1) (models.py:530-533): additive, so the backward pass is unchanged
        ops.add_noise_bf16(VI, bw, bnoise, hp.bottleneck_noise)
      h = VI
      vacts, vbits = [], []
      WV = hp.net_width_viewdirs
      for i, (d, concat) in enumerate(plan.view):
        e = plan.packed[('view', i)]
        out = self._buf((tag, 'vact', i if keep else i % 2), (M, WV), bf16)
        Bt = self._w(plan, e['f_off'], e['n_pad'], e['f_ld'])
        bias = flat[d.bias_off:d.bias_off + d.fan_out]
        dst = out if relu else activate((tag, 'vz', i if keep else i % 2), out, WV, vzs)
        if i == 0:
          ops.gemm_nt(VI, Bt, M=M, N=e['n_pad'], K1=plan.ldVI, bias=bias, n_bias=d.fan_out, relu=relu,
                      Cb=dst, ldcb=WV, nb=WV)
        elif concat:
          ops.gemm_nt(h, Bt, M=M, N=e['n_pad'], K1=WV, A2=VI, K2=plan.ldVI, bias=bias, n_bias=d.fan_out,
                      relu=relu, Cb=dst, ldcb=WV, nb=WV)
        else:
          ops.gemm_nt(h, Bt, M=M, N=e['n_pad'], K1=WV, bias=bias, n_bias=d.fan_out, relu=relu,
                      Cb=dst, ldcb=WV, nb=WV)
        if not relu:
          ops.act_fwd(hp.net_activation, dst, out)
        vacts.append(out)
        h = out
      e = plan.packed['rgb']
      raw_rgb = self._buf((tag, 'raw_rgb'), (M, 3), f32)
      d = plan.rgb
      ops.gemm_nt(h, self._w(plan, e['f_off'], e['n_pad'], e['f_ld']), M=M, N=e['n_pad'], K1=e['kpad'],
                  bias=flat[d.bias_off:d.bias_off + 3], n_bias=3, relu=False, Cf=raw_rgb, ldcf=3, f0=0, nf=3)
      res.update(VI=VI, vacts=vacts, raw_rgb=raw_rgb)
      if plan.diffuse_on:
        # models.py:584-602: tinted (or halved) specular + diffuse, tone-mapped; compositing then sees final colours
        res['raw_rgb_pre'] = raw_rgb
        res['raw_rgb'] = ops.ref_color_fwd(raw_rgb, res['small'], hp.rgb_premultiplier, hp.rgb_bias, hp.rgb_padding,
                                           hp.use_specular_tint)
    else:
      e = plan.packed['density']
      d = plan.density
      ops.gemm_nt(x, self._w(plan, e['f_off'], e['n_pad'], e['f_ld']), M=M, N=e['n_pad'], K1=plan.W,
                  bias=flat[d.bias_off:d.bias_off + 1], n_bias=1, relu=False, Cf=raw_density, ldcf=1, f0=0, nf=1)
    res['raw_density'] = raw_density
    return res

  def _head_gcol(self, plan: MLPPlan):
    """The merged head's weight gradient as the bottleneck's 256-column GEMM plus the density column as a vector (backward_level)."""
    W, bw = plan.W, plan.hp.bottleneck_width
    return bool(_HEAD_GCOL and not self._f32 and plan.has_rgb and plan.use_viewdirs and not plan.ref and len(plan.head_segs) == 2 and
                bw % 256 == 0 and W % 256 == 0 and W >= 512)

  def _panel_ok(self, plan: MLPPlan, M, keep):
    """True when this level's per-layer trunk runs in the panel layout: a ReLU trunk of width >= 512 (a multiple of 256: the
    narrower ones take the fused chain / weights-resident kernels) under the plain merged head of models.py:494-585, whose
    forward GEMM, dX GEMM and weight-gradient GEMM (with the density column as a vector) all read or write panel storage."""
    if not (_PANEL and not self._f32 and _USE_BITS and plan.hp.net_activation == 'relu' and plan.W % 256 == 0 and plan.W >= 512 and M % 256 == 0):
      return False
    if self._chain_ok(plan) or plan.tangent or plan.pn or plan.ref or not (plan.has_rgb and plan.use_viewdirs):
      return False
    if plan.ldF % 32 != 0 or plan.ldF < 192 or plan.packed['head']['n_pad'] % 256 != 0:
      return False
    if keep and plan.ldF % 256 != 0:
      # the layer-0 / skip-segment weight gradients are gemm_tn(feat, dY, K = ldF, b_layout = PANEL), and panel operands
      # need K % 256 == 0 (mnr_gemm_tn_bf16): ldF = 384 / 640 / 896 train on the row-major path (128 x 128 tiles)
      return False
    return (not keep) or self._head_gcol(plan)

  def _glo_table(self, flat):
    G = self.num_glo_features
    return flat[self.glo_off:self.glo_off + self.num_glo_embeddings * G].view(self.num_glo_embeddings, G)

  @_in_library
  def backward_level(self, lv, flat, grads, g_rgb_out, g_weights, g_expo=None, g_normals=None, g_npred=None, losses=None,
                     g_x_out=None, g_feat_out=None, g_tfeat_out=None):
    """VJP of one level w.r.t. the parameters: compositing -> heads -> trunk.
    grads: flat fp32 gradient vector (accumulated into).  g_normals / g_npred [M,3]: from the Ref-NeRF
    normal losses (train_utils.py:162-197).  losses: this level's data / interlevel / distortion losses, evaluated
    and differentiated inside the compositing VJP's launch (ops.composite_bwd).
    stop_level_grad = False (models.py:198-201): g_x_out [B, n] receives d loss / d (sigma * delta) of the compositing, and
    the list g_feat_out the bf16 [M, ldF] matrices whose sum is d loss / d features (one per trunk layer that reads the
    features: layer 0 and the skip layer); the list g_tfeat_out the [3 M, ldF] matrices whose sum is d loss / d (tangent feature
    rows) of the density-gradient normals' forward-mode network (`_tangent_backward`)."""
    plan: MLPPlan = lv['plan']
    hp = plan.hp
    M, n, tag = lv['M'], lv['n'], lv['tag']
    R = self._saved['rays']
    mlp = lv['mlp']
    acts = mlp['acts']
    x_last = acts[-1]
    W = plan.W
    D = len(plan.trunk)
    # Workspace of this pass, keyed by the stream it runs on: the proposal levels' backward runs on a side stream next to the
    # NeRF level's (train_utils.create_train_step), and a proposal MLP that takes this per-layer path (not chain-eligible:
    # non-ReLU activation, odd widths, MNR_FUSED_CHAIN=0) with the NeRF MLP's width and row count would otherwise share dA / dB /
    # dV / dHB / ... with it.  Levels that run one after the other on the same stream share their buffers.
    slot = 'nerf' if lv['level'] == self.num_levels - 1 else 'prop'

    def dy_buf(i):
      """dY of trunk layer i's output: two ping-pong buffers shared by the levels of a stream."""
      return self._buf(('bwd', slot, 'dA' if (D - 1 - i) % 2 == 0 else 'dB', W), (M, W), bf16)

    def dv_buf(i, nv):
      return self._buf(('bwd', slot, 'dV0' if (nv - 1 - i) % 2 == 0 else 'dV1', WV), (M, WV), bf16)

    if not mlp.get('chain'):
      dA = dy_buf(D - 1)

    def gslice(off, size):
      return grads[off:off + size]

    relu = hp.net_activation == 'relu'
    panel = bool(mlp.get('panel'))                        # the trunk's activations / masks / gradients are in panel storage
    PAN = ops.LAYOUT_PANEL
    lay_c = dict(c_layout=PAN) if panel else {}
    lay_ac = dict(a1_layout=PAN, c_layout=PAN) if panel else {}
    tn_a = dict(a_layout=PAN) if panel else {}
    tn_b = dict(b_layout=PAN) if panel else {}

    def act_vjp(z, d):
      """Non-ReLU activations: d (gradient w.r.t. a layer's activation, just written without a mask) *= act'(z)."""
      if not relu:
        ops.act_bwd(hp.net_activation, z, d)

    def mask_kw(i):
      """ReLU VJP of trunk layer i's output: 1-bit mask if available, else the saved activation."""
      if not relu:
        return {}
      if mlp['bits'][i] is not None:
        return dict(bits_in=mlp['bits'][i])
      return dict(mask=acts[i], ldmask=W)

    def feat_grad(i, dy, dy_panel=False):
      """d loss / d features through trunk layer i (0 or a skip layer): dY_i @ kernel_i[feature rows]^T -> bf16 [M, ldF]."""
      if g_feat_out is None:
        return
      e_ = plan.packed[('trunk', i)]
      out = self._buf(('bwd', slot, 'g_feat', len(g_feat_out)), (M, plan.ldF), bf16)
      ops.gemm_nt(dy, self._w(plan, e_['bf_off'], plan.ldF, e_['bf_ld']), M=M, N=plan.ldF, K1=e_['bf_ld'], Cb=out,
                  ldcb=plan.ldF, nb=plan.ldF, **(dict(a1_layout=PAN) if dy_panel else {}))
      g_feat_out.append(out)

    g_raw_grad = None
    if plan.has_rgb and not plan.use_viewdirs:
      g_raw_density, g_rgb = ops.composite_bwd(
          lv['ccfg'], lv['raw_density'], lv['tdist'], R.directions, lv['weights'], raw_rgb=lv['raw_rgb'],
          density_noise=lv['dnoise'], bg=lv['bg'], g_rgb_out=g_rgb_out, g_weights=g_weights, want_f32=True,
          exposure_scale=lv['expo'], g_exposure_scale=g_expo if lv['expo'] is not None else None, losses=losses,
          g_x_out=g_x_out)
      # the 4-column head [density | rgb]: dX into the trunk, dW / db scattered to the two Dense layers
      g4 = self._buf(('bwd', slot, 'g4'), (M, 4), f32)
      g4[:, 0].copy_(g_raw_density.view(M))
      g4[:, 1:4].copy_(g_rgb.view(M, 3))
      dn, dr = plan.density, plan.rgb
      w4 = self._buf(('bwd', slot, 'w4', W), (W, 4), f32)
      w4[:, 0].copy_(flat[dn.kernel_off:dn.kernel_off + W])
      w4[:, 1:4].copy_(flat[dr.kernel_off:dr.kernel_off + 3 * W].view(W, 3))
      t4 = self._buf(('bwd', slot, 't4', W), (W + 1, 4), f32)
      t4.zero_()
      ops.small_head_bwd(x_last, W, g4, w4, M=M, K=W, Cn=4, dX=dA, lddx=W, relu_mask=relu, dW=t4[:W].view(-1), db=t4[W])
      act_vjp(mlp['zs'][-1] if not relu else None, dA)
      for (d, c0) in plan.head_segs:
        ops.scatter_add(t4, 4, 0, c0, W, d.fan_out, gslice(d.kernel_off, W * d.fan_out), d.fan_out)
        ops.scatter_add(t4, 4, W, c0, 1, d.fan_out, gslice(d.bias_off, d.fan_out), d.fan_out)
    elif plan.has_rgb:
      bw = hp.bottleneck_width
      e = plan.packed['head']
      nh = e['nb_pad']
      # (columns beyond head_cols, and those of Ref-NeRF heads this MLP does not have, stay zero: keyed by module)
      dHB = self._buf(('bwd', slot, 'dHB', nh, plan.module_name), (M, nh), bf16, zero=True)
      # (the plain merged head [bottleneck | density] of 360.gin: the density column's weight gradient rides in the bottleneck's
      # dW GEMM as a vector, below; it then also leaves the compositing VJP as the fp32 vector that GEMM reads)
      # (for trunks of at least 512 columns: at 256 the merged N = 384 GEMM is six small tiles and the extra column buys nothing,
      # blender_256 1.764 / 1.767 M rays/s merged against 1.749 / 1.764 M, llff_raw 546.0 against 545.9 k)
      head_gcol = self._head_gcol(plan)
      assert head_gcol or not panel
      g_den_f32, g_rgb = ops.composite_bwd(
          lv['ccfg'], lv['raw_density'], lv['tdist'], R.directions, lv['weights'], raw_rgb=lv['raw_rgb'],
          density_noise=lv['dnoise'], bg=lv['bg'], g_rgb_out=g_rgb_out, g_weights=g_weights,
          g_den_bf16=dHB.view(-1)[bw:], ld_bf16=nh, want_f32=head_gcol, exposure_scale=lv['expo'],
          g_exposure_scale=g_expo if lv['expo'] is not None else None, losses=losses, g_x_out=g_x_out)
      g_raw_rgb = g_rgb.view(M, 3)
      if head_gcol:
        # the density column of dHB once more as a contiguous bf16 vector, and the density bias gradient from the strided column
        # (33 MB of 64-byte sectors): two small launches, issued HERE, before the proposal levels' persistent kernels fill
        # the CUs from the side stream (behind them a 25-us launch of 512 small workgroups took 0.3-0.6 ms to get its CUs,
        # with the main stream's next GEMM waiting for it: profiles/r3s3_step_seq_default.md)
        g_vec = self._buf(('bwd', slot, 'g_den_vec'), (M,), bf16)
        ops.cast_f32_to_bf16(g_den_f32.view(-1), 1, M, 1, g_vec, 1, 0)
        ops.colsum(dHB.view(-1)[bw:], M, 1, gslice(plan.density.bias_off, 1), ld=nh)
      if plan.diffuse_on:
        # colour combine VJP: -> d raw specular rgb, and the diffuse / tint columns of the head gradient
        g_raw_rgb = ops.ref_color_bwd(mlp['raw_rgb_pre'], mlp['small'], hp.rgb_premultiplier, hp.rgb_bias,
                                      hp.rgb_padding, hp.use_specular_tint, g_raw_rgb.contiguous(), dHB, bw + 4, bw + 7)
      # rgb Dense(3): dH, dW, db
      WV = hp.net_width_viewdirs
      vacts = mlp['vacts']
      d = plan.rgb
      NV = len(plan.view)
      dV0 = dv_buf(NV - 1, NV)
      h_last = vacts[-1]
      ops.small_head_bwd(h_last, WV, g_raw_rgb, flat[d.kernel_off:d.kernel_off + d.fan_in * 3].view(d.fan_in, 3),
                         M=M, K=WV, Cn=3, dX=dV0, lddx=WV, relu_mask=relu,
                         dW=gslice(d.kernel_off, d.fan_in * 3), db=gslice(d.bias_off, 3))
      act_vjp(mlp['vzs'][-1] if not relu else None, dV0)
      dy = dV0
      VI = mlp['VI']
      dVIa = dVIb = None
      want_glo = plan.glo > 0 and self._glo_cam is not None
      gGa = self._buf(('bwd', slot, 'gGa'), (M, plan.glo), f32) if want_glo else None
      gGb = None
      glo_kw = lambda t: dict(Cf=t, ldcf=plan.glo, f0=plan.glo_col, nf=plan.glo) if want_glo else {}
      for i in reversed(range(len(plan.view))):
        d, concat = plan.view[i]
        e = plan.packed[('view', i)]
        inp = VI if i == 0 else vacts[i - 1]
        in_w = plan.ldVI if i == 0 else WV
        # dW (rows of the first input segment; then the skip-concat rows), db
        ops.gemm_tn(inp, dy, gslice(d.kernel_off, d.fan_in * d.fan_out), M=M, K=in_w, N=WV,
                    lda=inp.stride(0), ldb=WV, ldc=d.fan_out,
                    k_valid=(plan.vi_width if i == 0 else WV), n_valid=d.fan_out,
                    bias_out=gslice(d.bias_off, d.fan_out), bias_n_valid=d.fan_out)
        if concat:
          ops.gemm_tn(VI, dy, gslice(d.kernel_off + WV * d.fan_out, plan.vi_width * d.fan_out), M=M,
                      K=plan.ldVI, N=WV, lda=plan.ldVI, ldb=WV, ldc=d.fan_out, k_valid=plan.vi_width,
                      n_valid=d.fan_out)
        if concat:
          # the view input also receives gradient through the skip concat (bottleneck, IDE / n.v / GLO columns)
          first_skip = dVIb is None
          tVI = self._buf(('bwd', slot, 'dVIb', 0 if first_skip else 1), (M, plan.ldVI), bf16)
          tG = self._buf(('bwd', slot, 'gGb', 0 if first_skip else 1), (M, plan.glo), f32) if want_glo else None
          B2 = self._w(plan, e['b2_off'], plan.ldVI, e['b_ld'])
          ops.gemm_nt(dy, B2, M=M, N=plan.ldVI, K1=e['b_ld'], Cb=tVI, ldcb=plan.ldVI, nb=plan.ldVI, **glo_kw(tG))
          if first_skip:
            dVIb, gGb = tVI, tG
          else:                                    # more than one skip layer: sum the contributions
            ops.add_cols_bf16(dVIb, tVI, dVIb, plan.ldVI)
            if want_glo:
              gGb.add_(tG)
        Bw = self._w(plan, e['b_off'], e['b_rows'], e['b_ld'])
        if i == 0:
          if plan.ref:
            dVIa = self._buf(('bwd', slot, 'dVIa'), (M, plan.ldVI), bf16)
            ops.gemm_nt(dy, Bw, M=M, N=e['b_rows'], K1=e['b_ld'], Cb=dVIa, ldcb=plan.ldVI, nb=plan.ldVI, **glo_kw(gGa))
          else:
            ops.gemm_nt(dy, Bw, M=M, N=e['b_rows'], K1=e['b_ld'], Cb=dHB, ldcb=nh, nb=bw, **glo_kw(gGa))
        else:
          other = dv_buf(i - 1, NV)
          if relu:
            ops.gemm_nt(dy, Bw, M=M, N=WV, K1=e['b_ld'], mask=vacts[i - 1], ldmask=WV, Cb=other, ldcb=WV, nb=WV)
          else:
            ops.gemm_nt(dy, Bw, M=M, N=WV, K1=e['b_ld'], Cb=other, ldcb=WV, nb=WV)
            act_vjp(mlp['vzs'][i - 1], other)
          dy = other
      if want_glo:
        G = plan.glo
        ops.glo_bwd(gGa, gGb, self._glo_cam, M // n, n, grads[self.glo_off:self.glo_off + self.num_glo_embeddings * G],
                    self.num_glo_embeddings, G)
      if dVIb is not None and not plan.ref:
        # bottleneck gradient through the skip concat (a view MLP deeper than skip_layer_dir)
        ops.add_cols_bf16(dHB, dVIb, dHB, bw)
      if plan.dn and g_normals is not None:
        g_raw_grad = ops.density_normals_bwd(mlp['raw_grad'], g_normals.view(M, 3))      # -> the tangent network, below
      if plan.pn and g_npred is not None:
        # VJP of normals_pred = -l2_normalize(grad_pred) into the grad_pred columns of the head gradient (zero without a loss on them)
        ops.pred_normals_bwd(mlp['small'], 1, g_npred.view(M, 3), dHB, bw + 1)
      if plan.ref:
        # IDE / reflection / normalisation VJP: fills the bottleneck (dVIa + dVIb), grad_pred and roughness
        # columns of dHB and returns the gradient w.r.t. d raw_density / d mean for the tangent network.
        g_raw_grad = ops.ref_head_bwd(mlp['small'], mlp.get('raw_grad'), R.viewdirs, n, plan.ide, hp.roughness_bias,
                                      dVIa, dVIb, bw, g_npred, g_normals, dHB, bw + 1, bw + 10, features=plan.features,
                                      deg_view=hp.deg_view)
      # merged head: dW, db, dX_last
      e = plan.packed['head']
      if head_gcol:
        # dW_bottleneck += x^T dHB[:, :bw] straight into the flat gradient (256x256 tiles), dw_density += x^T g as one more
        # column of the same launch (db_density: above)
        db_, dd_ = plan.bottleneck, plan.density
        ops.gemm_tn(x_last, dHB, gslice(db_.kernel_off, W * bw), M=M, K=W, N=bw, lda=W, ldb=nh, ldc=bw,
                    bias_out=gslice(db_.bias_off, bw), bias_n_valid=bw, gcol=g_vec, gcol_out=gslice(dd_.kernel_off, W), **tn_a)
      else:
        tmpW = self._buf(('bwd', slot, 'tmpW', W, nh), (W, nh), f32)
        tmpW.zero_()
        tmpb = self._buf(('bwd', slot, 'tmpb', nh), (nh,), f32)
        tmpb.zero_()
        ops.gemm_tn(x_last, dHB, tmpW, M=M, K=W, N=nh, lda=W, ldb=nh, ldc=nh, bias_out=tmpb,
                    bias_n_valid=plan.head_cols)
        for (d, c0) in plan.head_segs:
          ops.scatter_add(tmpW, nh, 0, c0, W, d.fan_out, gslice(d.kernel_off, W * d.fan_out), d.fan_out)
          ops.scatter_add(tmpb, nh, 0, c0, 1, d.fan_out, gslice(d.bias_off, d.fan_out), d.fan_out)
      Bw = self._w(plan, e['b_off'], _rup(W, 128), e['b_ld'])
      # (K = the head's columns rounded to the GEMM's 64-column K granule, not to the buffers' 128: 320 instead of 384 at 360.gin)
      ops.gemm_nt(dHB, Bw, M=M, N=_rup(W, 128), K1=_rup(plan.head_cols, 32 if (panel and _HEAD_K32) else 64), Cb=dA, ldcb=W, nb=W,
                  **mask_kw(len(acts) - 1), **lay_c)
      act_vjp(mlp['zs'][-1] if not relu else None, dA)
    else:
      g_raw_density, _ = ops.composite_bwd(
          lv['ccfg'], lv['raw_density'], lv['tdist'], R.directions, lv['weights'], density_noise=lv['dnoise'],
          bg=lv['bg'], g_rgb_out=g_rgb_out, g_weights=g_weights, want_f32=True, losses=losses, g_x_out=g_x_out)
      d = plan.density
      if mlp.get('chain'):
        # fused dX chain: head dW / db from the last activation, then every dY_i in one launch; dW_i = x_{i-1}^T dY_i below
        w_head = flat[d.kernel_off:d.kernel_off + W]
        ops.small_head_bwd(x_last, W, g_raw_density.view(M, 1), w_head.view(W, 1), M=M, K=W, Cn=1, dX=None,
                           relu_mask=False, dW=gslice(d.kernel_off, W), db=gslice(d.bias_off, 1))
        D = len(plan.trunk)
        # (keyed by level: the proposal levels' backward passes may run side by side on streams of their own)
        # (the last dY = mask * (g (x) w_head) is not stored when its only reader, the last layer's weight-gradient GEMM, can
        # build it from the factors: `_rank1_last`)
        r1 = _rank1_last(plan, D, W, g_feat_out is not None)
        dYs = [None if (r1 and i == D - 1) else self._buf(('bwd', slot, 'dYc', W, i, lv['level']), (M, W), bf16) for i in range(D)]
        Bws = [None] + [self._w(plan, plan.packed[('trunk', i)]['b_off'], _rup(W, 128), plan.packed[('trunk', i)]['b_ld'])
                        for i in range(1, D)]
        ops.mlp_chain_bwd(g_raw_density.view(M), w_head, mlp['bits'], Bws, dYs, M=M, W=W)
        feat = lv['feat']
        for i, (dl, concat) in enumerate(plan.trunk):
          inp, in_w, kv = (feat, plan.ldF, plan.F) if i == 0 else (acts[i - 1], W, W)
          ops.gemm_tn(inp, dYs[i], gslice(dl.kernel_off, kv * W), M=M, K=in_w, N=W, lda=in_w, ldb=W, ldc=W,
                      k_valid=kv, n_valid=W, bias_out=gslice(dl.bias_off, W), bias_n_valid=W,
                      rank1=(g_raw_density.view(M), w_head, mlp['bits'][i]) if dYs[i] is None else None)
          if concat:
            ops.gemm_tn(feat, dYs[i], gslice(dl.kernel_off + W * W, plan.F * W), M=M, K=plan.ldF, N=W,
                        lda=plan.ldF, ldb=W, ldc=W, k_valid=plan.F, n_valid=W)
          if i == 0 or concat:
            feat_grad(i, dYs[i])
        return
      ops.small_head_bwd(x_last, W, g_raw_density.view(M, 1), flat[d.kernel_off:d.kernel_off + W].view(W, 1),
                         M=M, K=W, Cn=1, dX=dA, lddx=W, relu_mask=relu,
                         dW=gslice(d.kernel_off, W), db=gslice(d.bias_off, 1))
      act_vjp(mlp['zs'][-1] if not relu else None, dA)
    feat = lv['feat']
    t_extras = None
    if g_raw_grad is not None:
      t_extras = self._tangent_backward(plan, flat, grads, mlp, feat, M, g_raw_grad, slot, g_tfeat_out)
    if mlp.get('chain_trunk'):
      # fused dX chain from the dY_last the head GEMMs left in dA; then dW_i = [x_{i-1} | feat]^T dY_i per layer
      dYs = [self._buf(('bwd', slot, 'dYc', W, i), (M, W), bf16) for i in range(D - 1)] + [None]
      Bws = [None] + [self._w(plan, plan.packed[('trunk', i)]['b_off'], _rup(W, 128), plan.packed[('trunk', i)]['b_ld'])
                      for i in range(1, D)]
      ops.mlp_chain_bwd(None, None, mlp['bits'], Bws, dYs, M=M, W=W, dY_in=dA)
      dYs[D - 1] = dA
      for i, (d, concat) in enumerate(plan.trunk):
        inp, in_w, kv = (feat, plan.ldF, plan.F) if i == 0 else (acts[i - 1], W, W)
        ops.gemm_tn(inp, dYs[i], gslice(d.kernel_off, kv * W), M=M, K=in_w, N=W, lda=in_w, ldb=W, ldc=W,
                    k_valid=kv, n_valid=W, bias_out=gslice(d.bias_off, W), bias_n_valid=W)
        if concat:
          ops.gemm_tn(feat, dYs[i], gslice(d.kernel_off + W * W, plan.F * W), M=M, K=plan.ldF, N=W,
                      lda=plan.ldF, ldb=W, ldc=W, k_valid=plan.F, n_valid=W)
        if i == 0 or concat:
          feat_grad(i, dYs[i])
      return
    # trunk: per layer its dW (independent of the dX chain: on the dW stream when that switch is on), then the dX GEMM the
    # next layer waits for
    dy = dA
    pair = bool(_PAIR_DXDW and panel and self.device.type == 'cuda' and (M // 256) % 8 == 0)
    if pair:
      cur_s = torch.cuda.current_stream(self.device)
      if getattr(self, '_dw_stream', None) is None:
        self._dw_stream = torch.cuda.Stream(device=self.device)
        self._half_cus = max(8, torch.cuda.get_device_properties(self.device).multi_processor_count // 2 // 8 * 8)
        self._pair_events = {}
      dw_done = None

    def pair_event(kind, i):
      """One cached event per (kind, trunk layer): nothing is allocated per step."""
      ev = self._pair_events.get((kind, i))
      if ev is None:
        ev = self._pair_events[(kind, i)] = torch.cuda.Event()
      return ev

    try:
      for i in reversed(range(len(plan.trunk))):
        d, concat = plan.trunk[i]
        e = plan.packed[('trunk', i)]
        if t_extras is not None:
          ops.add_cols_bf16(dy, t_extras[i], dy, W)        # the tangent network's act'' term of this layer's pre-activation
        if pair and i > 0:
          # dW_i on the side stream, dX_i on this one, half the chip each, both ascending through M
          ready = pair_event('ready', i)
          ready.record(cur_s)
          self._dw_stream.wait_event(ready)
          with torch.cuda.stream(self._dw_stream):
            ops.gemm_tn(acts[i - 1], dy, gslice(d.kernel_off, W * W), M=M, K=W, N=W, lda=W, ldb=W, ldc=W,
                        bias_out=gslice(d.bias_off, W), bias_n_valid=W, m_interleave=True, max_wgs=self._half_cus, **tn_a, **tn_b)
            if concat:
              ops.gemm_tn(feat, dy, gslice(d.kernel_off + W * W, plan.F * W), M=M, K=plan.ldF, N=W,
                          lda=plan.ldF, ldb=W, ldc=W, k_valid=plan.F, n_valid=W, **tn_b)
            done = pair_event('done', i)
            done.record(self._dw_stream)
          if concat:
            feat_grad(i, dy, dy_panel=panel)
          if dw_done is not None:
            cur_s.wait_event(dw_done)                      # dX_i overwrites the buffer dW_{i+1} read its dY from
          Bw = self._w(plan, e['b_off'], _rup(W, 128), e['b_ld'])
          other = dy_buf(i - 1)
          ops.gemm_nt(dy, Bw, M=M, N=_rup(W, 128), K1=e['b_ld'], Cb=other, ldcb=W, nb=W, max_wgs=self._half_cus,
                      **mask_kw(i - 1), **lay_ac)
          dw_done = done
          dy = other
          continue
        if pair and dw_done is not None:
          cur_s.wait_event(dw_done)
          dw_done = None
        if i == 0:
          ops.gemm_tn(feat, dy, gslice(d.kernel_off, plan.F * W), M=M, K=plan.ldF, N=W, lda=plan.ldF, ldb=W,
                      ldc=W, k_valid=plan.F, n_valid=W, bias_out=gslice(d.bias_off, W), bias_n_valid=W, **tn_b)
        else:
          ops.gemm_tn(acts[i - 1], dy, gslice(d.kernel_off, W * W), M=M, K=W, N=W, lda=W, ldb=W, ldc=W,
                      bias_out=gslice(d.bias_off, W), bias_n_valid=W, **tn_a, **tn_b)
          if concat:
            ops.gemm_tn(feat, dy, gslice(d.kernel_off + W * W, plan.F * W), M=M, K=plan.ldF, N=W,
                        lda=plan.ldF, ldb=W, ldc=W, k_valid=plan.F, n_valid=W, **tn_b)
        if i == 0 or concat:
          feat_grad(i, dy, dy_panel=panel)
        if i > 0:
          Bw = self._w(plan, e['b_off'], _rup(W, 128), e['b_ld'])
          other = dy_buf(i - 1)
          ops.gemm_nt(dy, Bw, M=M, N=_rup(W, 128), K1=e['b_ld'], Cb=other, ldcb=W, nb=W, walk_descending=bool(i & 1),
                      **mask_kw(i - 1), **lay_ac)
          act_vjp(mlp['zs'][i - 1] if not relu else None, other)
          dy = other
    finally:
      # whatever path leaves the loop: the side stream's writes into `grads` (fp32 atomics of the last dW launch) are ordered
      # against what the caller enqueues next on this stream (the gradient all-reduce, clip + Adam)
      if pair and dw_done is not None:
        cur_s.wait_event(dw_done)

  @_in_library
  def backward_prop_levels(self, lvs, flat, grads, g_weights, losses):
    """`backward_level` for ALL proposal levels in one pass (grouped buffers, `_props_group`): each level's compositing
    VJP (with its losses) writes its slice of one head-gradient vector, then ONE head VJP, ONE dX chain and ONE
    weight-gradient GEMM per layer run over the L*M rows.  The levels share PropMLP_0 (reference models.py:120-121), so
    the sums over rows are the same sums, in a different order."""
    Lg = len(lvs)
    plan: MLPPlan = lvs[0]['plan']
    M, W, D = lvs[0]['M'], plan.W, len(plan.trunk)
    Mall = Lg * M
    R = self._saved['rays']
    g_all = self._buf(('bwd', 'g_props', Lg), (Mall,), f32)
    for k, lv in enumerate(lvs):
      assert lv['group'] == (k, Lg) and lv['M'] == M and lv['plan'] is plan and lv['mlp'].get('chain')
      ops.composite_bwd(lv['ccfg'], lv['raw_density'], lv['tdist'], R.directions, lv['weights'],
                        density_noise=lv['dnoise'], bg=lv['bg'], g_rgb_out=None, g_weights=g_weights[k], want_f32=True,
                        losses=losses[k], g_raw_density_out=g_all[k * M:(k + 1) * M].view(lv['raw_density'].shape))

    def whole(name, cols, dtype):
      return self._buf((('lvl', 'props', Lg), name), (Mall,) + cols, dtype)

    acts = [whole(('act', i), (W,), bf16) for i in range(D)]
    bits = [whole(('bits', i), (W // 8,), torch.uint8) for i in range(D)]
    feat = whole('feat', (plan.ldF,), bf16)
    d = plan.density
    w_head = flat[d.kernel_off:d.kernel_off + W]
    ops.small_head_bwd(acts[-1], W, g_all.view(Mall, 1), w_head.view(W, 1), M=Mall, K=W, Cn=1, dX=None,
                       relu_mask=False, dW=grads[d.kernel_off:d.kernel_off + W], db=grads[d.bias_off:d.bias_off + 1])
    r1 = _rank1_last(plan, D, W, False)
    dYs = [None if (r1 and i == D - 1) else self._buf(('bwd', 'dYc', W, i, 'props'), (Mall, W), bf16) for i in range(D)]
    Bws = [None] + [self._w(plan, plan.packed[('trunk', i)]['b_off'], _rup(W, 128), plan.packed[('trunk', i)]['b_ld'])
                    for i in range(1, D)]
    ops.mlp_chain_bwd(g_all, w_head, bits, Bws, dYs, M=Mall, W=W)
    for i, (dl, concat) in enumerate(plan.trunk):
      inp, in_w, kv = (feat, plan.ldF, plan.F) if i == 0 else (acts[i - 1], W, W)
      ops.gemm_tn(inp, dYs[i], grads[dl.kernel_off:dl.kernel_off + kv * W], M=Mall, K=in_w, N=W, lda=in_w, ldb=W,
                  ldc=W, k_valid=kv, n_valid=W, bias_out=grads[dl.bias_off:dl.bias_off + W], bias_n_valid=W,
                  rank1=(g_all, w_head, bits[i]) if dYs[i] is None else None)
      if concat:
        o = dl.kernel_off + W * W
        ops.gemm_tn(feat, dYs[i], grads[o:o + plan.F * W], M=Mall, K=plan.ldF, N=W, lda=plan.ldF, ldb=W, ldc=W,
                    k_valid=plan.F, n_valid=W)

  def _tangent_backward(self, plan, flat, grads, mlp, feat, M, g_raw_grad, slot, g_tfeat_out=None):
    """Backward pass through the tangent network T_l = bits_l * (T_{l-1} W_l), raw_grad = T_last w_density
    (the "double backward" of the density-gradient normals): the network is linear in each W_l with
    fixed masks, so dW_l += T_{l-1}^T G_l and G_{l-1} = bits_{l-1} * (G_l W_l^T) on 3*M rows."""
    W = plan.W
    M3 = 3 * M
    T_acts, T_feat, bits = mlp['T_acts'], mlp['T_feat'], mlp['bits']

    def gslice(off, size):
      return grads[off:off + size]

    gA = self._buf(('bwd', slot, 'gTA', W), (M3, W), bf16)
    gB = self._buf(('bwd', slot, 'gTB', W), (M3, W), bf16)
    d = plan.density
    relu = plan.hp.net_activation == 'relu'
    # Non-ReLU activations: T_l = act'(z_l) * U_l depends on the primal pre-activation too; the term
    # d loss / d z_l = sum_c G_l * U_l * act''(z_l) comes back as `extras[l]` and joins the primal backward pass (backward_level)
    extras = None if relu else [self._buf(('bwd', slot, 'T_extra', W, i), (M, W), bf16) for i in range(len(plan.trunk))]
    # G_last = bits_last * (g_raw_grad[:, None] w_density^T);  dW_density += T_last^T g_raw_grad
    ops.small_head_bwd(T_acts[-1], W, g_raw_grad.view(M3, 1), flat[d.kernel_off:d.kernel_off + W].view(W, 1),
                       M=M3, K=W, Cn=1, dX=gA, lddx=W, relu_mask=False, dW=gslice(d.kernel_off, W), db=None,
                       bits=bits[-1] if relu else None, bits_row_mod=M if relu else 0)
    gy, other = gA, gB
    D = len(plan.trunk)
    Gs = None
    if _TANGENT_CHAIN and relu and D >= 2 and self._chain_ok(plan) and all(b is not None for b in bits):
      # every G_i in one masked-linear chain launch per direction (`_TANGENT_CHAIN`), kept for the weight-gradient GEMMs below
      Gs = [self._buf(('bwd', slot, 'gT', W, i), (M3, W), bf16) for i in range(D - 1)] + [gA]
      Bws = [None] + [self._w(plan, plan.packed[('trunk', i)]['b_off'], _rup(W, 128), plan.packed[('trunk', i)]['b_ld'])
                      for i in range(1, D)]
      for c in range(3):
        rows = slice(c * M, (c + 1) * M)
        ops.mlp_chain_bwd(None, None, bits, Bws, [g[rows] for g in Gs[:-1]] + [None], M=M, W=W, dY_in=gA[rows])
    for i in reversed(range(len(plan.trunk))):
      d, concat = plan.trunk[i]
      e = plan.packed[('trunk', i)]
      if Gs is not None:
        gy = Gs[i]
      if not relu:
        ops.act_tangent_bwd(plan.hp.net_activation, mlp['zs'][i], mlp['T_pre'][i], gy, extras[i])     # gy: d loss / d T_i -> d loss / d U_i
      if g_tfeat_out is not None and (i == 0 or concat):
        # stop_level_grad = False: the tangent feature rows depend on the sample positions: d loss / d T_feat through this layer's
        # feature rows, G_i @ kernel_i[feature rows]^T -> [3 M, ldF] (the backward image `bf` of `feat_grad`)
        gT = self._buf(('bwd', slot, 'g_tfeat', len(g_tfeat_out)), (M3, plan.ldF), bf16)
        ops.gemm_nt(gy, self._w(plan, e['bf_off'], plan.ldF, e['bf_ld']), M=M3, N=plan.ldF, K1=e['bf_ld'], Cb=gT, ldcb=plan.ldF,
                    nb=plan.ldF)
        g_tfeat_out.append(gT)
      if i == 0:
        ops.gemm_tn(T_feat, gy, gslice(d.kernel_off, plan.F * W), M=M3, K=plan.ldF, N=W, lda=plan.ldF, ldb=W,
                    ldc=W, k_valid=plan.F, n_valid=W)
      else:
        ops.gemm_tn(T_acts[i - 1], gy, gslice(d.kernel_off, W * W), M=M3, K=W, N=W, lda=W, ldb=W, ldc=W)
        if concat:
          ops.gemm_tn(T_feat, gy, gslice(d.kernel_off + W * W, plan.F * W), M=M3, K=plan.ldF, N=W,
                      lda=plan.ldF, ldb=W, ldc=W, k_valid=plan.F, n_valid=W)
        if Gs is None:
          Bw = self._w(plan, e['b_off'], _rup(W, 128), e['b_ld'])
          ops.gemm_nt(gy, Bw, M=M3, N=_rup(W, 128), K1=e['b_ld'], Cb=other, ldcb=W, nb=W,
                      **(dict(bits_in=bits[i - 1], bits_row_mod=M) if relu else {}))
          gy, other = other, gy
    return extras


# =============================================================================


def construct_model(rng, rays, config, device='cuda'):
  """models.py:315-338: returns (model, init_variables).  `rng`: int seed (flax init RNG)."""
  model = Model(config=config)
  model.build(device)
  flat = model.init_flat_params(seed=0 if rng is None else int(rng))
  variables = {'flat': flat, 'params': model.params_tree(flat)}
  model.bind(variables)
  return model, variables


def render_image(render_fn, rays, rng, config, verbose=True, world_size=1, rank=0):
  """models.py:625-706: render all pixels of an image in chunks of config.render_chunk_size.

  `render_fn(rng, chunk_rays)` returns (renderings, ray_history) for this rank's slice, already
  gathered across ranks (train_utils.create_render_fn does the all-gather of pixel buffers)."""
  height, width = rays.origins.shape[:2]
  num_rays = height * width
  rays = rays.map(lambda r: r.reshape((num_rays, -1)))
  chunks = []
  idx0s = range(0, num_rays, config.render_chunk_size)
  for i_chunk, idx0 in enumerate(idx0s):
    if verbose and i_chunk % max(1, len(idx0s) // 10) == 0:
      print(f'Rendering chunk {i_chunk}/{len(idx0s)-1}')
    chunk_rays = rays.map(lambda r: r[idx0:idx0 + config.render_chunk_size])
    actual = chunk_rays.origins.shape[0]
    rem = actual % world_size
    padding = 0
    if rem != 0:
      padding = world_size - rem
      chunk_rays = chunk_rays.map(lambda r: torch.cat([r, r[-1:].expand(padding, r.shape[-1])], 0))  # mode='edge'
    per = chunk_rays.origins.shape[0] // world_size
    local = chunk_rays.map(lambda r: r[rank * per:(rank + 1) * per].contiguous())
    chunk_renderings, _ = render_fn(rng, local)
    if padding:
      chunk_renderings = [{k: (v[:-padding] if not k.startswith('ray_') else v) for k, v in r.items()}
                          for r in chunk_renderings]
    chunk_rendering = dict(chunk_renderings[-1])
    for k in chunk_renderings[0]:
      if k.startswith('ray_'):
        chunk_rendering[k] = [r[k] for r in chunk_renderings]
    chunks.append(chunk_rendering)
  rendering = {}
  for k in chunks[0]:
    if k.startswith('ray_'):
      rendering[k] = [torch.cat([c[k][lv] for c in chunks], 0) for lv in range(len(chunks[0][k]))]
    else:
      z = torch.cat([c[k] for c in chunks], 0)
      rendering[k] = z.reshape((height, width) + tuple(z.shape[1:]))
  keys = [k for k in rendering if k.startswith('ray_')]
  if keys:
    nr = rendering[keys[0]][0].shape[0]
    g = torch.Generator().manual_seed(0)          # the reference uses random.PRNGKey(0)
    ray_idx = torch.randperm(nr, generator=g)[:config.vis_num_rays].to(rendering[keys[0]][0].device)
    for k in keys:
      rendering[k] = [r[ray_idx] for r in rendering[k]]
  return rendering
