"""ctypes binding of include/mnerf.h (cffi is not installed in this image).

The product path has NO CPU fallback: if libmnerf_hip.so is missing or a symbol
declared in the header is absent, importing/using the ops raises.
"""

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
# MNR_LIB_PATH: load another build of the same ABI (same-box A/B of two kernel versions; never a different implementation)
LIB_PATH = os.environ.get('MNR_LIB_PATH') or os.path.join(HERE, 'libmnerf_hip.so')
# The fp32-Dense DEBUG build (multinerf_amd/build.py, csrc/common.h MNR_DENSE_F32): the same kernel sources with float storage and
# plain fp32 Dense layers, loaded on demand for Model(dense_precision='fp32') and active inside `dense_f32()` only.
LIB_F32_PATH = os.path.join(HERE, 'libmnerf_hip_f32.so')
HEADER_PATH = os.path.join(os.path.dirname(HERE), 'include', 'mnerf.h')
# entry points of the layout-specific MFMA files that the fp32 build does not have (the host never calls them in that mode)
F32_ABSENT = ('mnr_mlp_chain_fwd', 'mnr_mlp_chain_fwd_ipe', 'mnr_mlp_chain_bwd')


def describe():
  """Which build is loaded: bench.py records it next to its numbers (an A/B library loaded through MNR_LIB_PATH or a
  build with extra hipcc flags must not pass for the shipped kernels)."""
  stamp = LIB_PATH + '.stamp'
  digest = None
  if os.path.exists(stamp):
    with open(stamp) as f:
      digest = f.read().strip()
  return {'path': os.path.relpath(LIB_PATH, os.path.dirname(HERE)), 'overridden': bool(os.environ.get('MNR_LIB_PATH')),
          'source_digest': digest, 'extra_hipcc_flags': os.environ.get('MNR_EXTRA_HIPCC_FLAGS', '')}

MNR_OK = 0
MNR_ERR_INVALID_ARGUMENT = -1

# enums of mnerf.h
RAYDIST = {None: 0, 'identity': 0, 'reciprocal': 1, 'piecewise': 2, 'log': 3, 'exp': 4, 'sqrt': 5,
           'square': 6}
ACT = {'sigmoid': 0, 'safe_exp': 1, 'softplus': 2, 'exp': 3, 'relu': 4, 'identity': 5}
DATA_LOSS = {'mse': 0, 'charb': 1, 'rawnerf': 2}
NET_ACT = {'softplus': 1, 'silu': 2}          # mnr_act_*_bf16 kinds ('relu' lives in the GEMM epilogues)

c_f32p = C.POINTER(C.c_float)
c_u16p = C.POINTER(C.c_uint16)
c_i32p = C.POINTER(C.c_int32)
vp = C.c_void_p


class ResampleCfg(C.Structure):
  _fields_ = [('n_prev', C.c_int), ('n_samples', C.c_int), ('use_dilation', C.c_int),
              ('dilation', C.c_float), ('domain_lo', C.c_float), ('domain_hi', C.c_float),
              ('anneal', C.c_float), ('resample_padding', C.c_float), ('single_jitter', C.c_int),
              ('max_jitter', C.c_float), ('raydist_fn', C.c_int)]


class IpeCfg(C.Structure):
  _fields_ = [('ray_shape', C.c_int), ('warp_contract', C.c_int), ('disable_integration', C.c_int),
              ('basis_k', C.c_int), ('min_deg', C.c_int), ('max_deg', C.c_int)]


class GemmNTArgs(C.Structure):
  _fields_ = [('A1', vp), ('lda1', C.c_int), ('K1', C.c_int),
              ('A2', vp), ('lda2', C.c_int), ('K2', C.c_int),
              ('Bt', vp), ('ldb', C.c_int),
              ('M', C.c_int64), ('N', C.c_int),
              ('bias', vp), ('n_bias', C.c_int), ('relu', C.c_int),
              ('mask', vp), ('ldmask', C.c_int),
              ('Cb', vp), ('ldcb', C.c_int), ('nb', C.c_int),
              ('Cf', vp), ('ldcf', C.c_int), ('f0', C.c_int), ('nf', C.c_int),
              ('mask_bits_out', vp), ('ld_bits_out', C.c_int),
              ('mask_bits_in', vp), ('ld_bits_in', C.c_int),
              ('bits_row_mod', C.c_int64), ('a1_layout', C.c_int), ('c_layout', C.c_int),
              ('vcol', vp), ('vcol_out', vp), ('vcol_bias', vp), ('walk_descending', C.c_int), ('max_wgs', C.c_int)]


class GemmTNArgs(C.Structure):
  _fields_ = [('A', vp), ('lda', C.c_int), ('K', C.c_int),
              ('B', vp), ('ldb', C.c_int), ('N', C.c_int),
              ('M', C.c_int64), ('C', vp), ('ldc', C.c_int),
              ('k_valid', C.c_int), ('n_valid', C.c_int),
              ('bias_out', vp), ('bias_n_valid', C.c_int), ('gcol', vp), ('gcol_out', vp),
              ('a_layout', C.c_int), ('b_layout', C.c_int), ('m_interleave', C.c_int), ('max_wgs', C.c_int),
              ('rank1_g', vp), ('rank1_w', vp), ('rank1_bits', vp), ('ld_rank1_bits', C.c_int)]


CHAIN_MAX_DEPTH = 8


class MlpChainFwdArgs(C.Structure):
  _fields_ = [('M', C.c_int64), ('W', C.c_int), ('depth', C.c_int),
              ('feat', vp), ('ld_feat', C.c_int), ('K0', C.c_int),
              ('Bt', vp * CHAIN_MAX_DEPTH), ('ldb', C.c_int * CHAIN_MAX_DEPTH), ('bias', vp * CHAIN_MAX_DEPTH),
              ('w_head', vp), ('b_head', vp), ('head_out', vp),
              ('acts', vp * CHAIN_MAX_DEPTH), ('bits', vp * CHAIN_MAX_DEPTH), ('skip_layer', C.c_int)]


CHAIN_IPE_GROUP_COLS = 192          # MNR_CHAIN_IPE_GROUP_COLS


class ChainIpeArgs(C.Structure):
  _fields_ = [('cfg', IpeCfg), ('n', C.c_int), ('tdist', vp), ('origins', vp), ('directions', vp), ('radii', vp),
              ('basis', vp)]


class MlpChainBwdArgs(C.Structure):
  _fields_ = [('M', C.c_int64), ('W', C.c_int), ('depth', C.c_int),
              ('g_head', vp), ('w_head', vp),
              ('bits', vp * CHAIN_MAX_DEPTH), ('Bw', vp * CHAIN_MAX_DEPTH), ('ldb', C.c_int * CHAIN_MAX_DEPTH),
              ('dY', vp * CHAIN_MAX_DEPTH), ('dY_in', vp)]


class PackDesc(C.Structure):
  _fields_ = [('src_off', C.c_int64), ('rows_in', C.c_int), ('cols_out', C.c_int),
              ('dst_off', C.c_int64), ('ld', C.c_int), ('row0', C.c_int), ('col0', C.c_int),
              ('transpose', C.c_int)]


class CompositeCfg(C.Structure):
  _fields_ = [('n', C.c_int), ('opaque_background', C.c_int), ('density_act', C.c_int),
              ('density_bias', C.c_float), ('density_noise_std', C.c_float), ('has_rgb', C.c_int),
              ('rgb_act', C.c_int), ('rgb_premultiplier', C.c_float), ('rgb_bias', C.c_float),
              ('rgb_padding', C.c_float), ('bg_mode', C.c_int), ('bg_value', C.c_float)]


class LevelBwdArgs(C.Structure):
  _fields_ = [('cfg', CompositeCfg), ('B', C.c_int64), ('B_valid', C.c_int64),
              ('raw_density', vp), ('density_noise', vp), ('raw_rgb', vp), ('tdist', vp), ('dirs', vp),
              ('bg', vp), ('exposure_scale', vp), ('weights', vp), ('g_rgb_out', vp), ('g_weights', vp),
              ('g_raw_density', vp), ('g_raw_density_bf16', vp), ('ld_bf16', C.c_int), ('g_raw_rgb', vp),
              ('g_exposure_scale', vp),
              ('data_loss_type', C.c_int), ('charb_padding', C.c_float), ('data_loss_mult', C.c_float),
              ('rgb_out', vp), ('gt', vp), ('lossmult', vp), ('lm_c', C.c_int), ('denom', vp), ('data_stats', vp),
              ('wloss_mode', C.c_int), ('wloss_mult', C.c_float), ('sdist', vp), ('n_ref', C.c_int), ('t_ref', vp),
              ('w_ref', vp), ('wloss_stat', vp), ('g_x', vp)]


class SdistBwdArgs(C.Structure):
  _fields_ = [('B', C.c_int64), ('B_valid', C.c_int64), ('n', C.c_int),
              ('sdist', vp), ('near', vp), ('far', vp), ('raydist_fn', C.c_int),
              ('g_x', vp), ('raw_density', vp), ('density_noise', vp), ('density_noise_std', C.c_float),
              ('density_bias', C.c_float), ('density_act', C.c_int), ('dirs', vp),
              ('g_t0', vp), ('g_t1', vp), ('distortion_mult', C.c_float), ('weights', vp),
              ('g_sdist_in', vp), ('g_sdist', vp)]


class IdeTables(C.Structure):
  _fields_ = [('T', C.c_int), ('lmax', C.c_int), ('m', vp), ('l', vp), ('sigma', vp), ('mat', vp)]


class AdamCfg(C.Structure):
  _fields_ = [('lr', C.c_float), ('b1', C.c_float), ('b2', C.c_float), ('eps', C.c_float),
              ('bias_corr1', C.c_float), ('bias_corr2', C.c_float), ('grad_max_val', C.c_float),
              ('grad_max_norm', C.c_float)]


i64, i32, f32 = C.c_int64, C.c_int, C.c_float
_PROTOS = {
    'mnr_abi_version': ([], i32),
    'mnr_device_info': ([i32, C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32], i32),
    'mnr_resample_level': ([C.POINTER(ResampleCfg), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp], i32),
    'mnr_resample_level_bwd': ([C.POINTER(ResampleCfg), i64, vp, vp, vp, vp, vp, vp, vp, vp], i32),
    'mnr_sorted_interp': ([i64, i32, i32, vp, vp, vp, vp, vp, vp], i32),
    'mnr_max_dilate_weights': ([i64, i32, vp, vp, f32, f32, f32, vp, vp, vp, vp], i32),
    'mnr_cast_rays_ipe': ([C.POINTER(IpeCfg), i64, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp], i32),
    'mnr_cast_rays_ipe_f32': ([C.POINTER(IpeCfg), i64, i32, vp, vp, vp, vp, vp, vp, vp], i32),
    'mnr_cast_rays_ipe_tangent': ([C.POINTER(IpeCfg), i64, i32, vp, vp, vp, vp, vp, vp, i32, vp], i32),
    'mnr_cast_rays_ipe_bwd': ([C.POINTER(IpeCfg), i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp], i32),
    'mnr_cast_rays_ipe_tangent_bwd': ([C.POINTER(IpeCfg), i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp], i32),
    'mnr_sdist_bwd': ([C.POINTER(SdistBwdArgs), vp], i32),
    'mnr_viewdir_enc_fill': ([i64, i32, vp, i32, vp, i32, i32, i32, vp], i32),
    'mnr_pixels_to_rays': ([i64, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp], i32),
    'mnr_glo_fill': ([i64, i32, i32, vp, vp, i32, vp, i32, i32, vp], i32),
    'mnr_glo_bwd': ([i64, i32, i32, vp, vp, vp, i32, vp, vp], i32),
    'mnr_gemm_nt_bf16': ([C.POINTER(GemmNTArgs), vp], i32),
    'mnr_gemm_tn_bf16': ([C.POINTER(GemmTNArgs), vp], i32),
    'mnr_mlp_chain_fwd': ([C.POINTER(MlpChainFwdArgs), vp], i32),
    'mnr_mlp_chain_bwd': ([C.POINTER(MlpChainBwdArgs), vp], i32),
    'mnr_mlp_chain_fwd_ipe': ([C.POINTER(MlpChainFwdArgs), C.POINTER(ChainIpeArgs), vp], i32),
    'mnr_colsum_bf16': ([vp, i32, i64, i32, vp, vp], i32),
    'mnr_pack_weights_bf16': ([vp, vp, i32, i32, vp, vp], i32),
    'mnr_scatter_add_f32': ([vp, i32, i32, i32, i32, i32, vp, i32, vp], i32),
    'mnr_act_fwd_bf16': ([i32, i64, vp, vp, vp], i32),
    'mnr_act_bwd_bf16': ([i32, i64, vp, vp, vp], i32),
    'mnr_act_tangent_fwd_bf16': ([i32, i64, vp, vp, vp, vp], i32),
    'mnr_act_tangent_bwd_bf16': ([i32, i64, vp, vp, vp, vp, vp], i32),
    'mnr_add_noise_bf16': ([i64, i32, vp, i32, vp, f32, vp], i32),
    'mnr_cast_f32_to_bf16': ([vp, i32, i64, i32, vp, i32, i32, vp], i32),
    'mnr_small_head_bwd': ([i64, i32, i32, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, i64, vp, i64, vp], i32),
    'mnr_composite_fwd': ([C.POINTER(CompositeCfg), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp], i32),
    'mnr_composite_bwd': ([C.POINTER(CompositeCfg), i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32,
                           vp, vp, vp], i32),
    'mnr_level_bwd': ([C.POINTER(LevelBwdArgs), vp], i32),
    'mnr_exposure_scale': ([i64, vp, vp, vp, vp, vp], i32),
    'mnr_exposure_scale_bwd': ([i64, vp, vp, vp, vp, vp], i32),
    'mnr_render_extras': ([i64, i32, vp, vp, vp, vp, vp], i32),
    'mnr_ref_head_fwd': ([i64, i32, vp, vp, vp, C.POINTER(IdeTables), i32, i32, f32, vp, i32, i32, i32, vp, vp, vp, vp], i32),
    'mnr_ref_head_bwd': ([i64, i32, vp, vp, vp, C.POINTER(IdeTables), i32, i32, f32, vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, vp, vp],
                         i32),
    'mnr_add_cols_bf16': ([i64, i32, vp, i32, vp, i32, vp, i32, vp], i32),
    'mnr_ref_color_fwd': ([i64, vp, vp, f32, f32, f32, i32, vp, vp], i32),
    'mnr_ref_color_bwd': ([i64, vp, vp, f32, f32, f32, i32, vp, vp, vp, i32, i32, i32, vp], i32),
    'mnr_ref_losses': ([i64, i32, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp], i32),
    'mnr_weighted_sum': ([i64, i32, i32, vp, vp, vp, vp], i32),
    'mnr_density_normals_fwd': ([i64, vp, vp, vp], i32),
    'mnr_density_normals_bwd': ([i64, vp, vp, vp, vp], i32),
    'mnr_pred_normals_fwd': ([i64, vp, i32, i32, vp, vp], i32),
    'mnr_pred_normals_bwd': ([i64, vp, i32, i32, vp, vp, i32, i32, vp], i32),
    'mnr_lossmult_sum': ([i64, vp, i32, vp, vp], i32),
    'mnr_render_metrics': ([i64, vp, vp, vp, vp, vp, vp, vp, vp, vp], i32),
    'mnr_data_loss': ([i32, f32, f32, i64, i64, vp, vp, vp, i32, vp, vp, vp, vp], i32),
    'mnr_interlevel_loss': ([f32, i64, i64, i32, vp, vp, i32, vp, vp, vp, vp, vp], i32),
    'mnr_distortion_loss': ([f32, i64, i64, i32, vp, vp, vp, vp, vp], i32),
    'mnr_lossfun_outer': ([i64, i32, vp, vp, i32, vp, vp, vp, vp], i32),
    'mnr_lossfun_distortion': ([i64, i32, vp, vp, vp, vp], i32),
    'mnr_weight_decay': ([vp, i64, i64, f32, vp, vp, vp, vp], i32),
    'mnr_grad_sqnorm': ([vp, i64, i64, f32, vp, vp], i32),
    'mnr_clip_adam': ([C.POINTER(AdamCfg), i64, i64, vp, vp, vp, vp, vp, vp], i32),
}

# include/mnerf_debug.h: development / test hooks (process-global A/B switches, small-grid test hooks, timelines).  The
# package never calls them; tests/ and tools/ reach them through debug().
_DEBUG_PROTOS = {
    'mnr_debug_gemm_timeline': ([vp], i32),
    'mnr_gemm_nt_set_persistent': ([i32], i32),
    'mnr_level_bwd_set_quad': ([i32], i32),
    'mnr_gemm_nt_set_pipelined': ([i32], i32),
    'mnr_gemm_nt_set_wres': ([i32], i32),
    'mnr_gemm_nt_panel_set_max_wgs': ([i32], i32),
    'mnr_debug_chain_timeline': ([vp], i32),
    'mnr_mlp_chain_set_deferred': ([i32], i32),
    'mnr_mlp_chain_set_max_wgs': ([i32], i32),
}

_lib = None


def header_symbols(path=None):
  """Every function name include/mnerf.h (or the header at `path`) declares."""
  with open(path or HEADER_PATH) as f:
    text = f.read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(mnr_[a-z0-9_]+)\s*\(', text)))


def _bind(path, absent=()):
  if not os.path.exists(path):
    raise RuntimeError(
        f'{path} is missing: the HIP extension has not been built. Run '
        '`python -m multinerf_amd.build` (needs hipcc, ROCm >= 7.0). There is no CPU fallback.')
  # torch must load ITS HIP runtime first: libmnerf_hip.so then binds to that same libamdhip64
  # instead of pulling a second copy from /opt/rocm (two runtimes in one process cannot share the device).
  import torch  # noqa: F401
  lib = C.CDLL(path)
  lib.mnr_last_error.restype = C.c_char_p
  lib.mnr_last_error.argtypes = []
  for name, (argtypes, restype) in _PROTOS.items():
    if name in absent:
      continue
    fn = getattr(lib, name)      # AttributeError if the symbol is missing
    fn.argtypes = argtypes
    fn.restype = restype
  return lib


_lib_f32 = None
_f32_depth = 0                    # > 0: inside dense_f32(): load() hands out the fp32-Dense debug build


def load():
  """Load libmnerf_hip.so (inside `dense_f32()`: libmnerf_hip_f32.so); raises (never falls back) when it is unavailable."""
  global _lib, _lib_f32
  if _f32_depth > 0:
    if _lib_f32 is None:
      _lib_f32 = _bind(LIB_F32_PATH, F32_ABSENT)
    return _lib_f32
  if _lib is None:
    _lib = _bind(LIB_PATH)
  return _lib


def f32_active():
  return _f32_depth > 0


class dense_f32:
  """Context: the C-ABI calls inside go to the fp32-Dense debug build, and what include/mnerf.h calls a bf16 matrix (uint16_t*)
  is a float matrix.  Re-entrant; `on=False` makes it a no-op (so that callers can write `with model.library():`)."""

  def __init__(self, on=True):
    self.on = bool(on)

  def __enter__(self):
    global _f32_depth
    if self.on:
      _f32_depth += 1
    return self

  def __exit__(self, *exc):
    global _f32_depth
    if self.on:
      _f32_depth -= 1
    return False


DEBUG_HEADER_PATH = os.path.join(os.path.dirname(HEADER_PATH), 'mnerf_debug.h')


def debug(handle=None):
  """The loaded library with the hooks of include/mnerf_debug.h bound (tests / tools only)."""
  lib = handle if handle is not None else load()
  for name, (argtypes, restype) in _DEBUG_PROTOS.items():
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
  return lib


def check(status):
  if status == MNR_OK:
    return
  msg = load().mnr_last_error().decode()
  if status == MNR_ERR_INVALID_ARGUMENT:
    raise ValueError(msg)
  raise RuntimeError(f'libmnerf_hip status {status}: {msg}')
