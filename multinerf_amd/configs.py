"""`Config` + config loading: the reference's gin surface (internal/configs.py).

Every field of the reference's `Config` dataclass (configs.py:49-172) is kept
with the same name and default so that existing .gin files and --gin_bindings
load unchanged; fields that only drive out-of-scope subsystems (datasets,
render paths, video, checkpoints -- SURVEY.md section 2) are carried but unread.
"""

import dataclasses
from typing import Any, Dict, Optional, Tuple

from multinerf_amd import gin


@gin.configurable
@dataclasses.dataclass
class Config:
  """Configuration flags (names/defaults: reference internal/configs.py:49-172)."""
  dataset_loader: str = 'llff'
  batching: str = 'all_images'
  batch_size: int = 16384
  patch_size: int = 1
  factor: int = 0
  load_alphabetical: bool = True
  forward_facing: bool = False
  render_path: bool = False
  llffhold: int = 8
  llff_use_all_images_for_training: bool = False
  use_tiffs: bool = False
  compute_disp_metrics: bool = False
  compute_normal_metrics: bool = False
  gc_every: int = 10000
  disable_multiscale_loss: bool = False
  randomized: bool = True
  near: float = 2.
  far: float = 6.
  checkpoint_dir: Optional[str] = None
  render_dir: Optional[str] = None
  data_dir: Optional[str] = None
  vocab_tree_path: Optional[str] = None
  render_chunk_size: int = 16384
  num_showcase_images: int = 5
  deterministic_showcase: bool = True
  vis_num_rays: int = 16
  vis_decimate: int = 0

  # train
  max_steps: int = 250000
  early_exit_steps: Optional[int] = None
  checkpoint_every: int = 25000
  print_every: int = 100
  train_render_every: int = 5000
  cast_rays_in_train_step: bool = False
  data_loss_type: str = 'charb'
  charb_padding: float = 0.001
  data_loss_mult: float = 1.0
  data_coarse_loss_mult: float = 0.
  interlevel_loss_mult: float = 1.0
  orientation_loss_mult: float = 0.0
  orientation_coarse_loss_mult: float = 0.0
  robustnerf_inlier_quantile: float = 0.5
  enable_robustnerf_loss: bool = False
  robustnerf_inner_patch_size: int = 8
  robustnerf_smoothed_filter_size: int = 3
  robustnerf_smoothed_inlier_quantile: float = 0.5
  robustnerf_inner_patch_inlier_quantile: float = 0.5
  orientation_loss_target: str = 'normals_pred'
  predicted_normal_loss_mult: float = 0.0
  predicted_normal_coarse_loss_mult: float = 0.0
  weight_decay_mults: Dict[str, Any] = dataclasses.field(default_factory=dict)

  lr_init: float = 0.002
  lr_final: float = 0.00002
  lr_delay_steps: int = 512
  lr_delay_mult: float = 0.01
  adam_beta1: float = 0.9
  adam_beta2: float = 0.999
  adam_eps: float = 1e-6
  grad_max_norm: float = 0.001
  grad_max_val: float = 0.
  distortion_loss_mult: float = 0.01

  # eval
  eval_only_once: bool = True
  eval_save_output: bool = True
  eval_save_ray_data: bool = False
  eval_render_interval: int = 1
  eval_dataset_limit: int = 2**31 - 1
  eval_quantize_metrics: bool = True
  eval_crop_borders: int = 0

  # render
  render_video_fps: int = 60
  render_video_crf: int = 18
  render_path_frames: int = 120
  z_variation: float = 0.
  z_phase: float = 0.
  render_dist_percentile: float = 0.5
  render_dist_curve_fn: str = 'log'
  render_path_file: Optional[str] = None
  render_job_id: int = 0
  render_num_jobs: int = 1
  render_resolution: Optional[Tuple[int, int]] = None
  render_focal: Optional[float] = None
  render_camtype: Optional[str] = None
  render_spherical: bool = False
  render_save_async: bool = True
  render_spline_keyframes: Optional[str] = None
  render_spline_n_interp: int = 30
  render_spline_degree: int = 5
  render_spline_smoothness: float = .03
  render_spline_interpolate_exposure: bool = False

  # raw
  rawnerf_mode: bool = False
  exposure_percentile: float = 97.
  num_border_pixels_to_mask: int = 0
  apply_bayer_mask: bool = False
  autoexpose_renders: bool = False
  eval_raw_affine_cc: bool = False


def load_config(gin_configs=None, gin_bindings=None, save_config=False):
  """internal/configs.py:183-192 (flags become arguments; absl is absent)."""
  gin.parse_config_files_and_bindings(gin_configs, gin_bindings, skip_unknown=True)
  config = Config()
  if save_config and config.checkpoint_dir:
    import os
    os.makedirs(config.checkpoint_dir, exist_ok=True)
    with open(os.path.join(config.checkpoint_dir, 'config.gin'), 'w') as f:
      f.write(gin.config_str())
  return config


# The BASELINE.json configs, as gin text this package ships (the user's own
# .gin files load through `load_config(gin_configs=[path])` just the same).
PRESETS = {
    # reference configs/360.gin
    '360': """
Config.dataset_loader = 'llff'
Config.near = 0.2
Config.far = 1e6
Config.factor = 4
Model.raydist_fn = @jnp.reciprocal
Model.opaque_background = True
PropMLP.warp_fn = @coord.contract
PropMLP.net_depth = 4
PropMLP.net_width = 256
PropMLP.disable_density_normals = True
PropMLP.disable_rgb = True
NerfMLP.warp_fn = @coord.contract
NerfMLP.net_depth = 8
NerfMLP.net_width = 1024
NerfMLP.disable_density_normals = True
""",
    # reference configs/blender_256.gin
    'blender_256': """
Config.dataset_loader = 'blender'
Config.batching = 'single_image'
Config.near = 2
Config.far = 6
Config.eval_render_interval = 5
Config.data_loss_type = 'mse'
Config.adam_eps = 1e-8
Model.num_levels = 2
Model.num_prop_samples = 128
Model.num_nerf_samples = 32
PropMLP.net_depth = 4
PropMLP.net_width = 256
PropMLP.basis_shape = 'octahedron'
PropMLP.basis_subdivisions = 1
PropMLP.disable_density_normals = True
PropMLP.disable_rgb = True
NerfMLP.net_depth = 8
NerfMLP.net_width = 256
NerfMLP.basis_shape = 'octahedron'
NerfMLP.basis_subdivisions = 1
NerfMLP.disable_density_normals = True
Config.distortion_loss_mult = 0.
NerfMLP.max_deg_point = 16
PropMLP.max_deg_point = 16
""",
    # reference configs/blender_refnerf.gin
    'blender_refnerf': """
Config.dataset_loader = 'blender'
Config.batching = 'single_image'
Config.near = 2
Config.far = 6
Config.eval_render_interval = 5
Config.compute_normal_metrics = True
Config.data_loss_type = 'mse'
Config.distortion_loss_mult = 0.0
Config.orientation_loss_mult = 0.1
Config.orientation_loss_target = 'normals_pred'
Config.predicted_normal_loss_mult = 3e-4
Config.orientation_coarse_loss_mult = 0.01
Config.predicted_normal_coarse_loss_mult = 3e-5
Config.interlevel_loss_mult = 0.0
Config.data_coarse_loss_mult = 0.1
Config.adam_eps = 1e-8
Model.num_levels = 2
Model.single_mlp = True
Model.num_prop_samples = 128
Model.num_nerf_samples = 128
Model.anneal_slope = 0.
Model.dilation_multiplier = 0.
Model.dilation_bias = 0.
Model.single_jitter = False
Model.resample_padding = 0.01
NerfMLP.net_depth = 8
NerfMLP.net_width = 256
NerfMLP.net_depth_viewdirs = 8
NerfMLP.basis_shape = 'octahedron'
NerfMLP.basis_subdivisions = 1
NerfMLP.disable_density_normals = False
NerfMLP.enable_pred_normals = True
NerfMLP.use_directional_enc = True
NerfMLP.use_reflections = True
NerfMLP.deg_view = 5
NerfMLP.enable_pred_roughness = True
NerfMLP.use_diffuse_color = True
NerfMLP.use_specular_tint = True
NerfMLP.use_n_dot_v = True
NerfMLP.bottleneck_width = 128
NerfMLP.density_bias = 0.5
NerfMLP.max_deg_point = 16
""",
    # reference configs/llff_raw.gin
    'llff_raw': """
Config.dataset_loader = 'llff'
Config.near = 0.
Config.far = 1.
Config.factor = 4
Config.forward_facing = True
Model.ray_shape = 'cylinder'
PropMLP.net_depth = 4
PropMLP.net_width = 256
PropMLP.basis_shape = 'octahedron'
PropMLP.basis_subdivisions = 1
PropMLP.disable_density_normals = True
PropMLP.disable_rgb = True
NerfMLP.net_depth = 8
NerfMLP.net_width = 256
NerfMLP.basis_shape = 'octahedron'
NerfMLP.basis_subdivisions = 1
NerfMLP.disable_density_normals = True
NerfMLP.max_deg_point = 16
PropMLP.max_deg_point = 16
Config.train_render_every = 5000
Config.rawnerf_mode = True
Config.data_loss_type = 'rawnerf'
Config.apply_bayer_mask = True
Model.learned_exposure_scaling = True
Model.num_levels = 2
Model.num_prop_samples = 128
Model.num_nerf_samples = 128
Model.opaque_background = True
NerfMLP.rgb_padding = 0.
NerfMLP.rgb_activation = @math.safe_exp
NerfMLP.rgb_bias = -5.
PropMLP.rgb_padding = 0.
PropMLP.rgb_activation = @math.safe_exp
PropMLP.rgb_bias = -5.
Config.interlevel_loss_mult = .0
Config.distortion_loss_mult = .01
Config.orientation_loss_mult = 0.
Config.data_coarse_loss_mult = 0.1
NerfMLP.density_noise = 1.
PropMLP.density_noise = 1.
Model.single_mlp = True
Model.anneal_slope = 0.
Model.dilation_multiplier = 0.
Model.dilation_bias = 0.
Model.single_jitter = False
NerfMLP.weight_init = 'glorot_uniform'
PropMLP.weight_init = 'glorot_uniform'
Config.batch_size = 16384
Config.render_chunk_size = 16384
Config.lr_init = 1e-3
Config.lr_final = 1e-5
Config.max_steps = 500000
Config.checkpoint_every = 25000
Config.lr_delay_steps = 2500
Config.lr_delay_mult = 0.01
Config.grad_max_norm = 0.1
Config.grad_max_val = 0.1
Config.adam_eps = 1e-8
""",
}


# reference configs/debug.gin: an OVERLAY (the reference passes it as a second --gin_configs file): a short schedule and
# small MLPs, among them PropMLP.net_width = 64.  load_preset('360+debug') binds it on top of a scene preset.
PRESETS['debug'] = """
Config.checkpoint_every = 1000
Config.print_every = 100
Config.train_render_every = 1000
Config.lr_delay_mult = 0.1
Config.lr_delay_steps = 500
Config.batch_size = 2048
Config.render_chunk_size = 2048
Config.lr_init = 5e-4
Config.lr_final = 5e-6
Config.factor = 4
Config.early_exit_steps = 3000
PropMLP.net_depth = 2
PropMLP.net_width = 64
NerfMLP.net_depth = 4
NerfMLP.net_width = 128
"""


def load_preset(name, gin_bindings=None):
  """Clear gin state, bind a named preset or several ('360+debug': left to right, like repeated --gin_configs) and the
  extra bindings, return Config."""
  gin.clear_config()
  for part in name.split('+'):
    gin.parse_config(PRESETS[part], skip_unknown=False)
  for b in (gin_bindings or []):
    gin.parse_config(b, skip_unknown=False)
  return Config()


# Importing this module must register every configurable the .gin files bind (gin drops bindings of
# unknown configurables when skip_unknown=True, internal/configs.py:185-186): pull in Model/NerfMLP/PropMLP.
from multinerf_amd import models as _models  # noqa: E402,F401
