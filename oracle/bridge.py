"""TEST INFRASTRUCTURE (see oracle/__init__.py): bridging between the product's hyper-parameter objects and the oracle's.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""

import dataclasses

import torch


def oracle_hparams(model):
  """Product hyper-parameter objects -> the oracle's dataclasses (same field names)."""
  from oracle import models as omodels
  def conv(src, cls):
    names = {f.name for f in dataclasses.fields(cls)}
    kw = {f.name: getattr(src, f.name) for f in dataclasses.fields(src) if f.name in names and f.name != 'config'}
    return cls(**kw)
  om = conv(model, omodels.Model)
  if model.config is not None:
    om.vis_num_rays = model.config.vis_num_rays
  return om, conv(model.nerf_hp, omodels.MLP), (None if model.single_mlp else conv(model.prop_hp, omodels.MLP))


def make_noise(model, B, seed=0):
  g = torch.Generator().manual_seed(seed)
  noise = {'u_jitter': {}, 'density_noise': {}, 'bg_rgbs': {}}
  for i in range(model.num_levels):
    n = model.num_prop_samples if i < model.num_levels - 1 else model.num_nerf_samples
    noise['u_jitter'][i] = torch.rand((B, 1 if model.single_jitter else n), generator=g)
    noise['density_noise'][i] = torch.randn((B, n), generator=g)
    noise['bg_rgbs'][i] = torch.rand((B, 3), generator=g)
  # bottleneck noise (models.py:530-533) from its own generator: the draws above stay what they were without it
  g2 = torch.Generator().manual_seed(seed + 1000)
  for i in range(model.num_levels):
    is_prop = i < model.num_levels - 1
    hp = model.prop_hp if is_prop else model.nerf_hp
    if hp.bottleneck_noise > 0 and not hp.disable_rgb and model.use_viewdirs:
      n = model.num_prop_samples if is_prop else model.num_nerf_samples
      noise.setdefault('bottleneck_noise', {})[i] = torch.randn((B, n, hp.bottleneck_width), generator=g2)
  return noise
