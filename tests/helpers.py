"""Shared test helpers: seeded synthetic rays and oracle<->product config bridging."""

import dataclasses

import numpy as np
import torch

from multinerf_amd import configs, gin, models, utils
from oracle import models as omodels


def synthetic_rays(B, seed=20200823, near=0.2, far=1e6, device='cpu'):
  """SURVEY.md section 8(d) 'Ray inputs (seeded)'."""
  rs = np.random.default_rng(seed)
  o = rs.uniform(-1, 1, (B, 3))
  tgt = rs.normal(0, 0.3, (B, 3))
  d = tgt - o
  d = d / np.linalg.norm(d, axis=-1, keepdims=True) * rs.uniform(1.0, 1.2, (B, 1))
  vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
  radii = rs.uniform(3e-4, 1e-3, (B, 1))
  ip = rs.uniform(-0.6, 0.6, (B, 2))
  cam = rs.integers(0, 200, (B, 1))
  rgb = rs.uniform(0, 1, (B, 3))
  f = lambda x: torch.as_tensor(x, dtype=torch.float32, device=device).contiguous()
  rays = utils.Rays(origins=f(o), directions=f(d), viewdirs=f(vd), radii=f(radii), imageplane=f(ip),
                    lossmult=f(np.ones((B, 1))), near=f(np.full((B, 1), near)), far=f(np.full((B, 1), far)),
                    cam_idx=torch.as_tensor(cam, dtype=torch.int32, device=device))
  return utils.Batch(rays=rays, rgb=f(rgb))


def oracle_hparams(model: models.Model):
  """Product hyper-parameter objects -> the oracle's dataclasses (same field names)."""
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
  return noise
