"""Seeded synthetic ray batches: the benchmark's and the tests' inputs.

No dataset ships with the image, so bench.py, __graft_entry__.smoke() and the parity tests all draw their rays
from here.  Shapes, ranges and the seed follow SURVEY.md 8(d); nothing here touches the oracle.
"""

import numpy as np
import torch

from multinerf_amd import utils


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


def procedural_scene_rays(B, seed, device='cpu', image_size=64):
  """SURVEY.md 8(d) 'PSNR': a procedural stand-in for the Blender scenes (no dataset in the image).

  A unit sphere at the origin shaded by its normal and a fixed light, in front of a white background
  (blender convention: near 2, far 6, cameras on a radius-4 sphere looking at the origin).  Rays are
  pixel rays of random cameras; ground-truth colour is the analytic ray/sphere intersection.
  """
  rs = np.random.default_rng(seed)
  # camera centres on the upper part of a radius-4 sphere
  z = rs.uniform(0.1, 0.9, (B, 1))
  phi = rs.uniform(0, 2 * np.pi, (B, 1))
  r = np.sqrt(1 - z * z)
  c = 4.0 * np.concatenate([r * np.cos(phi), r * np.sin(phi), z], -1)
  fwd = -c / np.linalg.norm(c, axis=-1, keepdims=True)
  up = np.array([[0., 0., 1.]])
  right = np.cross(fwd, up)
  right /= np.linalg.norm(right, axis=-1, keepdims=True)
  upv = np.cross(right, fwd)
  focal = 1.2 * image_size                       # ~45 degree field of view
  px = rs.uniform(-0.5, 0.5, (B, 2)) * image_size
  d = fwd * focal + right * px[:, :1] + upv * px[:, 1:]
  d = d / focal                                  # camera_utils convention: |d| ~ 1 at the image centre, un-normalised
  vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
  radii = np.full((B, 1), 2.0 / np.sqrt(12.0) / focal)          # camera_utils.py:604-610 pixel footprint
  # analytic render: unit sphere, normal shading
  b = np.sum(c * vd, -1)
  disc = b * b - (np.sum(c * c, -1) - 1.0)
  hit = disc > 0
  t = -b - np.sqrt(np.maximum(disc, 0))
  p = c + vd * t[:, None]
  nrm = p / np.maximum(np.linalg.norm(p, axis=-1, keepdims=True), 1e-9)
  light = np.array([0.3, 0.5, 0.8]) / np.linalg.norm([0.3, 0.5, 0.8])
  shade = 0.3 + 0.7 * np.clip(nrm @ light, 0, 1)[:, None]
  col = (0.5 + 0.5 * nrm) * shade
  rgb = np.where(hit[:, None], col, 1.0)
  f = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=device).contiguous()
  rays = utils.Rays(origins=f(c), directions=f(d), viewdirs=f(vd), radii=f(radii),
                    imageplane=f(px / image_size), lossmult=f(np.ones((B, 1))), near=f(np.full((B, 1), 2.0)),
                    far=f(np.full((B, 1), 6.0)),
                    cam_idx=torch.zeros((B, 1), dtype=torch.int32, device=device))
  return utils.Batch(rays=rays, rgb=f(rgb), alphas=f(hit.astype(np.float32)), normals=f(np.where(hit[:, None], nrm, 0.0)))


def unbounded_scene_rays(B, seed, device='cpu', image_size=96):
  """A procedural UNBOUNDED scene in the mip-NeRF 360 convention (configs/360.gin: near 0.2, far 1e6): content inside the
  unit ball (a shaded sphere), content far outside it (a checkered ground plane running to the horizon, so that samples
  live in the contracted region |x| > 1 of coord.contract) and a sky at infinity (painted by the opaque last interval of
  Model.opaque_background).  Cameras sit on a ring inside the ball and look inwards; rays are pixel rays of random cameras,
  ground-truth colours are analytic.  Used by the equal-step PSNR comparison (tests/golden/make_golden_psnr.py)."""
  rs = np.random.default_rng(seed)
  phi = rs.uniform(0, 2 * np.pi, (B, 1))
  c = np.concatenate([0.9 * np.cos(phi), 0.9 * np.sin(phi), rs.uniform(0.1, 0.4, (B, 1))], -1)
  look = np.concatenate([rs.normal(0, 0.05, (B, 2)), np.full((B, 1), 0.05)], -1)
  fwd = look - c
  fwd /= np.linalg.norm(fwd, axis=-1, keepdims=True)
  up = np.array([[0., 0., 1.]])
  right = np.cross(fwd, up)
  right /= np.linalg.norm(right, axis=-1, keepdims=True)
  upv = np.cross(right, fwd)
  focal = 0.9 * image_size                       # ~58 degree field of view
  px = rs.uniform(-0.5, 0.5, (B, 2)) * image_size
  d = (fwd * focal + right * px[:, :1] + upv * px[:, 1:]) / focal          # |d| ~ 1 at the image centre, un-normalised
  vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
  radii = np.full((B, 1), 2.0 / np.sqrt(12.0) / focal)                     # camera_utils.py:604-610 pixel footprint
  # analytic render along the unit view direction
  sc, sr = np.array([0.0, 0.0, 0.1]), 0.35
  oc = c - sc
  bq = np.sum(oc * vd, -1)
  disc = bq * bq - (np.sum(oc * oc, -1) - sr * sr)
  t_s = np.where(disc > 0, -bq - np.sqrt(np.maximum(disc, 0)), np.inf)
  t_s = np.where(t_s > 0, t_s, np.inf)
  zp = -0.3
  t_p = np.where(vd[:, 2] < -1e-6, (zp - c[:, 2]) / np.minimum(vd[:, 2], -1e-6), np.inf)
  light = np.array([0.3, 0.5, 0.8]) / np.linalg.norm([0.3, 0.5, 0.8])
  ps = c + vd * np.where(np.isfinite(t_s), t_s, 0.0)[:, None]
  nrm = (ps - sc) / sr
  col_s = (0.5 + 0.5 * nrm) * (0.3 + 0.7 * np.clip(nrm @ light, 0, 1))[:, None]
  pp = c + vd * np.where(np.isfinite(t_p), t_p, 0.0)[:, None]
  check = (np.floor(pp[:, 0] * 1.5) + np.floor(pp[:, 1] * 1.5)) % 2
  fade = np.exp(-np.where(np.isfinite(t_p), t_p, 0.0) / 30.0)[:, None]     # the checker washes out towards the horizon
  col_p = fade * np.where(check[:, None] > 0, [[0.85, 0.8, 0.7]], [[0.25, 0.3, 0.35]]) + (1 - fade) * np.array([[0.55, 0.55, 0.55]])
  sky = np.array([[0.55, 0.7, 0.95]]) * (0.55 + 0.45 * np.clip(vd[:, 2:3], 0, 1)) + np.array([[0.35, 0.25, 0.1]]) * (1 - np.clip(vd[:, 2:3] * 4, 0, 1))
  rgb = np.where((t_s < t_p)[:, None], col_s, np.where(np.isfinite(t_p)[:, None], col_p, np.clip(sky, 0, 1)))
  f = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=device).contiguous()
  rays = utils.Rays(origins=f(c), directions=f(d), viewdirs=f(vd), radii=f(radii), imageplane=f(px / image_size),
                    lossmult=f(np.ones((B, 1))), near=f(np.full((B, 1), 0.2)), far=f(np.full((B, 1), 1e6)),
                    cam_idx=torch.zeros((B, 1), dtype=torch.int32, device=device))
  return utils.Batch(rays=rays, rgb=f(np.clip(rgb, 0, 1)))
