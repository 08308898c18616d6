#!/usr/bin/env python
"""Goldens for camera_utils.transform_poses_pca on SEVERAL captures, by executing the reference's own function
(internal/camera_utils.py:191-227, pure NumPy: np.linalg.eig of the centres' scatter matrix).

    python tests/golden/make_golden_pca.py      # build container only; writes tests/golden/pca_poses.npz

The reference takes its axis signs from whatever LAPACK's general eigen-solver returns; the product (SVD + its own sign
rule, multinerf_amd/camera_utils.py:165) can only agree with that up to the sign ambiguity the reference's two fix-ups
(right-handedness, cameras' up-vector towards +z) leave open: a 180-degree turn about z.  tests/test_oracle_camera.py
counts, over these captures, how often the two frames are IDENTICAL and how often they differ by exactly that turn.
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (the jax / internal.configs / internal.utils stand-ins)

OUT = os.path.join(HERE, 'pca_poses.npz')


def captures():
  """12 pose sets: rings, forward-facing slabs, a near-degenerate line, random clouds; 5..60 cameras."""
  rs = np.random.RandomState(20240917)
  sets = []
  for k in range(12):
    n = int(rs.randint(5, 61))
    kind = k % 4
    if kind == 0:      # ring around an object, tilted
      ang = rs.uniform(0, 2 * np.pi, n)
      pos = np.stack([rs.uniform(2, 4) * np.cos(ang), rs.uniform(2, 4) * np.sin(ang), 0.4 * rs.normal(size=n)], 1)
    elif kind == 1:    # forward-facing slab
      pos = rs.normal(size=(n, 3)) * np.array([1.5, 0.8, 0.05])
    elif kind == 2:    # almost a line (two small spreads of nearly equal size)
      pos = rs.normal(size=(n, 3)) * np.array([3.0, 0.02, 0.021])
    else:              # generic cloud
      pos = rs.normal(size=(n, 3)) * rs.uniform(0.3, 3.0, size=3)
    q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
      q[:, 0] = -q[:, 0]
    pos = pos @ q.T + rs.normal(size=3) * 4
    rots = []
    for i in range(n):
      r, _ = np.linalg.qr(rs.normal(size=(3, 3)))
      if np.linalg.det(r) < 0:
        r[:, 0] = -r[:, 0]
      rots.append(r)
    sets.append(np.concatenate([np.stack(rots, 0), pos[:, :, None]], 2))
  return sets


def main():
  import types
  import dataclasses
  G.install_jax_standin()
  sys.path.insert(0, G.REF)
  cfg_stub = types.ModuleType('internal.configs')
  utils_stub = types.ModuleType('internal.utils')

  @dataclasses.dataclass
  class _Any:
    pass

  utils_stub.Pixels = utils_stub.Rays = _Any
  cfg_stub.Config = object
  sys.modules['internal.configs'] = cfg_stub
  sys.modules['internal.utils'] = utils_stub
  from internal import camera_utils
  g = {}
  for i, poses in enumerate(captures()):
    p, t = camera_utils.transform_poses_pca(poses.copy())
    g[f'in_{i}'], g[f'poses_{i}'], g[f'transform_{i}'] = poses, np.asarray(p), np.asarray(t)
  np.savez_compressed(OUT, **g)
  print(f'wrote {OUT}: {len(g)} arrays')


if __name__ == '__main__':
  main()
