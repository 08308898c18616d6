"""Host precompute of the IPE direction basis (reference internal/geopoly.py:78-124).

Run once per MLP at construction (models.py:388-389); the [K,3] result is
uploaded to the device and read by the IPE kernel.  Vertex ORDER is part of the
contract (it orders the encoding features and hence Dense_0's rows), so the
generation order (face-major, then barycentric (i,j) order), the keep-first
de-duplication and the keep-first-of-antipodal-pair rule follow the reference.
"""

import itertools

import numpy as np

_G = (1.0 + np.sqrt(5.0)) / 2.0


def _icosahedron():
  # geopoly.py:96-106
  v = np.array([(-1, 0, _G), (1, 0, _G), (-1, 0, -_G), (1, 0, -_G), (0, _G, 1), (0, _G, -1),
                (0, -_G, 1), (0, -_G, -1), (_G, 1, 0), (-_G, 1, 0), (_G, -1, 0),
                (-_G, -1, 0)], dtype=np.float64) / np.sqrt(_G + 2.0)
  f = [(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1), (8, 3, 10),
       (5, 3, 8), (5, 2, 3), (2, 7, 3), (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6),
       (0, 1, 6), (6, 1, 10), (9, 0, 11), (9, 11, 2), (9, 2, 5), (7, 2, 11)]
  return v, f


def _octahedron():
  # geopoly.py:108-112: faces = for each cube corner, the three axis vertices at
  # squared distance 2 from it, sorted.
  v = np.array([(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0)],
               dtype=np.float64)
  corners = np.array(list(itertools.product([-1, 1], repeat=3)), dtype=np.float64)
  d2 = ((corners[:, None, :] - v[None, :, :])**2).sum(-1)
  rows, cols = np.nonzero(d2 == 2)
  order = np.lexsort((cols, rows))
  flat = cols[order]                       # argwhere order: corner-major
  f = np.sort(flat.reshape(3, -1).T, axis=1)
  return v, [tuple(r) for r in f]


def _sqdist(a, b):
  """Pairwise squared distances by the reference's expansion (geopoly.py:21-30)."""
  na = (a * a).sum(1)
  nb = (b * b).sum(1)
  return np.maximum(0.0, na[:, None] + nb[None, :] - 2.0 * (a @ b.T))


def generate_basis(base_shape, angular_tesselation, remove_symmetries=True, eps=1e-4):
  """[K, 3] basis; callers use its transpose [3, K]."""
  if base_shape == 'icosahedron':
    verts, faces = _icosahedron()
  elif base_shape == 'octahedron':
    verts, faces = _octahedron()
  else:
    raise ValueError(f'base_shape {base_shape} not supported')
  v = angular_tesselation
  if not isinstance(v, int):
    raise ValueError(f'v {v} must an integer')
  if v < 1:
    raise ValueError(f'v {v} must be >= 1')
  bary = np.array([(i, j, v - i - j) for i in range(v + 1) for j in range(v + 1 - i)],
                  dtype=np.float64) / v
  pts = []
  for face in faces:
    p = bary @ verts[list(face), :]
    pts.append(p / np.sqrt((p * p).sum(1, keepdims=True)))
  pts = np.concatenate(pts, 0)
  # Keep a point iff it is the first one within eps (squared) of itself.
  d = _sqdist(pts, pts) <= eps
  first = d.argmax(1)
  keep = np.unique(first)
  pts = pts[keep]
  if remove_symmetries:
    anti = _sqdist(pts, -pts) < eps
    pts = pts[np.triu(anti).any(1)]
  return np.ascontiguousarray(pts[:, ::-1])
