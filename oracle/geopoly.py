"""Oracle restatement of reference internal/geopoly.py (TEST INFRASTRUCTURE ONLY).

Pure NumPy/fp64 host precompute: the tesselated-polyhedron direction basis that
`MLP.setup` builds once (models.py:388-389).  Row ORDER matters (it fixes the
IPE feature order and hence the first-layer weight layout), so de-duplication
and symmetry removal follow the reference's rules exactly.
"""

import itertools

import numpy as np

_PHI = (np.sqrt(5) + 1) / 2

# geopoly.py:96-106 -- icosahedron vertices (unnormalised) and faces.
_ICOSA_VERTS = np.array([(-1, 0, _PHI), (1, 0, _PHI), (-1, 0, -_PHI), (1, 0, -_PHI),
                         (0, _PHI, 1), (0, _PHI, -1), (0, -_PHI, 1), (0, -_PHI, -1),
                         (_PHI, 1, 0), (-_PHI, 1, 0), (_PHI, -1, 0), (-_PHI, -1, 0)])
_ICOSA_FACES = np.array([(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1),
                         (8, 10, 1), (8, 3, 10), (5, 3, 8), (5, 2, 3), (2, 7, 3),
                         (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6), (0, 1, 6),
                         (6, 1, 10), (9, 0, 11), (9, 11, 2), (9, 2, 5), (7, 2, 11)])
# geopoly.py:108-109 -- octahedron vertices.
_OCTA_VERTS = np.array([(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (-1, 0, 0),
                        (1, 0, 0)])


def compute_sq_dist(mat0, mat1=None):
  """geopoly.py:21-30 -- squared distances between all pairs of COLUMNS."""
  if mat1 is None:
    mat1 = mat0
  n0 = (mat0 * mat0).sum(axis=0)
  n1 = (mat1 * mat1).sum(axis=0)
  return np.maximum(0, n0[:, None] + n1[None, :] - 2 * (mat0.T @ mat1))


def compute_tesselation_weights(v):
  """geopoly.py:33-43 -- barycentric weights (i, j, v-i-j)/v, i-major order."""
  if v < 1:
    raise ValueError(f'v {v} must be >= 1')
  rows = [(i, j, v - i - j) for i in range(v + 1) for j in range(v + 1 - i)]
  return np.array(rows) / v


def tesselate_geodesic(base_verts, base_faces, v, eps=1e-4):
  """geopoly.py:46-75 -- subdivide each face, project to the sphere, dedupe.

  A vertex is kept iff it is the FIRST vertex (in face-major generation order)
  within sqrt(eps) of itself; survivors stay in generation order.
  """
  if not isinstance(v, int):
    raise ValueError(f'v {v} must an integer')
  bary = compute_tesselation_weights(v)
  chunks = []
  for face in base_faces:
    pts = bary @ base_verts[face, :]
    chunks.append(pts / np.linalg.norm(pts, axis=1, keepdims=True))
  verts = np.concatenate(chunks, axis=0)
  near = compute_sq_dist(verts.T) <= eps
  first_match = near.argmax(axis=1)  # lowest index with near[i, j] True
  return verts[np.unique(first_match), :]


def generate_basis(base_shape, angular_tesselation, remove_symmetries=True, eps=1e-4):
  """geopoly.py:78-124 -- returns [n, 3] (callers transpose, models.py:388-389)."""
  if base_shape == 'icosahedron':
    verts = tesselate_geodesic(_ICOSA_VERTS / np.sqrt(_PHI + 2), _ICOSA_FACES,
                               angular_tesselation)
  elif base_shape == 'octahedron':
    corners = np.array(list(itertools.product([-1, 1], repeat=3)))
    pairs = np.argwhere(compute_sq_dist(corners.T, _OCTA_VERTS.T) == 2)
    faces = np.sort(np.reshape(pairs[:, 1], [3, -1]).T, 1)
    verts = tesselate_geodesic(_OCTA_VERTS, faces, angular_tesselation)
  else:
    raise ValueError(f'base_shape {base_shape} not supported')
  if remove_symmetries:
    antipodal = compute_sq_dist(verts.T, -verts.T) < eps
    verts = verts[np.any(np.triu(antipodal), axis=1), :]
  return verts[:, ::-1]
