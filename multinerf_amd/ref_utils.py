"""Host precompute for the integrated directional encoding (reference internal/ref_utils.py:53-125).

Builds, once per MLP, the tables the IDE kernel reads: the (m, l) list (l = 2^i, m = 0..l), the
polynomial-in-z coefficient matrix of the spherical harmonics and the vMF attenuation rates
sigma_l = l(l+1)/2 (Ref-NeRF eqs. 6-8).  Pure Python/NumPy in float64, cast to fp32 for the device.
"""

import math

import numpy as np


def _gen_binom(a, k):
  """Generalised binomial coefficient C(a, k) for real a (ref_utils.py:53-55)."""
  out = 1.0
  for i in range(k):
    out *= (a - i)
  return out / math.factorial(k)


def _sph_harm_z_coeff(l, m, k):
  """Coefficient of z^k in the z-polynomial of Y_l^m (ref_utils.py:58-81)."""
  legendre = ((-1)**m * 2**l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m) *
              _gen_binom(0.5 * (l + k + m - 1.0), l))
  norm = math.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * math.pi * math.factorial(l + m)))
  return norm * legendre


def ide_tables(deg_view):
  """-> (m [T] int32, l [T] int32, mat [l_max+1, T] float32, sigma [T] float32)."""
  if deg_view > 5:
    raise ValueError('Only deg_view of at most 5 is numerically stable.')   # ref_utils.py:112-113
  ms, ls = [], []
  for i in range(deg_view):
    l = 2**i
    for m in range(l + 1):
      ms.append(m)
      ls.append(l)
  l_max = 2**(deg_view - 1)
  mat = np.zeros((l_max + 1, len(ms)), dtype=np.float64)
  for t, (m, l) in enumerate(zip(ms, ls)):
    for k in range(l - m + 1):
      mat[k, t] = _sph_harm_z_coeff(l, m, k)
  sigma = np.array([0.5 * l * (l + 1) for l in ls], dtype=np.float64)
  return (np.array(ms, dtype=np.int32), np.array(ls, dtype=np.int32), mat.astype(np.float32),
          sigma.astype(np.float32))
