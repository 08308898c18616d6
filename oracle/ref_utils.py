"""Oracle restatement of reference internal/ref_utils.py (TEST INFRASTRUCTURE ONLY)."""

import math as _pm

import numpy as np
import torch

from oracle import math as rmath

F32_EPS = float(np.finfo(np.float32).eps)


def reflect(viewdirs, normals):
  """ref_utils.py:22-37 -- u = 2 (n.v) n - v."""
  return 2.0 * torch.sum(normals * viewdirs, dim=-1, keepdim=True) * normals - viewdirs


def l2_normalize(x, eps=F32_EPS):
  """ref_utils.py:40-42."""
  return x / torch.sqrt(torch.clamp(torch.sum(x**2, dim=-1, keepdim=True), min=eps))


def compute_weighted_mae(weights, normals, normals_gt):
  """ref_utils.py:45-50 -- weighted mean angular error in degrees."""
  one_eps = 1 - F32_EPS
  return (weights * torch.arccos(
      torch.clamp((normals * normals_gt).sum(-1), -one_eps, one_eps))).sum() / weights.sum() * 180.0 / _pm.pi


def generalized_binomial_coeff(a, k):
  """ref_utils.py:53-55."""
  return np.prod(a - np.arange(k)) / _pm.factorial(k)


def assoc_legendre_coeff(l, m, k):
  """ref_utils.py:58-74 -- coefficient of cos^k sin^m in P_l^m(cos theta)."""
  return ((-1)**m * 2**l * _pm.factorial(l) / _pm.factorial(k) /
          _pm.factorial(l - k - m) *
          generalized_binomial_coeff(0.5 * (l + k + m - 1.0), l))


def sph_harm_coeff(l, m, k):
  """ref_utils.py:77-81."""
  return (np.sqrt((2.0 * l + 1.0) * _pm.factorial(l - m) /
                  (4.0 * np.pi * _pm.factorial(l + m))) * assoc_legendre_coeff(l, m, k))


def get_ml_array(deg_view):
  """ref_utils.py:84-96 -- (m, l) pairs, l = 2^i, m = 0..l."""
  ml_list = []
  for i in range(deg_view):
    l = 2**i
    for m in range(l + 1):
      ml_list.append((m, l))
  return np.array(ml_list).T


def ide_matrices(deg_view):
  """The constants of ref_utils.py:113-125: (ml_array [2,T], mat [l_max+1, T])."""
  if deg_view > 5:
    raise ValueError('Only deg_view of at most 5 is numerically stable.')
  ml_array = get_ml_array(deg_view)
  l_max = 2**(deg_view - 1)
  mat = np.zeros((l_max + 1, ml_array.shape[1]))
  for i, (m, l) in enumerate(ml_array.T):
    for k in range(l - m + 1):
      mat[k, i] = sph_harm_coeff(l, m, k)
  return ml_array, mat


def generate_ide_fn(deg_view):
  """ref_utils.py:99-159 -- integrated directional encoding (Ref-NeRF eqs. 6-8).

  Complex arithmetic is carried as explicit (real, imag) pairs so that it is
  differentiable in any torch dtype; (x+iy)^m is built by repeated complex
  multiplication.
  """
  ml_array, mat_np = ide_matrices(deg_view)

  def integrated_dir_enc_fn(xyz, kappa_inv):
    dtype = xyz.dtype
    mat = torch.as_tensor(mat_np, dtype=dtype)
    x = xyz[..., 0:1]
    y = xyz[..., 1:2]
    z = xyz[..., 2:3]
    vmz = torch.cat([z**i for i in range(mat.shape[0])], dim=-1)
    # (x + iy)^m for m = 0..max(m).
    m_max = int(ml_array[0].max())
    pr = [torch.ones_like(x)]
    pi = [torch.zeros_like(x)]
    for _ in range(m_max):
      r, i = pr[-1], pi[-1]
      pr.append(r * x - i * y)
      pi.append(r * y + i * x)
    vm_re = torch.cat([pr[m] for m in ml_array[0, :]], dim=-1)
    vm_im = torch.cat([pi[m] for m in ml_array[0, :]], dim=-1)
    zpart = rmath.matmul(vmz, mat)
    sigma = torch.as_tensor(0.5 * ml_array[1, :] * (ml_array[1, :] + 1), dtype=dtype)
    att = torch.exp(-sigma * kappa_inv)
    return torch.cat([vm_re * zpart * att, vm_im * zpart * att], dim=-1)

  return integrated_dir_enc_fn


def generate_dir_enc_fn(deg_view):
  """ref_utils.py:162-177."""
  ide = generate_ide_fn(deg_view)
  return lambda xyz: ide(xyz, torch.zeros_like(xyz[..., :1]))
