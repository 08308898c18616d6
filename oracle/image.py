"""Oracle restatement of the hot-path bits of reference internal/image.py (TEST INFRASTRUCTURE ONLY)."""

import math as _pm

import numpy as np
import torch

F32_EPS = float(np.finfo(np.float32).eps)


def mse_to_psnr(mse):
  """image.py:28-30."""
  return -10. / _pm.log(10.) * torch.log(torch.as_tensor(mse))


def psnr_to_mse(psnr):
  """image.py:33-35."""
  return torch.exp(-0.1 * _pm.log(10.) * torch.as_tensor(psnr))


def linear_to_srgb(linear, eps=None):
  """image.py:48-56."""
  if eps is None:
    eps = F32_EPS
  srgb0 = 323 / 25 * linear
  srgb1 = (211 * torch.clamp(linear, min=eps)**(5 / 12) - 11) / 200
  return torch.where(linear <= 0.0031308, srgb0, srgb1)


def srgb_to_linear(srgb, eps=None):
  """image.py:59-67."""
  if eps is None:
    eps = F32_EPS
  linear0 = 25 / 323 * srgb
  linear1 = torch.clamp((200 * srgb + 11) / 211, min=eps)**(12 / 5)
  return torch.where(srgb <= 0.04045, linear0, linear1)
