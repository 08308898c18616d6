"""Ray/batch containers and shard helpers (reference internal/utils.py:31-136).

`Rays`/`Pixels`/`Batch` keep the reference's field names; leaves are torch
tensors (device memory for the HIP kernels) instead of jax arrays.
"""

import dataclasses
from typing import Any, Optional

import torch


def _tree_map_dc(fn, obj):
  kw = {}
  for f in dataclasses.fields(obj):
    v = getattr(obj, f.name)
    if v is None:
      kw[f.name] = None
    elif dataclasses.is_dataclass(v):
      kw[f.name] = _tree_map_dc(fn, v)
    else:
      kw[f.name] = fn(v)
  return type(obj)(**kw)


@dataclasses.dataclass
class Pixels:
  """utils.py:31-41."""
  pix_x_int: Any
  pix_y_int: Any
  lossmult: Any
  near: Any
  far: Any
  cam_idx: Any
  exposure_idx: Optional[Any] = None
  exposure_values: Optional[Any] = None

  def map(self, fn):
    return _tree_map_dc(fn, self)


@dataclasses.dataclass
class Rays:
  """utils.py:44-57.  All tensors share their leading dims; last axis = channel."""
  origins: Any
  directions: Any
  viewdirs: Any
  radii: Any
  imageplane: Any
  lossmult: Any
  near: Any
  far: Any
  cam_idx: Any
  exposure_idx: Optional[Any] = None
  exposure_values: Optional[Any] = None

  def map(self, fn):
    return _tree_map_dc(fn, self)


def dummy_rays(include_exposure_idx=False, include_exposure_values=False, device='cpu'):
  """utils.py:60-79."""
  data_fn = lambda n: torch.zeros((1, n), device=device)
  kw = {}
  if include_exposure_idx:
    kw['exposure_idx'] = data_fn(1).to(torch.int32)
  if include_exposure_values:
    kw['exposure_values'] = data_fn(1)
  return Rays(origins=data_fn(3), directions=data_fn(3), viewdirs=data_fn(3),
              radii=data_fn(1), imageplane=data_fn(2), lossmult=data_fn(1),
              near=data_fn(1), far=data_fn(1), cam_idx=data_fn(1).to(torch.int32), **kw)


@dataclasses.dataclass
class Batch:
  """utils.py:82-89."""
  rays: Any
  rgb: Optional[Any] = None
  disps: Optional[Any] = None
  normals: Optional[Any] = None
  alphas: Optional[Any] = None

  def map(self, fn):
    return _tree_map_dc(fn, self)


def shard(xs, num_shards):
  """utils.py:125-128 -- [N, ...] -> [num_shards, N/num_shards, ...].

  The reference splits across jax.local_device_count(); here the caller names
  the shard count (world size) and each rank takes `shard(x, W)[rank]`.
  """
  fn = lambda x: x.reshape((num_shards, -1) + tuple(x.shape[1:]))
  return xs.map(fn) if hasattr(xs, 'map') else fn(xs)


def unshard(x, padding=0):
  """utils.py:131-136."""
  y = x.reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))
  if padding > 0:
    y = y[:-padding]
  return y
