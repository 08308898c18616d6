"""multinerf_amd/datasets.py on CPU: Blender loader (synthetic PNG scene written to tmp), pixel batches
(`cast_rays_in_train_step`: no ray kernel needed), batching modes, patch sampling, borders."""

import json
import math
import os

import numpy as np
import pytest
import torch

from multinerf_amd import configs, datasets, utils


def _write_blender_scene(root, n=3, size=8):
  from PIL import Image
  rs = np.random.default_rng(0)
  for split in ('train', 'test'):
    frames = []
    os.makedirs(os.path.join(root, split), exist_ok=True)
    for i in range(n):
      rgba = rs.integers(0, 256, (size, size, 4), dtype=np.uint8)
      Image.fromarray(rgba, 'RGBA').save(os.path.join(root, split, f'r_{i}.png'))
      nrm = rs.integers(0, 256, (size, size, 3), dtype=np.uint8)
      Image.fromarray(nrm, 'RGB').save(os.path.join(root, split, f'r_{i}_normal.png'))
      m = np.eye(4)
      m[:3, 3] = [0.1 * i, 0.2, 4.0]
      frames.append({'file_path': f'./{split}/r_{i}', 'transform_matrix': m.tolist()})
    with open(os.path.join(root, f'transforms_{split}.json'), 'w') as f:
      json.dump({'camera_angle_x': 0.7, 'frames': frames}, f)


def test_blender_loader_and_pixel_batches(tmp_path):
  root = str(tmp_path)
  _write_blender_scene(root)
  cfg = configs.load_preset('blender_256', ['Config.cast_rays_in_train_step = True', 'Config.batch_size = 32',
                                            'Config.compute_normal_metrics = True'])
  ds = datasets.Blender('train', root, cfg, device='cpu')
  assert ds.size == 3 and ds.height == 8 and ds.width == 8
  from PIL import Image
  rgba = np.asarray(Image.open(os.path.join(root, 'train', 'r_1.png')), dtype=np.float32) / 255.
  want = rgba[..., :3] * rgba[..., 3:] + (1 - rgba[..., 3:])                       # white background, datasets.py:553-554
  np.testing.assert_allclose(ds.images[1].numpy(), want, atol=1e-6)
  np.testing.assert_allclose(ds.alphas[1].numpy(), rgba[..., 3], atol=1e-6)
  assert abs(ds.focal - 0.5 * 8 / math.tan(0.35)) < 1e-4
  b = next(ds)
  assert isinstance(b.rays, utils.Pixels)                                          # fast path: pixels, not rays
  assert b.rays.pix_x_int.shape == (32,) and b.rgb.shape == (32, 3) and b.normals.shape == (32, 3)
  cam = b.rays.cam_idx[:, 0].long()
  assert len(torch.unique(cam)) == 1                                               # batching = 'single_image'
  np.testing.assert_array_equal(b.rgb.numpy(), ds.images[cam, b.rays.pix_y_int, b.rays.pix_x_int].numpy())
  assert float(b.rays.near[0]) == cfg.near and float(b.rays.far[0]) == cfg.far
  assert ds.peek() is ds.peek() and next(ds) is not None


def test_all_images_patches_and_borders(tmp_path):
  root = str(tmp_path)
  _write_blender_scene(root, n=4, size=16)
  cfg = configs.load_preset('blender_256', ['Config.cast_rays_in_train_step = True', 'Config.batch_size = 64',
                                            "Config.batching = 'all_images'", 'Config.patch_size = 2',
                                            'Config.num_border_pixels_to_mask = 3'])
  ds = datasets.Blender('train', root, cfg, device='cpu')
  b = next(ds)
  px, py, cam = b.rays.pix_x_int.view(16, 2, 2), b.rays.pix_y_int.view(16, 2, 2), b.rays.cam_idx.view(16, 2, 2)
  assert (px[:, :, 1] == px[:, :, 0] + 1).all() and (py[:, 1, :] == py[:, 0, :] + 1).all()   # 2x2 patches
  assert (cam == cam[:, :1, :1]).all() and len(torch.unique(cam)) > 1
  assert px.min() >= 3 and px.max() <= 16 - 3 - 1 and py.min() >= 3 and py.max() <= 16 - 3 - 1
  with pytest.raises(ValueError, match='too large'):
    cfg.patch_size, cfg.batch_size = 16, 64
    datasets.Blender('train', root, cfg, device='cpu')


def test_unknown_loader_fails_loudly():
  cfg = configs.load_preset('360')
  with pytest.raises(NotImplementedError, match='dataset_loader'):
    datasets.load_dataset('train', '/nonexistent', cfg, device='cpu')
