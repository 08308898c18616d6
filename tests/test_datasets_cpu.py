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
  cfg = configs.load_preset('360', ["Config.dataset_loader = 'dtu'"])
  with pytest.raises(NotImplementedError, match='dataset_loader'):
    datasets.load_dataset('train', '/nonexistent', cfg, device='cpu')


def _write_colmap_scene(root, n=9, size=(12, 10), model=4):
  """A COLMAP sparse model written with the published binary layout + JPEG-free PNG images."""
  import struct
  from PIL import Image
  rs = np.random.default_rng(1)
  w, h = size
  os.makedirs(os.path.join(root, 'sparse/0'), exist_ok=True)
  os.makedirs(os.path.join(root, 'images'), exist_ok=True)
  os.makedirs(os.path.join(root, 'images_2'), exist_ok=True)
  params = [50.0, 52.0, w / 2., h / 2., 0.01, -0.002, 0.0005, 0.0003]          # OPENCV: fx fy cx cy k1 k2 p1 p2
  with open(os.path.join(root, 'sparse/0/cameras.bin'), 'wb') as f:
    f.write(struct.pack('<Q', 1))
    f.write(struct.pack('<iiQQ', 1, model, w, h))
    f.write(struct.pack('<8d', *params))
  c2ws, names = [], []
  with open(os.path.join(root, 'sparse/0/images.bin'), 'wb') as f:
    f.write(struct.pack('<Q', n))
    for i in range(n):
      q = rs.normal(size=4)
      q /= np.linalg.norm(q)
      t = rs.normal(size=3)
      name = f'img_{n - i:02d}.png'                                                # not alphabetical in file order
      f.write(struct.pack('<i7di', i + 1, *q, *t, 1))
      f.write(name.encode() + b'\x00')
      f.write(struct.pack('<Q', 2))
      f.write(struct.pack('<ddq', 1.0, 2.0, -1) * 2)
      names.append(name)
      R = datasets._qvec_to_rotmat(q)
      w2c = np.eye(4)
      w2c[:3, :3], w2c[:3, 3] = R, t
      c2ws.append(np.linalg.inv(w2c)[:3, :4] @ np.diag([1, -1, -1, 1]))
      Image.fromarray(rs.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, 'images', name))
      Image.fromarray(rs.integers(0, 256, (h // 2, w // 2, 3), dtype=np.uint8)).save(os.path.join(root, 'images_2', name))
  return names, np.stack(c2ws, 0), params


def test_llff_colmap_loader(tmp_path):
  from multinerf_amd import camera_utils
  root = str(tmp_path)
  names, c2ws, prm = _write_colmap_scene(root)
  rot = datasets._qvec_to_rotmat(np.array([0.5, 0.5, 0.5, 0.5]))
  np.testing.assert_allclose(rot @ rot.T, np.eye(3), atol=1e-12)                   # a rotation (120 deg about (1,1,1))
  np.testing.assert_allclose(rot @ np.array([1., 0., 0.]), [0., 1., 0.], atol=1e-12)
  cfg = configs.load_preset('360', ['Config.factor = 2', 'Config.cast_rays_in_train_step = True', 'Config.batch_size = 16'])
  tr = datasets.load_dataset('train', root, cfg, device='cpu')
  te = datasets.load_dataset('test', root, cfg, device='cpu')
  assert te.size == 2 and tr.size == 7 and tr.height == 5 and tr.width == 6        # llffhold 8 of 9 images; images_2
  order = np.argsort(names)                                                        # load_alphabetical
  want_poses, _ = camera_utils.transform_poses_pca(c2ws[order].astype(np.float64))
  np.testing.assert_allclose(te.camtoworlds.numpy(), want_poses[[0, 8]], rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(np.abs(np.concatenate([tr.camtoworlds, te.camtoworlds])[:, :, 3]).max(), 1.0, rtol=1e-5)
  want_p2c = np.linalg.inv(np.array([[prm[0], 0, prm[2]], [0, prm[1], prm[3]], [0, 0, 1.]])) @ np.diag([2, 2, 1.])
  np.testing.assert_allclose(tr.pixtocams.numpy(), want_p2c, rtol=1e-6, atol=1e-9)
  assert tr.distortion_params == dict(k1=prm[4], k2=prm[5], k3=0., p1=prm[6], p2=prm[7])
  assert tr.camtype == camera_utils.ProjectionType.PERSPECTIVE
  b = next(tr)
  assert b.rays.pix_x_int.max() < 6 and b.rgb.shape == (16, 3)


def test_llff_forward_facing_ngp_poses(tmp_path):
  from PIL import Image
  from multinerf_amd import camera_utils
  root = str(tmp_path)
  os.makedirs(os.path.join(root, 'images'))
  rs = np.random.default_rng(2)
  frames, c2w = [], []
  for i in range(8):
    Image.fromarray(rs.integers(0, 256, (6, 8, 3), dtype=np.uint8)).save(os.path.join(root, 'images', f'{i}.png'))
    m = np.eye(4)
    m[:3, :3] += 0.05 * rs.normal(size=(3, 3))
    m[:3, 3] = rs.normal(size=3) * 0.3
    frames.append({'file_path': f'images/{i}.png', 'transform_matrix': m.tolist()})
    c2w.append(m[:3, :4])
  json.dump({'w': 8, 'h': 6, 'fl_x': 10.0, 'fl_y': 10.0, 'frames': frames}, open(os.path.join(root, 'transforms.json'), 'w'))
  bounds = np.concatenate([np.zeros((8, 15)), rs.uniform(2.0, 3.0, (8, 1)), rs.uniform(8.0, 9.0, (8, 1))], 1)
  np.save(os.path.join(root, 'poses_bounds.npy'), bounds)
  cfg = configs.load_preset('360', ['Config.factor = 0', 'Config.forward_facing = True', 'Config.batch_size = 8',
                                    'Config.cast_rays_in_train_step = True', 'Config.llff_use_all_images_for_training = True'])
  ds = datasets.load_dataset('train', root, cfg, device='cpu')
  assert ds.size == 8 and ds.pixtocam_ndc is not None and ds.cameras[3] is ds.pixtocam_ndc
  scale = 1. / (bounds[:, -2:].min() * .75)
  p = np.stack(c2w, 0).astype(np.float32).astype(np.float64)
  p[:, :3, 3] *= scale
  want, _ = camera_utils.recenter_poses(p)
  np.testing.assert_allclose(ds.camtoworlds.numpy(), want, rtol=1e-5, atol=1e-6)
