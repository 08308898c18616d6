"""Dataset providers for train.py / eval.py (reference internal/datasets.py:190-560).

MI355X-first differences from the reference's thread + queue + host-side NumPy ray casting: the images of a
scene (a few hundred MB) live in HBM, a train batch is pixel indices drawn on the device, colours are one
gather, and rays come from `camera_utils.cast_ray_batch` (csrc/camera.hip) -- either here or, with
`Config.cast_rays_in_train_step`, inside the train step exactly like the reference's fast path
(datasets.py:431-433).  Loaders: 'blender' (transforms_*.json + PNG, datasets.py:507-560) and 'procedural'
(an analytic scene for offline runs; tests/helpers.py uses the same scene) and 'llff' (COLMAP `sparse/0` binaries or
NGP `transforms.json` poses; forward-facing NDC and 360 scenes, datasets.py:563-712).  RawNeRF / DTU / TAT loaders need
rawpy or dataset-specific files that cannot be exercised here and are not restated.
"""

import json
import os

import numpy as np
import torch

from multinerf_amd import camera_utils, utils


class Dataset:
  """datasets.py:190-498 -- iterator of utils.Batch; `peek()`, `size`, `cameras`, `generate_ray_batch(i)`."""

  def __init__(self, split, data_dir, config, device='cuda', seed=0):
    self.split = split                                  # 'train' / 'test'
    self.data_dir = data_dir
    self.device = torch.device(device)
    self._batch_size = config.batch_size
    self._patch_size = max(config.patch_size, 1)
    if self._patch_size**2 > self._batch_size:
      raise ValueError(f'Patch size {self._patch_size}^2 too large for per-process batch size {self._batch_size}')
    if config.batching not in ('all_images', 'single_image'):
      raise ValueError(f'unknown batching {config.batching!r}')
    self._batching = config.batching
    self._load_disps = config.compute_disp_metrics
    self._load_normals = config.compute_normal_metrics
    self._num_border_pixels_to_mask = config.num_border_pixels_to_mask
    self._cast_rays_in_train_step = config.cast_rays_in_train_step
    self._test_camera_idx = 0
    self.near, self.far = config.near, config.far
    self.distortion_params = None
    self.pixtocam_ndc = None
    self.disp_images = self.normal_images = self.alphas = None
    self.camtype = camera_utils.ProjectionType.PERSPECTIVE
    self.images = self.camtoworlds = self.pixtocams = None
    self.height = self.width = self.focal = None
    self._gen = torch.Generator(device=self.device).manual_seed(20200823 + seed)
    self._load_renderings(config)
    to = lambda a, dt=torch.float32: None if a is None else torch.as_tensor(np.asarray(a)).to(self.device, dt)
    self.images = to(self.images)
    self.disp_images, self.normal_images, self.alphas = to(self.disp_images), to(self.normal_images), to(self.alphas)
    self.camtoworlds = to(self.camtoworlds)[..., :3, :4].contiguous()
    self.pixtocams = to(self.pixtocams)
    self._n_examples = self.camtoworlds.shape[0]
    self.cameras = (self.pixtocams, self.camtoworlds, self.distortion_params, self.pixtocam_ndc)
    self._peeked = None

  @property
  def size(self):
    return self._n_examples

  def _load_renderings(self, config):
    raise NotImplementedError

  def __iter__(self):
    return self

  def __next__(self):
    if self._peeked is not None:
      b, self._peeked = self._peeked, None
      return b
    return self._next_train() if self.split == 'train' else self._next_test()

  def peek(self):
    if self._peeked is None:
      self._peeked = next(self)
    return self._peeked

  def _make_ray_batch(self, pix_x_int, pix_y_int, cam_idx, lossmult=None):
    """datasets.py:382-450."""
    shape = pix_x_int.shape
    bs = lambda v, dt=torch.float32: torch.as_tensor(v, dtype=dt, device=self.device).expand(shape)[..., None].contiguous()
    pixels = utils.Pixels(pix_x_int=pix_x_int, pix_y_int=pix_y_int,
                          lossmult=bs(1.) if lossmult is None else lossmult, near=bs(self.near), far=bs(self.far),
                          cam_idx=bs(cam_idx, torch.int32))
    if self._cast_rays_in_train_step and self.split == 'train':
      rays = pixels
    else:
      rays = camera_utils.cast_ray_batch(self.cameras, pixels, self.camtype)
    ci = torch.as_tensor(cam_idx, device=self.device).expand(shape).long()
    batch = dict(rays=rays, rgb=self.images[ci, pix_y_int, pix_x_int])
    if self._load_disps:
      batch['disps'] = self.disp_images[ci, pix_y_int, pix_x_int]
    if self._load_normals:
      batch['normals'] = self.normal_images[ci, pix_y_int, pix_x_int]
      batch['alphas'] = self.alphas[ci, pix_y_int, pix_x_int]
    return utils.Batch(**batch)

  def _next_train(self):
    """datasets.py:452-488: random pixels (patches) of random cameras, drawn on the device."""
    num_patches = self._batch_size // self._patch_size**2
    lo = self._num_border_pixels_to_mask
    hi = self._num_border_pixels_to_mask + self._patch_size - 1
    ri = lambda a, b, shape: torch.randint(a, b, shape, generator=self._gen, device=self.device)
    px = ri(lo, self.width - hi, (num_patches, 1, 1))
    py = ri(lo, self.height - hi, (num_patches, 1, 1))
    dx, dy = camera_utils.pixel_coordinates(self._patch_size, self._patch_size, self.device)
    px, py = px + dx, py + dy
    if self._batching == 'all_images':
      cam = ri(0, self._n_examples, (num_patches, 1, 1)).expand(px.shape)
    else:
      cam = ri(0, self._n_examples, (1,)).expand(px.shape)
    b = self._make_ray_batch(px.reshape(-1), py.reshape(-1), cam.reshape(-1))
    return b

  def generate_ray_batch(self, cam_idx):
    """datasets.py:490-502: all pixels of one camera, shaped [H, W, ...]."""
    px, py = camera_utils.pixel_coordinates(self.width, self.height, self.device)
    return self._make_ray_batch(px, py, int(cam_idx))

  def _next_test(self):
    cam_idx = self._test_camera_idx
    self._test_camera_idx = (self._test_camera_idx + 1) % self._n_examples
    return self.generate_ray_batch(cam_idx)


class Blender(Dataset):
  """datasets.py:507-560 (PNG path; `use_tiffs` and `_disp.tiff` need a TIFF reader that is not installed)."""

  def _load_renderings(self, config):
    from PIL import Image
    if config.render_path:
      raise ValueError('render_path cannot be used for the blender dataset.')
    if config.use_tiffs or self._load_disps:
      raise NotImplementedError('TIFF inputs (use_tiffs / disparity maps) need a TIFF reader')
    with open(os.path.join(self.data_dir, f'transforms_{self.split}.json')) as fp:
      meta = json.load(fp)
    images, normals, cams = [], [], []

    def get_img(path):
      im = Image.open(path)
      if config.factor > 1:                         # image.downsample: area average over factor x factor blocks
        a = np.asarray(im, dtype=np.float32)
        h, w = a.shape[0] // config.factor * config.factor, a.shape[1] // config.factor * config.factor
        a = a[:h, :w].reshape(h // config.factor, config.factor, w // config.factor, config.factor, -1).mean((1, 3))
        return a
      return np.asarray(im, dtype=np.float32)

    for frame in meta['frames']:
      fprefix = os.path.join(self.data_dir, frame['file_path'])
      images.append(get_img(fprefix + '.png') / 255.)
      if self._load_normals:
        normals.append(get_img(fprefix + '_normal.png')[..., :3] * 2. / 255. - 1.)
      cams.append(np.array(frame['transform_matrix'], dtype=np.float32))
    images = np.stack(images, 0)
    if self._load_normals:
      self.normal_images = np.stack(normals, 0)
      self.alphas = images[..., -1]
    rgb, alpha = images[..., :3], images[..., -1:]
    self.images = rgb * alpha + (1. - alpha)        # white background
    self.height, self.width = self.images.shape[1:3]
    self.camtoworlds = np.stack(cams, 0)
    self.focal = .5 * self.width / np.tan(.5 * float(meta['camera_angle_x']))
    self.pixtocams = camera_utils.get_pixtocam(self.focal, self.width, self.height).numpy()


class Procedural(Dataset):
  """An analytic stand-in for the Blender scenes: a normal-shaded unit sphere on white, cameras on a radius-4
  sphere looking at the origin (near 2, far 6).  `data_dir` is ignored."""

  NUM_TRAIN, NUM_TEST, SIZE = 40, 6, 96

  def _load_renderings(self, config):
    n = self.NUM_TRAIN if self.split == 'train' else self.NUM_TEST
    rs = np.random.default_rng(7 if self.split == 'train' else 8)
    H = W = self.SIZE // max(config.factor, 1)
    focal = 1.2 * W
    z = rs.uniform(0.1, 0.9, n)
    phi = rs.uniform(0, 2 * np.pi, n)
    c = 4.0 * np.stack([np.sqrt(1 - z * z) * np.cos(phi), np.sqrt(1 - z * z) * np.sin(phi), z], -1)
    fwd = -c / np.linalg.norm(c, axis=-1, keepdims=True)
    right = np.cross(fwd, np.array([[0., 0., 1.]]))
    right /= np.linalg.norm(right, axis=-1, keepdims=True)
    up = np.cross(right, fwd)
    c2w = np.stack([right, up, -fwd, c], -1)       # OpenGL: columns x, y, z (camera looks along -z), position
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    d_cam = np.stack([(xs + .5 - W / 2) / focal, -(ys + .5 - H / 2) / focal, -np.ones_like(xs, dtype=np.float64)], -1)
    imgs, nrms, alphas = [], [], []
    light = np.array([0.3, 0.5, 0.8]) / np.linalg.norm([0.3, 0.5, 0.8])
    for i in range(n):
      d = d_cam @ c2w[i, :, :3].T
      d /= np.linalg.norm(d, axis=-1, keepdims=True)
      b = d @ c[i]
      disc = b * b - (c[i] @ c[i] - 1.0)
      hit = disc > 0
      t = -b - np.sqrt(np.maximum(disc, 0))
      p = c[i] + d * t[..., None]
      nrm = p / np.maximum(np.linalg.norm(p, axis=-1, keepdims=True), 1e-9)
      col = (0.5 + 0.5 * nrm) * (0.3 + 0.7 * np.clip(nrm @ light, 0, 1)[..., None])
      imgs.append(np.where(hit[..., None], col, 1.0))
      nrms.append(np.where(hit[..., None], nrm, 0.0))
      alphas.append(hit.astype(np.float32))
    self.images = np.stack(imgs, 0)
    if self._load_normals:
      self.normal_images, self.alphas = np.stack(nrms, 0), np.stack(alphas, 0)
    self.height, self.width, self.focal = H, W, focal
    self.camtoworlds = c2w
    self.pixtocams = camera_utils.get_pixtocam(focal, W, H).numpy()


# ----------------------------------------------------------------------------- COLMAP / NGP poses for LLFF


_COLMAP_MODELS = {0: ('SIMPLE_PINHOLE', 3), 1: ('PINHOLE', 4), 2: ('SIMPLE_RADIAL', 4), 3: ('RADIAL', 5),
                  4: ('OPENCV', 8), 5: ('OPENCV_FISHEYE', 8)}


def read_colmap_binary(colmap_dir):
  """cameras.bin + images.bin of a COLMAP sparse model (the published binary layout: little-endian, counts as
  uint64; the reference reads it through pycolmap.SceneManager, datasets.py:55-78).  Returns
  (cameras {id: (model_id, w, h, params)}, images [(name, qvec wxyz, tvec, camera_id)] in file order)."""
  import struct
  cams = {}
  with open(os.path.join(colmap_dir, 'cameras.bin'), 'rb') as f:
    (n,) = struct.unpack('<Q', f.read(8))
    for _ in range(n):
      cid, model, w, h = struct.unpack('<iiQQ', f.read(24))
      if model not in _COLMAP_MODELS:
        raise NotImplementedError(f'COLMAP camera model {model}')
      npar = _COLMAP_MODELS[model][1]
      cams[cid] = (model, w, h, struct.unpack(f'<{npar}d', f.read(8 * npar)))
  images = []
  with open(os.path.join(colmap_dir, 'images.bin'), 'rb') as f:
    (n,) = struct.unpack('<Q', f.read(8))
    for _ in range(n):
      vals = struct.unpack('<i7di', f.read(64))
      qvec, tvec, cid = np.array(vals[1:5]), np.array(vals[5:8]), vals[8]
      name = b''
      while True:
        ch = f.read(1)
        if ch in (b'\x00', b''):
          break
        name += ch
      (npts,) = struct.unpack('<Q', f.read(8))
      f.seek(24 * npts, 1)                           # (x, y, point3D_id) per 2-D point: not needed
      images.append((name.decode(), qvec, tvec, cid))
  return cams, images


def _qvec_to_rotmat(q):
  w, x, y, z = q
  return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                   [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                   [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def load_colmap_posedata(colmap_dir):
  """NeRFSceneManager.process (datasets.py:62-149): shared intrinsics of camera 1, world-to-camera -> camera-to-world,
  COLMAP (right, down, fwd) -> NeRF (right, up, back), distortion parameters by camera model."""
  cams, images = read_colmap_binary(colmap_dir)
  model, _, _, prm = cams[1]
  if model in (0, 2, 3):
    fx = fy = prm[0]
    cx, cy = prm[1], prm[2]
    extra = prm[3:]
  else:
    fx, fy, cx, cy = prm[:4]
    extra = prm[4:]
  pixtocam = np.linalg.inv(camera_utils.intrinsic_matrix(fx, fy, cx, cy).numpy())
  w2c = []
  for _, q, t, _ in images:
    m = np.eye(4)
    m[:3, :3], m[:3, 3] = _qvec_to_rotmat(q), t
    w2c.append(m)
  poses = np.linalg.inv(np.stack(w2c, 0))[:, :3, :4] @ np.diag([1, -1, -1, 1])
  names = [im[0] for im in images]
  camtype = camera_utils.ProjectionType.PERSPECTIVE
  if model in (0, 1):
    params = None
  elif model == 2:
    params = dict(k1=extra[0], k2=0., k3=0., p1=0., p2=0.)
  elif model == 3:
    params = dict(k1=extra[0], k2=extra[1], k3=0., p1=0., p2=0.)
  elif model == 4:
    params = dict(k1=extra[0], k2=extra[1], k3=0., p1=extra[2], p2=extra[3])
  else:
    params = dict(k1=extra[0], k2=extra[1], k3=extra[2], k4=extra[3])
    camtype = camera_utils.ProjectionType.FISHEYE
  return names, poses, pixtocam, params, camtype


def load_blender_posedata(data_dir, split=None):
  """datasets.py:152-186: poses from `transforms[_split].json` (Blender / NGP layout)."""
  suffix = '' if split is None else f'_{split}'
  with open(os.path.join(data_dir, f'transforms{suffix}.json')) as fp:
    meta = json.load(fp)
  names, poses = [], []
  for frame in meta['frames']:
    if os.path.exists(os.path.join(data_dir, frame['file_path'])):
      names.append(frame['file_path'].split('/')[-1])
      poses.append(np.array(frame['transform_matrix'], dtype=np.float32))
  poses = np.stack(poses, 0)
  w, h = meta['w'], meta['h']
  cx, cy = meta.get('cx', w / 2.), meta.get('cy', h / 2.)
  fx = meta['fl_x'] if 'fl_x' in meta else 0.5 * w / np.tan(0.5 * float(meta['camera_angle_x']))
  fy = meta['fl_y'] if 'fl_y' in meta else 0.5 * h / np.tan(0.5 * float(meta['camera_angle_y']))
  pixtocam = np.linalg.inv(camera_utils.intrinsic_matrix(fx, fy, cx, cy).numpy())
  coeffs = ['k1', 'k2', 'p1', 'p2']
  params = None if not any(c in meta for c in coeffs) else {c: meta.get(c, 0.) for c in coeffs}
  return names, poses, pixtocam, params, camera_utils.ProjectionType.PERSPECTIVE


class LLFF(Dataset):
  """datasets.py:563-712 for ordinary (non-raw) captures: COLMAP `sparse/0` or NGP `transforms.json` poses, images in
  `images[_factor]`, forward-facing scenes in NDC (recenter + bound rescale) or 360 scenes (PCA alignment into the unit
  cube), every `llffhold`-th image held out.  Render paths (spiral / ellipse / spline) and RawNeRF inputs are not restated."""

  def _load_renderings(self, config):
    from PIL import Image
    if config.rawnerf_mode:
      raise NotImplementedError('RawNeRF inputs need rawpy (raw_utils.load_raw_dataset)')
    if config.render_path:
      raise NotImplementedError('render paths are out of scope (DESIGN.md section 7)')
    factor = config.factor if config.factor > 0 else 1
    suffix = f'_{config.factor}' if config.factor > 0 else ''
    colmap_dir = os.path.join(self.data_dir, 'sparse/0/')
    pose_data = load_colmap_posedata(colmap_dir) if os.path.exists(colmap_dir) else load_blender_posedata(self.data_dir)
    image_names, poses, pixtocam, distortion_params, camtype = pose_data
    if config.load_alphabetical:
      inds = np.argsort(image_names)
      image_names = [image_names[i] for i in inds]
      poses = poses[inds]
    pixtocam = pixtocam @ np.diag([factor, factor, 1.])
    self.pixtocams = pixtocam.astype(np.float32)
    self.focal = 1. / self.pixtocams[0, 0]
    self.distortion_params = distortion_params
    self.camtype = camtype
    colmap_image_dir = os.path.join(self.data_dir, 'images')
    image_dir = os.path.join(self.data_dir, 'images' + suffix)
    for d in (image_dir, colmap_image_dir):
      if not os.path.exists(d):
        raise ValueError(f'Image folder {d} does not exist.')
    colmap_to_image = dict(zip(sorted(os.listdir(colmap_image_dir)), sorted(os.listdir(image_dir))))
    images = np.stack([np.asarray(Image.open(os.path.join(image_dir, colmap_to_image[f])), dtype=np.float32)[..., :3]
                       for f in image_names], 0) / 255.
    posefile = os.path.join(self.data_dir, 'poses_bounds.npy')
    bounds = np.load(posefile)[:, -2:] if os.path.exists(posefile) else np.array([0.01, 1.])
    self.colmap_to_world_transform = np.eye(4)
    poses = np.array(poses, dtype=np.float64)
    if config.forward_facing:
      self.pixtocam_ndc = torch.as_tensor(self.pixtocams.reshape(-1, 3, 3)[0])
      scale = 1. / (bounds.min() * .75)
      poses[:, :3, 3] *= scale
      self.colmap_to_world_transform = np.diag([scale] * 3 + [1])
      poses, transform = camera_utils.recenter_poses(poses)
      self.colmap_to_world_transform = transform @ self.colmap_to_world_transform
    else:
      poses, transform = camera_utils.transform_poses_pca(poses)
      self.colmap_to_world_transform = transform
    self.poses = poses
    all_indices = np.arange(images.shape[0])
    train_indices = all_indices if config.llff_use_all_images_for_training else all_indices % config.llffhold != 0
    indices = {'test': all_indices[all_indices % config.llffhold == 0], 'train': train_indices}[self.split]
    self.images = images[indices]
    self.camtoworlds = poses[indices]
    self.height, self.width = self.images.shape[1:3]


dataset_dict = {'blender': Blender, 'llff': LLFF, 'procedural': Procedural}


def load_dataset(split, train_dir, config, device='cuda'):
  """datasets.py:40-52."""
  if config.dataset_loader not in dataset_dict:
    raise NotImplementedError(f'dataset_loader {config.dataset_loader!r}: only {sorted(dataset_dict)} are restated '
                              '(the others need COLMAP / rawpy data)')
  return dataset_dict[config.dataset_loader](split, train_dir, config, device=device)
