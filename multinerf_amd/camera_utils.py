"""Ray generation on device: the reference's camera_utils.pixels_to_rays / cast_ray_batch
(internal/camera_utils.py:514-688) behind the same names, backed by mnr_pixels_to_rays (csrc/camera.hip).

Only the per-pixel leaves are here; pose utilities / camera paths stay out of scope.
"""

import ctypes as C
import enum

import numpy as np
import torch

from multinerf_amd import _lib as L
from multinerf_amd import ops, utils


class ProjectionType(enum.Enum):
  """camera_utils.py:514-517."""
  PERSPECTIVE = 'perspective'
  FISHEYE = 'fisheye'


_DIST_KEYS = ('k1', 'k2', 'k3', 'k4', 'p1', 'p2')


def pixels_to_rays(pix_x_int, pix_y_int, pixtocams, camtoworlds, distortion_params=None, pixtocam_ndc=None,
                   camtype=ProjectionType.PERSPECTIVE, cam_idx=None):
  """camera_utils.py:520-631.  pixtocams [3,3] or [N,3,3] (+ cam_idx [...] selecting the camera per pixel; the
  reference indexes the stacks before the call, `batch_index` camera_utils.py:665).  Device tensors in,
  (origins, directions, viewdirs, radii, imageplane) out, shaped like the pixel arrays."""
  shape = tuple(pix_x_int.shape)
  dev = pix_x_int.device
  if not ops._on_device(pix_x_int):
    raise ValueError('pixels_to_rays: device tensors required (the HIP path has no CPU fallback)')
  px = pix_x_int.reshape(-1).to(torch.int32).contiguous()
  py = pix_y_int.reshape(-1).to(torch.int32).contiguous()
  B = px.numel()
  P = pixtocams.to(device=dev, dtype=torch.float32).reshape(-1, 3, 3).contiguous()
  Cw = camtoworlds.to(device=dev, dtype=torch.float32)[..., :3, :4].reshape(-1, 3, 4).contiguous()
  ncam = max(P.shape[0], Cw.shape[0])
  if P.shape[0] != ncam:
    P = P.expand(ncam, 3, 3).contiguous()
  if Cw.shape[0] != ncam:
    Cw = Cw.expand(ncam, 3, 4).contiguous()
  ci = None
  if ncam > 1:
    if cam_idx is None:
      raise ValueError('pixels_to_rays: stacked cameras need cam_idx')
    ci = cam_idx.reshape(-1).to(torch.int32).contiguous()
  dist = None
  if distortion_params is not None:
    unknown = set(distortion_params) - set(_DIST_KEYS)
    if unknown:
      raise TypeError(f'unexpected distortion parameters {sorted(unknown)}')      # **kwargs error of the reference
    dist = (C.c_float * 6)(*[float(distortion_params.get(k, 0.0)) for k in _DIST_KEYS])
  ndc = None
  if pixtocam_ndc is not None:
    ndc = torch.as_tensor(pixtocam_ndc).to(device=dev, dtype=torch.float32).reshape(3, 3).contiguous()
  f32 = torch.float32
  origins = torch.empty((B, 3), dtype=f32, device=dev)
  directions = torch.empty((B, 3), dtype=f32, device=dev)
  viewdirs = torch.empty((B, 3), dtype=f32, device=dev)
  radii = torch.empty((B, 1), dtype=f32, device=dev)
  imageplane = torch.empty((B, 2), dtype=f32, device=dev)
  p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
  L.check(ops.lib().mnr_pixels_to_rays(
      B, p(px), p(py), p(ci), ncam, p(P), p(Cw), C.cast(dist, C.c_void_p) if dist is not None else None, p(ndc),
      1 if ProjectionType(camtype) == ProjectionType.FISHEYE else 0, p(origins), p(directions), p(viewdirs), p(radii),
      p(imageplane), ops._stream()))
  r = lambda t, c: t.reshape(shape + (c,))
  return r(origins, 3), r(directions, 3), r(viewdirs, 3), r(radii, 1), r(imageplane, 2)


def cast_ray_batch(cameras, pixels, camtype=ProjectionType.PERSPECTIVE, xnp=None):
  """camera_utils.py:634-688: utils.Pixels -> utils.Rays (`xnp` is accepted and ignored: arrays are torch)."""
  pixtocams, camtoworlds, distortion_params, pixtocam_ndc = cameras
  origins, directions, viewdirs, radii, imageplane = pixels_to_rays(
      pixels.pix_x_int, pixels.pix_y_int, pixtocams, camtoworlds, distortion_params=distortion_params,
      pixtocam_ndc=pixtocam_ndc, camtype=camtype, cam_idx=pixels.cam_idx[..., 0])
  return utils.Rays(origins=origins, directions=directions, viewdirs=viewdirs, radii=radii, imageplane=imageplane,
                    lossmult=pixels.lossmult, near=pixels.near, far=pixels.far, cam_idx=pixels.cam_idx,
                    exposure_idx=pixels.exposure_idx, exposure_values=pixels.exposure_values)


def intrinsic_matrix(fx, fy, cx, cy):
  """camera_utils.py:398-408."""
  return torch.tensor([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]], dtype=torch.float64)


def get_pixtocam(focal, width, height):
  """camera_utils.py:411-417 (inverted in fp64, returned fp32)."""
  return torch.linalg.inv(intrinsic_matrix(focal, focal, width * .5, height * .5)).float()


def pixel_coordinates(width, height, device='cpu'):
  """camera_utils.py:420-424: (x, y) integer grids, each [height, width]."""
  y, x = torch.meshgrid(torch.arange(height, device=device), torch.arange(width, device=device), indexing='ij')
  return x, y


# ----------------------------------------------------------------------------- pose utilities (host, NumPy):
# scene normalisation done once at dataset load time (camera_utils.py:101-156,191-227), not on the per-ray path.


def pad_poses(p):
  """camera_utils.py:101-104: [..., 3, 4] -> [..., 4, 4] with a [0, 0, 0, 1] bottom row."""
  bottom = np.broadcast_to([0, 0, 0, 1.], p[..., :1, :4].shape)
  return np.concatenate([p[..., :3, :4], bottom], axis=-2)


def unpad_poses(p):
  """camera_utils.py:107-109."""
  return p[..., :3, :4]


def normalize(x):
  """camera_utils.py:139-141."""
  return x / np.linalg.norm(x)


def viewmatrix(lookdir, up, position):
  """camera_utils.py:129-136."""
  vec2 = normalize(lookdir)
  vec0 = normalize(np.cross(up, vec2))
  vec1 = normalize(np.cross(vec2, vec0))
  return np.stack([vec0, vec1, vec2, position], axis=1)


def average_pose(poses):
  """camera_utils.py:120-126."""
  position = poses[:, :3, 3].mean(0)
  z_axis = poses[:, :3, 2].mean(0)
  up = poses[:, :3, 1].mean(0)
  return viewmatrix(z_axis, up, position)


def recenter_poses(poses):
  """Express all cameras in the frame of their average camera (what camera_utils.py:112-117 computes).

  Returns (poses [N,3,4], world -> average-camera transform [4,4]).  The average frame [R | c] is rigid, so its
  inverse is written down directly, [R^T | -R^T c], instead of inverting the padded 4x4.
  """
  mean_frame = average_pose(poses)
  r_t = mean_frame[:3, :3].T
  world_to_mean = np.eye(4)
  world_to_mean[:3, :3] = r_t
  world_to_mean[:3, 3] = -r_t @ mean_frame[:3, 3]
  return np.einsum('ij,njk->nik', world_to_mean[:3, :3], poses[:, :3, :4]) + \
      np.concatenate([np.zeros((3, 3)), world_to_mean[:3, 3:4]], 1)[None], world_to_mean


def focus_point_fn(poses):
  """The point closest, in the least-squares sense, to every camera's optical axis (camera_utils.py:144-156).

  With unit axis directions d_i through centres o_i the distance to axis i is |P_i (x - o_i)|, P_i = I - d_i d_i^T an
  orthogonal projector (P_i^T P_i = P_i), so the normal equations are (sum P_i) x = sum P_i o_i.
  """
  d = poses[:, :3, 2]
  o = poses[:, :3, 3]
  proj = np.eye(3)[None] - d[:, :, None] * d[:, None, :]
  proj = np.einsum('nji,njk->nik', proj, proj)              # P^T P: equals P for unit d, kept general for un-normalised axes
  return np.linalg.solve(proj.sum(0), np.einsum('nij,nj->i', proj, o))


def transform_poses_pca(poses):
  """Scene normalisation for unbounded captures (what camera_utils.py:191-227 computes): rotate the principal axes of
  the camera centres onto x, y, z (largest spread first), keep the cameras' mean up-vector pointing to +z... i.e. a
  positive z component of the mean y axis, and scale the centres into [-1, 1]^3.  Returns (poses, 4x4 transform).

  A principal axis is only defined up to sign, and the two fix-ups below (handedness, up-vector) leave a 180-degree turn
  about z open.  Checkpoints and render paths are expressed in this frame, so the signs have to be the reference's: they are
  whatever LAPACK's general eigen-solver returns for the 3x3 scatter matrix of the centres, which is why this function asks
  the same solver the same question (`np.linalg.eig` of c^T c, camera_utils.py:205) instead of an SVD with a sign rule of its
  own (rounds 1-3: identical to the reference on 11 of the 12 captures of tests/golden/pca_poses.npz, diag(-1, -1, 1) off on
  one; now 12 of 12, tests/test_oracle_camera.py).
  """
  centres = poses[:, :3, 3]
  mean = centres.mean(0)
  c = centres - mean
  spread, vecs = np.linalg.eig(c.T @ c)
  axes = vecs[:, np.argsort(spread)[::-1]].T                              # rows: principal directions, largest spread first
  if np.linalg.det(axes) < 0:                                           # keep a right-handed frame
    axes[2] = -axes[2]
  world_to_pca = np.eye(4)
  world_to_pca[:3, :3] = axes
  world_to_pca[:3, 3] = -axes @ mean
  out = np.einsum('ij,njk->nik', axes, poses[:, :3, :4])
  out[:, :, 3] += world_to_pca[:3, 3]
  if out[:, 2, 1].mean() < 0:                                           # cameras upside down on average: turn about x
    turn = np.diag([1., -1., -1.])
    out = np.einsum('ij,njk->nik', turn, out)
    world_to_pca[:3] = turn @ world_to_pca[:3]
  shrink = 1.0 / np.abs(out[:, :, 3]).max()
  out[:, :, 3] *= shrink
  world_to_pca[:3] *= shrink
  return out, world_to_pca
