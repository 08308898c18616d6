"""CPU restatement of the ray-generation leaves of internal/camera_utils.py (TEST INFRASTRUCTURE ONLY).

Pinned against the reference's own source (executed with xnp = numpy) by
tests/golden/make_golden.py -> tests/golden/leaves.npz (keys cam_*).  Works on torch tensors of any
float dtype; every function cites the reference lines it restates.
"""

import enum

import torch


class ProjectionType(enum.Enum):
  """camera_utils.py:514-517."""
  PERSPECTIVE = 'perspective'
  FISHEYE = 'fisheye'


def convert_to_ndc(origins, directions, pixtocam, near=1.):
  """camera_utils.py:32-98."""
  t = -(near + origins[..., 2]) / directions[..., 2]
  origins = origins + t[..., None] * directions
  dx, dy, dz = directions.unbind(-1)
  ox, oy, oz = origins.unbind(-1)
  xmult = 1. / pixtocam[0, 2]
  ymult = 1. / pixtocam[1, 2]
  origins_ndc = torch.stack([xmult * ox / oz, ymult * oy / oz, -torch.ones_like(oz)], -1)
  infinity_ndc = torch.stack([xmult * dx / dz, ymult * dy / dz, torch.ones_like(oz)], -1)
  return origins_ndc, infinity_ndc - origins_ndc


def _compute_residual_and_jacobian(x, y, xd, yd, k1=0.0, k2=0.0, k3=0.0, k4=0.0, p1=0.0, p2=0.0):
  """camera_utils.py:427-474."""
  r = x * x + y * y
  d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
  fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
  fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
  d_r = (k1 + r * (2.0 * k2 + r * (3.0 * k3 + r * 4.0 * k4)))
  d_x = 2.0 * x * d_r
  d_y = 2.0 * y * d_r
  fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
  fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
  fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
  fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
  return fx, fy, fx_x, fx_y, fy_x, fy_y


def _radial_and_tangential_undistort(xd, yd, k1=0, k2=0, k3=0, k4=0, p1=0, p2=0, eps=1e-9, max_iterations=10):
  """camera_utils.py:477-511 (Newton iterations from the distorted point)."""
  x, y = xd.clone(), yd.clone()
  for _ in range(max_iterations):
    fx, fy, fx_x, fx_y, fy_x, fy_y = _compute_residual_and_jacobian(x, y, xd, yd, k1, k2, k3, k4, p1, p2)
    denominator = fy_x * fx_y - fx_x * fy_y
    x_numerator = fx * fy_y - fy * fx_y
    y_numerator = fy * fx_x - fx * fy_x
    ok = denominator.abs() > eps
    x = x + torch.where(ok, x_numerator / denominator, torch.zeros_like(denominator))
    y = y + torch.where(ok, y_numerator / denominator, torch.zeros_like(denominator))
  return x, y


def pixels_to_rays(pix_x_int, pix_y_int, pixtocams, camtoworlds, distortion_params=None, pixtocam_ndc=None,
                   camtype=ProjectionType.PERSPECTIVE):
  """camera_utils.py:520-631 -> (origins, directions, viewdirs, radii, imageplane)."""
  dt = pixtocams.dtype
  px, py = pix_x_int.to(dt), pix_y_int.to(dt)

  def pix_to_dir(x, y):
    return torch.stack([x + .5, y + .5, torch.ones_like(x)], -1)

  pixel_dirs_stacked = torch.stack([pix_to_dir(px, py), pix_to_dir(px + 1, py), pix_to_dir(px, py + 1)], 0)
  mat_vec_mul = lambda A, b: torch.matmul(A, b[..., None])[..., 0]
  camera_dirs_stacked = mat_vec_mul(pixtocams, pixel_dirs_stacked)
  if distortion_params is not None:
    x, y = _radial_and_tangential_undistort(camera_dirs_stacked[..., 0], camera_dirs_stacked[..., 1],
                                            **distortion_params)
    camera_dirs_stacked = torch.stack([x, y, torch.ones_like(x)], -1)
  if camtype == ProjectionType.FISHEYE:
    theta = torch.sqrt(torch.sum(camera_dirs_stacked[..., :2]**2, -1))
    theta = torch.clamp(theta, max=torch.pi)
    sin_theta_over_theta = torch.sin(theta) / theta
    camera_dirs_stacked = torch.stack([camera_dirs_stacked[..., 0] * sin_theta_over_theta,
                                       camera_dirs_stacked[..., 1] * sin_theta_over_theta,
                                       torch.cos(theta)], -1)
  camera_dirs_stacked = camera_dirs_stacked * torch.tensor([1., -1., -1.], dtype=dt)   # OpenCV -> OpenGL
  imageplane = camera_dirs_stacked[0, ..., :2]
  directions_stacked = mat_vec_mul(camtoworlds[..., :3, :3], camera_dirs_stacked)
  directions, dx, dy = directions_stacked[0], directions_stacked[1], directions_stacked[2]
  origins = camtoworlds[..., :3, -1].expand(directions.shape)
  viewdirs = directions / torch.linalg.norm(directions, dim=-1, keepdim=True)
  if pixtocam_ndc is None:
    dx_norm = torch.linalg.norm(dx - directions, dim=-1)
    dy_norm = torch.linalg.norm(dy - directions, dim=-1)
  else:
    origins_dx, _ = convert_to_ndc(origins, dx, pixtocam_ndc)
    origins_dy, _ = convert_to_ndc(origins, dy, pixtocam_ndc)
    origins, directions = convert_to_ndc(origins, directions, pixtocam_ndc)
    dx_norm = torch.linalg.norm(origins_dx - origins, dim=-1)
    dy_norm = torch.linalg.norm(origins_dy - origins, dim=-1)
  radii = (0.5 * (dx_norm + dy_norm))[..., None] * 2 / (12.0**0.5)
  return origins, directions, viewdirs, radii, imageplane


def cast_ray_batch(cameras, pixels, camtype=ProjectionType.PERSPECTIVE, rays_cls=None):
  """camera_utils.py:634-688.  `pixels` has the utils.Pixels fields; returns rays_cls(**fields) (or a dict)."""
  pixtocams, camtoworlds, distortion_params, pixtocam_ndc = cameras
  cam_idx = pixels.cam_idx[..., 0].long()
  batch_index = lambda arr: arr if arr.ndim == 2 else arr[cam_idx]
  origins, directions, viewdirs, radii, imageplane = pixels_to_rays(
      pixels.pix_x_int, pixels.pix_y_int, batch_index(pixtocams), batch_index(camtoworlds),
      distortion_params=distortion_params, pixtocam_ndc=pixtocam_ndc, camtype=camtype)
  fields = dict(origins=origins, directions=directions, viewdirs=viewdirs, radii=radii, imageplane=imageplane,
                lossmult=pixels.lossmult, near=pixels.near, far=pixels.far, cam_idx=pixels.cam_idx,
                exposure_idx=pixels.exposure_idx, exposure_values=pixels.exposure_values)
  return rays_cls(**fields) if rays_cls is not None else fields
