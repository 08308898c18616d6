"""mnr_pixels_to_rays (csrc/camera.hip) vs the committed goldens of the reference's camera_utils and vs the
CPU oracle on a larger seeded batch; train_step with Config.cast_rays_in_train_step.  -m gpu."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multinerf_amd import camera_utils, configs, models, train_utils, utils
from oracle import camera_utils as ocam
from tests import helpers
from tests.test_oracle_camera import FIELDS, _cases, _pixels


@pytest.fixture(scope='module', autouse=True)
def _gpu():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')


def _to_dev_cams(cams):
  P, C, dist, ndc = cams
  return (P.cuda(), C.cuda(), dist, None if ndc is None else ndc.cuda())


@pytest.mark.parametrize('name', ['persp', 'single', 'dist', 'fisheye', 'ndc'])
def test_cast_ray_batch_vs_reference_golden(golden, name):
  cams, ct = _cases(golden, torch.float32)[name]
  pix = _pixels(golden).map(lambda t: t.cuda())
  rays = camera_utils.cast_ray_batch(_to_dev_cams(cams), pix, camera_utils.ProjectionType(ct.value))
  for f in FIELDS:
    got = getattr(rays, f).cpu().numpy()
    want = golden[f'cam_{name}_{f}']
    assert got.shape == want.shape, f
    # fp32 kernel vs the reference evaluated in fp64: the fp32 oracle sits at the same distance
    np.testing.assert_allclose(got, want, rtol=3e-4, atol=3e-5, err_msg=f)


@pytest.mark.parametrize('name', ['persp', 'dist', 'fisheye', 'ndc'])
def test_pixels_to_rays_vs_oracle_large(golden, name):
  cams, ct = _cases(golden, torch.float32)[name]
  g = torch.Generator().manual_seed(5)
  shape = (37, 29)                     # arbitrary leading dims, like an image
  px = torch.randint(0, 64, shape, generator=g)
  py = torch.randint(0, 48, shape, generator=g)
  ci = torch.randint(0, 5, shape + (1,), generator=g)
  B = (37, 29, 1)
  pix = utils.Pixels(pix_x_int=px, pix_y_int=py, lossmult=torch.ones(B), near=torch.zeros(B), far=torch.ones(B),
                     cam_idx=ci)
  want = ocam.cast_ray_batch(cams, pix, ct)
  rays = camera_utils.cast_ray_batch(_to_dev_cams(cams), pix.map(lambda t: t.cuda()),
                                     camera_utils.ProjectionType(ct.value))
  for f in FIELDS:
    got = getattr(rays, f).cpu()
    assert got.shape == want[f].shape, f
    err = (got - want[f]).abs().max().item()
    scale = want[f].abs().max().item()
    print(f'{name} {f}: max abs err {err:.2e} (scale {scale:.2e})')
    assert err <= 2e-5 * max(1.0, scale), f


def test_train_step_casts_rays_on_device():
  """Config.cast_rays_in_train_step: batch.rays holds utils.Pixels; same step as with pre-cast rays."""
  cfg = configs.load_preset('blender_256', ['Config.cast_rays_in_train_step = True', 'NerfMLP.net_width = 128',
                                            'PropMLP.net_width = 128', 'NerfMLP.bottleneck_width = 128'])
  model = models.Model(config=cfg)
  model.build('cuda')
  flat = model.init_flat_params(seed=2)
  B = 64
  g = torch.Generator().manual_seed(9)
  focal, W, H = 80.0, 64, 64
  pixtocam = torch.linalg.inv(torch.tensor([[focal, 0, W / 2.], [0, focal, H / 2.], [0, 0, 1.]]))
  c2w = torch.eye(4)[:3].clone()
  c2w[2, 3] = 4.0
  cameras = (pixtocam.cuda(), c2w.cuda(), None, None)
  one = lambda v: torch.full((B, 1), v)
  pix = utils.Pixels(pix_x_int=torch.randint(0, W, (B,), generator=g), pix_y_int=torch.randint(0, H, (B,), generator=g),
                     lossmult=one(1.0), near=one(2.0), far=one(6.0), cam_idx=torch.zeros((B, 1), dtype=torch.int32))
  rgb = torch.rand((B, 3), generator=g)
  noise = helpers.make_noise(model, B)
  outs = []
  for mode in ('pixels', 'rays'):
    cfg.cast_rays_in_train_step = mode == 'pixels'
    state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone(), 'params': None})
    step = train_utils.create_train_step(model, cfg)
    pd = pix.map(lambda t: t.cuda())
    rays_in = pd if mode == 'pixels' else camera_utils.cast_ray_batch(cameras, pd)
    batch = utils.Batch(rays=rays_in, rgb=rgb.cuda())
    state, stats, _ = step(0, state, batch, cameras, 0.5, 0.0, noise=noise, return_grads=True)
    outs.append((stats.materialize()['loss'], stats['_grads'].clone()))
  assert abs(outs[0][0] - outs[1][0]) < 1e-6
  a, b = outs[0][1].double(), outs[1][1].double()
  assert ((a - b).norm() / b.norm()).item() < 1e-3       # same rays -> same step up to atomics order
