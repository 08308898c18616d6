"""Model(dense_precision='fp32') on the MI355X: the fp32-Dense debug build against the oracle in float64.  -m gpu.

The product runs its Dense layers as bf16 x bf16 -> fp32 MFMAs and stores activations and their gradients in bf16; its
composed parity tests (tests/test_gpu_model.py, tests/test_gpu_zz_fullsize.py, tests/test_gpu_sampling_grad.py) therefore
carry tolerances of 1e-2 ... 4e-1 against an oracle that emulates that rounding.  This file runs the SAME host code and the
SAME kernel sources (sampling, featurisation, compositing, losses, every hand-written VJP, the tangent network, the sampling
gradients) from libmnerf_hip_f32.so: float storage (csrc/common.h, MNR_DENSE_F32) and plain-FMA Dense layers
(csrc/dense_f32.inc) = the reference's jax-cpu precision (flax Dense in fp32, reference internal/models.py:436-437,
math.py:21-23).  What is left is fp32 arithmetic, and every output is held against the oracle evaluated in FLOAT64 on the same
float32 inputs (the oracle is pinned to the reference's own code in float64, tests/test_oracle_models_golden.py):

  forward    sdist 3e-5, weights 1e-4, rgb 1e-4 (absolute)
  gradients  relative L2 per top-level module <= max(GRAD_TOL = 2e-4, 2 x |oracle_fp32 - oracle_fp64|): where fp32 arithmetic
             itself costs more than 2e-4 the kernels must be about as close to float64 as the fp32 oracle is; both distances
             are printed.
  ReLU kinks fp32 cannot promise the SIGN of a pre-activation that is within its evaluation noise of 0, and a unit that takes the
             other side for one sample is not a 1e-7 error but a different (equally valid) subgradient: one such unit moves a
             small-batch gradient by 1e-3, and a few per million units do (counted and printed).  Both oracles therefore take the
             kernels' side of every kink (oracle.models.mlp_apply `relu_sides`, tests/helpers.py kernel_relu_sides), and the test
             asserts that float64's own sign disagrees on at most 5e-5 of the units, all with |z| < 5e-3 (typical |z| ~ 1).

The same cases as the bf16 product's: tests/test_gpu_model.CASES, configs/360.gin AS IS at full width, and the composed
stop_level_grad = False cases incl. 360.gin AS IS, which the bf16 product can only hold to the oracle's own bf16 cost (0.41).
"""

import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from multinerf_amd import configs, models, train_utils
from oracle import models as omodels
from oracle import train_utils as otrain
from tests import helpers
from tests.test_gpu_model import CASES, _setup
from tests.test_gpu_sampling_grad import COMPOSED

GRAD_TOL = 2e-4
FWD_TOL = dict(sdist=3e-5, weights=1e-4, rgb=1e-4)


@pytest.fixture(scope='module', autouse=True)
def _gpu():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')


def _check_step(tag, cfg, model, hp, params, flat, batch, tf, noise, grad_tol=GRAD_TOL, forward=True):
  om, on, op = hp
  assert model.dense_precision == 'fp32' and not model._chain_ok(model.prop_plan)
  dev = lambda t: t.cuda()
  noise_d = None if noise is None else {k: {lv: dev(t) for lv, t in d.items()} for k, d in noise.items()}
  if forward:
    r_64, h_64 = omodels.model_apply(om, on, op, helpers.to_float64(params), helpers.to_float64(batch.rays), tf, True, zero_glo=False,
                                     noise=helpers.to_float64(noise))
    rend, hist = model.apply({'flat': flat}, None, batch.rays.map(dev), tf, True, zero_glo=False, noise=noise_d)
    torch.cuda.synchronize()
    for lv in range(model.num_levels):
      e_s = (hist[lv]['sdist'].cpu().double() - h_64[lv]['sdist']).abs().max().item()
      e_w = (hist[lv]['weights'].cpu().double() - h_64[lv]['weights']).abs().max().item()
      print(f'F32MODE {tag} level {lv}: |sdist - oracle_fp64| {e_s:.2e}, |weights - oracle_fp64| {e_w:.2e}')
      assert e_s <= FWD_TOL['sdist'] and e_w <= FWD_TOL['weights'], (lv, e_s, e_w)
    e_rgb = (rend[-1]['rgb'].cpu().double() - r_64[-1]['rgb']).abs().max().item()
    print(f'F32MODE {tag}: |rgb - oracle_fp64| {e_rgb:.2e}')
    # (density-gradient normals, models.py:478-492, are a DERIVATIVE of the ReLU network: discontinuous across a kink, so a unit
    # whose pre-activation is within fp32 rounding of 0 moves the colour of its sample; the train-step check below takes the
    # kernels' side of every kink, this standalone forward check does not)
    assert e_rgb <= (1e-3 if any(p.tangent for p in model._plans) else FWD_TOL['rgb']), e_rgb
  state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone().cuda(), 'params': None})
  _, stats, _ = train_utils.create_train_step(model, cfg)(0, state, batch.map(dev), None, tf, 0.0, noise=noise_d, return_grads=True)
  torch.cuda.synchronize()
  s = stats.materialize()
  # the float64 oracle takes the kernels' side of every ReLU kink (tests/helpers.py kernel_relu_sides: fp32 cannot promise the
  # sign of a pre-activation that is within its rounding of 0, and one flipped unit of one sample moves a small-batch gradient
  # by 1e-3); the plain fp32 oracle, for the cost column, does not
  B = batch.rays.origins.shape[0]
  sides = helpers.kernel_relu_sides(model, B)
  stats_64, grads_64 = helpers.oracle_train_step_f64(params, om, on, op, cfg, batch, tf, noise, relu_sides=sides)
  _, _, _, grads_32 = otrain.train_step(params, otrain.init_opt_state(params), om, on, op, cfg, batch, tf, noise=noise,
                                        relu_sides={k: (None if v is None else {'masks': v['masks']}) for k, v in sides.items()})
  g_64 = helpers.flat_from_tree_f64(model, grads_64)
  g_32 = model.flat_from_tree(grads_32, device='cpu').double()
  assert abs(s['loss'] - float(stats_64['loss'])) <= 5e-5 * abs(float(stats_64['loss'])) + 1e-7, (s['loss'], float(stats_64['loss']))
  return helpers.check_fp32_mode_gradient(model, stats['_grads'], g_64, g_32, tag, grad_tol=grad_tol)


@pytest.mark.parametrize('name,extra,B', CASES)
def test_train_step_parity_fp32_mode(name, extra, B):
  """tests/test_gpu_model.py::test_train_step_parity's cases (all five BASELINE configurations and their thirty-two variants),
  forward and one train step, in the fp32-Dense mode against the float64 oracle."""
  cfg, m_bf, hp, params, _, batch = _setup(name, extra, B)
  model = models.Model(config=cfg, dense_precision='fp32').build('cuda')
  flat = model.flat_from_tree(params)
  _check_step(f'{name}{extra[:2]}', cfg, model, hp, params, flat, batch, 0.3, helpers.make_noise(model, B))


def test_full_width_train_step_gradient_fp32_mode():
  """configs/360.gin AS IS (1024-wide NeRF trunk, 256-wide proposal MLP, 9,007,493 parameters) on 256 rays: the configuration of
  tests/test_gpu_zz_fullsize.py::test_full_width_train_step_gradient_is_the_oracles, whose bf16 arm holds trunk layer 0 to
  0.135 / 0.31.  (On the simulator: reduced widths.)"""
  sim = os.environ.get('MNR_TESTS_ON_SIMULATOR') == '1'
  extra = ['NerfMLP.net_width = 128', 'PropMLP.net_width = 128', 'Model.num_prop_samples = 32', 'Model.num_nerf_samples = 32'] if sim else []
  B = 8 if sim else 256
  cfg, m_bf, hp, params, _, batch = _setup('360', extra, B, seed=0)
  model = models.Model(config=cfg, dense_precision='fp32').build('cuda')
  assert sim or model.num_params == 9007493
  flat = model.flat_from_tree(params)
  _check_step('360 full width', cfg, model, hp, params, flat, batch, 0.3, helpers.make_noise(model, B))


@pytest.mark.parametrize('name,preset,bindings,B,strict', COMPOSED)
def test_train_step_through_the_sampling_fp32_mode(name, preset, bindings, B, strict):
  """Model.stop_level_grad = False (models.py:198-201): tests/test_gpu_sampling_grad.py's composed cases, incl. configs/360.gin AS
  IS at twelve encoding degrees, where the bf16 product can only be held to the oracle's own bf16-vs-fp32 distance (0.41 for
  PropMLP_0, `strict=False` there).  In fp32 every case is strict."""
  if os.environ.get('MNR_TESTS_ON_SIMULATOR') == '1':
    bindings = bindings + ['NerfMLP.net_width = 128', 'PropMLP.net_width = 128', 'Model.num_prop_samples = 32', 'Model.num_nerf_samples = 16']
    B = 4
  cfg, m_bf, hp, params, _, batch = _setup(preset, bindings + ['Model.stop_level_grad = False'], B, seed=5)
  model = models.Model(config=cfg, dense_precision='fp32').build('cuda')
  flat = model.flat_from_tree(params)
  _check_step(f'sampling {name}', cfg, model, hp, params, flat, batch, 0.4, helpers.make_noise(model, B), forward=False)
