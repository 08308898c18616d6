"""The fused per-level Dense chain (csrc/fused_mlp.hip) against the per-layer GEMM path it replaces.  -m gpu.

Both paths feed the same bf16 operands to the same MFMA instruction in the same k order and apply the same bias / ReLU /
rounding, so the layer activations, their 1-bit masks and the dX chain must agree BIT FOR BIT; the Dense(1) head sums
256 products in a different order (fp32) and the weight gradients go through fp32 atomics: those are held to 1e-5.
The composed tests (tests/test_gpu_model.py) hold the fused path to the oracle.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multinerf_amd import configs, models, train_utils
from oracle import models as omodels
from tests import helpers


@pytest.fixture(scope='module', autouse=True)
def _gpu():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')


def _run(use_chain, cfg, flat, batch, noise, monkeypatch):
  monkeypatch.setattr(models, '_USE_CHAIN', use_chain)
  model = models.Model(config=cfg).build('cuda')
  state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone().cuda(), 'params': None})
  step = train_utils.create_train_step(model, cfg)
  _, stats, _ = step(0, state, batch.map(lambda t: t.cuda()), None, 0.4, 0.0,
                     noise={k: {lv: t.cuda() for lv, t in d.items()} for k, d in noise.items()}, return_grads=True)
  torch.cuda.synchronize()
  saved = model._saved['levels']
  out = dict(grads=stats['_grads'].cpu(), loss=stats.materialize()['loss'], model=model)
  for li, lv in enumerate(saved[:-1]):
    assert bool(lv['mlp'].get('chain')) == use_chain
    out[li] = dict(acts=[a.cpu().clone() for a in lv['mlp']['acts']], bits=[b.cpu().clone() for b in lv['mlp']['bits']],
                   raw=lv['mlp']['raw_density'].cpu().clone(), sdist=lv['sdist'].cpu().clone(), weights=lv['weights'].cpu().clone())
  return out


@pytest.mark.parametrize('W,bindings,B', [
    (256, ['NerfMLP.net_width = 128'], 24),                                            # 360.gin's PropMLP as is: K0 = 512
    (128, ['NerfMLP.net_width = 128', 'PropMLP.net_width = 128', 'PropMLP.net_depth = 3'], 16),
    (256, ['NerfMLP.net_width = 128', 'PropMLP.basis_shape = "octahedron"', 'PropMLP.basis_subdivisions = 1',
           'PropMLP.max_deg_point = 16', 'PropMLP.net_depth = 2'], 8),                 # K0 = 128, two layers
])
def test_fused_chain_equals_per_layer_path(W, bindings, B, monkeypatch):
  cfg = configs.load_preset('360', bindings)
  m0 = models.Model(config=cfg).build('cuda')
  assert m0.prop_plan.W == W and m0._chain_ok(m0.prop_plan)
  om, on, op = helpers.oracle_hparams(m0)
  params = omodels.init_params(om, on, op, seed=5)
  g = torch.Generator().manual_seed(6)
  for mod in params.values():
    for d in mod.values():
      if isinstance(d, dict) and 'bias' in d:
        d['bias'] = 0.05 * torch.randn(d['bias'].shape, generator=g)
  flat = m0.flat_from_tree(params)
  batch = helpers.synthetic_rays(B, near=cfg.near, far=cfg.far)
  noise = helpers.make_noise(m0, B)
  a = _run(True, cfg, flat, batch, noise, monkeypatch)
  b = _run(False, cfg, flat, batch, noise, monkeypatch)
  # level 0 sees identical inputs on both paths: bitwise activations and masks, head to fp32 rounding
  for i, (x, y) in enumerate(zip(a[0]['acts'], b[0]['acts'])):
    assert torch.equal(x.view(torch.int16), y.view(torch.int16)), f'level 0 activation {i}'
  for i, (x, y) in enumerate(zip(a[0]['bits'], b[0]['bits'])):
    assert torch.equal(x, y), f'level 0 mask bits {i}'
  scale = b[0]['raw'].abs().max().item()
  err = (a[0]['raw'] - b[0]['raw']).abs().max().item()
  print(f'W={W}: head |fused - per-layer| = {err:.2e} (|raw| <= {scale:.2e})')
  assert err <= 2e-6 * max(scale, 1.0)
  # the dX chain on identical inputs (level 0's saved masks, a seeded head gradient): bitwise against
  # small_head_bwd + one masked NT GEMM per layer
  model = a['model']
  plan = model.prop_plan
  lv = model._saved['levels'][0]
  M, D = lv['M'], len(plan.trunk)
  bits = lv['mlp']['bits']
  acts = lv['mlp']['acts']
  g_head = (torch.randn((M,), generator=torch.Generator().manual_seed(7)) * 0.01).cuda()
  w_head = flat.cuda()[plan.density.kernel_off:plan.density.kernel_off + W].contiguous()
  dYs = [torch.empty((M, W), dtype=torch.bfloat16, device='cuda') for _ in range(D)]
  Bws = [None] + [model._w(plan, plan.packed[('trunk', i)]['b_off'], models._rup(W, 128), plan.packed[('trunk', i)]['b_ld'])
                  for i in range(1, D)]
  ops = models.ops
  ops.mlp_chain_bwd(g_head, w_head, bits, Bws, dYs, M=M, W=W)
  ref = torch.empty((M, W), dtype=torch.bfloat16, device='cuda')
  ops.small_head_bwd(acts[-1], W, g_head.view(M, 1), w_head.view(W, 1), M=M, K=W, Cn=1, dX=ref, lddx=W, relu_mask=True)
  torch.cuda.synchronize()
  assert torch.equal(ref.view(torch.int16), dYs[-1].view(torch.int16)), 'dY of the last layer'
  for i in reversed(range(1, D)):
    nxt = torch.empty((M, W), dtype=torch.bfloat16, device='cuda')
    ops.gemm_nt(ref, Bws[i], M=M, N=models._rup(W, 128), K1=plan.packed[('trunk', i)]['b_ld'], Cb=nxt, ldcb=W, nb=W,
                bits_in=bits[i - 1])
    torch.cuda.synchronize()
    assert torch.equal(nxt.view(torch.int16), dYs[i - 1].view(torch.int16)), f'dY of layer {i - 1}'
    ref = nxt
  # end to end: downstream of the head's last-ulp differences everything stays close
  np.testing.assert_allclose(a[1]['sdist'].numpy(), b[1]['sdist'].numpy(), atol=2e-5)
  assert abs(a['loss'] - b['loss']) <= 1e-3 * abs(b['loss'])
  for name, lo, hi in model.modules:
    x, y = a['grads'][lo:hi].double(), b['grads'][lo:hi].double()
    rel = ((x - y).norm() / (y.norm() + 1e-30)).item()
    print(f'W={W} {name}: |g(fused) - g(per-layer)| / |g| = {rel:.2e}')
    # (samples move with the head's last ulp: not bitwise.  The proposal MLP's own gradient stays within 1e-3; what is
    # downstream of the moved samples sees bf16 roundings flip, 3-7 % at these 8-24 ray batches)
    assert rel < (2e-2 if name == 'PropMLP_0' else 0.2), (name, rel)


@pytest.mark.parametrize('preset,bindings,B', [
    ('llff_raw', [], 8),                                   # 96 -> 256 x 8, skip concat into layer 5 (K = 256 + 128), heads + view MLP
    ('blender_256', [], 8),                                # the same trunk behind a fused proposal level
    ('llff_raw', ['NerfMLP.net_width = 128', 'NerfMLP.net_depth = 4', 'NerfMLP.skip_layer = 2'], 8),   # W = 128, skip into layer 3
])
def test_fused_trunk_with_skip_equals_per_layer_path(preset, bindings, B, monkeypatch):
  """The 256-wide NeRF trunks (models.py:441-465 incl. the skip concat of :458-459) on the fused chain: activations and
  masks of every layer bit for bit against the per-layer GEMMs on identical inputs, the dX chain from a given dY_last
  bit for bit against one masked NT GEMM per layer, and the whole train step close to the per-layer path."""
  cfg = configs.load_preset(preset, bindings)
  m0 = models.Model(config=cfg).build('cuda')
  plan = m0.nerf_plan
  skips = [i for i, (_, c) in enumerate(plan.trunk) if c]
  if len(skips) != 1:
    pytest.skip(f'{len(skips)} skip layers')
  assert m0._chain_ok(plan) and plan.has_rgb
  om, on, op = helpers.oracle_hparams(m0)
  params = omodels.init_params(om, on, op, seed=5)
  g = torch.Generator().manual_seed(6)
  for mname, mod in params.items():
    if mname in ('exposure_scaling_offsets', 'Embed_0'):
      continue
    for d in mod.values():
      d['bias'] = 0.05 * torch.randn(d['bias'].shape, generator=g)
  flat = m0.flat_from_tree(params)
  batch = helpers.synthetic_rays(B, near=cfg.near, far=cfg.far)
  if cfg.rawnerf_mode:
    batch.rays.exposure_idx = torch.randint(0, 5, (B, 1), generator=g).to(torch.int32)
    batch.rays.exposure_values = 0.5 + torch.rand((B, 1), generator=g)
    batch.rays.lossmult = (torch.rand((B, 3), generator=g) > 0.4).float()
  noise = helpers.make_noise(m0, B)

  def run(use_chain):
    monkeypatch.setattr(models, '_USE_CHAIN', use_chain)
    model = models.Model(config=cfg).build('cuda')
    state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone().cuda(), 'params': None})
    step = train_utils.create_train_step(model, cfg)
    _, stats, _ = step(0, state, batch.map(lambda t: t.cuda()), None, 0.4, 0.0,
                       noise={k: {lv: t.cuda() for lv, t in d.items()} for k, d in noise.items()}, return_grads=True)
    torch.cuda.synchronize()
    lvs = model._saved['levels']
    lv = [l for l in lvs if l['plan'] is model.nerf_plan][0]          # first level that runs the NeRF MLP
    assert bool(lv['mlp'].get('chain_trunk')) == use_chain
    return dict(grads=stats['_grads'].cpu(), loss=stats.materialize()['loss'], model=model, lv=lv, first=lvs.index(lv),
                acts=[a.cpu().clone() for a in lv['mlp']['acts']], bits=[b.cpu().clone() for b in lv['mlp']['bits']])

  a, b = run(True), run(False)
  if a['first'] == 0:
    # level 0 sees identical inputs on both paths: every layer bit for bit (incl. the skip layer and what follows it)
    for i, (x, y) in enumerate(zip(a['acts'], b['acts'])):
      assert torch.equal(x.view(torch.int16), y.view(torch.int16)), f'activation {i}'
    for i, (x, y) in enumerate(zip(a['bits'], b['bits'])):
      assert torch.equal(x, y), f'mask bits {i}'
  # the dX chain from a seeded dY_last on the chain run's own masks: bitwise against one masked NT GEMM per layer
  model, lv = a['model'], a['lv']
  W, M, D = plan.W, lv['M'], len(plan.trunk)
  bits = lv['mlp']['bits']
  ops = models.ops
  gl = torch.Generator().manual_seed(7)
  keep = ((bits[-1].cpu().int()[..., None] >> torch.arange(8)) & 1).reshape(M, W).bool()
  dy_last = (torch.randn((M, W), generator=gl) * 0.01 * keep).to(torch.bfloat16).cuda()
  dYs = [torch.empty((M, W), dtype=torch.bfloat16, device='cuda') for _ in range(D - 1)] + [None]
  Bws = [None] + [model._w(model.nerf_plan, model.nerf_plan.packed[('trunk', i)]['b_off'], models._rup(W, 128),
                           model.nerf_plan.packed[('trunk', i)]['b_ld']) for i in range(1, D)]
  ops.mlp_chain_bwd(None, None, bits, Bws, dYs, M=M, W=W, dY_in=dy_last)
  ref = dy_last
  for i in reversed(range(1, D)):
    nxt = torch.empty((M, W), dtype=torch.bfloat16, device='cuda')
    ops.gemm_nt(ref, Bws[i], M=M, N=models._rup(W, 128), K1=model.nerf_plan.packed[('trunk', i)]['b_ld'], Cb=nxt, ldcb=W, nb=W,
                bits_in=bits[i - 1])
    torch.cuda.synchronize()
    assert torch.equal(nxt.view(torch.int16), dYs[i - 1].view(torch.int16)), f'dY of layer {i - 1}'
    ref = nxt
  # end to end
  assert abs(a['loss'] - b['loss']) <= 1e-3 * abs(b['loss'])
  for name, lo, hi in model.modules:
    x, y = a['grads'][lo:hi].double(), b['grads'][lo:hi].double()
    rel = ((x - y).norm() / (y.norm() + 1e-30)).item()
    print(f'{preset} {name}: |g(fused trunk) - g(per-layer)| / |g| = {rel:.2e}')
    assert rel < (1e-4 if a['first'] == 0 and cfg.interlevel_loss_mult == 0 and model.single_mlp else 0.2), (name, rel)


@pytest.mark.parametrize('bindings,B,n', [
    ([], 12, 64),                                                                      # 360.gin's PropMLP: K = 21, L = 12, W = 256, contract
    (['PropMLP.net_width = 128', 'PropMLP.net_depth = 2', 'PropMLP.warp_fn = None', 'PropMLP.basis_shape = "octahedron"',
      'PropMLP.basis_subdivisions = 1', 'PropMLP.max_deg_point = 16', 'Model.ray_shape = "cylinder"'], 8, 32),   # K = 9? / L = 16, W = 128, no warp
])
def test_chain_with_in_kernel_ipe_equals_feature_matrix_path(bindings, B, n):
  """mnr_mlp_chain_fwd_ipe (layer 0's IPE features built in LDS, four degrees at a time; render.py:103-127 + coord.py:102-133
  fused with models.py:441-465) against mnr_cast_rays_ipe + mnr_mlp_chain_fwd.  The features are the same bf16 values; only
  layer 0's fp32 accumulation order differs (group-major K), so its bf16 output differs by at most one ulp on a small fraction
  of the elements, and the head stays within 5e-3 of its scale (mean error < 1e-4)."""
  cfg = configs.load_preset('360', ['NerfMLP.net_width = 128'] + bindings)
  model = models.Model(config=cfg).build('cuda')
  plan = model.prop_plan
  assert model._ipe_chain_ok(plan)
  hp, W, D = plan.hp, plan.W, len(plan.trunk)
  flat = model.init_flat_params(seed=3)
  g = torch.Generator().manual_seed(4)
  for d in plan.dense:
    flat[d.bias_off:d.bias_off + d.fan_out] = (0.05 * torch.randn((d.fan_out,), generator=g)).cuda()
  model.pack_weights(flat, ipe=True)
  ops = models.ops
  M = B * n
  origins = (torch.randn((B, 3), generator=g) * 0.3).cuda()
  directions = torch.nn.functional.normalize(torch.randn((B, 3), generator=g), dim=-1).cuda() * 1.3
  radii = (0.002 + 0.002 * torch.rand((B,), generator=g)).cuda()
  tdist = torch.cumsum(0.02 + torch.rand((B, n + 1), generator=g) * 0.3, dim=-1).cuda().contiguous()     # in and beyond the unit ball
  kw = dict(ray_shape=model.ray_shape, warp_contract=(hp.warp_fn == 'contract'), min_deg=hp.min_deg_point, max_deg=hp.max_deg_point)
  feat = ops.cast_rays_ipe(tdist, origins, directions, radii, plan.basis_dev, ld_feat=plan.ldF, **kw)
  bias = lambda d: flat[d.bias_off:d.bias_off + d.fan_out]
  lay = lambda key: model._w(plan, plan.packed[key]['f_off'], plan.packed[key]['n_pad'], plan.packed[key]['f_ld'])
  ref_layers = [(lay(('trunk', i)), bias(d)) for i, (d, _) in enumerate(plan.trunk)]
  ipe_layers = [(lay('trunk0_ipe'), bias(plan.trunk[0][0]))] + ref_layers[1:]
  w_head = lay('density')[0]
  b_head = flat[plan.density.bias_off:plan.density.bias_off + 1]
  # layer 0 alone
  x_ref = torch.empty((M, W), dtype=torch.bfloat16, device='cuda')
  x_ipe = torch.empty((M, W), dtype=torch.bfloat16, device='cuda')
  ops.mlp_chain_fwd(feat, plan.ldF, ref_layers[:1], M=M, W=W, acts=[x_ref])
  ops.mlp_chain_fwd_ipe(tdist, origins, directions, radii, plan.basis_dev, ipe_layers[:1], M=M, W=W, act_last=x_ipe, **kw)
  torch.cuda.synchronize()
  a, b = x_ref.cpu().float(), x_ipe.cpu().float()
  frac = (a != b).float().mean().item()
  # one bf16 ulp of the larger value, or (cancellation: a pre-activation near zero) 1e-5 of the layer's scale
  excess = ((a - b).abs() - torch.maximum(torch.maximum(a.abs(), b.abs()) * 2.0 ** -7, torch.full_like(a, 1e-5 * a.abs().max().item()))).max().item()
  print(f'layer 0: {frac:.2e} of the bf16 outputs differ; max |diff| {(a - b).abs().max().item():.2e} (|x| <= {a.abs().max().item():.2e})')
  assert excess <= 0 and frac < 5e-3
  assert x_ref.float().abs().max().item() > 0.1                                      # (not a comparison of zeros)
  # the whole chain with its density head
  h_ref = torch.empty((M,), dtype=torch.float32, device='cuda')
  h_ipe = torch.empty((M,), dtype=torch.float32, device='cuda')
  ops.mlp_chain_fwd(feat, plan.ldF, ref_layers, M=M, W=W, w_head=w_head, b_head=b_head, head_out=h_ref)
  ops.mlp_chain_fwd_ipe(tdist, origins, directions, radii, plan.basis_dev, ipe_layers, M=M, W=W, w_head=w_head, b_head=b_head,
                        head_out=h_ipe, **kw)
  torch.cuda.synchronize()
  scale = h_ref.abs().max().item()
  err = (h_ref - h_ipe).abs()
  print(f'head: max |fused - matrix| = {err.max().item():.2e}, mean {err.mean().item():.2e} (|raw| <= {scale:.2e})')
  assert err.max().item() <= 5e-3 * scale and err.mean().item() <= 1e-4 * scale


def test_render_with_in_kernel_ipe_equals_feature_matrix_path(monkeypatch):
  """Model.__call__ without a backward pass (render / eval): the proposal levels on mnr_mlp_chain_fwd_ipe against the same
  forward pass with the feature matrices (MNR_FUSED_IPE=0); the composed oracle parity of this path is
  tests/test_gpu_model.py::test_forward_parity, which runs it by default."""
  cfg = configs.load_preset('360', ['NerfMLP.net_width = 128'])
  B = 24
  batch = helpers.synthetic_rays(B, near=cfg.near, far=cfg.far)
  out = {}
  for on in (True, False):
    monkeypatch.setattr(models, '_FUSED_IPE', on)
    model = models.Model(config=cfg).build('cuda')
    flat = model.init_flat_params(seed=3)
    calls = []
    real = models.ops.mlp_chain_fwd_ipe
    monkeypatch.setattr(models.ops, 'mlp_chain_fwd_ipe', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    renderings, hist = model.apply({'flat': flat}, None, batch.rays.map(lambda t: t.cuda()), 0.5, True)
    torch.cuda.synchronize()
    monkeypatch.setattr(models.ops, 'mlp_chain_fwd_ipe', real)
    assert len(calls) == (2 if on else 0)
    out[on] = (renderings[-1]['rgb'].cpu(), [h['sdist'].cpu() for h in hist], renderings[-1]['distance_median'].cpu())
  rgb_a, sd_a, dm_a = out[True]
  rgb_b, sd_b, dm_b = out[False]
  print(f'rgb |fused - matrix| = {(rgb_a - rgb_b).abs().max().item():.2e}, sdist {max((x - y).abs().max().item() for x, y in zip(sd_a, sd_b)):.2e}')
  # (the densities differ in their last bf16-induced digits, the resampled positions move with them)
  for x, y in zip(sd_a, sd_b):
    np.testing.assert_allclose(x.numpy(), y.numpy(), atol=1e-3)
  np.testing.assert_allclose(rgb_a.numpy(), rgb_b.numpy(), atol=5e-3)
  np.testing.assert_allclose(dm_a.numpy(), dm_b.numpy(), rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize('W,K0,depth,tiles,skip', [(128, 128, 2, 20, 0), (256, 64, 3, 12, 0), (128, 64, 3, 10, 2)])
def test_chain_small_grids_change_no_bit(W, K0, depth, tiles, skip):
  """The chain kernels' persistent loop at sizes the simulator finishes: with at most n workgroups (mnr_mlp_chain_set_max_wgs,
  include/mnerf_debug.h) every workgroup walks several tiles, and that must not change one bit of the forward pass (activations,
  masks, head) or of the dX chain.  skip > 0: that layer reads [x | features] (models.py:458-459), the feature segment streamed
  a second time per tile."""
  ops = models.ops
  dbg = ops.L.debug()
  M = tiles * 256
  g = torch.Generator().manual_seed(11)
  bf = torch.bfloat16
  feat = (torch.rand((M, K0), generator=g) * 2 - 1).to(bf).cuda()
  fan_in = lambda i: K0 if i == 0 else (W + K0 if i == skip else W)
  layers = [(((torch.rand((W, fan_in(i)), generator=g) * 2 - 1) * (6.0 / fan_in(i)) ** 0.5).to(bf).cuda(),
             (0.05 * torch.randn((W,), generator=g)).cuda()) for i in range(depth)]
  wh = ((torch.rand((W,), generator=g) * 2 - 1) * 0.15).to(bf).cuda()
  bh = torch.full((1,), 0.01).cuda()
  gh = (torch.randn((M,), generator=g) * 0.01).cuda()
  Bw = [None] + [layers[i][0][:, :W].t().contiguous() for i in range(1, depth)]     # (the dX chain runs over the x segment)

  def run():
    out = torch.empty((M,), device='cuda')
    acts = [torch.empty((M, W), dtype=bf, device='cuda') for _ in range(depth)]
    bits = [torch.empty((M, W // 8), dtype=torch.uint8, device='cuda') for _ in range(depth)]
    dY = [torch.empty((M, W), dtype=bf, device='cuda') for _ in range(depth)]
    ops.mlp_chain_fwd(feat, K0, layers, M=M, W=W, w_head=wh, b_head=bh, head_out=out, acts=acts, bits=bits, skip_layer=skip)
    ops.mlp_chain_bwd(gh, wh.float(), bits, Bw, dY, M=M, W=W)
    torch.cuda.synchronize()
    return [out.cpu()] + [a.cpu().view(torch.int16) for a in acts] + [b.cpu() for b in bits] + [d.cpu().view(torch.int16) for d in dY]

  ref = run()
  assert ref[1].float().abs().max().item() > 0                                      # (not a comparison of zeros)
  try:
    for max_wgs in (24, 5, 1):
      ops.L.check(dbg.mnr_mlp_chain_set_max_wgs(max_wgs))
      got = run()
      for i, (x, y) in enumerate(zip(ref, got)):
        assert torch.equal(x, y), f'max_wgs {max_wgs}: output {i} differs'
  finally:
    ops.L.check(dbg.mnr_mlp_chain_set_max_wgs(0))
