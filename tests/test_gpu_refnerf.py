"""Ref-NeRF kernels (refnerf.hip, tangent features) vs the oracle and its autograd.  -m gpu."""

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import coord as ocoord
from oracle import image as oimage
from oracle import ref_utils as oref
from oracle import render as orender
from oracle import train_utils as otrain


@pytest.fixture(scope='module')
def ops():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from multinerf_amd import ops as _ops
  _ops.lib()
  return _ops


def dev(x):
  return x.contiguous().cuda()


def _head_ref(small, raw_grad, viewdirs_s, rb):
  """Oracle composition of models.py:492-563 for already-flattened samples (viewdirs per sample)."""
  gp = small[:, 1:4]
  npred = -oref.l2_normalize(gp)
  nrm = -oref.l2_normalize(raw_grad.T)
  rough = torch.nn.functional.softplus(small[:, 10] + rb)
  refdirs = oref.reflect(-viewdirs_s, npred)
  ide = oref.generate_ide_fn(5)(refdirs, rough[:, None])
  ndv = (npred * viewdirs_s).sum(-1, keepdim=True)
  return nrm, npred, rough, torch.cat([ide, ndv], -1)


def test_ref_head_fwd_bwd(ops):
  gen = torch.Generator().manual_seed(21)
  B, n = 23, 16
  M = B * n
  small = torch.randn((M, 11), generator=gen)
  small[5, 1:4] = 0.0                      # zero predicted gradient: the eps clamp of l2_normalize
  raw_grad = torch.randn((3, M), generator=gen) * 3
  v = torch.randn((B, 3), generator=gen)
  v = v / v.norm(dim=-1, keepdim=True)
  vs = v[:, None, :].expand(B, n, 3).reshape(M, 3)
  rb = -1.0
  tabs = ops.IdeTablesDev(5, 'cuda')
  vi = torch.full((M, 256), 3.0, dtype=torch.bfloat16).cuda()
  nrm, npred, rough = ops.ref_head_fwd(dev(small), dev(raw_grad), dev(v), n, tabs, rb, vi, 128, 256)

  s64, rg64, vs64 = small.double().requires_grad_(True), raw_grad.double().requires_grad_(True), vs.double()
  nrm_r, npred_r, rough_r, enc_r = _head_ref(s64, rg64, vs64, rb)
  np.testing.assert_allclose(nrm.cpu().numpy(), nrm_r.detach().numpy(), rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(npred.cpu().numpy(), npred_r.detach().numpy(), rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(rough.cpu().numpy(), rough_r.detach().numpy(), rtol=1e-5, atol=1e-6)
  enc = vi.cpu().float()[:, 128:128 + 73]
  # bf16 storage (2^-8 relative) on top of the fp32 evaluation of the degree-16 polynomials.
  np.testing.assert_allclose(enc.numpy(), enc_r.detach().numpy(), rtol=2**-7, atol=2e-3)
  assert (vi.cpu().float()[:, :128] == 3.0).all() and (vi.cpu().float()[:, 201:] == 0.0).all()

  # VJP vs autograd (fp64 oracle)
  g_enc = torch.randn((M, 73), generator=gen) * 0.1
  g_np_l = torch.randn((M, 3), generator=gen) * 0.1
  g_n_l = torch.randn((M, 3), generator=gen) * 0.1
  obj = (enc_r * g_enc.double()).sum() + (npred_r * g_np_l.double()).sum() + (nrm_r * g_n_l.double()).sum()
  obj.backward()
  dvi_a = torch.zeros((M, 256), dtype=torch.bfloat16)
  dvi_b = torch.zeros((M, 256), dtype=torch.bfloat16)
  ga = (g_enc * 0.6).to(torch.bfloat16)
  gb = (g_enc - ga.float()).to(torch.bfloat16)
  dvi_a[:, 128:201], dvi_b[:, 128:201] = ga, gb
  g_used = ga.float() + gb.float()
  dhb = torch.zeros((M, 256), dtype=torch.bfloat16).cuda()
  g_rg = ops.ref_head_bwd(dev(small), dev(raw_grad), dev(v), n, tabs, rb, dev(dvi_a), dev(dvi_b), 128,
                          dev(g_np_l), dev(g_n_l), dhb, 129, 138)
  # autograd with the bf16-rounded upstream actually fed to the kernel
  s2, rg2 = small.double().requires_grad_(True), raw_grad.double().requires_grad_(True)
  nrm2, npred2, rough2, enc2 = _head_ref(s2, rg2, vs64, rb)
  ((enc2 * g_used.double()).sum() + (npred2 * g_np_l.double()).sum() + (nrm2 * g_n_l.double()).sum()).backward()
  np.testing.assert_allclose(g_rg.cpu().numpy(), rg2.grad.numpy(), rtol=1e-4, atol=1e-6)
  got = dhb.cpu().float()
  ref_gp, ref_r = s2.grad[:, 1:4], s2.grad[:, 10]
  sc = ref_gp.abs().max().item()
  np.testing.assert_allclose(got[:, 129:132].numpy(), ref_gp.numpy(), rtol=2e-2, atol=2e-3 * sc)
  np.testing.assert_allclose(got[:, 138].numpy(), ref_r.numpy(), rtol=2e-2, atol=2e-3 * ref_r.abs().max().item())
  assert (got[:, :129] == 0).all() and (got[:, 132:138] == 0).all()


def _head_ref_features(small, raw_grad, viewdirs_s, rb, F, deg, ops):
  """models.py:492-563 for a feature set (MNR_REF_* bits): the oracle's leaves composed the way the reference composes them."""
  npred = -oref.l2_normalize(small[:, 1:4]) if F & ops.REF_PRED_NORMALS else None
  nrm = -oref.l2_normalize(raw_grad.T) if F & ops.REF_DENSITY_NORMALS else None
  nu = npred if npred is not None else nrm
  rough = torch.nn.functional.softplus(small[:, 10] + rb) if F & ops.REF_ROUGHNESS else None
  d = oref.reflect(-viewdirs_s, nu) if F & ops.REF_REFLECT else viewdirs_s
  enc = oref.generate_ide_fn(deg)(d, rough[:, None]) if F & ops.REF_IDE else ocoord.pos_enc(d, 0, deg, True)
  cols = [enc] + ([(nu * viewdirs_s).sum(-1, keepdim=True)] if F & ops.REF_N_DOT_V else [])
  return nrm, npred, rough, torch.cat(cols, -1)


# (predicted normals, density normals, reflect, IDE, n.v, roughness) = bits 1, 2, 4, 8, 16, 32
@pytest.mark.parametrize('F,deg', [(63, 5), (62, 5), (47, 5), (1 | 2 | 4 | 16, 4), (1 | 2 | 4 | 16 | 32, 4), (1 | 2 | 16 | 32, 4),
                                   (2 | 4, 3), (32, 4), (1 | 16, 2)])
def test_ref_head_feature_sets(ops, F, deg):
  """Every feature set the reference can run (models.py:468-563): the complete head without predicted normals (reflections about
  the density gradient's normals), without n.v, the positional encoding of the reflection direction with and without a
  roughness head, the view direction's encoding next to n.v, roughness as an output only ..."""
  gen = torch.Generator().manual_seed(100 + F)
  B, n = 9, 16
  M = B * n
  small = torch.randn((M, 11), generator=gen)
  raw_grad = torch.randn((3, M), generator=gen) * 3
  v = torch.randn((B, 3), generator=gen)
  v = v / v.norm(dim=-1, keepdim=True)
  vs = v[:, None, :].expand(B, n, 3).reshape(M, 3)
  rb = -1.0
  tabs = ops.IdeTablesDev(deg, 'cuda') if F & ops.REF_IDE else None
  has_dn, has_pn = bool(F & ops.REF_DENSITY_NORMALS), bool(F & ops.REF_PRED_NORMALS)
  vi = torch.full((M, 256), 3.0, dtype=torch.bfloat16).cuda()
  nrm, npred, rough = ops.ref_head_fwd(dev(small), dev(raw_grad) if has_dn else None, dev(v), n, tabs, rb, vi, 128, 256,
                                       features=F, deg_view=deg)
  s2, rg2, vs64 = small.double().requires_grad_(True), raw_grad.double().requires_grad_(True), vs.double()
  nrm_r, npred_r, rough_r, enc_r = _head_ref_features(s2, rg2, vs64, rb, F, deg, ops)
  for got, want in ((nrm, nrm_r), (npred, npred_r), (rough, rough_r)):
    assert (got is None) == (want is None)
    if got is not None:
      np.testing.assert_allclose(got.cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-6)
  E = enc_r.shape[1]
  assert E == (2 * tabs.c.T if F & ops.REF_IDE else 3 + 6 * deg) + (1 if F & ops.REF_N_DOT_V else 0)
  got_vi = vi.cpu().float()
  np.testing.assert_allclose(got_vi[:, 128:128 + E].numpy(), enc_r.detach().numpy(), rtol=2**-7, atol=2e-3)
  assert (got_vi[:, :128] == 3.0).all() and (got_vi[:, 128 + E:] == 0.0).all()

  g_enc = torch.randn((M, E), generator=gen) * 0.1
  g_np_l = torch.randn((M, 3), generator=gen) * 0.1 if has_pn else None
  g_n_l = torch.randn((M, 3), generator=gen) * 0.1 if has_dn else None
  dvi_a = torch.zeros((M, 256), dtype=torch.bfloat16)
  dvi_a[:, 128:128 + E] = g_enc.to(torch.bfloat16)
  g_used = dvi_a[:, 128:128 + E].double()
  obj = (enc_r * g_used).sum()
  if has_pn:
    obj = obj + (npred_r * g_np_l.double()).sum()
  if has_dn:
    obj = obj + (nrm_r * g_n_l.double()).sum()
  if obj.requires_grad:
    obj.backward()
  dhb = torch.full((M, 256), 7.0, dtype=torch.bfloat16).cuda()
  g_rg = ops.ref_head_bwd(dev(small), dev(raw_grad) if has_dn else None, dev(v), n, tabs, rb, dev(dvi_a), None, 128,
                          dev(g_np_l) if has_pn else None, dev(g_n_l) if has_dn else None, dhb, 129, 138, features=F, deg_view=deg)
  assert (g_rg is None) == (not has_dn)
  if has_dn:
    want = rg2.grad.numpy() if rg2.grad is not None else np.zeros((3, M))
    np.testing.assert_allclose(g_rg.cpu().numpy(), want, rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(want).max())))
  got = dhb.cpu().float()
  sg = s2.grad if s2.grad is not None else torch.zeros_like(s2)
  if has_pn:
    ref_gp = sg[:, 1:4]
    np.testing.assert_allclose(got[:, 129:132].numpy(), ref_gp.numpy(), rtol=2e-2, atol=2e-3 * max(ref_gp.abs().max().item(), 1e-6))
  else:
    assert (got[:, 129:132] == 7.0).all()                # columns of a head that is off are not touched
  if F & ops.REF_ROUGHNESS:
    ref_r = sg[:, 10]
    np.testing.assert_allclose(got[:, 138].numpy(), ref_r.numpy(), rtol=2e-2, atol=2e-3 * max(ref_r.abs().max().item(), 1e-6))
    if not F & ops.REF_IDE:
      assert (got[:, 138] == 0).all()                    # roughness as an output only (models.py:521-523, 550)
  else:
    assert (got[:, 138] == 7.0).all()
  assert (got[:, 132:138] == 7.0).all()


def test_ref_head_refuses_what_the_reference_cannot_run(ops):
  M, n = 32, 16
  small, v = torch.zeros((M, 11)).cuda(), torch.ones((2, 3)).cuda()
  vi = torch.zeros((M, 128), dtype=torch.bfloat16).cuda()
  tabs = ops.IdeTablesDev(5, 'cuda')
  rg = torch.zeros((3, M)).cuda()
  with pytest.raises(ValueError, match='Normals must be computed for reflection directions'):
    ops.ref_head_fwd(small, None, v, n, None, 0.0, vi, 0, 128, features=ops.REF_REFLECT, deg_view=4)
  with pytest.raises(ValueError, match='IDE needs the predicted roughness and reflection directions'):
    ops.ref_head_fwd(small, rg, v, n, tabs, 0.0, vi, 0, 128, features=ops.REF_DENSITY_NORMALS | ops.REF_REFLECT | ops.REF_IDE)
  with pytest.raises(ValueError, match='IDE needs the predicted roughness and reflection directions'):
    ops.ref_head_fwd(small, rg, v, n, tabs, 0.0, vi, 0, 128, features=ops.REF_DENSITY_NORMALS | ops.REF_ROUGHNESS | ops.REF_IDE)


@pytest.mark.parametrize('use_tint', [True, False])
def test_ref_color_fwd_bwd(ops, use_tint):
  gen = torch.Generator().manual_seed(22)
  M = 999
  raw_rgb = torch.randn((M, 3), generator=gen) * 2
  small = torch.randn((M, 11), generator=gen) * 2
  small[:7, 4:7] = -12.0                   # tiny diffuse: exercises the linear branch of linear_to_srgb
  raw_rgb[:7] = -12.0
  pad = 0.001
  rr, sm = raw_rgb.double().requires_grad_(True), small.double().requires_grad_(True)
  spec = torch.sigmoid(rr)
  # (models.py:592-595: the tint, or 0.5 without a tint head)
  lin = (torch.sigmoid(sm[:, 7:10]) if use_tint else 0.5) * spec + torch.sigmoid(sm[:, 4:7] - math.log(3.0))
  ref = torch.clamp(oimage.linear_to_srgb(lin), 0.0, 1.0) * (1 + 2 * pad) - pad
  out = ops.ref_color_fwd(dev(raw_rgb), dev(small), 1.0, 0.0, pad, use_tint)
  np.testing.assert_allclose(out.cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-6)
  g = torch.randn((M, 3), generator=gen)
  (ref * g.double()).sum().backward()
  dhb = torch.zeros((M, 256), dtype=torch.bfloat16).cuda()
  g_rr = ops.ref_color_bwd(dev(raw_rgb), dev(small), 1.0, 0.0, pad, use_tint, dev(g), dhb, 132, 135)
  np.testing.assert_allclose(g_rr.cpu().numpy(), rr.grad.numpy(), rtol=2e-4, atol=1e-6)
  got = dhb.cpu().float()
  np.testing.assert_allclose(got[:, 132:135].numpy(), sm.grad[:, 4:7].numpy(), rtol=2**-7, atol=1e-4)
  if use_tint:
    np.testing.assert_allclose(got[:, 135:138].numpy(), sm.grad[:, 7:10].numpy(), rtol=2**-7, atol=1e-4)
  else:
    assert (got[:, 135:138] == 0).all()                      # no tint head: its columns are not touched


def test_ref_losses_and_weighted_sum(ops):
  gen = torch.Generator().manual_seed(23)
  B, Bv, n = 50, 47, 24
  w = torch.rand((B, n), generator=gen) * 0.05
  nrm = torch.randn((B, n, 3), generator=gen)
  nrm = nrm / nrm.norm(dim=-1, keepdim=True)
  npred = torch.randn((B, n, 3), generator=gen)
  npred = npred / npred.norm(dim=-1, keepdim=True)
  v = torch.randn((B, 3), generator=gen)
  v = v / v.norm(dim=-1, keepdim=True)

  class Cfg:
    orientation_loss_target = 'normals_pred'
    orientation_coarse_loss_mult, orientation_loss_mult = 0.01, 0.1
    predicted_normal_coarse_loss_mult, predicted_normal_loss_mult = 3e-5, 3e-4

  class Obj:
    pass

  rays, model = Obj(), Obj()
  rays.viewdirs, model.num_levels = v[:Bv], 2
  wv, nv, pv = (w[:Bv].clone().requires_grad_(True), nrm[:Bv].clone().requires_grad_(True),
                npred[:Bv].clone().requires_grad_(True))
  hist = [{'weights': wv, 'normals': nv, 'normals_pred': pv}]      # one level = the fine level when num_levels == 1
  model.num_levels = 1
  lo = otrain.orientation_loss(rays, model, hist, Cfg)
  lp = otrain.predicted_normal_loss(model, hist, Cfg)
  (lo + lp).backward()
  stats = torch.zeros(2).cuda()
  g_w = torch.zeros((B, n)).cuda()
  g_n, g_np = ops.ref_losses(0.1, 3e-4, True, dev(w), dev(nrm.reshape(-1, 3)), dev(npred.reshape(-1, 3)), dev(v), stats,
                             g_w, True, B_valid=Bv)
  np.testing.assert_allclose(stats.cpu().numpy(), [lo.item(), lp.item()], rtol=1e-4)
  np.testing.assert_allclose(g_w.cpu()[:Bv].numpy(), wv.grad.numpy(), rtol=1e-4, atol=1e-9)
  np.testing.assert_allclose(g_n.cpu().reshape(B, n, 3)[:Bv].numpy(), nv.grad.numpy(), rtol=1e-4, atol=1e-10)
  np.testing.assert_allclose(g_np.cpu().reshape(B, n, 3)[:Bv].numpy(), pv.grad.numpy(), rtol=1e-4, atol=1e-10)
  assert (g_w.cpu()[Bv:] == 0).all()
  ws = ops.weighted_sum(dev(w), dev(nrm))
  np.testing.assert_allclose(ws.cpu().numpy(), (w[..., None] * nrm).sum(1).numpy(), rtol=1e-5, atol=1e-7)


def test_normals_on_their_own(ops):
  """The two partial Ref-NeRF mixes of round 4 at kernel level: predicted normals alone (mnr_pred_normals_fwd / _bwd from three
  columns of the head's fp32 side output into the head's bf16 gradient matrix), density-gradient normals alone
  (mnr_density_normals_fwd / _bwd on the tangent network's component-major raw_grad), and the orientation loss with only one of the
  two fields (mnr_ref_losses with normals = NULL / normals_pred = NULL) against the oracle and its autograd
  (models.py:492,498, ref_utils.py:40-42, train_utils.py:162-178)."""
  gen = torch.Generator().manual_seed(31)
  M = 777
  small = torch.randn((M, 4), generator=gen)
  small[5, 1:4] = 0.0                                     # a zero vector: l2_normalize clamps the norm (ref_utils.py:41)
  x = small[:, 1:4].clone().requires_grad_(True)
  npred_o = -oref.l2_normalize(x)
  g = torch.randn((M, 3), generator=gen)
  (npred_o * g).sum().backward()
  npred = ops.pred_normals_fwd(dev(small), 1)
  np.testing.assert_allclose(npred.cpu().numpy(), npred_o.detach().numpy(), rtol=1e-5, atol=1e-6)
  dhb = torch.zeros((M, 16), dtype=torch.bfloat16).cuda()
  ops.pred_normals_bwd(dev(small), 1, dev(g), dhb, 9)
  got = dhb.float().cpu()
  assert (got[:, :9] == 0).all() and (got[:, 12:] == 0).all()
  np.testing.assert_allclose(got[:, 9:12].numpy(), x.grad.bfloat16().float().numpy(), rtol=1e-2, atol=1e-6)   # (bf16 columns)
  raw = torch.randn((3, M), generator=gen)
  raw[:, 7] = 0.0
  r = raw.clone().requires_grad_(True)
  nrm_o = -oref.l2_normalize(r.t())
  (nrm_o * g).sum().backward()
  nrm = ops.density_normals_fwd(dev(raw))
  np.testing.assert_allclose(nrm.cpu().numpy(), nrm_o.detach().numpy(), rtol=1e-5, atol=1e-6)
  g_raw = ops.density_normals_bwd(dev(raw), dev(g))
  np.testing.assert_allclose(g_raw.cpu().numpy(), r.grad.numpy(), rtol=1e-4, atol=1e-6)

  # the orientation loss with one field only
  B, Bv, n = 40, 37, 16
  w = torch.rand((B, n), generator=gen) * 0.05
  nv = torch.randn((B, n, 3), generator=gen)
  nv = nv / nv.norm(dim=-1, keepdim=True)
  v = torch.randn((B, 3), generator=gen)
  v = v / v.norm(dim=-1, keepdim=True)

  class Obj:
    pass

  for target, field in (('normals_pred', 'normals_pred'), ('normals', 'normals')):
    class Cfg:
      orientation_loss_target = target
      orientation_coarse_loss_mult, orientation_loss_mult = 0.01, 0.1
    rays, model = Obj(), Obj()
    rays.viewdirs, model.num_levels = v[:Bv], 1
    wv, fv = w[:Bv].clone().requires_grad_(True), nv[:Bv].clone().requires_grad_(True)
    hist = [{'weights': wv, 'normals': None, 'normals_pred': None}]
    hist[0][field] = fv
    lo = otrain.orientation_loss(rays, model, hist, Cfg)
    lo.backward()
    stats = torch.zeros(2).cuda()
    g_w = torch.zeros((B, n)).cuda()
    is_pred = target == 'normals_pred'
    g_n, g_np = ops.ref_losses(0.1, 0.0, is_pred, dev(w), None if is_pred else dev(nv.reshape(-1, 3)),
                               dev(nv.reshape(-1, 3)) if is_pred else None, dev(v), stats, g_w, True, B_valid=Bv)
    assert (g_n is None) == is_pred and (g_np is None) == (not is_pred)
    np.testing.assert_allclose(stats.cpu().numpy(), [lo.item(), 0.0], rtol=1e-4, atol=1e-12)
    np.testing.assert_allclose(g_w.cpu()[:Bv].numpy(), wv.grad.numpy(), rtol=1e-4, atol=1e-9)
    got_g = (g_np if is_pred else g_n).cpu().reshape(B, n, 3)[:Bv]
    np.testing.assert_allclose(got_g.numpy(), fv.grad.numpy(), rtol=1e-4, atol=1e-10)
  # what the reference refuses, the C ABI refuses: the predicted-normal loss needs both fields
  with pytest.raises(ValueError, match='without density-gradient normals'):
    ops.ref_losses(0.1, 3e-4, True, dev(w), None, dev(nv.reshape(-1, 3)), dev(v), torch.zeros(2).cuda(), None, False, B_valid=Bv)


def test_tangent_features(ops):
  """d(features)/d(mean_c) (forward mode) vs autograd of the oracle's IPE w.r.t. the means."""
  from multinerf_amd import geopoly
  gen = torch.Generator().manual_seed(24)
  B, n, maxdeg = 20, 16, 16
  o = torch.rand((B, 3), generator=gen) * 2 - 1
  d = torch.randn((B, 3), generator=gen)
  d = d / d.norm(dim=-1, keepdim=True)
  radii = torch.full((B, 1), 5e-4)
  tdist = 2.0 + 4.0 * torch.sort(torch.rand((B, n + 1), generator=gen), -1).values
  basis = torch.as_tensor(geopoly.generate_basis('octahedron', 1), dtype=torch.float32)
  means, covs = orender.cast_rays(tdist.double(), o.double(), d.double(), radii.double(), 'cone', diag=False)
  bT = basis.T.contiguous().double()

  def feats(mu):
    lm, lv = ocoord.lift_and_diagonalize(mu, covs, bT)
    return ocoord.integrated_pos_enc(lm, lv, 0, maxdeg)

  F = 2 * 3 * maxdeg
  tang = ops.cast_rays_ipe_tangent(dev(tdist), dev(o), dev(d), dev(radii.reshape(-1)), dev(basis), ray_shape='cone',
                                   min_deg=0, max_deg=maxdeg, ld_feat=128).cpu().float()
  M = B * n
  assert tang.shape == (3 * M, 128) and (tang[:, F:] == 0).all()
  for c in range(3):
    e = torch.zeros_like(means)
    e[..., c] = 1
    _, jvp = torch.func.jvp(feats, (means,), (e,))
    ref = jvp.reshape(M, F)
    got = tang[c * M:(c + 1) * M, :F].double()
    scale = ref.abs().max(0, keepdim=True).values.clamp_min(1e-3)
    # bf16 storage + the 4-degree sincos recurrence; columns compared relative to their own scale.
    assert ((got - ref).abs() / scale).max().item() < 2e-2


@pytest.mark.parametrize('far', [False, True])
def test_tangent_features_under_the_contraction(ops, far):
  """The same rows with warp_fn = contract (models.py:445-446 applies it INSIDE predict_density, so value_and_grad, :478-481,
  differentiates through it): d(features)/d(mean_c) then carries the contraction's Jacobian and, through J cov J^T with the
  covariance an input, its derivative -- against autograd (jvp) of the oracle's track_linearize(contract) + lift + IPE."""
  from multinerf_amd import geopoly
  gen = torch.Generator().manual_seed(25)
  B, n, maxdeg = 20, 16, 12
  o = torch.rand((B, 3), generator=gen) * 2 - 1
  d = torch.randn((B, 3), generator=gen)
  d = d / d.norm(dim=-1, keepdim=True) * (1.0 + 0.2 * torch.rand((B, 1), generator=gen))
  radii = torch.full((B, 1), 5e-4)
  c = torch.sort(torch.rand((B, n + 1), generator=gen), -1).values
  # near: t in [0.2, 3] (means on both sides of the unit sphere); far: reciprocal spacing out to t = 500
  tdist = (1.0 / (c / 500.0 + (1 - c) / 0.2)).flip(-1) if far else 0.2 + 2.8 * c
  tdist = torch.sort(tdist, -1).values.contiguous()
  basis = torch.as_tensor(geopoly.generate_basis('icosahedron', 2), dtype=torch.float32)
  means, covs = orender.cast_rays(tdist.double(), o.double(), d.double(), radii.double(), 'cone', diag=False)
  bT = basis.T.contiguous().double()

  def feats(mu):
    mu2, cv2 = ocoord.track_linearize(ocoord.contract, mu, covs)          # the covariance is an input, held fixed
    lm, lv = ocoord.lift_and_diagonalize(mu2, cv2, bT)
    return ocoord.integrated_pos_enc(lm, lv, 0, maxdeg)

  K = basis.shape[0]
  F = 2 * K * maxdeg
  ld = (F + 127) // 128 * 128
  tang = ops.cast_rays_ipe_tangent(dev(tdist), dev(o), dev(d), dev(radii.reshape(-1)), dev(basis), ray_shape='cone',
                                   min_deg=0, max_deg=maxdeg, ld_feat=ld, warp_contract=True).cpu().float()
  M = B * n
  assert tang.shape == (3 * M, ld) and (tang[:, F:] == 0).all()
  inside = (means.norm(dim=-1) <= 1).reshape(M)
  assert far or (inside.any() and (~inside).any())
  for cdir in range(3):
    e = torch.zeros_like(means)
    e[..., cdir] = 1
    _, jvp = torch.func.jvp(feats, (means,), (e,))
    ref = jvp.reshape(M, F)
    got = tang[cdir * M:(cdir + 1) * M, :F].double()
    scale = ref.abs().max(0, keepdim=True).values.clamp_min(1e-3)
    err = ((got - ref).abs() / scale).max().item()
    print(f'tangent rows under the contraction, far={far}, d/d mean_{cdir}: max column-relative error {err:.2e}')
    assert err < 2e-2
