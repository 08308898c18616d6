"""Parity of every HIP kernel (through the C ABI) against the CPU oracle.  -m gpu.

Tolerances are stated per test.  Integer artefacts (sample indices, searchsorted
windows) are compared exactly.
"""

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import coord as ocoord
from oracle import math as omath
from oracle import models as omodels
from oracle import render as orender
from oracle import stepfun as ostepfun
from oracle import train_utils as otrain


@pytest.fixture(scope='module')
def ops():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from multinerf_amd import ops as _ops
  _ops.lib()
  return _ops


@pytest.fixture(params=[1, 0], ids=['quad', 'lane'])
def level_bwd_kernel(request, ops):
  """Both level-backward kernels: four lanes per ray (default where 16 rays fit LDS) and lane per ray (long rays, A/B)."""
  ops.L.check(ops.L.debug().mnr_level_bwd_set_quad(request.param))
  yield request.param
  ops.L.check(ops.L.debug().mnr_level_bwd_set_quad(1))


def dev(x):
  return x.contiguous().cuda()


def rand_stepfun(gen, B, n, lo=0.0, hi=1.0):
  t = torch.cumsum(torch.rand((B, n + 1), generator=gen) + 1e-3, -1)
  t = (t - t[:, :1]) / (t[:, -1:] - t[:, :1]) * (hi - lo) + lo
  w = torch.softmax(torch.randn((B, n), generator=gen) * 2, -1)
  return t.float(), w.float()


# ----------------------------------------------------------------------------- sampling


def test_sorted_interp_bit_exact_indices(ops):
  gen = torch.Generator().manual_seed(1)
  B, nc, nu = 257, 191, 64
  t, w = rand_stepfun(gen, B, nc - 1)
  cw = ostepfun.integrate_weights(w)
  u = torch.sort(torch.rand((B, nu), generator=gen) * (1 - 1e-6), -1).values
  ref, ref_idx = omath.sorted_interp(u, cw, t, return_index=True)
  out, idx = ops.sorted_interp(dev(u), dev(cw), dev(t))
  assert torch.equal(idx.cpu(), ref_idx), 'sample indices must be bit-exact given identical (u, cw)'
  # same (u, cw, t) and contraction off => identical fp32 arithmetic
  np.testing.assert_array_equal(out.cpu().numpy(), ref.numpy())


def test_sorted_interp_out_of_range_and_ties(ops):
  xp = torch.tensor([[0.0, 0.2, 0.2, 0.2, 0.7, 1.0]])
  fp = torch.tensor([[1.0, 2.0, 3.0, 4.0, 5.0, 6.0]])
  x = torch.tensor([[-0.5, 0.0, 0.1, 0.2, 0.5, 1.0, 1.5]])
  ref, ref_idx = omath.sorted_interp(x, xp, fp, return_index=True)
  out, idx = ops.sorted_interp(dev(x), dev(xp), dev(fp))
  assert torch.equal(idx.cpu(), ref_idx)
  np.testing.assert_array_equal(out.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize('n,dilation,domain', [(64, 0.0103125, (0., 1.)), (64, 0.00262207, (0., 1.)),
                                               (8, 0.53, (-math.inf, math.inf)), (128, 0.0064, (0., 1.))])
def test_max_dilate_weights(ops, n, dilation, domain):
  gen = torch.Generator().manual_seed(2)
  B = 131
  t, w = rand_stepfun(gen, B, n)
  w[3, 5:9] = 0.0
  td_ref, wd_ref = ostepfun.max_dilate_weights(t, w, dilation, domain=domain, renormalize=True)
  td, wd = ops.max_dilate_weights(dev(t), dev(w), dilation, domain)
  np.testing.assert_array_equal(td.cpu().numpy(), td_ref.numpy())   # merge == sort, same fp32 ops
  # weights: renormalisation sums in a different order (sequential vs pairwise): 1e-6 relative.
  np.testing.assert_allclose(wd.cpu().numpy(), wd_ref.numpy(), rtol=2e-6, atol=1e-9)


def _resample_ref(sd, w, u_jit, near, far, *, n, use_dil, dil, anneal, pad, single, raydist):
  if use_dil:
    sd, w = ostepfun.max_dilate_weights(sd, w, dil, domain=(0., 1.), renormalize=True)
    sd, w = sd[..., 1:-1], w[..., 1:-1]
  logits = ostepfun.resample_logits(sd, w, anneal, pad)
  s, idx = ostepfun.sample_intervals(u_jit, sd, logits, n, single_jitter=single, domain=(0., 1.),
                                     return_index=True)
  _, s_to_t = ocoord.construct_ray_warps(raydist, near, far)
  return s, s_to_t(s), idx


def _u_base(n, jittered):
  eps = float(np.finfo(np.float32).eps)
  if jittered:
    u_max = eps + (1 - eps) / n
    return torch.linspace(0, 1 - u_max, n), (1 - u_max) / (n - 1) - eps
  pad = 1 / (2 * n)
  return torch.linspace(pad, 1. - pad - eps, n), 0.0


@pytest.mark.parametrize('case,B', [('level0_det', 200), ('level1_360_jit1', 200), ('level2_360_jit1', 200), ('nodil_jitn', 200), ('b256_dil', 200),
                                    # every level of configs/360.gin at the benchmark's batch (round-3 verdict: 200 rays were the whole
                                    # evidence for "bit-exact indices"): 1.0 M / 1.0 M / 0.5 M indices, half of the rays with the peaked
                                    # weights a trained proposal network produces
                                    ('level0_jit1', 16384), ('level1_360_jit1', 16384), ('level2_360_jit1', 16384)])
def test_resample_level(ops, case, B):
  gen = torch.Generator().manual_seed(3 + B)
  cfgs = {
      'level0_det': dict(np_=1, n=64, use_dil=False, dil=0.5025, anneal=0.9090909, pad=0.0, single=True, jit=False, raydist='reciprocal', near=0.2, far=1e6),
      'level0_jit1': dict(np_=1, n=64, use_dil=False, dil=0.5025, anneal=0.9090909, pad=0.0, single=True, jit=True, raydist='reciprocal', near=0.2, far=1e6),
      'level1_360_jit1': dict(np_=64, n=64, use_dil=True, dil=0.0103125, anneal=0.9090909, pad=0.0, single=True, jit=True, raydist='reciprocal', near=0.2, far=1e6),
      'level2_360_jit1': dict(np_=64, n=32, use_dil=True, dil=0.00262207, anneal=1.0, pad=0.0, single=True, jit=True, raydist='reciprocal', near=0.2, far=1e6),
      'nodil_jitn': dict(np_=128, n=128, use_dil=False, dil=0.0, anneal=1.0, pad=0.01, single=False, jit=True, raydist=None, near=2.0, far=6.0),
      'b256_dil': dict(np_=128, n=32, use_dil=True, dil=0.0064, anneal=0.5, pad=0.0, single=True, jit=True, raydist=None, near=2.0, far=6.0),
  }
  c = cfgs[case]
  if c['np_'] == 1:
    sd = torch.tensor([[0., 1.]]).repeat(B, 1)
    w = torch.ones((B, 1))
  else:
    sd, w = rand_stepfun(gen, B, c['np_'])
    if B > 1000:
      # half of the rays: one or two narrow peaks over a floor of tiny weights (a surface hit), incl. exact zeros
      peaked = torch.softmax(torch.randn((B, c['np_']), generator=gen) * 12, -1)
      peaked[peaked < 1e-7] = 0.0
      w = torch.where((torch.arange(B) % 2 == 0)[:, None], w, peaked)
    w = w * 0.98   # alpha-compositing weights sum to <= 1
  near = torch.full((B, 1), c['near'])
  far = torch.full((B, 1), c['far'])
  u_base, max_jitter = _u_base(c['n'], c['jit'])
  u_jit = None
  if c['jit']:
    u_jit = torch.rand((B, 1 if c['single'] else c['n']), generator=gen)
  s_ref, t_ref, idx_ref = _resample_ref(sd, w, u_jit, near, far, n=c['n'], use_dil=c['use_dil'], dil=c['dil'],
                                        anneal=c['anneal'], pad=c['pad'], single=c['single'], raydist=c['raydist'])
  s, t, idx = ops.resample_level(dev(sd), dev(w), dev(u_base), None if u_jit is None else dev(u_jit),
                                 dev(near), dev(far), n_samples=c['n'], use_dilation=c['use_dil'],
                                 dilation=c['dil'], domain=(0., 1.), anneal=c['anneal'], resample_padding=c['pad'],
                                 single_jitter=c['single'], max_jitter=max_jitter, raydist_fn=c['raydist'],
                                 want_idx=True)
  mismatch = (idx.cpu() != idx_ref).float().mean().item()
  # Sample indices are bit-exact (north_star) BY CONSTRUCTION: every operation between the inputs and the index is an
  # individually rounded IEEE fp32 +, -, *, / or comparison in a documented order (blocked sums; the path's own exp / log,
  # csrc/resample.hip rs_exp / rs_log = oracle/math.py kexp / klog), restated operation for operation in the oracle.
  print(f'{case} B={B}: index mismatch rate {mismatch:.2e} ({int((idx.cpu() != idx_ref).sum())} of {idx_ref.numel()})')
  assert mismatch == 0
  # ... and, REPORTED rather than assumed zero (SURVEY hard part 3 iii): the same kernel output against the oracle evaluated in
  # the REFERENCE's association order (left-to-right sums / cumsum, the host library's exp / log: stepfun.py:146,156)
  with ostepfun.reference_order():
    s_ro, _, idx_ro = _resample_ref(sd, w, u_jit, near, far, n=c['n'], use_dil=c['use_dil'], dil=c['dil'],
                                    anneal=c['anneal'], pad=c['pad'], single=c['single'], raydist=c['raydist'])
  mism_ro = int((idx.cpu() != idx_ro).sum())
  print(f'{case} B={B}: vs the oracle in REFERENCE order: index mismatches {mism_ro} of {idx_ro.numel()} '
        f'({mism_ro / idx_ro.numel():.2e}); max |s - s_ref_order| {(s.cpu() - s_ro).abs().max().item():.2e}')
  assert mism_ro / idx_ro.numel() < (5e-3 if B < 1000 else 1e-4)
  # (positions where the index agrees; a sample that lands in the neighbouring bin sits a bin width away: up to 1.4e-3 with the
  # peaked weights of the full-batch cases)
  ok_ro = torch.nn.functional.pad(idx.cpu() == idx_ro, (1, 1), value=True)
  edge_ok = ok_ro[..., :-1] & ok_ro[..., 1:]                      # interval edges whose two neighbouring samples agree
  d_ok = (s.cpu() - s_ro)[edge_ok].abs().max().item()
  print(f'{case} B={B}: max |s - s_ref_order| where the indices agree {d_ok:.2e}')
  # measured <= 6.6e-5 at 200 rays (level2_360_jit1); the peaked weights of the full-batch cases have bins whose CDF step is a few
  # ulps, inside which (u - cw0) / (cw1 - cw0) turns a 1-ulp softmax difference into a fraction of the (narrow) bin
  assert d_ok < (2e-4 if B < 1000 else 2e-3)
  assert (s.cpu() - s_ro).abs().max().item() < 5e-3
  same = (idx.cpu() == idx_ref).all(-1)
  # (u - cw0)/(cw1 - cw0) amplifies the <=1-ulp softmax differences inside narrow bins: 5e-5 in s.
  print(f'{case} B={B}: max |s - s_kernel_order| {(s.cpu() - s_ref).abs().max().item():.2e}')
  np.testing.assert_allclose(s.cpu().numpy(), s_ref.numpy(), atol=5e-5, rtol=0)
  rel = ((t.cpu() - t_ref).abs() / t_ref.abs().clamp_min(1e-6))
  # reciprocal warp amplifies 1-ulp s differences near s=1 (t ~ 1e5..1e6): relative tolerance there.
  assert rel[same].max().item() < 2e-2 if c['raydist'] == 'reciprocal' else rel.max().item() < 1e-5
  assert torch.isfinite(t).all()


def test_resample_rejects_single_sample(ops):
  B = 4
  z = torch.zeros((B, 2)).cuda()
  with pytest.raises(ValueError, match='num_samples must be > 1'):
    ops.resample_level(z, torch.ones((B, 1)).cuda(), torch.zeros(1).cuda(), None, torch.ones(B).cuda(),
                       torch.ones(B).cuda(), n_samples=1, use_dilation=False, dilation=0., domain=(0., 1.),
                       anneal=1., resample_padding=0., single_jitter=True, max_jitter=0., raydist_fn=None)


# ----------------------------------------------------------------------------- features


def _rays(gen, B):
  o = torch.rand((B, 3), generator=gen) * 2 - 1
  tgt = torch.randn((B, 3), generator=gen) * 0.3
  d = tgt - o
  d = d / d.norm(dim=-1, keepdim=True) * (1.0 + 0.2 * torch.rand((B, 1), generator=gen))
  radii = 3e-4 + 7e-4 * torch.rand((B, 1), generator=gen)
  return o.float(), d.float(), radii.float()


@pytest.mark.parametrize('shape,contract,basis_name,maxdeg', [('cone', True, ('icosahedron', 2), 12),
                                                              ('cone', False, ('octahedron', 1), 16),
                                                              ('cylinder', False, ('octahedron', 1), 16),
                                                              # an even number of directions (6) and a wide one (46: three passes of the
                                                              # bf16 kernel's (sample, direction pair) loop)
                                                              ('cone', True, ('icosahedron', 1), 10),
                                                              ('cone', False, ('icosahedron', 3), 4)])
def test_cast_rays_ipe(ops, shape, contract, basis_name, maxdeg):
  from multinerf_amd import geopoly
  gen = torch.Generator().manual_seed(4)
  B, n = 96, 32
  o, d, radii = _rays(gen, B)
  if contract:
    s = torch.sort(torch.rand((B, n + 1), generator=gen), -1).values
    s[:, 0], s[:, -1] = 0.0, 1.0 - 2**-23
    tdist = 1.0 / (s / 1e6 + (1 - s) / 0.2)
  else:
    tdist = 2.0 + 4.0 * torch.sort(torch.rand((B, n + 1), generator=gen), -1).values
  basis = torch.as_tensor(geopoly.generate_basis(*basis_name), dtype=torch.float32)
  means, covs = orender.cast_rays(tdist, o, d, radii, shape, diag=False)
  if contract:
    means, covs = ocoord.track_linearize(ocoord.contract, means, covs)
  lm, lv = ocoord.lift_and_diagonalize(means, covs, basis.T.contiguous())
  ref = ocoord.integrated_pos_enc(lm, lv, 0, maxdeg).reshape(B * n, -1)
  ref64 = ocoord.integrated_pos_enc(lm.double(), lv.double(), 0, maxdeg).reshape(B * n, -1)

  nfeat = 2 * basis.shape[0] * maxdeg
  ld = (nfeat + 63) // 64 * 64
  feat, gm, gc = ops.cast_rays_ipe(dev(tdist), dev(o), dev(d), dev(radii.reshape(-1)), dev(basis), ray_shape=shape,
                                   warp_contract=contract, min_deg=0, max_deg=maxdeg, ld_feat=ld,
                                   want_gaussians=True)
  f32 = ops.cast_rays_ipe_f32(dev(tdist), dev(o), dev(d), dev(radii.reshape(-1)), dev(basis), ray_shape=shape,
                              warp_contract=contract, min_deg=0, max_deg=maxdeg)
  # Gaussians: J cov J^T cancels catastrophically for far samples (cov ~ 1e10, J ~ 1e-6), so the fp32
  # oracle itself is far from fp64 there; the kernel is held to the oracle's own distance from fp64.
  m64, c64 = orender.cast_rays(tdist.double(), o.double(), d.double(), radii.double(), shape, diag=False)
  if contract:
    m64, c64 = ocoord.track_linearize(ocoord.contract, m64, c64)
  m64, c64 = m64.reshape(-1, 3), c64.reshape(-1, 9)
  m_ref = means.reshape(-1, 3).double()
  c_ref = covs.reshape(-1, 9).double()
  em_k = (gm.cpu().double() - m64).abs().max().item()
  em_o = (m_ref - m64).abs().max().item()
  assert em_k <= max(4 * em_o, 2e-6), (em_k, em_o)
  sc = c64.abs().max(-1, keepdim=True).values.clamp_min(1e-30)
  ec_k = ((gc.cpu().double() - c64).abs() / sc).max().item()
  ec_o = ((c_ref - c64).abs() / sc).max().item()
  print(f'cov rel err: kernel {ec_k:.2e} fp32-oracle {ec_o:.2e}')
  assert ec_k <= max(4 * ec_o, 1e-5), (ec_k, ec_o)
  # Features: the fp32 oracle itself is only accurate to |mean| 2^deg 2^-24 in the sine argument
  # (reference tests/coord_test.py:112-127 uses per-degree tolerances for the same reason), so the
  # kernel is held to the oracle's own distance from fp64, per degree.
  K = basis.shape[0]
  got = f32.cpu().double()
  bT = basis.T.contiguous().double()
  lmk, lvk = ocoord.lift_and_diagonalize(gm.cpu().double().reshape(B, n, 3), gc.cpu().double().reshape(B, n, 3, 3), bT)
  ref64_k = ocoord.integrated_pos_enc(lmk, lvk, 0, maxdeg).reshape(B * n, -1)      # fp64 from the kernel's Gaussians
  lmo, lvo = ocoord.lift_and_diagonalize(means.double(), covs.double(), bT)
  ref64_o = ocoord.integrated_pos_enc(lmo, lvo, 0, maxdeg).reshape(B * n, -1)      # fp64 from the oracle's Gaussians
  for l in range(maxdeg):
    cols = [h * K * maxdeg + l * K + k for h in (0, 1) for k in range(K)]
    e_kernel = (got[:, cols] - ref64_k[:, cols]).abs().max().item()
    e_oracle = (ref.double()[:, cols] - ref64_o[:, cols]).abs().max().item()
    assert e_kernel <= max(4 * e_oracle, 2e-6 * 2**l + 1e-6), (l, e_kernel, e_oracle)
  # end to end vs fp64: bounded by the sensitivity to the fp32 warp noise measured above.
  e2e = (got - ref64).abs().max().item()
  print(f'features end-to-end max |kernel - fp64| = {e2e:.2e}; fp32 oracle: {(ref.double() - ref64).abs().max().item():.2e}')
  assert e2e < 2e-2
  # bf16 rows = rounding of the fp32 features; padding columns are zero.
  fb = feat.cpu().float()
  np.testing.assert_allclose(fb[:, :nfeat].numpy(), f32.cpu().to(torch.bfloat16).float().numpy(), atol=0, rtol=0)
  assert (fb[:, nfeat:] == 0).all()


def test_cast_rays_rejects_bad_shape(ops):
  z = torch.zeros((2, 3)).cuda()
  with pytest.raises(ValueError, match='ray_shape'):
    ops.cast_rays_ipe(torch.zeros((2, 3)).cuda(), z, z, torch.zeros(2).cuda(), torch.eye(3).cuda(),
                      ray_shape='sphere', warp_contract=False, min_deg=0, max_deg=4, ld_feat=64)


def test_exposure_scale_bwd(ops):
  """models.py:257-267's VJP w.r.t. exposure_scaling_offsets: g_offsets[idx[b], c] += ev[b] * g[b, c] for idx[b] > 0; the
  kernel sums the lanes of a wave that share an index before its atomics (few distinct indices per batch)."""
  gen = torch.Generator().manual_seed(9)
  B = 1000 + 37
  idx = torch.randint(-1, 6, (B,), generator=gen).to(torch.int32)
  idx[100:164] = 3                                    # a whole wave on one index
  idx[200:264] = torch.arange(64).to(torch.int32) + 7  # and one with 64 different ones
  ev = 0.5 + torch.rand((B,), generator=gen)
  g = torch.randn((B, 3), generator=gen)
  out = torch.full((100, 3), 0.25).cuda()
  ops.exposure_scale_bwd(dev(ev), dev(idx), dev(g), out.view(-1), B)
  want = torch.full((100, 3), 0.25, dtype=torch.float64)
  for b in range(B):
    if idx[b] > 0:
      want[idx[b]] += (ev[b] * g[b]).double()
  np.testing.assert_allclose(out.cpu().double().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('n', [8, 13, 32])       # (the kernel stores one evaluation into groups of 8 samples: whole, ragged, several)
def test_viewdir_enc_fill(ops, n):
  gen = torch.Generator().manual_seed(5)
  B = 37
  v = torch.randn((B, 3), generator=gen)
  v = (v / v.norm(dim=-1, keepdim=True)).float()
  ref = ocoord.pos_enc(v, 0, 4, append_identity=True)
  dst = torch.full((B * n, 320), 7.0, dtype=torch.bfloat16).cuda()
  ops.viewdir_enc_fill(dev(v), n, 4, dst, 256, 320)
  out = dst.cpu().float().reshape(B, n, 320)
  assert (out[..., :256] == 7.0).all()
  np.testing.assert_allclose(out[..., 256:283].numpy(), ref[:, None, :].expand(B, n, 27).to(torch.bfloat16).float().numpy(),
                             atol=8e-3)   # device sinf vs host sinf may differ by 1 ulp before bf16 rounding
  assert (out[..., 283:] == 0).all()


# ----------------------------------------------------------------------------- dense


def _bf(x):
  return x.to(torch.bfloat16)


@pytest.mark.parametrize('M,N,K1,K2', [(256, 128, 64, 0), (1024, 256, 512, 0), (512, 1024, 1024, 512),
                                       (384, 128, 320, 0)])
def test_gemm_nt(ops, M, N, K1, K2):
  gen = torch.Generator().manual_seed(6)
  A1 = _bf(torch.randn((M, K1), generator=gen))
  A2 = _bf(torch.randn((M, K2), generator=gen)) if K2 else None
  Bt = _bf(torch.randn((N, K1 + K2), generator=gen) / math.sqrt(K1 + K2))   # asymmetric, transpose-detecting
  bias = torch.randn((N,), generator=gen)
  A = A1.float() if A2 is None else torch.cat([A1.float(), A2.float()], -1)
  ref = A.double() @ Bt.double().T + bias.double()
  Cb = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
  ops.gemm_nt(dev(A1), dev(Bt), M=M, N=N, K1=K1, A2=None if A2 is None else dev(A2), K2=K2, bias=dev(bias),
              n_bias=N, relu=True, Cb=Cb, ldcb=N, nb=N)
  got = Cb.cpu().double()
  want = torch.relu(ref)
  # fp32 accumulation of exact bf16 products, one bf16 rounding at the end: 2^-8 relative + tiny abs.
  np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2**-7, atol=1e-2)
  # fp32 output window + partial bf16 window + bias bound + mask
  Cf = torch.zeros((M, 3), dtype=torch.float32).cuda()
  Cb2 = torch.full((M, N), 9.0, dtype=torch.bfloat16).cuda()
  mask = _bf((torch.rand((M, N), generator=gen) > 0.5).float())
  ops.gemm_nt(dev(A1), dev(Bt), M=M, N=N, K1=K1, A2=None if A2 is None else dev(A2), K2=K2, bias=dev(bias),
              n_bias=70, relu=False, mask=dev(mask), ldmask=N, Cb=Cb2, ldcb=N, nb=66, Cf=Cf, ldcf=3, f0=65, nf=3)
  b2 = bias.double().clone()
  b2[70:] = 0
  want2 = (A.double() @ Bt.double().T + b2) * mask.double()
  np.testing.assert_allclose(Cf.cpu().double().numpy(), want2[:, 65:68].numpy(), rtol=1e-4, atol=2e-4)
  got2 = Cb2.cpu().double()
  np.testing.assert_allclose(got2[:, :66].numpy(), want2[:, :66].numpy(), rtol=2**-7, atol=1e-2)
  assert (got2[:, 66:] == 9.0).all()


def test_gemm_nt_bit_masks(ops):
  """Forward epilogue writes (out > 0) bits; the dX epilogue consuming them == the bf16-mask path."""
  gen = torch.Generator().manual_seed(66)
  M, N, K = 512, 256, 128
  A = _bf(torch.randn((M, K), generator=gen))
  Bt = _bf(torch.randn((N, K), generator=gen) / math.sqrt(K))
  act = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
  bits = torch.zeros((M, N // 8), dtype=torch.uint8).cuda()
  ops.gemm_nt(dev(A), dev(Bt), M=M, N=N, K1=K, relu=True, Cb=act, ldcb=N, nb=N, bits_out=bits)
  want = (act.cpu().float() > 0).reshape(M, N // 8, 8)
  got = ((bits.cpu().int()[..., None] >> torch.arange(8)) & 1).bool()
  assert torch.equal(got, want)
  G = _bf(torch.randn((M, N), generator=gen))
  W2 = _bf(torch.randn((N, N), generator=gen) / math.sqrt(N))
  d1 = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
  d2 = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
  ops.gemm_nt(dev(G), dev(W2), M=M, N=N, K1=N, mask=act, ldmask=N, Cb=d1, ldcb=N, nb=N)
  ops.gemm_nt(dev(G), dev(W2), M=M, N=N, K1=N, bits_in=bits, Cb=d2, ldcb=N, nb=N)
  assert torch.equal(d1.cpu(), d2.cpu())
  # 128-wide (small-tile) configuration
  act3 = torch.zeros((M, 128), dtype=torch.bfloat16).cuda()
  bits3 = torch.zeros((M, 16), dtype=torch.uint8).cuda()
  ops.gemm_nt(dev(A), dev(Bt[:128].contiguous()), M=M, N=128, K1=K, relu=True, Cb=act3, ldcb=128, nb=128, bits_out=bits3)
  got3 = ((bits3.cpu().int()[..., None] >> torch.arange(8)) & 1).bool()
  assert torch.equal(got3, (act3.cpu().float() > 0).reshape(M, 16, 8))


@pytest.mark.parametrize('K1,K2', [(64, 0), (128, 0), (1024, 0), (512, 256)])
def test_gemm_nt_pipelined_loop_is_bitwise_the_two_stage_loop(ops, K1, K2):
  """The hand-pipelined K loop of the 256x256 tiles (default) against the two-stage loop it replaced
  (mnr_gemm_nt_set_pipelined(0)): forward layer (bias, ReLU, bit masks, [A1|A2]) and the dX layer reading those masks, in
  a persistent launch capped at 8 workgroups (each walks two of the 16 virtual tiles), with fewer K-tiles than pipeline
  stages (K = 64), every tail flavour (K = 128) and long loops.  (Trunk-sized shapes: tools/nt_pipe_probe.py,
  profiles/r2_nt_pipe_probe.txt.)"""
  gen = torch.Generator().manual_seed(67)
  M, N = 1024, 512
  A1 = dev(_bf(torch.relu(torch.randn((M, K1), generator=gen))))
  A2 = dev(_bf(torch.randn((M, K2), generator=gen))) if K2 else None
  Bt = dev(_bf(torch.randn((N, K1 + K2), generator=gen) / math.sqrt(K1 + K2)))
  bias = dev(0.1 * torch.randn((N,), generator=gen))
  G = dev(_bf(torch.randn((M, N), generator=gen)))
  W2 = dev(_bf(torch.randn((N, N), generator=gen) / math.sqrt(N)))
  outs = []
  try:
    ops.L.check(ops.L.debug().mnr_gemm_nt_set_persistent(-8))
    for pipe in (1, 0):
      ops.L.check(ops.L.debug().mnr_gemm_nt_set_pipelined(pipe))
      act = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
      bits = torch.zeros((M, N // 8), dtype=torch.uint8).cuda()
      dx = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
      ops.gemm_nt(A1, Bt, M=M, N=N, K1=K1, A2=A2, K2=K2, bias=bias, n_bias=N, relu=True, Cb=act, ldcb=N, nb=N, bits_out=bits)
      ops.gemm_nt(G, W2, M=M, N=N, K1=N, bits_in=bits, Cb=dx, ldcb=N, nb=N)
      outs.append((act.cpu().view(torch.int16), bits.cpu(), dx.cpu().view(torch.int16)))
  finally:
    ops.L.check(ops.L.debug().mnr_gemm_nt_set_pipelined(1))
    ops.L.check(ops.L.debug().mnr_gemm_nt_set_persistent(1))
  for a, b in zip(*outs):
    assert torch.equal(a, b)
  A = A1.cpu().float() if A2 is None else torch.cat([A1.cpu().float(), A2.cpu().float()], -1)
  want = torch.relu(A.double() @ Bt.cpu().double().T + bias.cpu().double())
  np.testing.assert_allclose(outs[0][0].view(torch.bfloat16).double().numpy(), want.numpy(), rtol=2**-7, atol=1e-2)


@pytest.mark.parametrize('K1,K2,a1_panel', [(192, 0, False), (256, 64, True), (1024, 0, True), (512, 512, True), (320, 0, False)])
def test_gemm_nt_panel_kernel_is_bitwise_the_tiled_kernel(ops, K1, K2, a1_panel):
  """The panel-layout NT kernel (csrc/gemm_blk.hip: results stored as 1-KiB blocks straight from the accumulators, one LDS-DMA
  pipeline across output tiles, the epilogue interleaved with the next tile's first k-step, ReLU masks in tile order) against the
  tiled kernel: forward layer (bias, ReLU, masks, [A1 | A2]) and the dX layer reading those masks, after un-blocking bit for bit;
  launches capped at 8 workgroups (each walks three of the 24 tiles) and uncapped; the shortest K the kernel takes, both A1
  layouts, the head dX shape (K = 320 from a row-major operand); and the values against an fp64 reference."""
  gen = torch.Generator().manual_seed(68)
  M, N = 1536, 1024
  A1 = _bf(torch.relu(torch.randn((M, K1), generator=gen)))
  A2 = dev(_bf(torch.randn((M, K2), generator=gen))) if K2 else None
  Bt = dev(_bf(torch.randn((N, K1 + K2), generator=gen) / math.sqrt(K1 + K2)))
  bias = dev(0.1 * torch.randn((N,), generator=gen))
  G = _bf(torch.randn((M, N), generator=gen))
  W2 = dev(_bf(torch.randn((N, N), generator=gen) / math.sqrt(N)))
  PAN = ops.LAYOUT_PANEL
  act0 = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
  bits0 = torch.zeros((M, N // 8), dtype=torch.uint8).cuda()
  dx0 = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
  ops.gemm_nt(dev(A1), Bt, M=M, N=N, K1=K1, A2=A2, K2=K2, bias=bias, n_bias=N, relu=True, Cb=act0, ldcb=N, nb=N, bits_out=bits0)
  ops.gemm_nt(dev(G), W2, M=M, N=N, K1=N, bits_in=bits0, Cb=dx0, ldcb=N, nb=N)
  A1p = dev(ops.to_panel(A1) if a1_panel else A1)
  try:
    for cap in (8, 0):
      ops.L.check(ops.L.debug().mnr_gemm_nt_panel_set_max_wgs(cap))
      act = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
      bits = torch.zeros((M * N // 8,), dtype=torch.uint8).cuda()
      dx = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
      ops.gemm_nt(A1p, Bt, M=M, N=N, K1=K1, A2=A2, K2=K2, bias=bias, n_bias=N, relu=True, Cb=act, ldcb=N, nb=N, bits_out=bits,
                  a1_layout=PAN if a1_panel else 0, c_layout=PAN)
      ops.gemm_nt(dev(ops.to_panel(G)), W2, M=M, N=N, K1=N, bits_in=bits, Cb=dx, ldcb=N, nb=N, a1_layout=PAN, c_layout=PAN)
      assert torch.equal(ops.from_panel(act).view(torch.int16), act0.view(torch.int16)), cap
      assert torch.equal(ops.bits_from_tile_order(bits, M, N), bits0), cap
      assert torch.equal(ops.from_panel(dx).view(torch.int16), dx0.view(torch.int16)), cap
  finally:
    ops.L.check(ops.L.debug().mnr_gemm_nt_panel_set_max_wgs(0))
  A = A1.float() if A2 is None else torch.cat([A1.float(), A2.cpu().float()], -1)
  want = torch.relu(A.double() @ Bt.cpu().double().T + bias.cpu().double())
  np.testing.assert_allclose(act0.cpu().double().numpy(), want.numpy(), rtol=2**-7, atol=1e-2)
  # the merged head behind a panel trunk: the tiled kernel reading a panel-layout activation into row-major outputs
  if K2 == 0 and K1 % 256 == 0:
    Nh = 512
    Bh = dev(_bf(torch.randn((Nh, K1), generator=gen) / math.sqrt(K1)))
    bh = dev(0.1 * torch.randn((Nh,), generator=gen))
    outs = []
    for lay in (0, PAN):
      vi = torch.zeros((M, 384), dtype=torch.bfloat16).cuda()
      den = torch.zeros((M,), dtype=torch.float32).cuda()
      ops.gemm_nt(dev(ops.to_panel(A1)) if lay else dev(A1), Bh, M=M, N=Nh, K1=K1, bias=bh, n_bias=257, relu=False, Cb=vi, ldcb=384, nb=256,
                  Cf=den, ldcf=1, f0=256, nf=1, a1_layout=lay)
      outs.append((vi.cpu().view(torch.int16), den.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('ap,bp', [(1, 1), (0, 1), (1, 0)])
def test_gemm_tn_panel_operands(ops, ap, bp):
  """Weight gradients from panel-layout operands (gemm_tn_body.inc A_PANEL / B_PANEL): against fp64 and against the row-major
  call (same sums, the order of the atomics aside), the fused bias gradient, the k_valid bound of the feature rows, and with a
  panel A and a row-major B the extra vector column of the merged head (gemm_tn_gcol_kernel)."""
  gen = torch.Generator().manual_seed(19)
  M, K, N = 8192 + 64, 512, 256
  ldb = 384 if not bp else N
  A = _bf(torch.randn((M, K), generator=gen))
  Bfull = _bf(torch.randn((M, ldb), generator=gen))
  Bm = Bfull[:, :N]
  gvec = _bf(0.37 * torch.randn((M,), generator=gen)) if (ap and not bp) else None
  ref = A.double().T @ Bm.double()
  PAN = ops.LAYOUT_PANEL
  Ad = dev(ops.to_panel(A) if ap else A)
  Bd = dev(ops.to_panel(Bm.contiguous()) if bp else Bfull)
  Cout = torch.ones((K, N), dtype=torch.float32).cuda()
  bsum = torch.zeros((N,)).cuda()
  gout = torch.zeros((K,)).cuda() if gvec is not None else None
  ops.gemm_tn(Ad, Bd, Cout, M=M, K=K, N=N, lda=K, ldb=ldb, ldc=N, k_valid=K - 8, bias_out=bsum, bias_n_valid=N,
              gcol=None if gvec is None else dev(gvec), gcol_out=gout, a_layout=PAN if ap else 0, b_layout=PAN if bp else 0)
  got = Cout.cpu().double()
  np.testing.assert_allclose(got[:K - 8].numpy(), (ref + 1)[:K - 8].numpy(), rtol=1e-4, atol=1e-3 * math.sqrt(M))
  assert (got[K - 8:] == 1).all()
  np.testing.assert_allclose(bsum.cpu().double().numpy(), Bm.double().sum(0).numpy(), rtol=1e-5, atol=1e-3 * math.sqrt(M))
  if gvec is not None:
    want_g = A.double().T @ gvec.double()
    np.testing.assert_allclose(gout.cpu().double()[:K - 8].numpy(), want_g[:K - 8].numpy(), rtol=1e-4, atol=1e-3 * math.sqrt(M))


def test_gemm_nt_rejects_bad_shapes(ops):
  a = torch.zeros((128, 64), dtype=torch.bfloat16).cuda()
  with pytest.raises(ValueError, match='multiple of 128'):
    ops.gemm_nt(a, a, M=100, N=128, K1=64, Cb=a, ldcb=64, nb=64)
  with pytest.raises(ValueError, match='multiples of 64'):
    ops.gemm_nt(a, a, M=128, N=128, K1=48, Cb=a, ldcb=64, nb=64)
  bits = torch.zeros((128, 16), dtype=torch.uint8).cuda()
  with pytest.raises(ValueError, match='no bias and no ReLU'):
    ops.gemm_nt(a, a, M=128, N=128, K1=64, Cb=a, ldcb=64, nb=64, bits_in=bits, relu=True)
  with pytest.raises(ValueError, match='bits_row_mod'):
    ops.gemm_nt(a, a, M=128, N=128, K1=64, Cb=a, ldcb=64, nb=64, bits_in=bits, bits_row_mod=64)


@pytest.mark.parametrize('M,K,N', [(64, 128, 128), (4096, 256, 128), (8192, 512, 256), (2048 + 64, 128, 384)])
def test_gemm_tn(ops, M, K, N):
  gen = torch.Generator().manual_seed(7)
  A = _bf(torch.randn((M, K), generator=gen))
  Bm = _bf(torch.randn((M, N), generator=gen))
  ref = A.double().T @ Bm.double()
  Cout = torch.ones((K, N), dtype=torch.float32).cuda()
  bsum = torch.full((N,), 2.0).cuda()
  ops.gemm_tn(dev(A), dev(Bm), Cout, M=M, K=K, N=N, bias_out=bsum, bias_n_valid=N - 7)
  np.testing.assert_allclose(Cout.cpu().double().numpy(), (ref + 1).numpy(), rtol=1e-4, atol=1e-3 * math.sqrt(M))
  want_b = Bm.double().sum(0) + 2
  want_b[N - 7:] = 2
  np.testing.assert_allclose(bsum.cpu().double().numpy(), want_b.numpy(), rtol=1e-5, atol=1e-3 * math.sqrt(M))
  # bounds: only a [k_valid, n_valid] window is touched, with a wider ldc
  C2 = torch.zeros((K, N + 8), dtype=torch.float32).cuda()
  ops.gemm_tn(dev(A), dev(Bm), C2, M=M, K=K, N=N, k_valid=K - 5, n_valid=N - 3)
  got = C2.cpu().double()
  np.testing.assert_allclose(got[:K - 5, :N - 3].numpy(), ref[:K - 5, :N - 3].numpy(), rtol=1e-4, atol=1e-3 * math.sqrt(M))
  assert (got[K - 5:] == 0).all() and (got[:, N - 3:] == 0).all()


@pytest.mark.parametrize('M,K,N,ldb', [(4096, 256, 256, 384), (8192 + 64, 512, 256, 256), (2048, 256, 512, 512)])
def test_gemm_tn_extra_column(ops, M, K, N, ldb):
  """mnr_gemm_tn_bf16 with `gcol`: one more column of B as an fp32 vector (the density head's gradient next to the
  bottleneck's, models.py:460 / :527; a contiguous bf16 vector): gcol_out[k] += sum_m A[m,k] g[m], exact products in fp32 like every other
  column; C, the fused bias gradient and the k_valid bound unchanged by it."""
  gen = torch.Generator().manual_seed(17)
  A = _bf(torch.randn((M, K), generator=gen))
  Bfull = _bf(torch.randn((M, ldb), generator=gen))
  g = torch.randn((M,), generator=gen) * 0.37
  ref = A.double().T @ Bfull[:, :N].double()
  ref_g = A.double().T @ _bf(g).double()
  Cout = torch.ones((K, N), dtype=torch.float32).cuda()
  bsum = torch.zeros((N,)).cuda()
  gout = torch.full((K,), 3.0).cuda()
  kv = K - 5
  ops.gemm_tn(dev(A), dev(Bfull), Cout, M=M, K=K, N=N, ldb=ldb, k_valid=kv, bias_out=bsum, bias_n_valid=N, gcol=dev(_bf(g)), gcol_out=gout)
  got = Cout.cpu().double()
  np.testing.assert_allclose(got[:kv].numpy(), (ref[:kv] + 1).numpy(), rtol=1e-4, atol=1e-3 * math.sqrt(M))
  assert (got[kv:] == 1).all()
  np.testing.assert_allclose(bsum.cpu().double().numpy(), Bfull[:, :N].double().sum(0).numpy(), rtol=1e-5, atol=1e-3 * math.sqrt(M))
  gg = gout.cpu().double()
  np.testing.assert_allclose(gg[:kv].numpy(), (ref_g[:kv] + 3).numpy(), rtol=1e-4, atol=1e-3 * math.sqrt(M))
  assert (gg[kv:] == 3).all()
  with pytest.raises(ValueError, match='gcol'):
    ops.gemm_tn(dev(A), dev(Bfull), Cout, M=M, K=K, N=128, ldb=ldb, gcol=dev(_bf(g)), gcol_out=gout)


@pytest.mark.parametrize('M,K,N,pitch', [(16384, 256, 256, 32), (8192 + 64, 512, 512, 80), (128, 256, 256, 32),
                                         pytest.param(1 << 20, 256, 256, 32, id='proposal_level_rows')])
def test_gemm_tn_rank1_b_operand(ops, M, K, N, pitch):
  """mnr_gemm_tn_args.rank1_*: B[m, n] = bit ? bf16(g[m] w[n]) : 0 built in LDS from the factors (the proposal MLP's last dY,
  models.py:457-460: relu'(z) * (g_density (x) w_density)), against fp64 of the same bf16 matrix and against the launch that
  reads it stored (same image, same order inside a workgroup: equal up to the order of the fp32 atomics); the fused bias
  gradient and the k_valid / n_valid window; two n-tiles and a mask pitch wider than the tile, a single 64-row step, and the
  360.gin proposal level's own row count."""
  gen = torch.Generator().manual_seed(29)
  A = dev(_bf(torch.randn((M, K), generator=gen)))
  g = dev(0.1 * torch.randn((M,), generator=gen))
  w = dev(torch.randn((N,), generator=gen))
  bits = dev(torch.randint(0, 256, (M, pitch), generator=gen, dtype=torch.uint8))
  bit = ((bits[:, :N // 8, None].int() >> torch.arange(8, device=bits.device)) & 1).reshape(M, N).bool()
  Bm = torch.where(bit, g[:, None] * w[None, :], torch.zeros((), device=g.device)).bfloat16()
  ref = A.double().T @ Bm.double()
  kv, nv = K - 8, N - 3
  C0, b0 = torch.ones((K, N), dtype=torch.float32).cuda(), torch.zeros((N,)).cuda()
  C1, b1 = torch.ones((K, N), dtype=torch.float32).cuda(), torch.zeros((N,)).cuda()
  ops.gemm_tn(A, Bm, C0, M=M, K=K, N=N, k_valid=kv, n_valid=nv, bias_out=b0, bias_n_valid=N)
  ops.gemm_tn(A, None, C1, M=M, K=K, N=N, k_valid=kv, n_valid=nv, bias_out=b1, bias_n_valid=N, rank1=(g, w, bits))
  got = C1.double()
  np.testing.assert_allclose(got[:kv, :nv].cpu().numpy(), (ref + 1)[:kv, :nv].cpu().numpy(), rtol=1e-4, atol=1e-4 * math.sqrt(M))
  assert (got[kv:] == 1).all() and (got[:, nv:] == 1).all()
  np.testing.assert_allclose(b1.double().cpu().numpy(), Bm.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * math.sqrt(M))
  rel = ((C1 - C0).double().norm() / (C0 - 1).double().norm()).item()
  print(f'rank1 vs stored B at M = {M}: |dC| / |C| = {rel:.2e}')
  assert rel < 1e-6
  with pytest.raises(ValueError, match='rank1'):
    ops.gemm_tn(A, None, C1, M=M, K=K, N=N, rank1=(g, w, bits), gcol=dev(_bf(torch.zeros(M))), gcol_out=torch.zeros((K,)).cuda())


@pytest.mark.parametrize('K1,K2,bits', [(1024, 0, False), (1024, 512, False), (1024, 0, True)])
def test_gemm_nt_trunk_shapes(ops, K1, K2, bits):
  """The 360.gin trunk's own GEMM shapes (1024-wide layers, the 1536-wide skip layer, a dX layer with 1-bit masks) at
  M = 65536 rows = 256 output tiles per column block: every persistent workgroup walks several tiles, the pipelined K
  loop runs its steady-state, tail and cross-segment bodies.  Reference: fp64 matmul of the same bf16 operands (on the
  device: 2 x 10^11 MACs are seconds of CPU otherwise)."""
  if not torch.cuda.is_available() or not dev(torch.zeros(1)).is_cuda:
    pytest.skip('needs a real device for the fp64 reference')
  gen = torch.Generator().manual_seed(60)
  M, N = 65536, 1024
  A1 = dev(_bf(torch.randn((M, K1), generator=gen)))
  A2 = dev(_bf(torch.randn((M, K2), generator=gen))) if K2 else None
  Bt = dev(_bf(torch.randn((N, K1 + K2), generator=gen) / math.sqrt(K1 + K2)))
  bias = dev(torch.randn((N,), generator=gen))
  A = A1 if A2 is None else torch.cat([A1, A2], -1)
  Cb = torch.zeros((M, N), dtype=torch.bfloat16).cuda()
  if not bits:
    ref = torch.relu(A.double() @ Bt.double().T + bias.double())
    bo = torch.zeros((M, N // 8), dtype=torch.uint8).cuda()
    ops.gemm_nt(A1, Bt, M=M, N=N, K1=K1, A2=A2, K2=K2, bias=bias, n_bias=N, relu=True, Cb=Cb, ldcb=N, nb=N, bits_out=bo)
    got_bits = ((bo.int()[..., None] >> torch.arange(8, device=bo.device)) & 1).bool()
    assert torch.equal(got_bits, (Cb.float() > 0).reshape(M, N // 8, 8))
  else:
    mask = torch.rand((M, N), generator=gen) > 0.5
    mb = dev((mask.reshape(M, N // 8, 8).int() << torch.arange(8)).sum(-1).to(torch.uint8))
    ref = (A.double() @ Bt.double().T) * dev(mask).double()
    ops.gemm_nt(A1, Bt, M=M, N=N, K1=K1, Cb=Cb, ldcb=N, nb=N, bits_in=mb)
  err = (Cb.double() - ref).abs()
  tol = 2.0 ** -7 * ref.abs() + 1e-2
  bad = int((err > tol).sum())
  print(f'gemm_nt {M}x{N}x{K1}+{K2} bits_in={bits}: max abs err {err.max().item():.3e}, '
        f'max err / (2^-8 |ref| + 1e-3) {(err / (2.0 ** -8 * ref.abs() + 1e-3)).max().item():.2f}')
  assert bad == 0, bad


@pytest.mark.parametrize('K1,K2', [(1024, 0), (1024, 512), (512, 0)])
def test_gemm_nt_vector_column_at_the_head_shape(ops, K1, K2):
  """mnr_gemm_nt_args.vcol at the merged head's shape (M = 65536 rows of a panel-storage trunk activation, N = 256 bottleneck
  columns + the density column as a vector; also behind a two-segment input): the bf16 result bitwise the plain launch's, the
  vector column bitwise the fp32 side column of the merged 257-column operand and within fp32 rounding of an fp64 reference."""
  if not torch.cuda.is_available() or not dev(torch.zeros(1)).is_cuda:
    pytest.skip('a trunk-sized launch: needs a real device (tests/test_sim_gemm.py runs the kernel on the simulator)')
  gen = torch.Generator().manual_seed(61)
  M = 65536
  K = K1 + K2
  X = dev(_bf(torch.relu(torch.randn((M, K1), generator=gen))))
  X2 = dev(_bf(torch.randn((M, K2), generator=gen))) if K2 else None
  Bt = _bf(torch.randn((512, K), generator=gen) / math.sqrt(K))
  Bt[257:] = 0
  Bt = dev(Bt)
  bias = dev(torch.randn((257,), generator=gen))
  Xp = ops.to_panel(X)
  C0 = torch.zeros((M, 384), dtype=torch.bfloat16).cuda()
  f0 = torch.zeros((M,), dtype=torch.float32).cuda()
  ops.gemm_nt(Xp, Bt, M=M, N=512, K1=K1, A2=X2, K2=K2, bias=bias, n_bias=257, Cb=C0, ldcb=384, nb=256, Cf=f0, ldcf=1, f0=256, nf=1,
              a1_layout=ops.LAYOUT_PANEL)
  C1 = torch.zeros((M, 384), dtype=torch.bfloat16).cuda()
  v1 = torch.zeros((M,), dtype=torch.float32).cuda()
  ops.gemm_nt(Xp, Bt[:256], M=M, N=256, K1=K1, A2=X2, K2=K2, bias=bias, n_bias=256, Cb=C1, ldcb=384, nb=256,
              vcol=Bt[256], vcol_out=v1, vcol_bias=bias[256:257], a1_layout=ops.LAYOUT_PANEL)
  torch.cuda.synchronize()
  assert torch.equal(C1.view(torch.int16), C0.view(torch.int16))
  assert torch.equal(v1, f0)
  A = X if X2 is None else torch.cat([X, X2], -1)
  if A.is_cuda:
    ref = A.double() @ Bt[256].double() + bias[256].double()
    err = (v1.double() - ref).abs().max().item()
    print(f'gemm_nt vcol {M}x256x{K1}+{K2}: max |vector column - fp64| {err:.3e}')
    assert err < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('K', [1024, 512])
def test_gemm_tn_trunk_shapes(ops, K):
  """The trunk's weight-gradient shapes: [1024 (or the 512 padded feature columns), 1024] outputs reduced over 65536
  rows (16 / 8 output tiles x 16 / 32 M-splits of the 256-wide tile), bias gradient fused; vs fp64 on the device."""
  if not torch.cuda.is_available() or not dev(torch.zeros(1)).is_cuda:
    pytest.skip('needs a real device for the fp64 reference')
  gen = torch.Generator().manual_seed(70)
  M, N = 65536, 1024
  A = dev(_bf(torch.randn((M, K), generator=gen)))
  Bm = dev(_bf(torch.randn((M, N), generator=gen)))
  ref = A.double().T @ Bm.double()
  Cout = torch.ones((K, N), dtype=torch.float32).cuda()
  bsum = torch.zeros((N,)).cuda()
  kv = K - 8 if K == 512 else K                       # 504 valid feature columns of the 512
  ops.gemm_tn(A, Bm, Cout, M=M, K=K, N=N, k_valid=kv, n_valid=N, bias_out=bsum, bias_n_valid=N)
  got = Cout.double()
  # fp32 accumulation of exact products over 65536 terms of unit variance: |err| ~ 2^-24 * sqrt(M) * |partial sums|
  np.testing.assert_allclose(got[:kv].cpu().numpy(), (ref[:kv] + 1).cpu().numpy(), rtol=1e-4, atol=1e-3 * math.sqrt(M))
  assert (got[kv:] == 1).all()
  np.testing.assert_allclose(bsum.double().cpu().numpy(), Bm.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3 * math.sqrt(M))
  rel = ((got[:kv] - 1 - ref[:kv]).norm() / ref[:kv].norm()).item()
  print(f'gemm_tn {K}x{N} over M={M}: relative L2 error {rel:.2e}')
  assert rel < 1e-5


def test_colsum_pack_scatter_cast_smallhead(ops):
  gen = torch.Generator().manual_seed(8)
  M, N = 3000, 320
  X = _bf(torch.randn((M, N), generator=gen))
  out = torch.zeros(N).cuda()
  ops.colsum(dev(X), M, 257, out)
  ref = X.double().sum(0)
  np.testing.assert_allclose(out.cpu().double()[:257].numpy(), ref[:257].numpy(), rtol=1e-5, atol=1e-3)
  assert (out.cpu()[257:] == 0).all()

  # pack: a [in=5,out=7] kernel, plain at (1,2) of a [8,16] matrix and transposed at (0,3) of a [16,8] one.
  import ctypes as C
  from multinerf_amd import _lib as L
  params = torch.randn(100, generator=gen)
  descs = (L.PackDesc * 2)(L.PackDesc(10, 5, 7, 0, 16, 1, 2, 0), L.PackDesc(10, 5, 7, 128, 8, 0, 3, 1))
  dbytes = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).cuda()
  dst = torch.zeros(128 + 128, dtype=torch.bfloat16).cuda()
  ops.pack_weights(dev(params), dbytes, 2, 35, dst)
  W = params[10:45].reshape(5, 7)
  d = dst.cpu().float()
  a = d[:128].reshape(8, 16)
  b = d[128:].reshape(16, 8)
  wb = W.to(torch.bfloat16).float()
  assert torch.equal(a[1:6, 2:9], wb) and a.abs().sum() == wb.abs().sum()
  assert torch.equal(b[0:7, 3:8], wb.T) and b.abs().sum() == wb.abs().sum()

  src = torch.randn((6, 10), generator=gen)
  dstf = torch.ones((3, 4)).cuda()
  ops.scatter_add(dev(src), 10, 2, 5, 3, 4, dstf, 4)
  np.testing.assert_allclose(dstf.cpu().numpy(), (1 + src[2:5, 5:9]).numpy(), rtol=1e-6)

  g = torch.randn((M, 3), generator=gen)
  cb = torch.zeros((M, 8), dtype=torch.bfloat16).cuda()
  ops.cast_f32_to_bf16(dev(g), 3, M, 3, cb, 8, 2)
  assert torch.equal(cb.cpu()[:, 2:5], g.to(torch.bfloat16)) and (cb.cpu()[:, 5:] == 0).all()

  K = 128
  H = _bf(torch.relu(torch.randn((M, K), generator=gen)))
  Wk = torch.randn((K, 3), generator=gen)
  dX = torch.zeros((M, K), dtype=torch.bfloat16).cuda()
  dW = torch.zeros((K, 3)).cuda()
  db = torch.zeros(3).cuda()
  ops.small_head_bwd(dev(H), K, dev(g), dev(Wk), M=M, K=K, Cn=3, dX=dX, lddx=K, relu_mask=True, dW=dW, db=db)
  ref_dx = (g.double() @ Wk.double().T) * (H.double() > 0)
  np.testing.assert_allclose(dX.cpu().double().numpy(), ref_dx.numpy(), rtol=2**-7, atol=1e-3)
  np.testing.assert_allclose(dW.cpu().double().numpy(), (H.double().T @ g.double()).numpy(), rtol=1e-4, atol=1e-2)
  np.testing.assert_allclose(db.cpu().double().numpy(), g.double().sum(0).numpy(), rtol=1e-5, atol=1e-3)


# ----------------------------------------------------------------------------- compositing


@pytest.mark.parametrize('n,opaque,has_rgb,rgb_act,pad,noise', [(32, True, True, 'sigmoid', 0.001, False),
                                                                (64, True, False, 'sigmoid', 0.001, False),
                                                                (30, True, True, 'sigmoid', 0.001, False),     # not a multiple of the 4 lanes per ray
                                                                (128, False, True, 'safe_exp', 0.0, True)])
def test_composite_fwd_bwd(ops, level_bwd_kernel, n, opaque, has_rgb, rgb_act, pad, noise):
  gen = torch.Generator().manual_seed(9)
  B = 150
  raw_d = torch.randn((B, n), generator=gen) * 2
  raw_rgb = torch.randn((B, n, 3), generator=gen)
  tdist = torch.cumsum(torch.rand((B, n + 1), generator=gen) * 0.1 + 1e-3, -1) + 0.2
  dirs = torch.randn((B, 3), generator=gen)
  bg = torch.rand((B, 3), generator=gen)
  expo = (0.5 + torch.rand((B, 3), generator=gen)) if rgb_act == 'safe_exp' else None
  dnoise = torch.randn((B, n), generator=gen) if noise else None
  premult, rgb_bias, dbias = 1.0, (-5.0 if rgb_act == 'safe_exp' else 0.0), -1.0
  g_out = torch.randn((B, 3), generator=gen)
  g_w = torch.randn((B, n), generator=gen) * 0.1

  rd = raw_d.clone().requires_grad_(True)
  rr = raw_rgb.clone().requires_grad_(True)
  raw = rd + (dnoise if noise else 0.0)
  density = torch.nn.functional.softplus(raw + dbias)
  w_ref, _, _ = orender.compute_alpha_weights(density, tdist, dirs, opaque_background=opaque)
  if has_rgb:
    act = omodels._ACT[rgb_act]
    rgb = act(premult * rr + rgb_bias) * (1 + 2 * pad) - pad
    if expo is not None:
      rgb = rgb * expo[:, None, :]
  else:
    rgb = torch.zeros((B, n, 3))
  rend = orender.volumetric_rendering(rgb, w_ref, tdist, bg, tdist[:, -1:], False)
  obj = (rend['rgb'] * g_out).sum() + (w_ref * g_w).sum()
  obj.backward()

  cfg = ops.composite_cfg(n, opaque_background=opaque, density_act='softplus', density_bias=dbias,
                          density_noise_std=1.0 if noise else 0.0, has_rgb=has_rgb, rgb_act=rgb_act,
                          rgb_premultiplier=premult, rgb_bias=rgb_bias, rgb_padding=pad, bg_mode=1, bg_value=0.0)
  kw = dict(raw_rgb=dev(raw_rgb) if has_rgb else None, density_noise=dev(dnoise) if noise else None, bg=dev(bg),
            exposure_scale=dev(expo) if expo is not None else None)
  den, rgb_s, w, rgb_out, acc = ops.composite_fwd(cfg, dev(raw_d), dev(tdist), dev(dirs), **kw)
  np.testing.assert_allclose(den.cpu().numpy(), density.detach().numpy(), rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(w.cpu().numpy(), w_ref.detach().numpy(), rtol=2e-5, atol=1e-6)
  np.testing.assert_allclose(rgb_out.cpu().numpy(), rend['rgb'].detach().numpy(), rtol=2e-5, atol=2e-6)
  np.testing.assert_allclose(acc.cpu().numpy(), w_ref.detach().sum(-1).numpy(), rtol=2e-5, atol=1e-6)
  if has_rgb:
    np.testing.assert_allclose(rgb_s.cpu().numpy(), rgb.detach().numpy(), rtol=2e-5, atol=1e-6)
  gbf = torch.zeros((B * n, 8), dtype=torch.bfloat16).cuda()
  g_rd, g_rr = ops.composite_bwd(cfg, dev(raw_d), dev(tdist), dev(dirs), w, g_rgb_out=dev(g_out), g_weights=dev(g_w),
                                 g_den_bf16=gbf.view(-1)[3:], ld_bf16=8, **kw)
  sc = rd.grad.abs().max().item()
  np.testing.assert_allclose(g_rd.cpu().numpy(), rd.grad.numpy(), rtol=2e-4, atol=2e-5 * sc)
  assert torch.equal(gbf.cpu()[:, 3].reshape(B, n), g_rd.cpu().to(torch.bfloat16))
  if has_rgb:
    np.testing.assert_allclose(g_rr.cpu().numpy(), rr.grad.numpy(), rtol=2e-4, atol=2e-5 * rr.grad.abs().max().item())


def test_render_extras(ops):
  gen = torch.Generator().manual_seed(10)
  B, n = 77, 32
  density = torch.exp(torch.randn((B, n), generator=gen))
  density[5] = 0
  tdist = torch.cumsum(torch.rand((B, n + 1), generator=gen) * 0.1 + 1e-3, -1) + 0.2
  dirs = torch.randn((B, 3), generator=gen)
  w, _, _ = orender.compute_alpha_weights(density, tdist, dirs, opaque_background=False)
  far = torch.full((B, 1), 1e6)
  ref = orender.volumetric_rendering(torch.zeros((B, n, 3)), w, tdist, 0.5, far, True)
  out = ops.render_extras(dev(w), dev(tdist), dev(far.reshape(-1))).cpu()
  for i, k in enumerate(['distance_mean', 'distance_percentile_5', 'distance_median', 'distance_percentile_95']):
    np.testing.assert_allclose(out[:, i].numpy(), ref[k].numpy(), rtol=2e-4, atol=1e-5, err_msg=k)


# ----------------------------------------------------------------------------- losses / optimiser


def test_losses(ops):
  gen = torch.Generator().manual_seed(11)
  B, Bv, n, ne = 300, 290, 32, 64
  t, w = rand_stepfun(gen, B, n)
  te, we = rand_stepfun(gen, B, ne)
  w, we = w * 0.9, we * 0.95
  np.testing.assert_allclose(ops.lossfun_outer(dev(t), dev(w), dev(te), dev(we)).cpu().numpy(),
                             ostepfun.lossfun_outer(t, w, te, we).numpy(), rtol=1e-3, atol=1e-7)
  np.testing.assert_allclose(ops.lossfun_distortion(dev(t), dev(w)).cpu().numpy(),
                             ostepfun.lossfun_distortion(t, w).numpy(), rtol=1e-5, atol=1e-8)
  # interlevel + distortion with gradients, B_valid < B (padded rays contribute nothing)
  wv = w[:Bv].clone().requires_grad_(True)
  wev = we[:Bv].clone().requires_grad_(True)
  li = 1.0 * torch.mean(ostepfun.lossfun_outer(t[:Bv], wv.detach(), te[:Bv], wev))
  ld = 0.01 * torch.mean(ostepfun.lossfun_distortion(t[:Bv], wv))
  (li + ld).backward()
  stats = torch.zeros(2).cuda()
  g_we = torch.zeros((B, ne)).cuda()
  g_w = torch.zeros((B, n)).cuda()
  ops.interlevel_loss(1.0, dev(t), dev(w), dev(te), dev(we), stats[0:1], g_we, B_valid=Bv)
  ops.distortion_loss(0.01, dev(t), dev(w), stats[1:2], g_w, B_valid=Bv)
  np.testing.assert_allclose(stats.cpu().numpy(), [li.item(), ld.item()], rtol=1e-4)
  # the window sums add/subtract in a different order than autograd's scatter: 1e-3 of the largest entry.
  np.testing.assert_allclose(g_we.cpu()[:Bv].numpy(), wev.grad.numpy(), rtol=1e-3,
                             atol=1e-3 * wev.grad.abs().max().item())
  np.testing.assert_allclose(g_w.cpu()[:Bv].numpy(), wv.grad.numpy(), rtol=1e-4, atol=1e-10)
  assert (g_we.cpu()[Bv:] == 0).all() and (g_w.cpu()[Bv:] == 0).all()

  class Cfg:
    disable_multiscale_loss = False
    charb_padding = 0.001
    data_coarse_loss_mult = 0.0
    data_loss_mult = 1.0
    compute_disp_metrics = False
    compute_normal_metrics = False

  class Obj:
    pass

  for loss_type, lm_c in [('charb', 1), ('mse', 1), ('rawnerf', 3)]:
    Cfg.data_loss_type = loss_type
    rgb = (torch.rand((B, 3), generator=gen) * 1.3).requires_grad_(True)
    gt = torch.rand((B, 3), generator=gen)
    lm = torch.rand((B, lm_c), generator=gen)
    batch, rays = Obj(), Obj()
    batch.rgb, rays.lossmult = gt[:Bv], lm[:Bv]
    loss, st = otrain.compute_data_loss(batch, [{'rgb': rgb[:Bv]}], rays, Cfg)
    loss.backward()
    denom = torch.zeros(1).cuda()
    ops.lossmult_sum(dev(lm), Bv, denom)
    stats = torch.zeros(2).cuda()
    g = ops.data_loss(loss_type, 0.001, 1.0, dev(rgb.detach()), dev(gt), dev(lm), denom, stats, B_valid=Bv)
    np.testing.assert_allclose(stats.cpu().numpy(), [st['mses'][0].item(), loss.item()], rtol=1e-4)
    np.testing.assert_allclose(g.cpu().numpy(), rgb.grad.numpy(), rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize('mode,loss_type,lm_c,has_rgb', [('interlevel', 'charb', 1, False), ('distortion', 'charb', 1, True),
                                                         ('distortion', 'rawnerf', 3, True), (None, 'mse', 1, True)])
def test_level_bwd_fuses_losses_and_compositing_vjp(ops, level_bwd_kernel, mode, loss_type, lm_c, has_rgb):
  """mnr_level_bwd: data loss + interlevel | distortion loss + compositing VJP in ONE launch, against torch autograd of
  the oracle's composed objective (train_utils.py:72-159 on render.py:130-213), B_valid < B, with an upstream d / d weights
  (the Ref-NeRF normal losses' path) on top."""
  gen = torch.Generator().manual_seed(21)
  B, Bv, n, n_ref = 137, 130, 64 if mode == 'interlevel' else 32, 32
  raw_d = torch.randn((B, n), generator=gen) * 2
  raw_rgb = torch.randn((B, n, 3), generator=gen)
  sdist, _ = rand_stepfun(gen, B, n)
  tdist = 0.2 + 5.8 * sdist
  t_ref, w_ref = rand_stepfun(gen, B, n_ref)
  w_ref = w_ref * 0.9
  dirs = torch.randn((B, 3), generator=gen)
  bg = torch.rand((B, 3), generator=gen)
  gt = torch.rand((B, 3), generator=gen)
  lm = torch.rand((B, lm_c), generator=gen)
  g_w_up = torch.randn((B, n), generator=gen) * 0.01
  g_w_up[Bv:] = 0                                         # (upstream gradients of padded rays are zero by construction)
  dbias, pad, dmult, wmult = -1.0, 0.001, 1.0, (1.0 if mode == 'interlevel' else 0.01)

  rd = raw_d.clone().requires_grad_(True)
  rr = raw_rgb.clone().requires_grad_(True)
  density = torch.nn.functional.softplus(rd + dbias)
  w, _, _ = orender.compute_alpha_weights(density, tdist, dirs, opaque_background=True)
  rgb = torch.sigmoid(rr) * (1 + 2 * pad) - pad if has_rgb else torch.zeros((B, n, 3))
  rend = orender.volumetric_rendering(rgb, w, tdist, bg, tdist[:, -1:], False)

  class Cfg:
    disable_multiscale_loss = False
    charb_padding = 0.001
    data_coarse_loss_mult = 0.0
    data_loss_mult = dmult
    compute_disp_metrics = False
    compute_normal_metrics = False
    data_loss_type = loss_type

  class Obj:
    pass

  batch, rays = Obj(), Obj()
  batch.rgb, rays.lossmult = gt[:Bv], lm[:Bv]
  dloss, st = otrain.compute_data_loss(batch, [{'rgb': rend['rgb'][:Bv]}], rays, Cfg)
  if mode == 'interlevel':
    wloss = wmult * torch.mean(ostepfun.lossfun_outer(t_ref[:Bv], w_ref[:Bv], sdist[:Bv], w[:Bv]))
  elif mode == 'distortion':
    wloss = wmult * torch.mean(ostepfun.lossfun_distortion(sdist[:Bv], w[:Bv]))
  else:
    wloss = torch.zeros(())
  (dloss + wloss + (w * g_w_up).sum()).backward()

  cfg = ops.composite_cfg(n, opaque_background=True, density_act='softplus', density_bias=dbias, density_noise_std=0.0,
                          has_rgb=has_rgb, rgb_act='sigmoid', rgb_premultiplier=1.0, rgb_bias=0.0, rgb_padding=pad, bg_mode=1,
                          bg_value=0.0)
  kw = dict(raw_rgb=dev(raw_rgb) if has_rgb else None, bg=dev(bg))
  _, _, w_k, rgb_out, _ = ops.composite_fwd(cfg, dev(raw_d), dev(tdist), dev(dirs), **kw)
  denom = torch.zeros(1).cuda()
  ops.lossmult_sum(dev(lm), Bv, denom)
  stats = torch.zeros(3).cuda()
  wspec = None if mode is None else dict(mode=mode, mult=wmult, sdist=dev(sdist), t_ref=dev(t_ref), w_ref=dev(w_ref), stat=stats[2:3])
  g_rd, g_rr = ops.composite_bwd(cfg, dev(raw_d), dev(tdist), dev(dirs), w_k, g_weights=dev(g_w_up), **kw,
                                 losses=dict(B_valid=Bv, data=dict(type=loss_type, charb_padding=0.001, mult=dmult, rgb_out=rgb_out,
                                                                  gt=dev(gt), lossmult=dev(lm), denom=denom, stats=stats[0:2]),
                                             weights=wspec))
  np.testing.assert_allclose(stats.cpu().numpy(), [st['mses'][0].item(), dloss.item(), wloss.item()], rtol=2e-4, atol=1e-9)
  sc = rd.grad.abs().max().item()
  np.testing.assert_allclose(g_rd.cpu().numpy(), rd.grad.numpy(), rtol=1e-3, atol=2e-4 * sc)
  if has_rgb:
    np.testing.assert_allclose(g_rr.cpu().numpy(), rr.grad.numpy(), rtol=1e-3, atol=2e-4 * rr.grad.abs().max().item())
  assert (g_rd.cpu()[Bv:] == 0).all()                     # padded rays: no loss, no gradient


def test_clip_adam(ops):
  gen = torch.Generator().manual_seed(12)
  P = 100003
  seg = [(0, 60000), (60000, P)]
  params = torch.randn(P, generator=gen)
  grad = torch.randn(P, generator=gen) * 1e-2
  grad[7] = float('nan')
  mu = torch.randn(P, generator=gen) * 1e-3
  nu = torch.rand(P, generator=gen) * 1e-5

  class Cfg:
    adam_beta1, adam_beta2, adam_eps = 0.9, 0.999, 1e-6
    lr_init, lr_final, max_steps, lr_delay_steps, lr_delay_mult = 2e-3, 2e-5, 250000, 512, 0.01
    grad_max_norm, grad_max_val = 1e-3, 0.05

  tree_p = {'a': {'k': params[:60000].clone()}, 'b': {'k': params[60000:].clone()}}
  tree_g = {'a': {'k': grad[:60000].clone()}, 'b': {'k': grad[60000:].clone()}}
  g = otrain.clip_gradients(tree_g, Cfg)
  g = otrain.tree_map(torch.nan_to_num, g)
  state = {'count': 41, 'mu': {'a': {'k': mu[:60000].clone()}, 'b': {'k': mu[60000:].clone()}},
           'nu': {'a': {'k': nu[:60000].clone()}, 'b': {'k': nu[60000:].clone()}}}
  new_p, new_s = otrain.adam_update(tree_p, g, state, Cfg)
  dp, dg, dmu, dnu = dev(params), dev(grad), dev(mu), dev(nu)
  lr = float(otrain.lr_fn(Cfg, 41))
  for (b, e) in seg:
    sq = torch.zeros(1).cuda()
    ops.grad_sqnorm(dg, b, e, Cfg.grad_max_val, sq)
    ops.clip_adam(dg, dp, dmu, dnu, b, e, sq, lr=lr, b1=0.9, b2=0.999, eps=1e-6, step=42,
                  grad_max_val=Cfg.grad_max_val, grad_max_norm=Cfg.grad_max_norm)
  ref_p = torch.cat([new_p['a']['k'], new_p['b']['k']])
  # NaN gradient: the reference's norm is NaN for that module (mult = NaN -> nan_to_num(0)); the kernel
  # reproduces this because the NaN enters the sum of squares the same way.
  np.testing.assert_allclose(dp.cpu().numpy(), ref_p.numpy(), rtol=1e-5, atol=1e-7)
  np.testing.assert_allclose(dmu.cpu().numpy(), torch.cat([new_s['mu']['a']['k'], new_s['mu']['b']['k']]).numpy(), rtol=1e-5, atol=1e-9)
  np.testing.assert_allclose(dnu.cpu().numpy(), torch.cat([new_s['nu']['a']['k'], new_s['nu']['b']['k']]).numpy(), rtol=1e-5, atol=1e-12)


# ----------------------------------------------------------------------------- small ABI leaves added with
# GLO / Ref-NeRF / metrics / weight decay (each against a few lines of torch)

def test_glo_fill_and_bwd(ops):
  g = torch.Generator().manual_seed(3)
  B, n, G, E, ld, col0 = 9, 4, 5, 17, 32, 8
  table = torch.randn((E, G), generator=g).cuda()
  cam = torch.randint(0, E, (B,), generator=g).to(torch.int32).cuda()
  dst = torch.full((B * n, ld), 7.0, dtype=torch.bfloat16).cuda()
  ops.glo_fill(table, cam, B, n, dst, col0)
  want = table[cam.long()].repeat_interleave(n, 0).to(torch.bfloat16)
  assert torch.equal(dst[:, col0:col0 + G], want)
  assert (dst[:, :col0] == 7).all() and (dst[:, col0 + G:] == 7).all()          # only its own columns
  ops.glo_fill(table, None, B, n, dst, col0)                                      # zero_glo
  assert (dst[:, col0:col0 + G] == 0).all()
  ga = torch.randn((B * n, G), generator=g).cuda()
  gb = torch.randn((B * n, G), generator=g).cuda()
  grad = torch.zeros((E, G)).cuda()
  ops.glo_bwd(ga, gb, cam, B, n, grad.view(-1), E, G)
  ref = torch.zeros((E, G)).cuda().index_add_(0, cam.long(), (ga + gb).view(B, n, G).sum(1))
  np.testing.assert_allclose(grad.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_add_cols_bf16(ops):
  g = torch.Generator().manual_seed(4)
  a = torch.randn((64, 48), generator=g).to(torch.bfloat16).cuda()
  b = torch.randn((64, 40), generator=g).to(torch.bfloat16).cuda()
  dst = torch.zeros((64, 56), dtype=torch.bfloat16).cuda()
  ops.add_cols_bf16(a, b, dst, 32)
  assert torch.equal(dst[:, :32], (a[:, :32].float() + b[:, :32].float()).to(torch.bfloat16))
  assert (dst[:, 32:] == 0).all()
  ops.add_cols_bf16(a, None, dst, 16)
  assert torch.equal(dst[:, :16], a[:, :16])


def test_render_metrics(ops):
  g = torch.Generator().manual_seed(5)
  B = 777
  dm, disp = torch.rand(B, generator=g) * 5, torch.rand(B, generator=g)
  acc, al = torch.rand(B, generator=g), torch.rand(B, generator=g)
  n, ngt = torch.randn((B, 3), generator=g), torch.randn((B, 3), generator=g)
  out = torch.zeros(2).cuda()
  ops.render_metrics(B, distance_mean=dm.cuda(), disps=disp.cuda(), acc=acc.cuda(), alphas=al.cuda(), normals=n.cuda(),
                     normals_gt=ngt.cuda(), out_disp=out[0:1], out_normal=out[1:2])
  want_d = ((1 / (1 + dm) - disp)**2).mean()
  w = acc * al
  nn = n / n.norm(dim=-1, keepdim=True)
  gg = ngt / ngt.norm(dim=-1, keepdim=True)
  want_n = (w * torch.arccos(torch.clamp((nn * gg).sum(-1), -1 + 1.2e-7, 1 - 1.2e-7))).sum() / w.sum() * 180 / math.pi
  np.testing.assert_allclose(out.cpu().numpy(), [want_d.item(), want_n.item()], rtol=2e-5)


def test_weight_decay(ops):
  g = torch.Generator().manual_seed(6)
  p = torch.randn(5000, generator=g).cuda()
  grad = torch.randn(5000, generator=g).cuda()
  g0 = grad.clone()
  out = torch.zeros(2).cuda()
  ops.weight_decay(p, 100, 4100, 0.25, grad, out[0:1], out[1:2])
  sq = (p[100:4100].double()**2).sum().item()
  np.testing.assert_allclose(out.cpu().numpy(), [0.25 * sq, sq], rtol=1e-5)
  want = g0.clone()
  want[100:4100] += 0.5 * p[100:4100]
  np.testing.assert_allclose(grad.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_resample_nan_logits(ops):
  """train_frac = 0 (anneal = 0) with an exactly-zero weight: 0 * log(0) = NaN propagates through softmax / CDF as
  in the reference (jnp.max / jnp.minimum keep NaN) and sorted_interp then returns the first fence-post."""
  gen = torch.Generator().manual_seed(11)
  B, np_, n = 70, 64, 32
  sd, w = rand_stepfun(gen, B, np_)
  w = w * 0.9
  w[::2, 7] = 0.0                                   # every other ray has a zero weight
  near, far = torch.full((B, 1), 2.0), torch.full((B, 1), 6.0)
  u_base, mj = _u_base(n, False)
  s_ref, t_ref, idx_ref = _resample_ref(sd, w, None, near, far, n=n, use_dil=False, dil=0.0, anneal=0.0, pad=0.0,
                                        single=True, raydist=None)
  s, t, idx = ops.resample_level(dev(sd), dev(w), dev(u_base), None, dev(near), dev(far), n_samples=n,
                                 use_dilation=False, dilation=0.0, domain=(0., 1.), anneal=0.0, resample_padding=0.0,
                                 single_jitter=True, max_jitter=mj, raydist_fn=None, want_idx=True)
  assert torch.isfinite(s_ref).all()              # the reference resolves the NaN CDF to the first fence-post
  np.testing.assert_allclose(s.cpu().numpy(), s_ref.numpy(), atol=2e-6)
  assert (s.cpu()[::2] == sd[::2, :1]).all()      # NaN rays: every fence-post equals t[0]
  assert torch.equal(idx.cpu(), idx_ref)
