"""The GEMM kernels' SOURCE (multinerf_amd/csrc/gemm.hip) run through tools/hipsim, the host-side functional simulator.

What this is: index arithmetic, LDS swizzles, fragment layouts, K-loop pipeline bookkeeping (counted vmcnt waits, raw
barriers) and the epilogues of the shipped kernels, executed lane by lane on the CPU against a plain fp32 reference, with
LDS-DMA landing as early and as late as the hardware may deliver it and with different wave orders.  What it is not: a
product path (the package never loads the simulator) or evidence about the hardware; tests/test_gpu_kernels.py is that.
The kernels here are the ones the GPU suite validates, so agreement also pins the simulator's restated instruction
semantics (MFMA lane layout, ds_read_b64_tr_b16, DPP quad_perm, global_load_lds).
"""

import shutil

import ctypes

import numpy as np
import pytest
import torch

from tests import sim_helpers as S

pytestmark = pytest.mark.skipif(
    not (shutil.which('clang++') or __import__('os').path.exists('/opt/rocm/lib/llvm/bin/clang++')), reason='needs clang++')

# (LDS-DMA lands late, fiber order): eager/forward, late/forward, eager/reverse, late/shuffled
MODES = [(0, 0), (1, 0), (0, -1), (1, 11)]
C_ULL = ctypes.c_ulonglong


@pytest.fixture(scope='module')
def sim():
  lib = S.load_sim()
  yield lib
  lib.mnr_gemm_nt_set_persistent(1)
  lib.mnr_gemm_nt_set_wres(1)


def _packbits(x):
  return torch.from_numpy(np.packbits((x > 0).numpy(), axis=1, bitorder='little'))


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('persist', [0, -8])
def test_nt_forward_layer(sim, persist, mode):
  """Forward layer on the 256x256 tile: [A1|A2] W^T + b, ReLU, bf16 output + 1-bit ReLU masks; one workgroup per tile and
  a persistent launch (8 workgroups walking 16 virtual tiles: the LDS hand-over between a tile's epilogue and the next
  tile's prologue, the running tile index)."""
  g = torch.Generator().manual_seed(2)
  M, N, K1, K2 = 2304, 256, 192, 64                              # 9 M tiles -> 16 virtual workgroups (7 of them idle)
  A1 = torch.randn((M, K1), generator=g).bfloat16()
  A2 = torch.randn((M, K2), generator=g).bfloat16()
  Bt = (torch.randn((N, K1 + K2), generator=g) * 0.1).bfloat16()
  bias = torch.randn(N, generator=g)
  sim.mnr_gemm_nt_set_wres(0)
  sim.mnr_gemm_nt_set_persistent(persist)
  sim.hipsim_reset(*mode)
  try:
    Cb, _, bits = S.sim_gemm_nt(sim, A1, Bt, A2=A2, bias=bias, relu=True, bits_out=True)
  finally:
    sim.mnr_gemm_nt_set_persistent(1)
    sim.mnr_gemm_nt_set_wres(1)
  ref = torch.relu(torch.cat([A1, A2], 1).float() @ Bt.float().T + bias)
  np.testing.assert_allclose(Cb.float().numpy(), ref.numpy(), atol=2e-2, rtol=1e-2)       # bf16 output rounding
  assert torch.equal(bits, _packbits(Cb.float()))


@pytest.mark.parametrize('mode', [MODES[1], MODES[3]])
@pytest.mark.parametrize('K1,K2', [(64, 0), (128, 0), (64, 64), (128, 192)])
def test_nt_pipelined_loop_short_k_and_two_stage_loop(sim, K1, K2, mode):
  """The hand-pipelined K loop (BK = 32, 4 stages; default) with fewer K-tiles than stages / in every tail flavour, against the
  reference and bitwise against the two-stage BK = 64 loop it replaced (mnr_gemm_nt_set_pipelined(0))."""
  g = torch.Generator().manual_seed(5)
  M, N = 512, 256
  A1 = torch.randn((M, K1), generator=g).bfloat16()
  A2 = torch.randn((M, K2), generator=g).bfloat16() if K2 else None
  Bt = (torch.randn((N, K1 + K2), generator=g) * 0.1).bfloat16()
  bias = torch.randn(N, generator=g)
  sim.mnr_gemm_nt_set_wres(0)
  sim.mnr_gemm_nt_set_persistent(-8)
  try:
    sim.hipsim_reset(*mode)
    Cb, _, bits = S.sim_gemm_nt(sim, A1, Bt, A2=A2, bias=bias, relu=True, bits_out=True)
    sim.mnr_gemm_nt_set_pipelined(0)
    sim.hipsim_reset(*mode)
    Cb0, _, bits0 = S.sim_gemm_nt(sim, A1, Bt, A2=A2, bias=bias, relu=True, bits_out=True)
    assert torch.equal(Cb.view(torch.int16), Cb0.view(torch.int16)) and torch.equal(bits, bits0)
  finally:
    sim.mnr_gemm_nt_set_pipelined(1)
    sim.mnr_gemm_nt_set_persistent(1)
    sim.mnr_gemm_nt_set_wres(1)
  A = torch.cat([A1, A2], 1) if K2 else A1
  ref = torch.relu(A.float() @ Bt.float().T + bias)
  np.testing.assert_allclose(Cb.float().numpy(), ref.numpy(), atol=2e-2, rtol=1e-2)
  assert torch.equal(bits, _packbits(Cb.float()))


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('persist', [0, -8])
def test_nt_dx_layer_with_bit_masks(sim, persist, mode):
  """dX layer: (dY W) masked by the forward layer's bits (with the tangent rows' modulo), fp32 side output."""
  g = torch.Generator().manual_seed(5)
  M, N, K = 1024, 512, 128
  dY = torch.randn((M, K), generator=g).bfloat16()
  Wt = (torch.randn((N, K), generator=g) * 0.1).bfloat16()
  keep = torch.rand((256, N), generator=g) > 0.5
  bits = _packbits(keep.float())
  sim.mnr_gemm_nt_set_persistent(persist)
  sim.hipsim_reset(*mode)
  try:
    Cb, Cf, _ = S.sim_gemm_nt(sim, dY, Wt, bits_in=bits, bits_row_mod=256, out_f32=(8, 11))
  finally:
    sim.mnr_gemm_nt_set_persistent(1)
  ref = (dY.float() @ Wt.float().T) * keep.repeat(4, 1)
  np.testing.assert_allclose(Cb.float().numpy(), ref.numpy(), atol=2e-2, rtol=1e-2)
  np.testing.assert_allclose(Cf.numpy(), ref[:, 8:19].numpy(), atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize('K', [64, 128, 192, 256])
def test_nt_weights_resident_persistent_kernel(sim, K):
  """gemm_nt_wres_kernel (weights in registers, persistent workgroups walking the M tiles, activation tiles as one
  pipeline across tile boundaries): bitwise the tiled kernel's result, forward (partial bias, ReLU, bit masks) and dX
  (bit masks with the tangent rows' modulo), with 1, 2 and 3 workgroups for 5 tiles, in every mode."""
  g = torch.Generator().manual_seed(K)
  M, N = 1280, 256
  A = torch.randn((M, K), generator=g).bfloat16()
  Bt = (torch.randn((N, K + 8), generator=g) * 0.1).bfloat16()[:, :K]
  bias = torch.randn(200, generator=g)
  keep = torch.rand((256, N), generator=g) > 0.5
  bits_in = _packbits(keep.float())
  sim.mnr_gemm_nt_set_wres(0)
  sim.hipsim_reset(0, 0)
  want_f, _, want_b = S.sim_gemm_nt(sim, A, Bt, bias=bias, relu=True, bits_out=True)
  want_d, _, _ = S.sim_gemm_nt(sim, A, Bt, bits_in=bits_in, bits_row_mod=256)
  n0 = (C_ULL * 4)()
  try:
    for wgs, mode in [(5, MODES[0]), (2, MODES[1]), (3, MODES[2]), (2, MODES[3])]:
      sim.hipsim_reset(*mode)
      sim.mnr_gemm_nt_set_wres(wgs)
      got_f, _, got_b = S.sim_gemm_nt(sim, A, Bt, bias=bias, relu=True, bits_out=True)
      got_d, _, _ = S.sim_gemm_nt(sim, A, Bt, bits_in=bits_in, bits_row_mod=256)
      sim.hipsim_stats(n0)
      assert torch.equal(got_f, want_f) and torch.equal(got_b, want_b) and torch.equal(got_d, want_d), (wgs, mode)
      assert n0[3] != 0
  finally:
    sim.mnr_gemm_nt_set_wres(1)


@pytest.mark.parametrize('mode', MODES[:2])
def test_nt_small_tile_heads(sim, mode):
  """128x128 tile (N not a multiple of 256): partial-width bf16 output, bf16 mask operand, no bias."""
  g = torch.Generator().manual_seed(6)
  M, N, K = 384, 128, 320
  A = torch.randn((M, K), generator=g).bfloat16()
  Bt = (torch.randn((N, K), generator=g) * 0.1).bfloat16()
  mask = torch.randn((M, N), generator=g).bfloat16()
  sim.hipsim_reset(*mode)
  Cb, _, _ = S.sim_gemm_nt(sim, A, Bt, mask=mask, nb=96)
  ref = (A.float() @ Bt.float().T) * (mask.float() > 0)
  np.testing.assert_allclose(Cb.float().numpy(), ref[:, :96].numpy(), atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize('shape', [(512, 256, 256), (1024, 128, 384), (320, 128, 128)])
@pytest.mark.parametrize('mode', MODES)
def test_tn_weight_gradient(sim, shape, mode):
  """dW += X^T dY and db += column sums (the all-ones MFMA), both tile sizes, M not a multiple of the split."""
  M, K, N = shape
  g = torch.Generator().manual_seed(M)
  A = torch.randn((M, K), generator=g).bfloat16()
  B = torch.randn((M, N), generator=g).bfloat16()
  acc = torch.ones((K, N))
  db = torch.zeros(N)
  sim.hipsim_reset(*mode)
  S.sim_gemm_tn(sim, A, B, acc, bias_out=db, k_valid=K - 3, n_valid=N - 5)
  ref = 1.0 + A.float().T @ B.float()
  ref[K - 3:, :] = 1.0
  ref[:, N - 5:] = 1.0
  np.testing.assert_allclose(acc.numpy(), ref.numpy(), atol=1e-3, rtol=1e-5)
  np.testing.assert_allclose(db.numpy(), B.float().sum(0).numpy(), atol=1e-3, rtol=1e-5)


def test_small_head_bwd_and_colsum(sim):
  """The non-MFMA kernels of gemm.hip: rgb-head VJP (dX with bit masks, dW / db through workgroup partials), colsum."""
  g = torch.Generator().manual_seed(8)
  M, K, Cn = 1000, 128, 3
  H = torch.randn((M, K), generator=g).bfloat16()
  gr = torch.randn((M, Cn), generator=g)
  W = torch.randn((K, Cn), generator=g)
  keep = torch.rand((M, K), generator=g) > 0.3
  bits = _packbits(keep.float())
  dX = torch.zeros((M, K), dtype=torch.bfloat16)
  dW, db = torch.zeros((K, Cn)), torch.zeros(Cn)
  scratch = torch.zeros(1 << 16)
  sim.hipsim_reset(0, 0)
  S.sim_check(sim, sim.mnr_small_head_bwd(M, K, Cn, S.ptr(H), K, S.ptr(gr), S.ptr(W), S.ptr(dX), K, 0, S.ptr(dW), S.ptr(db),
                                            S.ptr(bits), bits.stride(0), 0, S.ptr(scratch), scratch.numel(), None))
  np.testing.assert_allclose(dX.float().numpy(), ((gr @ W.T) * keep).numpy(), atol=2e-2, rtol=1e-2)
  np.testing.assert_allclose(dW.numpy(), (H.float().T @ gr).numpy(), atol=1e-3, rtol=1e-4)
  np.testing.assert_allclose(db.numpy(), gr.sum(0).numpy(), atol=1e-3, rtol=1e-4)
  out = torch.zeros(K)
  S.sim_check(sim, sim.mnr_colsum_bf16(S.ptr(H), K, M, K - 2, S.ptr(out), None))
  np.testing.assert_allclose(out[:K - 2].numpy(), H.float().sum(0)[:K - 2].numpy(), atol=1e-3, rtol=1e-4)
  assert float(out[K - 2:].abs().max()) == 0.0


# ----------------------------------------------------------------------------- the simulator catches what it claims to


def _selftest(sim, bug, mode, rounds=4):
  src = torch.randn(rounds * 1024, generator=torch.Generator().manual_seed(1))
  s = src.view(rounds, 4, 64, 4)
  want = torch.stack([s[:, (t // 64 + 1) % 4, t % 64, 0].sum() + s[:, (t // 64 + 1) % 4, t % 64, 3].sum() for t in range(256)])
  dst = torch.zeros(256)
  sim.hipsim_reset(*mode)
  sim.hipsim_selftest(bug, S.ptr(src), S.ptr(dst), rounds, 1)
  failed = bool(sim.hipsim_failed())
  msg = sim.hipsim_error().decode()
  sim.hipsim_reset(0, 0)
  return bool(torch.allclose(dst, want, atol=1e-5)) and not failed, msg


def test_simulator_detects_seeded_synchronisation_bugs(sim):
  """tools/hipsim/selftest.hip: the correct kernel passes in every mode; a read without the vmcnt wait shows with DMA
  landing late; a missing barrier and an early restage show through run-ahead scheduling; a lane-varying LDS-DMA base
  is flagged."""
  for mode in MODES:
    assert _selftest(sim, 0, mode)[0]
  assert _selftest(sim, 1, (0, 0))[0]                  # invisible while DMA lands at issue ...
  assert not _selftest(sim, 1, (1, 0))[0]              # ... stale LDS when it lands at the wait
  for mode in MODES:
    assert not _selftest(sim, 2, mode)[0]
    assert not _selftest(sim, 3, mode)[0]
    ok, msg = _selftest(sim, 4, mode)
    assert not ok and 'wave-uniform' in msg


def test_simulator_flags_lds_overflow_and_bad_arguments(sim):
  """The C ABI's argument checks run unchanged on the host build."""
  A = torch.zeros((100, 64), dtype=torch.bfloat16)
  Bt = torch.zeros((128, 64), dtype=torch.bfloat16)
  sim.hipsim_reset(0, 0)
  with pytest.raises(RuntimeError, match='multiple of 128'):
    S.sim_gemm_nt(sim, A, Bt)


# ---- panel layout (include/mnerf.h MNR_LAYOUT_PANEL): csrc/gemm_blk.hip and the panel operands of the tiled kernels ----

from multinerf_amd import ops as _ops  # noqa: E402  (layout helpers only: pure torch)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('a1_panel', [0, 1])
def test_nt_panel_kernel_forward_and_dx(sim, a1_panel, mode):
  """gemm_nt_panel_kernel: one LDS-DMA pipeline across the tiles a persistent workgroup walks, the epilogue of tile t
  interleaved with the first k-step of tile t + 1, results / masks in panel / tile order: BITWISE the tiled kernel's result after
  un-blocking (forward with a skip segment, bias, ReLU, masks; dX with the masks it wrote), with LDS-DMA landing late / early and
  different wave orders, 8 and 16 workgroups over 12 tiles (walks of one and two tiles, padded M-tile slots)."""
  g = torch.Generator().manual_seed(3)
  M, N, K1, K2 = 768, 512, 192, 64
  A1 = torch.randn((M, K1), generator=g).bfloat16()
  A2 = torch.randn((M, K2), generator=g).bfloat16()
  Bt = (torch.randn((N, K1 + K2), generator=g) * 0.1).bfloat16()
  bias = torch.randn(N, generator=g)
  dY = torch.randn((M, 256), generator=g).bfloat16()
  Wt = (torch.randn((N, 256), generator=g) * 0.1).bfloat16()
  sim.mnr_gemm_nt_set_wres(0)
  try:
    sim.hipsim_reset(*mode)
    C0, _, b0 = S.sim_gemm_nt(sim, A1, Bt, A2=A2, bias=bias, relu=True, bits_out=True)
    sim.hipsim_reset(*mode)
    D0, _, _ = S.sim_gemm_nt(sim, dY, Wt, bits_in=b0)
    for wgs in (8, 16):
      sim.mnr_gemm_nt_panel_set_max_wgs(wgs)
      sim.hipsim_reset(*mode)
      Cp, _, bp = S.sim_gemm_nt(sim, _ops.to_panel(A1) if a1_panel else A1, Bt, A2=A2, bias=bias, relu=True, bits_out=True,
                                a1_layout=a1_panel, c_layout=1)
      assert torch.equal(_ops.from_panel(Cp).view(torch.int16), C0.view(torch.int16))
      assert torch.equal(_ops.bits_from_tile_order(bp.view(-1), M, N), b0)
      sim.hipsim_reset(*mode)
      # (the second launch of each pair walks the M-tiles in descending order: mnr_gemm_nt_args.walk_descending)
      Dp, _, _ = S.sim_gemm_nt(sim, _ops.to_panel(dY) if a1_panel else dY, Wt, bits_in=bp.view(-1), a1_layout=a1_panel, c_layout=1,
                               walk_descending=True)
      assert torch.equal(_ops.from_panel(Dp).view(torch.int16), D0.view(torch.int16))
  finally:
    sim.mnr_gemm_nt_panel_set_max_wgs(0)
    sim.mnr_gemm_nt_set_wres(1)
  ref = torch.relu(torch.cat([A1, A2], 1).float() @ Bt.float().T + bias)
  np.testing.assert_allclose(C0.float().numpy(), ref.numpy(), atol=3e-2, rtol=1e-2)


@pytest.mark.parametrize('mode', [MODES[1], MODES[3]])
def test_nt_vector_column_next_to_a_256_wide_result(sim, mode):
  """mnr_gemm_nt_args.vcol (the density head next to the bottleneck, models.py:460 / :527): one more output column supplied as a
  vector, computed by one extra MFMA per wave and k-step on row blocks taken in rotated order.  The 256-wide bf16 result must be
  bitwise the plain launch's, the vector column bitwise the fp32 side column of the merged 257-column operand (the same products
  in the same order), with two activation segments, several tiles per workgroup, late DMA and shuffled wave orders."""
  g = torch.Generator().manual_seed(6)
  M, K1, K2 = 4096, 192, 64                        # 16 tiles on 8 workgroups: every workgroup walks two
  X = torch.relu(torch.randn((M, K1), generator=g)).bfloat16()
  X2 = torch.randn((M, K2), generator=g).bfloat16()
  Bt = (torch.randn((512, K1 + K2), generator=g) * 0.1).bfloat16()
  Bt[257:] = 0
  bias = torch.randn(257, generator=g)
  sim.mnr_gemm_nt_set_persistent(-8)
  try:
    sim.hipsim_reset(*mode)
    C0, F0, _ = S.sim_gemm_nt(sim, _ops.to_panel(X), Bt, A2=X2, bias=bias, nb=256, out_f32=(256, 1), a1_layout=1)
    sim.hipsim_reset(*mode)
    C1, v1, _ = S.sim_gemm_nt(sim, _ops.to_panel(X), Bt[:256].contiguous(), A2=X2, bias=bias[:256].contiguous(), a1_layout=1,
                              vcol=Bt[256].contiguous(), vcol_bias=bias[256:257].contiguous())
  finally:
    sim.mnr_gemm_nt_set_persistent(1)
  assert torch.equal(C1.view(torch.int16), C0[:, :256].view(torch.int16))
  assert torch.equal(v1, F0[:, 0])
  ref = torch.cat([X, X2], 1).float() @ Bt[256].float() + bias[256]
  np.testing.assert_allclose(v1.numpy(), ref.numpy(), atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize('mode', [MODES[1], MODES[3]])
def test_nt_tiled_kernel_reads_a_panel_activation(sim, mode):
  """The merged head behind a panel-layout trunk: the pipelined tiled kernel with A1 in panel storage, row-major bf16 result
  narrower than N plus an fp32 side column: bitwise the row-major call."""
  g = torch.Generator().manual_seed(4)
  M, N, K = 512, 512, 256
  X = torch.randn((M, K), generator=g).bfloat16()
  Bt = (torch.randn((N, K), generator=g) * 0.1).bfloat16()
  bias = torch.randn(257, generator=g)
  sim.hipsim_reset(*mode)
  C0, F0, _ = S.sim_gemm_nt(sim, X, Bt, bias=bias, nb=256, out_f32=(256, 1))
  sim.hipsim_reset(*mode)
  C1, F1, _ = S.sim_gemm_nt(sim, _ops.to_panel(X), Bt, bias=bias, nb=256, out_f32=(256, 1), a1_layout=1)
  assert torch.equal(C0.view(torch.int16), C1.view(torch.int16)) and torch.equal(F0, F1)


@pytest.mark.parametrize('mode', [MODES[1], MODES[2]])
@pytest.mark.parametrize('ap,bp', [(1, 1), (0, 1), (1, 0)])
def test_tn_panel_operands(sim, ap, bp, mode):
  """Weight gradients from panel-layout activations / gradients (gemm_tn_body.inc A_PANEL / B_PANEL: a 32-row stage of a panel
  operand is 16 consecutive 1-KiB blocks, padded by 128 bytes per block in LDS): the row-major call's sums, the bias gradient,
  and with (panel A, row-major B) the extra vector column of the merged head."""
  g = torch.Generator().manual_seed(6)
  M, K, N = 512, 512, 256
  A = torch.randn((M, K), generator=g).bfloat16()
  B = torch.randn((M, N), generator=g).bfloat16()
  gcol = torch.randn((M,), generator=g).bfloat16() if (ap and not bp) else None
  C0 = torch.zeros((K, N))
  C1 = torch.zeros((K, N))
  db0, db1 = torch.zeros(N), torch.zeros(N)
  g0, g1 = (torch.zeros(K), torch.zeros(K)) if gcol is not None else (None, None)
  sim.hipsim_reset(*mode)
  S.sim_gemm_tn(sim, A, B, C0, bias_out=db0, gcol=gcol, gcol_out=g0)
  sim.hipsim_reset(*mode)
  S.sim_gemm_tn(sim, _ops.to_panel(A) if ap else A, _ops.to_panel(B) if bp else B, C1, bias_out=db1, a_layout=ap, b_layout=bp,
                gcol=gcol, gcol_out=g1)
  ref = A.float().T @ B.float()
  np.testing.assert_allclose(C1.numpy(), ref.numpy(), atol=2e-3, rtol=1e-4)
  np.testing.assert_allclose(C1.numpy(), C0.numpy(), atol=1e-3, rtol=1e-5)      # (the atomics' order differs between launches)
  np.testing.assert_allclose(db1.numpy(), B.float().sum(0).numpy(), atol=2e-3, rtol=1e-4)
  if gcol is not None:
    np.testing.assert_allclose(g1.numpy(), (A.float().T @ gcol.float()).numpy(), atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(g1.numpy(), g0.numpy(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize('layouts', [(0, 0), (1, 1), (1, 0)])
def test_tn_block_cyclic_m_splits(sim, layouts):
  """mnr_gemm_tn_args.m_interleave / max_wgs (the dW launch of a paired dX + dW, models._PAIR_DXDW): the M-splits take the
  256-row M-tiles block-cyclically instead of contiguously -- the same sums over rows in another order; row-major and panel
  operands, the bias gradient and the extra column along."""
  ap, bp = layouts
  g = torch.Generator().manual_seed(11)
  M, K, N = 2048, 256, 512
  A = torch.randn((M, K), generator=g).bfloat16()
  B = torch.randn((M, N), generator=g).bfloat16()
  gcol = torch.randn(M, generator=g).bfloat16() if (ap, bp) == (1, 0) else None
  ref, refb = A.float().T @ B.float(), B.float().sum(0)
  sim.hipsim_reset(1, 3)
  for il, cap in ((False, 0), (True, 8), (True, 16)):
    C, b = torch.zeros((K, N)), torch.zeros(N)
    gout = torch.zeros(K) if gcol is not None else None
    S.sim_gemm_tn(sim, _ops.to_panel(A) if ap else A, _ops.to_panel(B) if bp else B, C, bias_out=b, a_layout=ap, b_layout=bp,
                  gcol=gcol, gcol_out=gout, m_interleave=il, max_wgs=cap)
    np.testing.assert_allclose(C.numpy(), ref.numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(b.numpy(), refb.numpy(), rtol=1e-5, atol=1e-3)
    if gcol is not None:
      np.testing.assert_allclose(gout.numpy(), (A.float().T @ gcol.float()).numpy(), rtol=1e-4, atol=1e-2)
  # 3 M-tiles do not divide among the 8 splits the launcher picks for this shape: the flag is performance only and the launch
  # runs with contiguous splits (the split count depends on the width and the CU count, a caller cannot gate on it)
  C3 = torch.zeros((K, N))
  S.sim_gemm_tn(sim, A[:768], B[:768], C3, m_interleave=True, max_wgs=16)
  np.testing.assert_allclose(C3.numpy(), (A[:768].float().T @ B[:768].float()).numpy(), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('late_dma', [0, 1])
def test_tn_rank1_b_operand_built_in_the_kernel(sim, late_dma):
  """mnr_gemm_tn_args.rank1_*: B[m, n] = bit ? bf16(g[m] w[n]) : 0 expanded in LDS from the factors (the proposal MLP's last dY,
  models.py:460 behind the ReLU of :457) must give what the stored matrix gives, value for value: same B image, same order of
  the sums.  Several splits, two n-tiles, K = 512 (two k-tiles share an n-tile), splits of 1-3 M-tiles (the prologue / tail
  variants of the stage loop), the DMA data arriving as late as the counted waits allow."""
  g = torch.Generator().manual_seed(23)
  for M, K, N, cap in ((2048, 256, 256, 0), (1536, 512, 512, 16), (256, 256, 256, 8), (4096, 256, 256, 8)):
    A = torch.randn((M, K), generator=g).bfloat16()
    gh = torch.randn(M, generator=g) * 0.1
    w = torch.randn(N, generator=g)
    bits = torch.randint(0, 256, (M, N // 8 + 4), generator=g, dtype=torch.uint8)            # (a pitch wider than the tile)
    bit = ((bits[:, :N // 8, None].int() >> torch.arange(8)) & 1).reshape(M, N).bool()
    B = torch.where(bit, gh[:, None] * w[None, :], torch.zeros(())).bfloat16()
    sim.hipsim_reset(late_dma, 5)
    C0, b0 = torch.zeros((K, N)), torch.zeros(N)
    S.sim_gemm_tn(sim, A, B, C0, bias_out=b0, max_wgs=cap, m_interleave=(M == 4096))
    C1, b1 = torch.zeros((K, N)), torch.zeros(N)
    S.sim_gemm_tn(sim, A, None, C1, bias_out=b1, max_wgs=cap, m_interleave=(M == 4096), rank1=(gh, w, bits))
    np.testing.assert_allclose(C0.numpy(), (A.float().T @ B.float()).numpy(), rtol=1e-5, atol=2e-3)
    assert torch.equal(C1, C0) and torch.equal(b1, b0), (M, K, N, (C1 - C0).abs().max().item())
