"""Torch-tensor front ends for the C ABI (include/mnerf.h).

torch is plumbing here: it owns device memory and the stream; all arithmetic on
the hot path happens inside libmnerf_hip.so.  Every function validates device,
dtype and contiguity and then passes raw device pointers.
"""

import ctypes as C
import os

import torch

from multinerf_amd import _lib as L

bf16 = torch.bfloat16
f32 = torch.float32


def _ptr(t):
  return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_device(t):
  return t.is_cuda


def act_dtype():
  """Storage type of the Dense layers' activation / gradient / packed-weight matrices: bf16, or fp32 inside
  `_lib.dense_f32()` (the fp32-Dense debug build, Model(dense_precision='fp32'))."""
  return f32 if L.f32_active() else bf16


def _chk(t, dtype, name, allow_none=False):
  if dtype is bf16 and L.f32_active():
    dtype = f32
  if t is None:
    if allow_none:
      return
    raise ValueError(f'{name} is None')
  if not _on_device(t):
    raise ValueError(f'{name} must be a device tensor (the HIP path has no CPU fallback)')
  if t.dtype != dtype:
    raise ValueError(f'{name} must be {dtype}, is {t.dtype}')
  if not t.is_contiguous():
    raise ValueError(f'{name} must be contiguous')


def lib():
  """The loaded C-ABI library (include/mnerf.h).  No process-global switches are applied: the development hooks of
  include/mnerf_debug.h are reached through multinerf_amd._lib.debug() by tests and tools only."""
  return L.load()


class _GemmProfile:
  """Optional HIP-event timing of kernel launches on the launch stream (bench.py roofline): GEMMs under the tag
  'gemm', the bandwidth-bound kernels under their own tags together with their algorithmic HBM bytes."""

  def __init__(self):
    self.on = False
    self.events = []

  def enable(self):
    self.on, self.events = True, []
    self.base = torch.cuda.Event(enable_timing=True)     # time origin for the busy-time union across streams
    self.base.record(torch.cuda.current_stream())

  def disable(self):
    self.on = False

  def start(self):
    if not self.on:
      return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    return e0

  def stop(self, e0, tag='gemm', nbytes=0):
    if e0 is None:
      return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record(torch.cuda.current_stream())
    self.events.append((e0, e1, tag, nbytes))

  def collect(self):
    """-> (total GEMM ms, GEMM launches); call after a synchronize.  Other tags: collect_by_tag() first."""
    by = self.collect_by_tag()
    g = by.get('gemm', dict(ms=0.0, launches=0))
    return g['ms'], g['launches']

  def collect_by_tag(self):
    """-> {tag: {'ms', 'launches', 'bytes'}}; empties the event list."""
    out, spans = {}, {}
    for a, b, tag, nbytes in self.events:
      d = out.setdefault(tag, dict(ms=0.0, launches=0, bytes=0))
      d['ms'] += a.elapsed_time(b)
      d['launches'] += 1
      d['bytes'] += nbytes
      spans.setdefault(tag, []).append((self.base.elapsed_time(a), self.base.elapsed_time(b)))
    # 'busy_ms': length of the UNION of the tag's launch intervals (= 'ms' when the launches are serial on one stream;
    # less when two streams run kernels of the tag side by side, multinerf_amd/streams.py)
    for tag, iv in spans.items():
      iv.sort()
      busy, end = 0.0, None
      for a0, b0 in iv:
        if end is None or a0 > end:
          busy += b0 - a0
          end = b0
        elif b0 > end:
          busy += b0 - end
          end = b0
      out[tag]['busy_ms'] = busy
    self.events = []
    return out


PROFILE = _GemmProfile()


# ----------------------------------------------------------------------------- sampling


def resample_level(sdist_prev, w_prev, u_base, jitter, near, far, *, n_samples, use_dilation,
                   dilation, domain, anneal, resample_padding, single_jitter, max_jitter,
                   raydist_fn, want_idx=False):
  for t, nm in ((sdist_prev, 'sdist_prev'), (w_prev, 'w_prev'), (u_base, 'u_base'),
                (near, 'near'), (far, 'far')):
    _chk(t, f32, nm)
  _chk(jitter, f32, 'jitter', allow_none=True)
  B, n_prev = w_prev.shape
  assert sdist_prev.shape == (B, n_prev + 1) and u_base.numel() == n_samples
  assert near.numel() == B and far.numel() == B
  if jitter is not None:
    assert jitter.numel() == (B if single_jitter else B * n_samples)
  cfg = L.ResampleCfg(n_prev, n_samples, int(use_dilation), float(dilation), float(domain[0]),
                      float(domain[1]), float(anneal), float(resample_padding), int(single_jitter),
                      float(max_jitter), L.RAYDIST[raydist_fn])
  dev = w_prev.device
  sdist = torch.empty((B, n_samples + 1), dtype=f32, device=dev)
  tdist = torch.empty((B, n_samples + 1), dtype=f32, device=dev)
  idx = torch.empty((B, n_samples), dtype=torch.int32, device=dev) if want_idx else None
  _e = PROFILE.start()
  L.check(lib().mnr_resample_level(C.byref(cfg), B, _ptr(sdist_prev), _ptr(w_prev), _ptr(u_base),
                                   _ptr(jitter), _ptr(near), _ptr(far), _ptr(sdist), _ptr(tdist),
                                   _ptr(idx), _stream()))
  PROFILE.stop(_e, 'resample', 4 * (sdist_prev.numel() + w_prev.numel() + 2 * sdist_prev.shape[0] * (n_samples + 1)))
  return (sdist, tdist, idx) if want_idx else (sdist, tdist)


def resample_level_bwd(sdist_prev, w_prev, u_base, jitter, g_sdist, *, n_samples, use_dilation, dilation, domain, anneal,
                       resample_padding, single_jitter, max_jitter, raydist_fn, g_sdist_prev=None, g_w_prev=None):
  """VJP of `resample_level` w.r.t. (sdist_prev, w_prev) given g_sdist = d loss / d sdist (Model.stop_level_grad = False)."""
  for t, nm in ((sdist_prev, 'sdist_prev'), (w_prev, 'w_prev'), (u_base, 'u_base'), (g_sdist, 'g_sdist')):
    _chk(t, f32, nm)
  _chk(jitter, f32, 'jitter', allow_none=True)
  B, n_prev = w_prev.shape
  assert sdist_prev.shape == (B, n_prev + 1) and u_base.numel() == n_samples and g_sdist.shape == (B, n_samples + 1)
  if jitter is not None:
    assert jitter.numel() == (B if single_jitter else B * n_samples)
  cfg = L.ResampleCfg(n_prev, n_samples, int(use_dilation), float(dilation), float(domain[0]),
                      float(domain[1]), float(anneal), float(resample_padding), int(single_jitter),
                      float(max_jitter), L.RAYDIST[raydist_fn])
  dev = w_prev.device
  if g_sdist_prev is None:
    g_sdist_prev = torch.empty((B, n_prev + 1), dtype=f32, device=dev)
  if g_w_prev is None:
    g_w_prev = torch.empty((B, n_prev), dtype=f32, device=dev)
  _chk(g_sdist_prev, f32, 'g_sdist_prev')
  _chk(g_w_prev, f32, 'g_w_prev')
  assert g_sdist_prev.shape == (B, n_prev + 1) and g_w_prev.shape == (B, n_prev)
  L.check(lib().mnr_resample_level_bwd(C.byref(cfg), B, _ptr(sdist_prev), _ptr(w_prev), _ptr(u_base), _ptr(jitter),
                                       _ptr(g_sdist), _ptr(g_sdist_prev), _ptr(g_w_prev), _stream()))
  return g_sdist_prev, g_w_prev


def sorted_interp(u, cw, t):
  for x, nm in ((u, 'u'), (cw, 'cw'), (t, 't')):
    _chk(x, f32, nm)
  B, nc = cw.shape
  nu = u.shape[1]
  out = torch.empty_like(u)
  idx = torch.empty(u.shape, dtype=torch.int32, device=u.device)
  L.check(lib().mnr_sorted_interp(B, nc, nu, _ptr(u), _ptr(cw), _ptr(t), _ptr(out), _ptr(idx), _stream()))
  return out, idx


def max_dilate_weights(t, w, dilation, domain):
  _chk(t, f32, 't')
  _chk(w, f32, 'w')
  B, n = w.shape
  t_out = torch.empty((B, 3 * n + 1), dtype=f32, device=w.device)
  w_out = torch.empty((B, 3 * n), dtype=f32, device=w.device)
  scratch = torch.empty((B, n), dtype=f32, device=w.device)
  L.check(lib().mnr_max_dilate_weights(B, n, _ptr(t), _ptr(w), float(dilation), float(domain[0]),
                                       float(domain[1]), _ptr(t_out), _ptr(w_out), _ptr(scratch), _stream()))
  return t_out, w_out


# ----------------------------------------------------------------------------- features


def _ipe_cfg(ray_shape, warp_contract, disable_integration, basis, min_deg, max_deg):
  if ray_shape not in ('cone', 'cylinder'):
    raise ValueError('ray_shape must be \'cone\' or \'cylinder\'')   # render.py:124
  return L.IpeCfg(0 if ray_shape == 'cone' else 1, int(bool(warp_contract)), int(bool(disable_integration)),
                  basis.shape[0], int(min_deg), int(max_deg))


def cast_rays_ipe(tdist, origins, directions, radii, basis, *, ray_shape, warp_contract, min_deg,
                  max_deg, ld_feat, disable_integration=False, out=None, want_gaussians=False):
  """-> bf16 features [B*n, ld_feat] (+ optional post-warp means [B*n,3], covs [B*n,9])."""
  for x, nm in ((tdist, 'tdist'), (origins, 'origins'), (directions, 'directions'), (radii, 'radii'),
                (basis, 'basis')):
    _chk(x, f32, nm)
  B, n1 = tdist.shape
  n = n1 - 1
  cfg = _ipe_cfg(ray_shape, warp_contract, disable_integration, basis, min_deg, max_deg)
  dev = tdist.device
  if out is None:
    out = torch.empty((B * n, ld_feat), dtype=act_dtype(), device=dev)
  _chk(out, bf16, 'out')
  assert out.shape == (B * n, ld_feat)
  means = covs = None
  if want_gaussians:
    means = torch.empty((B * n, 3), dtype=f32, device=dev)
    covs = torch.empty((B * n, 9), dtype=f32, device=dev)
  _e = PROFILE.start()
  L.check(lib().mnr_cast_rays_ipe(C.byref(cfg), B, n, _ptr(tdist), _ptr(origins), _ptr(directions),
                                  _ptr(radii), _ptr(basis), _ptr(out), ld_feat, _ptr(means), _ptr(covs),
                                  _stream()))
  PROFILE.stop(_e, 'ipe', 4 * tdist.numel() + 2 * out.numel())
  return (out, means, covs) if want_gaussians else out


def cast_rays_ipe_f32(tdist, origins, directions, radii, basis, *, ray_shape, warp_contract, min_deg,
                      max_deg, disable_integration=False):
  for x, nm in ((tdist, 'tdist'), (origins, 'origins'), (directions, 'directions'), (radii, 'radii'),
                (basis, 'basis')):
    _chk(x, f32, nm)
  B, n1 = tdist.shape
  n = n1 - 1
  cfg = _ipe_cfg(ray_shape, warp_contract, disable_integration, basis, min_deg, max_deg)
  nfeat = 2 * basis.shape[0] * (max_deg - min_deg)
  out = torch.empty((B * n, nfeat), dtype=f32, device=tdist.device)
  L.check(lib().mnr_cast_rays_ipe_f32(C.byref(cfg), B, n, _ptr(tdist), _ptr(origins), _ptr(directions),
                                      _ptr(radii), _ptr(basis), _ptr(out), _stream()))
  return out


def cast_rays_ipe_tangent(tdist, origins, directions, radii, basis, *, ray_shape, min_deg, max_deg, ld_feat,
                          out=None, warp_contract=False, disable_integration=False):
  """-> bf16 [3*B*n, ld_feat]: rows c*B*n + s = d(features of sample s)/d(mean_c), mean = the pre-warp mean."""
  for x, nm in ((tdist, 'tdist'), (origins, 'origins'), (directions, 'directions'), (radii, 'radii'),
                (basis, 'basis')):
    _chk(x, f32, nm)
  B, n1 = tdist.shape
  n = n1 - 1
  cfg = _ipe_cfg(ray_shape, warp_contract, disable_integration, basis, min_deg, max_deg)
  if out is None:
    out = torch.empty((3 * B * n, ld_feat), dtype=act_dtype(), device=tdist.device)
  _chk(out, bf16, 'out')
  L.check(lib().mnr_cast_rays_ipe_tangent(C.byref(cfg), B, n, _ptr(tdist), _ptr(origins), _ptr(directions),
                                          _ptr(radii), _ptr(basis), _ptr(out), ld_feat, _stream()))
  return out


def cast_rays_ipe_bwd(tdist, origins, directions, radii, basis, g_feat_a, g_feat_b=None, *, ray_shape, warp_contract, min_deg,
                      max_deg, disable_integration=False, g_t0=None, g_t1=None):
  """VJP of `cast_rays_ipe` w.r.t. the interval ends: g_feat_a (+ g_feat_b) bf16 [B*n, ld] -> (g_t0, g_t1) fp32 [B*n]."""
  for x, nm in ((tdist, 'tdist'), (origins, 'origins'), (directions, 'directions'), (radii, 'radii'), (basis, 'basis')):
    _chk(x, f32, nm)
  _chk(g_feat_a, bf16, 'g_feat_a')
  _chk(g_feat_b, bf16, 'g_feat_b', allow_none=True)
  B, n1 = tdist.shape
  n = n1 - 1
  ld = g_feat_a.stride(0)
  assert g_feat_a.shape[0] == B * n and (g_feat_b is None or (g_feat_b.shape[0] == B * n and g_feat_b.stride(0) == ld))
  cfg = _ipe_cfg(ray_shape, warp_contract, disable_integration, basis, min_deg, max_deg)
  dev = tdist.device
  if g_t0 is None:
    g_t0 = torch.empty((B * n,), dtype=f32, device=dev)
  if g_t1 is None:
    g_t1 = torch.empty((B * n,), dtype=f32, device=dev)
  _chk(g_t0, f32, 'g_t0')
  _chk(g_t1, f32, 'g_t1')
  assert g_t0.numel() == B * n and g_t1.numel() == B * n
  L.check(lib().mnr_cast_rays_ipe_bwd(C.byref(cfg), B, n, _ptr(tdist), _ptr(origins), _ptr(directions), _ptr(radii), _ptr(basis),
                                      _ptr(g_feat_a), _ptr(g_feat_b), ld, _ptr(g_t0), _ptr(g_t1), _stream()))
  return g_t0, g_t1


def cast_rays_ipe_tangent_bwd(tdist, origins, directions, radii, basis, g_T_a, g_T_b, g_t0, g_t1, *, ray_shape, warp_contract, min_deg,
                              max_deg, disable_integration=False):
  """VJP of `cast_rays_ipe_tangent` w.r.t. the interval ends: g_T_a (+ g_T_b) bf16 [3*B*n, ld] -> g_t0, g_t1 fp32 [B*n] += ..."""
  for x, nm in ((tdist, 'tdist'), (origins, 'origins'), (directions, 'directions'), (radii, 'radii'), (basis, 'basis'),
                (g_t0, 'g_t0'), (g_t1, 'g_t1')):
    _chk(x, f32, nm)
  _chk(g_T_a, bf16, 'g_T_a')
  _chk(g_T_b, bf16, 'g_T_b', allow_none=True)
  B, n1 = tdist.shape
  n = n1 - 1
  ld = g_T_a.stride(0)
  assert g_T_a.shape[0] == 3 * B * n and (g_T_b is None or (g_T_b.shape[0] == 3 * B * n and g_T_b.stride(0) == ld))
  assert g_t0.numel() == B * n and g_t1.numel() == B * n
  cfg = _ipe_cfg(ray_shape, warp_contract, disable_integration, basis, min_deg, max_deg)
  L.check(lib().mnr_cast_rays_ipe_tangent_bwd(C.byref(cfg), B, n, _ptr(tdist), _ptr(origins), _ptr(directions), _ptr(radii),
                                              _ptr(basis), _ptr(g_T_a), _ptr(g_T_b), ld, _ptr(g_t0), _ptr(g_t1), _stream()))
  return g_t0, g_t1


def sdist_bwd(sdist, near, far, raydist_fn, *, B_valid=None, g_x=None, raw_density=None, density_noise=None, density_noise_std=0.0,
              density_bias=0.0, density_act='softplus', dirs=None, g_t0=None, g_t1=None, distortion_mult=0.0, weights=None,
              g_sdist_in=None, out=None):
  """d loss / d sdist [B, n+1] of one level (Model.stop_level_grad = False): see mnr_sdist_bwd in include/mnerf.h."""
  _chk(sdist, f32, 'sdist')
  B, n1 = sdist.shape
  n = n1 - 1
  for x, nm in ((near, 'near'), (far, 'far')):
    _chk(x, f32, nm)
  for x, nm in ((g_x, 'g_x'), (raw_density, 'raw_density'), (density_noise, 'density_noise'), (dirs, 'dirs'), (g_t0, 'g_t0'),
                (g_t1, 'g_t1'), (weights, 'weights'), (g_sdist_in, 'g_sdist_in')):
    _chk(x, f32, nm, allow_none=True)
  assert near.numel() == B and far.numel() == B
  assert g_x is None or (g_x.numel() == B * n and raw_density is not None and raw_density.numel() == B * n and dirs is not None)
  assert g_sdist_in is None or g_sdist_in.shape == (B, n1)
  if out is None:
    out = torch.empty((B, n1), dtype=f32, device=sdist.device)
  _chk(out, f32, 'out')
  a = L.SdistBwdArgs()
  a.B, a.B_valid, a.n = B, (B if B_valid is None else int(B_valid)), n
  a.sdist, a.near, a.far, a.raydist_fn = _ptr(sdist), _ptr(near), _ptr(far), L.RAYDIST[raydist_fn]
  a.g_x, a.raw_density, a.density_noise = _ptr(g_x), _ptr(raw_density), _ptr(density_noise)
  a.density_noise_std, a.density_bias, a.density_act = float(density_noise_std), float(density_bias), L.ACT[density_act]
  a.dirs, a.g_t0, a.g_t1 = _ptr(dirs), _ptr(g_t0), _ptr(g_t1)
  a.distortion_mult, a.weights = float(distortion_mult), _ptr(weights)
  a.g_sdist_in, a.g_sdist = _ptr(g_sdist_in), _ptr(out)
  L.check(lib().mnr_sdist_bwd(C.byref(a), _stream()))
  return out


def viewdir_enc_fill(viewdirs, n, deg_view, dst, col0, col_end):
  _chk(viewdirs, f32, 'viewdirs')
  _chk(dst, bf16, 'dst')
  B = viewdirs.shape[0]
  assert dst.shape[0] == B * n
  L.check(lib().mnr_viewdir_enc_fill(B, n, _ptr(viewdirs), deg_view, _ptr(dst), dst.stride(0), col0, col_end,
                                     _stream()))


# ----------------------------------------------------------------------------- dense


def glo_fill(table, cam_idx, B, n, dst, col0):
  """cam_idx None == zero_glo."""
  _chk(table, f32, 'table')
  _chk(cam_idx, torch.int32, 'cam_idx', allow_none=True)
  _chk(dst, bf16, 'dst')
  E, G = table.shape
  L.check(lib().mnr_glo_fill(B, n, G, _ptr(table), _ptr(cam_idx), E, _ptr(dst), dst.stride(0), col0, _stream()))


def glo_bwd(g_a, g_b, cam_idx, B, n, grad_table, num_embeddings, G):
  _chk(g_a, f32, 'g_a')
  _chk(g_b, f32, 'g_b', allow_none=True)
  _chk(cam_idx, torch.int32, 'cam_idx')
  _chk(grad_table, f32, 'grad_table')
  L.check(lib().mnr_glo_bwd(B, n, G, _ptr(g_a), _ptr(g_b), _ptr(cam_idx), num_embeddings, _ptr(grad_table), _stream()))


LAYOUT_ROWMAJOR, LAYOUT_PANEL = 0, 1      # include/mnerf.h MNR_LAYOUT_*


def to_panel(x):
  """Row-major [M, W] -> the same storage in MNR_LAYOUT_PANEL (include/mnerf.h): 1-KiB blocks of 32 rows x 16 columns,
  element (m, n) at ((m // 32) * (W // 16) + n // 16) * 512 + (m % 32) * 16 + n % 16.  Layout conversions are for tests and
  tools: on the product path every producer and consumer of a panel matrix is one of the library's own GEMMs."""
  M, W = x.shape
  assert M % 32 == 0 and W % 16 == 0
  return x.reshape(M // 32, 32, W // 16, 16).permute(0, 2, 1, 3).contiguous().view(M, W)


def from_panel(x):
  """Inverse of `to_panel`."""
  M, W = x.shape
  assert M % 32 == 0 and W % 16 == 0
  return x.reshape(M // 32, W // 16, 32, 16).permute(0, 2, 1, 3).contiguous().view(M, W)


def bits_to_tile_order(bits, N):
  """Row-major 1-bit masks [M, N/8] (bit n & 7 of byte [m, n // 8]) -> the TILE order of a panel-result GEMM (include/mnerf.h:
  8 KiB per 256 x 256 tile, 16 bytes per thread); returns a flat uint8 tensor of M * N / 8 bytes.  Tests / tools only."""
  M = bits.shape[0]
  assert M % 256 == 0 and N % 256 == 0 and bits.shape[1] == N // 8
  # row = (mt, wm, i, r), byte column = (nt, wn, j, h, kh)   ->   (mt, nt, wm, wn, kh, r, j, i, h)
  v = bits.reshape(M // 256, 2, 4, 32, N // 256, 4, 2, 2, 2)
  return v.permute(0, 4, 1, 5, 8, 3, 6, 2, 7).contiguous().view(-1)


def bits_from_tile_order(tile_bits, M, N):
  """Inverse of `bits_to_tile_order`: -> [M, N/8]."""
  v = tile_bits.reshape(M // 256, N // 256, 2, 4, 2, 32, 2, 4, 2)      # (mt, nt, wm, wn, kh, r, j, i, h)
  return v.permute(0, 2, 7, 5, 1, 3, 6, 8, 4).contiguous().view(M, N // 8)


def gemm_nt(A1, Bt, *, M, N, K1, A2=None, K2=0, lda1=None, lda2=None, ldb=None, bias=None, n_bias=0,
            relu=False, mask=None, ldmask=0, Cb=None, ldcb=0, nb=0, Cf=None, ldcf=0, f0=0, nf=0,
            bits_out=None, bits_in=None, bits_row_mod=0, a1_layout=LAYOUT_ROWMAJOR, c_layout=LAYOUT_ROWMAJOR,
            vcol=None, vcol_out=None, vcol_bias=None, walk_descending=False, max_wgs=0):
  """C[M,N] = epilogue([A1|A2] @ Bt^T).  Pointers may be views with explicit leading dimensions.  a1_layout / c_layout:
  LAYOUT_PANEL for the wide trunk's activations and gradients (include/mnerf.h; the bits are then in tile order)."""
  _chk(A1, bf16, 'A1')
  _chk(Bt, bf16, 'Bt')
  _chk(A2, bf16, 'A2', allow_none=True)
  _chk(mask, bf16, 'mask', allow_none=True)
  _chk(bias, f32, 'bias', allow_none=True)
  _chk(Cb, bf16, 'Cb', allow_none=True)
  _chk(Cf, f32, 'Cf', allow_none=True)
  a = L.GemmNTArgs()
  a.A1, a.lda1, a.K1 = A1.data_ptr(), lda1 if lda1 else A1.stride(0), K1
  a.A2, a.lda2, a.K2 = (A2.data_ptr() if A2 is not None else None), (lda2 if lda2 else (A2.stride(0) if A2 is not None else 0)), K2
  a.Bt, a.ldb = Bt.data_ptr(), ldb if ldb else Bt.stride(0)
  a.M, a.N = M, N
  a.bias, a.n_bias, a.relu = (bias.data_ptr() if bias is not None else None), n_bias, int(relu)
  a.mask, a.ldmask = (mask.data_ptr() if mask is not None else None), ldmask
  a.Cb, a.ldcb, a.nb = (Cb.data_ptr() if Cb is not None else None), ldcb, nb
  a.Cf, a.ldcf, a.f0, a.nf = (Cf.data_ptr() if Cf is not None else None), ldcf, f0, nf
  for t_, nm_ in ((bits_out, 'bits_out'), (bits_in, 'bits_in')):
    _chk(t_, torch.uint8, nm_, allow_none=True)
  a.mask_bits_out, a.ld_bits_out = (bits_out.data_ptr(), bits_out.stride(0)) if bits_out is not None else (None, 0)
  a.mask_bits_in, a.ld_bits_in = (bits_in.data_ptr(), bits_in.stride(0)) if bits_in is not None else (None, 0)
  a.bits_row_mod = bits_row_mod
  a.a1_layout, a.c_layout = a1_layout, c_layout
  a.walk_descending = int(bool(walk_descending))
  a.max_wgs = int(max_wgs)
  _chk(vcol, bf16, 'vcol', allow_none=True)
  _chk(vcol_out, f32, 'vcol_out', allow_none=True)
  assert (vcol is None) == (vcol_out is None)
  _chk(vcol_bias, f32, 'vcol_bias', allow_none=True)
  a.vcol, a.vcol_out = (vcol.data_ptr() if vcol is not None else None), (vcol_out.data_ptr() if vcol_out is not None else None)
  a.vcol_bias = vcol_bias.data_ptr() if vcol_bias is not None else None
  _e = PROFILE.start()
  L.check(lib().mnr_gemm_nt_bf16(C.byref(a), _stream()))
  PROFILE.stop(_e)


def gemm_tn(A, B, Cout, *, M, K, N, lda=None, ldb=None, ldc=None, k_valid=None, n_valid=None, bias_out=None,
            bias_n_valid=0, gcol=None, gcol_out=None, a_layout=LAYOUT_ROWMAJOR, b_layout=LAYOUT_ROWMAJOR, m_interleave=False,
            max_wgs=0, rank1=None):
  """Cout[k,n] += sum_m A[m,k] B[m,n]; optionally bias_out[n] += sum_m B[m,n] (fused bias gradient) and
  gcol_out[k] += sum_m A[m,k] gcol[m] (one more column of B given as a contiguous bf16 vector [M]).
  rank1 = (g [M] fp32, w [N] fp32, bits [M, N / 8] uint8) with B = None: B[m,n] = bit ? bf16(g[m] w[n]) : 0 is built inside
  the kernel (the proposal MLP's last dY, mnr_gemm_tn_args.rank1_*)."""
  _chk(bias_out, f32, 'bias_out', allow_none=True)
  _chk(gcol, bf16, 'gcol', allow_none=True)
  _chk(gcol_out, f32, 'gcol_out', allow_none=True)
  assert (gcol is None) == (gcol_out is None) and (gcol is None or (gcol.numel() == M and gcol.is_contiguous()))
  _chk(A, bf16, 'A')
  _chk(B, bf16, 'B', allow_none=rank1 is not None)
  _chk(Cout, f32, 'C')
  a = L.GemmTNArgs()
  a.A, a.lda, a.K = A.data_ptr(), lda if lda else A.stride(0), K
  if rank1 is not None:
    g, w, bits = rank1
    _chk(g, f32, 'rank1 g')
    _chk(w, f32, 'rank1 w')
    _chk(bits, torch.uint8, 'rank1 bits')
    assert B is None and g.numel() == M and g.is_contiguous() and w.numel() >= N and w.is_contiguous()
    assert bits.dim() == 2 and bits.shape[0] == M and bits.shape[1] * 8 >= N and bits.stride(1) == 1
    a.B, a.ldb, a.N = None, N, N
    a.rank1_g, a.rank1_w, a.rank1_bits, a.ld_rank1_bits = g.data_ptr(), w.data_ptr(), bits.data_ptr(), bits.stride(0)
  else:
    a.B, a.ldb, a.N = B.data_ptr(), ldb if ldb else B.stride(0), N
  a.M = M
  a.C, a.ldc = Cout.data_ptr(), ldc if ldc else Cout.stride(0)
  a.k_valid = K if k_valid is None else k_valid
  a.n_valid = N if n_valid is None else n_valid
  a.bias_out = bias_out.data_ptr() if bias_out is not None else None
  a.bias_n_valid = bias_n_valid
  a.gcol = gcol.data_ptr() if gcol is not None else None
  a.gcol_out = gcol_out.data_ptr() if gcol_out is not None else None
  a.a_layout, a.b_layout = a_layout, b_layout
  a.m_interleave, a.max_wgs = int(bool(m_interleave)), int(max_wgs)
  _e = PROFILE.start()
  L.check(lib().mnr_gemm_tn_bf16(C.byref(a), _stream()))
  PROFILE.stop(_e)


def mlp_chain_fwd(feat, K0, layers, *, M, W, w_head=None, b_head=None, head_out=None, acts=None, bits=None, skip_layer=0):
  """Fused Dense + ReLU chain with a Dense(1) head (csrc/fused_mlp.hip): `layers` = [(Bt [W, ldb] bf16, bias [W] fp32)],
  layer 0 reading feat [M, ld_feat] over K0 columns.  acts / bits: optional per-layer outputs (training)."""
  depth = len(layers)
  if not 1 <= depth <= L.CHAIN_MAX_DEPTH:
    raise ValueError(f'mlp_chain_fwd: depth {depth}')
  _chk(feat, bf16, 'feat')
  _chk(w_head, bf16, 'w_head', allow_none=True)
  _chk(b_head, f32, 'b_head', allow_none=True)
  _chk(head_out, f32, 'head_out', allow_none=True)
  a = L.MlpChainFwdArgs()
  a.M, a.W, a.depth, a.skip_layer = M, W, depth, int(skip_layer)
  a.feat, a.ld_feat, a.K0 = feat.data_ptr(), feat.stride(0), K0
  for i, (Bt, bias) in enumerate(layers):
    _chk(Bt, bf16, f'Bt[{i}]')
    _chk(bias, f32, f'bias[{i}]')
    assert Bt.shape[0] >= W and bias.numel() == W
    a.Bt[i], a.ldb[i], a.bias[i] = Bt.data_ptr(), Bt.stride(0), bias.data_ptr()
    if acts is not None and acts[i] is not None:
      _chk(acts[i], bf16, f'acts[{i}]')
      assert acts[i].shape == (M, W)
      a.acts[i] = acts[i].data_ptr()
    if bits is not None and bits[i] is not None:
      _chk(bits[i], torch.uint8, f'bits[{i}]')
      assert bits[i].shape == (M, W // 8)
      a.bits[i] = bits[i].data_ptr()
  if w_head is not None:
    assert w_head.numel() >= W and head_out is not None and head_out.numel() == M
    a.w_head, a.head_out = w_head.data_ptr(), head_out.data_ptr()
    a.b_head = b_head.data_ptr() if b_head is not None else None
  _e = PROFILE.start()
  L.check(lib().mnr_mlp_chain_fwd(C.byref(a), _stream()))
  PROFILE.stop(_e)


def mlp_chain_fwd_ipe(tdist, origins, directions, radii, basis, layers, *, M, W, ray_shape, warp_contract, min_deg,
                      max_deg, disable_integration=False, w_head=None, b_head=None, head_out=None, act_last=None):
  """The inference chain with layer 0's IPE features produced in the kernel (csrc/fused_mlp.hip, mnr_mlp_chain_fwd_ipe):
  no feature matrix.  layers[0][0] is the group-major reordered kernel^T (include/mnerf.h); tdist [B, n+1] with B*n = M."""
  depth = len(layers)
  if not 1 <= depth <= L.CHAIN_MAX_DEPTH:
    raise ValueError(f'mlp_chain_fwd_ipe: depth {depth}')
  for x, nm in ((tdist, 'tdist'), (origins, 'origins'), (directions, 'directions'), (radii, 'radii'), (basis, 'basis')):
    _chk(x, f32, nm)
  _chk(w_head, bf16, 'w_head', allow_none=True)
  _chk(b_head, f32, 'b_head', allow_none=True)
  _chk(head_out, f32, 'head_out', allow_none=True)
  B, n1 = tdist.shape
  assert B * (n1 - 1) == M and origins.shape[0] == B and directions.shape[0] == B and radii.numel() == B
  a = L.MlpChainFwdArgs()
  a.M, a.W, a.depth, a.skip_layer = M, W, depth, 0
  for i, (Bt, bias) in enumerate(layers):
    _chk(Bt, bf16, f'Bt[{i}]')
    _chk(bias, f32, f'bias[{i}]')
    assert Bt.shape[0] >= W and bias.numel() == W
    a.Bt[i], a.ldb[i], a.bias[i] = Bt.data_ptr(), Bt.stride(0), bias.data_ptr()
  if act_last is not None:
    _chk(act_last, bf16, 'act_last')
    assert act_last.shape == (M, W)
    a.acts[depth - 1] = act_last.data_ptr()
  if w_head is not None:
    assert w_head.numel() >= W and head_out is not None and head_out.numel() == M
    a.w_head, a.head_out = w_head.data_ptr(), head_out.data_ptr()
    a.b_head = b_head.data_ptr() if b_head is not None else None
  q = L.ChainIpeArgs()
  q.cfg = _ipe_cfg(ray_shape, warp_contract, disable_integration, basis, min_deg, max_deg)
  q.n = n1 - 1
  q.tdist, q.origins, q.directions, q.radii, q.basis = _ptr(tdist), _ptr(origins), _ptr(directions), _ptr(radii), _ptr(basis)
  _e = PROFILE.start()
  L.check(lib().mnr_mlp_chain_fwd_ipe(C.byref(a), C.byref(q), _stream()))
  PROFILE.stop(_e)


def mlp_chain_bwd(g_head, w_head, bits, Bws, dYs, *, M, W, dY_in=None):
  """The dX chain of the fused Dense stack: dY[last] = mask * (g_head (x) w_head), dY[i-1] = mask_{i-1} * (dY[i] W_i^T).
  Bws[i] (i >= 1): [W, ldb] bf16 kernel as stored (rows = inputs); Bws[0] unused."""
  depth = len(bits)
  assert len(dYs) == depth and len(Bws) == depth
  a = L.MlpChainBwdArgs()
  a.M, a.W, a.depth = M, W, depth
  if dY_in is not None:
    _chk(dY_in, bf16, 'dY_in')
    assert dY_in.shape == (M, W) and dYs[depth - 1] is None
    a.dY_in = dY_in.data_ptr()
  else:
    _chk(g_head, f32, 'g_head')
    _chk(w_head, f32, 'w_head')
    assert g_head.numel() == M and w_head.numel() == W
    a.g_head, a.w_head = g_head.data_ptr(), w_head.data_ptr()
  for i in range(depth):
    _chk(bits[i], torch.uint8, f'bits[{i}]')
    assert bits[i].shape == (M, W // 8)
    a.bits[i] = bits[i].data_ptr()
    if dYs[i] is not None:
      _chk(dYs[i], bf16, f'dY[{i}]')
      assert dYs[i].shape == (M, W)
      a.dY[i] = dYs[i].data_ptr()
    if i >= 1:
      _chk(Bws[i], bf16, f'Bw[{i}]')
      a.Bw[i], a.ldb[i] = Bws[i].data_ptr(), Bws[i].stride(0)
  _e = PROFILE.start()
  L.check(lib().mnr_mlp_chain_bwd(C.byref(a), _stream()))
  PROFILE.stop(_e)


def colsum(X, M, n_valid, out, ld=None):
  _chk(X, bf16, 'X')
  _chk(out, f32, 'out')
  L.check(lib().mnr_colsum_bf16(_ptr(X), ld if ld else X.stride(0), M, n_valid, _ptr(out), _stream()))


def pack_weights(params, descs_dev, n_desc, max_elems, dst):
  _chk(params, f32, 'params')
  _chk(dst, bf16, 'dst')
  L.check(lib().mnr_pack_weights_bf16(_ptr(params), _ptr(descs_dev), n_desc, max_elems, _ptr(dst), _stream()))


def scatter_add(src, ld_src, row0, col0, rows, cols, dst, ld_dst):
  _chk(src, f32, 'src')
  _chk(dst, f32, 'dst')
  L.check(lib().mnr_scatter_add_f32(_ptr(src), ld_src, row0, col0, rows, cols, _ptr(dst), ld_dst, _stream()))


def cast_f32_to_bf16(src, ld_src, M, n, dst, ld_dst, col0):
  _chk(src, f32, 'src')
  _chk(dst, bf16, 'dst')
  L.check(lib().mnr_cast_f32_to_bf16(_ptr(src), ld_src, M, n, _ptr(dst), ld_dst, col0, _stream()))


def act_fwd(kind, z, a):
  """a = act(z) for a non-ReLU net_activation ('softplus' | 'silu'); z, a contiguous bf16 of the same shape."""
  _chk(z, bf16, 'z')
  _chk(a, bf16, 'a')
  assert z.shape == a.shape
  L.check(lib().mnr_act_fwd_bf16(L.NET_ACT[kind], z.numel(), _ptr(z), _ptr(a), _stream()))


def act_bwd(kind, z, d):
  """d *= act'(z) in place: gradient w.r.t. the activation -> w.r.t. the pre-activation."""
  _chk(z, bf16, 'z')
  _chk(d, bf16, 'd')
  assert z.shape == d.shape
  L.check(lib().mnr_act_bwd_bf16(L.NET_ACT[kind], z.numel(), _ptr(z), _ptr(d), _stream()))


def act_tangent_fwd(kind, z, U, T):
  """T = act'(z) * U: the tangent network of the density-gradient normals behind a non-ReLU activation; z [M, W], U, T [3M, W]."""
  for x, nm in ((z, 'z'), (U, 'U'), (T, 'T')):
    _chk(x, bf16, nm)
  assert U.shape == T.shape and U.numel() == 3 * z.numel()
  L.check(lib().mnr_act_tangent_fwd_bf16(L.NET_ACT[kind], z.numel(), _ptr(z), _ptr(U), _ptr(T), _stream()))


def act_tangent_bwd(kind, z, U, G, extra):
  """G (d loss / d T, [3M, W]) *= act'(z) in place; extra [M, W] = sum_c G_c * U_c * act''(z) (joins the primal gradient of z)."""
  for x, nm in ((z, 'z'), (U, 'U'), (G, 'G'), (extra, 'extra')):
    _chk(x, bf16, nm)
  assert U.shape == G.shape and U.numel() == 3 * z.numel() and extra.shape == z.shape
  L.check(lib().mnr_act_tangent_bwd_bf16(L.NET_ACT[kind], z.numel(), _ptr(z), _ptr(U), _ptr(G), _ptr(extra), _stream()))


def add_noise_bf16(X, cols, noise, scale):
  """X[:, :cols] += scale * noise (fp32 add, one bf16 rounding); X bf16 [M, ld], noise fp32 [M, cols]."""
  _chk(noise, f32, 'noise')
  if X.dtype != act_dtype() or not _on_device(X):
    raise ValueError(f'X must be a {act_dtype()} device tensor')
  M = X.shape[0]
  assert noise.shape == (M, cols)
  L.check(lib().mnr_add_noise_bf16(M, cols, _ptr(X), X.stride(0), _ptr(noise), float(scale), _stream()))


_HEAD_SCRATCH = {}


def _head_scratch(device):
  """Workspace for the per-workgroup dW / db partials of mnr_small_head_bwd (4 MiB per device AND stream, allocated
  once: two levels' backward passes may run side by side on different streams, multinerf_amd/streams.py)."""
  key = (device, getattr(_stream(), 'value', None))
  t = _HEAD_SCRATCH.get(key)
  if t is None:
    t = _HEAD_SCRATCH[key] = torch.empty(1 << 20, dtype=f32, device=device)
  return t


def small_head_bwd(H, ldh, g, W, *, M, K, Cn, dX=None, lddx=0, relu_mask=True, dW=None, db=None, bits=None,
                   bits_row_mod=0):
  _chk(bits, torch.uint8, 'bits', allow_none=True)
  _chk(H, bf16, 'H')
  _chk(g, f32, 'g')
  _chk(W, f32, 'W')
  _e = PROFILE.start()
  scratch = _head_scratch(H.device)
  L.check(lib().mnr_small_head_bwd(M, K, Cn, _ptr(H), ldh, _ptr(g), _ptr(W), _ptr(dX), lddx, int(relu_mask),
                                   _ptr(dW), _ptr(db), _ptr(bits), bits.stride(0) if bits is not None else 0,
                                   bits_row_mod, _ptr(scratch), scratch.numel(), _stream()))
  PROFILE.stop(_e, 'small_head_bwd', 2 * M * K * (2 if dX is not None else 1) + 4 * M * Cn)


# ----------------------------------------------------------------------------- compositing


def composite_cfg(n, *, opaque_background, density_act, density_bias, density_noise_std, has_rgb, rgb_act,
                  rgb_premultiplier, rgb_bias, rgb_padding, bg_mode, bg_value):
  return L.CompositeCfg(n, int(opaque_background), L.ACT[density_act], float(density_bias),
                        float(density_noise_std), int(has_rgb), L.ACT[rgb_act], float(rgb_premultiplier),
                        float(rgb_bias), float(rgb_padding), int(bg_mode), float(bg_value))


def composite_fwd(cfg, raw_density, tdist, dirs, *, raw_rgb=None, density_noise=None, bg=None,
                  exposure_scale=None, want_acc=True):
  _chk(raw_density, f32, 'raw_density')
  _chk(tdist, f32, 'tdist')
  _chk(dirs, f32, 'dirs')
  for x, nm in ((raw_rgb, 'raw_rgb'), (density_noise, 'density_noise'), (bg, 'bg'),
                (exposure_scale, 'exposure_scale')):
    _chk(x, f32, nm, allow_none=True)
  B, n = raw_density.shape
  dev = raw_density.device
  density = torch.empty((B, n), dtype=f32, device=dev)
  weights = torch.empty((B, n), dtype=f32, device=dev)
  rgb = torch.empty((B, n, 3), dtype=f32, device=dev) if cfg.has_rgb else None
  rgb_out = torch.empty((B, 3), dtype=f32, device=dev)
  acc = torch.empty((B,), dtype=f32, device=dev) if want_acc else None
  _e = PROFILE.start()
  L.check(lib().mnr_composite_fwd(C.byref(cfg), B, _ptr(raw_density), _ptr(density_noise), _ptr(raw_rgb),
                                  _ptr(tdist), _ptr(dirs), _ptr(bg), _ptr(exposure_scale), _ptr(density),
                                  _ptr(rgb), _ptr(weights), _ptr(rgb_out), _ptr(acc), _stream()))
  PROFILE.stop(_e, 'composite_fwd', 4 * (raw_density.numel() * (2 + 2 + (6 if raw_rgb is not None else 0))))
  return density, rgb, weights, rgb_out, acc


def composite_bwd(cfg, raw_density, tdist, dirs, weights, *, raw_rgb=None, density_noise=None, bg=None,
                  exposure_scale=None, g_rgb_out=None, g_weights=None, g_den_bf16=None, ld_bf16=0,
                  want_f32=True, g_exposure_scale=None, losses=None, g_raw_density_out=None, g_x_out=None):
  """Compositing VJP; with `losses` the level's training losses are fused in front of it (mnr_level_bwd):
  g_raw_density_out: optional [B, n] fp32 destination for d loss / d raw_density (else a fresh tensor).
  g_x_out: optional [B, n] fp32 destination for d loss / d (sigma * delta) (Model.stop_level_grad = False, mnr_sdist_bwd).
  losses = dict(B_valid=..., data=dict(type, charb_padding, mult, rgb_out, gt, lossmult, denom, stats) | None,
                weights=dict(mode='interlevel'|'distortion', mult, sdist, t_ref, w_ref, stat) | None)."""
  B, n = raw_density.shape
  dev = raw_density.device
  for x, nm in ((raw_density, 'raw_density'), (tdist, 'tdist'), (dirs, 'dirs'), (weights, 'weights')):
    _chk(x, f32, nm)
  for x, nm in ((g_rgb_out, 'g_rgb_out'), (g_weights, 'g_weights')):
    _chk(x, f32, nm, allow_none=True)
  _chk(g_den_bf16, bf16, 'g_den_bf16', allow_none=True)
  if g_raw_density_out is not None:
    _chk(g_raw_density_out, f32, 'g_raw_density_out')
    assert want_f32 and g_raw_density_out.shape == (B, n) and g_raw_density_out.is_contiguous()
    g_raw_density = g_raw_density_out
  else:
    g_raw_density = torch.empty((B, n), dtype=f32, device=dev) if want_f32 else None
  g_raw_rgb = torch.empty((B, n, 3), dtype=f32, device=dev) if cfg.has_rgb else None
  a = L.LevelBwdArgs()
  a.cfg = cfg
  a.B = B
  a.B_valid = B if losses is None else int(losses['B_valid'])
  p = lambda t: None if t is None else t.data_ptr()
  a.raw_density, a.density_noise, a.raw_rgb, a.tdist, a.dirs = p(raw_density), p(density_noise), p(raw_rgb), p(tdist), p(dirs)
  a.bg, a.exposure_scale, a.weights, a.g_rgb_out, a.g_weights = p(bg), p(exposure_scale), p(weights), p(g_rgb_out), p(g_weights)
  a.g_raw_density, a.g_raw_density_bf16, a.ld_bf16 = p(g_raw_density), p(g_den_bf16), ld_bf16
  a.g_raw_rgb, a.g_exposure_scale = p(g_raw_rgb), p(g_exposure_scale)
  a.data_loss_type, a.wloss_mode = -1, 0
  if g_x_out is not None:
    _chk(g_x_out, f32, 'g_x_out')
    assert g_x_out.shape == (B, n) and g_x_out.is_contiguous()
    a.g_x = p(g_x_out)
  keep = []
  if losses is not None and losses.get('data') is not None:
    d = losses['data']
    if d['type'] not in L.DATA_LOSS:
      raise ValueError(f"unsupported data_loss_type {d['type']!r}")
    for k in ('rgb_out', 'gt', 'lossmult', 'denom', 'stats'):
      _chk(d[k], f32, 'data.' + k)
    assert d['rgb_out'].shape[0] == B and d['gt'].shape[0] == B and d['lossmult'].shape[0] == B
    a.data_loss_type, a.charb_padding, a.data_loss_mult = L.DATA_LOSS[d['type']], float(d['charb_padding']), float(d['mult'])
    a.rgb_out, a.gt, a.lossmult, a.lm_c = p(d['rgb_out']), p(d['gt']), p(d['lossmult']), d['lossmult'].shape[-1]
    a.denom, a.data_stats = p(d['denom']), p(d['stats'])
    keep.append(d)
  if losses is not None and losses.get('weights') is not None:
    w = losses['weights']
    _chk(w['sdist'], f32, 'weights.sdist')
    _chk(w['stat'], f32, 'weights.stat')
    assert w['sdist'].shape == (B, n + 1)
    a.wloss_mult, a.sdist, a.wloss_stat = float(w['mult']), p(w['sdist']), p(w['stat'])
    if w['mode'] == 'interlevel':
      _chk(w['t_ref'], f32, 'weights.t_ref')
      _chk(w['w_ref'], f32, 'weights.w_ref')
      assert w['t_ref'].shape[0] == B and w['t_ref'].shape[1] == w['w_ref'].shape[1] + 1
      a.wloss_mode, a.n_ref, a.t_ref, a.w_ref = 1, w['w_ref'].shape[1], p(w['t_ref']), p(w['w_ref'])
    elif w['mode'] == 'distortion':
      a.wloss_mode = 2
    else:
      raise ValueError(f"weights.mode {w['mode']!r}")
    keep.append(w)
  _e = PROFILE.start()
  L.check(lib().mnr_level_bwd(C.byref(a), _stream()))
  PROFILE.stop(_e, 'composite_bwd', 4 * (raw_density.numel() * (3 + (3 if raw_rgb is not None else 0))) + (2 + (12 if raw_rgb is not None else 0)) * raw_density.numel())
  return g_raw_density, g_raw_rgb


def exposure_scale(exposure_values, exposure_idx, offsets):
  _chk(exposure_values, f32, 'exposure_values')
  _chk(exposure_idx, torch.int32, 'exposure_idx')
  _chk(offsets, f32, 'offsets', allow_none=True)
  B = exposure_values.numel()
  out = torch.empty((B, 3), dtype=f32, device=exposure_values.device)
  L.check(lib().mnr_exposure_scale(B, _ptr(exposure_values), _ptr(exposure_idx), _ptr(offsets), _ptr(out), _stream()))
  return out


def exposure_scale_bwd(exposure_values, exposure_idx, g_scale, g_offsets, B_valid):
  _chk(g_scale, f32, 'g_scale')
  _chk(g_offsets, f32, 'g_offsets')
  L.check(lib().mnr_exposure_scale_bwd(B_valid, _ptr(exposure_values), _ptr(exposure_idx), _ptr(g_scale),
                                       _ptr(g_offsets), _stream()))


def render_extras(weights, tdist, t_far):
  for x, nm in ((weights, 'weights'), (tdist, 'tdist'), (t_far, 't_far')):
    _chk(x, f32, nm)
  B, n = weights.shape
  out = torch.empty((B, 4), dtype=f32, device=weights.device)
  L.check(lib().mnr_render_extras(B, n, _ptr(weights), _ptr(tdist), _ptr(t_far), _ptr(out), _stream()))
  return out


# ----------------------------------------------------------------------------- Ref-NeRF


class IdeTablesDev:
  """Device copies of ref_utils.ide_tables(deg_view) + the C struct that points at them."""

  def __init__(self, deg_view, device):
    from multinerf_amd import ref_utils
    m, l, mat, sigma = ref_utils.ide_tables(deg_view)
    self.T = len(m)
    self.m = torch.as_tensor(m, dtype=torch.int32, device=device)
    self.l = torch.as_tensor(l, dtype=torch.int32, device=device)
    self.mat = torch.as_tensor(mat, dtype=f32, device=device).contiguous()
    self.sigma = torch.as_tensor(sigma, dtype=f32, device=device)
    self.c = L.IdeTables(self.T, mat.shape[0] - 1, self.m.data_ptr(), self.l.data_ptr(), self.sigma.data_ptr(),
                         self.mat.data_ptr())


# mnr_ref_head_fwd / _bwd feature bits (include/mnerf.h MNR_REF_*)
REF_PRED_NORMALS, REF_DENSITY_NORMALS, REF_REFLECT, REF_IDE, REF_N_DOT_V, REF_ROUGHNESS = 1, 2, 4, 8, 16, 32
REF_ALL = 63


def ref_head_fwd(small, raw_grad, viewdirs, n, tabs, roughness_bias, vi, col0, col_end, features=REF_ALL, deg_view=0):
  """-> (normals, normals_pred, roughness); None for the parts `features` switches off."""
  for x, nm in ((small, 'small'), (viewdirs, 'viewdirs')):
    _chk(x, f32, nm)
  _chk(raw_grad, f32, 'raw_grad', allow_none=True)
  _chk(vi, bf16, 'vi')
  M = small.shape[0]
  dev = small.device
  normals = torch.empty((M, 3), dtype=f32, device=dev) if features & REF_DENSITY_NORMALS else None
  npred = torch.empty((M, 3), dtype=f32, device=dev) if features & REF_PRED_NORMALS else None
  rough = torch.empty((M,), dtype=f32, device=dev) if features & REF_ROUGHNESS else None
  L.check(lib().mnr_ref_head_fwd(M, n, _ptr(small), _ptr(raw_grad), _ptr(viewdirs), C.byref(tabs.c) if tabs is not None else None,
                                 int(features), int(deg_view), float(roughness_bias), _ptr(vi), vi.stride(0), col0, col_end,
                                 _ptr(normals), _ptr(npred), _ptr(rough), _stream()))
  return normals, npred, rough


def ref_head_bwd(small, raw_grad, viewdirs, n, tabs, roughness_bias, dvi_a, dvi_b, col0, g_npred, g_n, dhb, col_gp,
                 col_rough, features=REF_ALL, deg_view=0):
  """-> g_raw_grad [3, M] (None without density-gradient normals)."""
  M = small.shape[0]
  _chk(dvi_a, bf16, 'dvi_a')
  _chk(dvi_b, bf16, 'dvi_b', allow_none=True)
  _chk(dhb, bf16, 'dhb')
  g_raw_grad = torch.empty((3, M), dtype=f32, device=small.device) if features & REF_DENSITY_NORMALS else None
  L.check(lib().mnr_ref_head_bwd(M, n, _ptr(small), _ptr(raw_grad), _ptr(viewdirs), C.byref(tabs.c) if tabs is not None else None,
                                 int(features), int(deg_view), float(roughness_bias), _ptr(dvi_a), _ptr(dvi_b), dvi_a.stride(0), col0,
                                 _ptr(g_npred), _ptr(g_n), _ptr(dhb), dhb.stride(0), col_gp, col_rough,
                                 _ptr(g_raw_grad), _stream()))
  return g_raw_grad


def add_cols_bf16(a, b, dst, cols):
  """dst[:, :cols] = a[:, :cols] + b[:, :cols] (row strides taken from the tensors)."""
  for x, nm in ((a, 'a'), (dst, 'dst')):
    if x.dtype != act_dtype() or not _on_device(x):
      raise ValueError(f'{nm} must be a {act_dtype()} device tensor')
  M = a.shape[0]
  L.check(lib().mnr_add_cols_bf16(M, cols, _ptr(a), a.stride(0), _ptr(b), b.stride(0) if b is not None else 0,
                                  _ptr(dst), dst.stride(0), _stream()))


def ref_color_fwd(raw_rgb, small, premult, rgb_bias, pad, use_tint):
  _chk(raw_rgb, f32, 'raw_rgb')
  _chk(small, f32, 'small')
  M = small.shape[0]
  out = torch.empty((M, 3), dtype=f32, device=small.device)
  L.check(lib().mnr_ref_color_fwd(M, _ptr(raw_rgb), _ptr(small), float(premult), float(rgb_bias), float(pad),
                                  int(use_tint), _ptr(out), _stream()))
  return out


def ref_color_bwd(raw_rgb, small, premult, rgb_bias, pad, use_tint, g_rgb, dhb, col_diffuse, col_tint):
  _chk(g_rgb, f32, 'g_rgb')
  _chk(dhb, bf16, 'dhb')
  M = small.shape[0]
  g_raw_rgb = torch.empty((M, 3), dtype=f32, device=small.device)
  L.check(lib().mnr_ref_color_bwd(M, _ptr(raw_rgb), _ptr(small), float(premult), float(rgb_bias), float(pad),
                                  int(use_tint), _ptr(g_rgb), _ptr(g_raw_rgb), _ptr(dhb), dhb.stride(0), col_diffuse,
                                  col_tint, _stream()))
  return g_raw_rgb


def density_normals_fwd(raw_grad):
  """normals = -l2_normalize(raw_grad) (models.py:492); raw_grad [3, M] component-major."""
  _chk(raw_grad, f32, 'raw_grad')
  M = raw_grad.shape[1]
  out = torch.empty((M, 3), dtype=f32, device=raw_grad.device)
  L.check(lib().mnr_density_normals_fwd(M, _ptr(raw_grad), _ptr(out), _stream()))
  return out


def density_normals_bwd(raw_grad, g_normals):
  _chk(raw_grad, f32, 'raw_grad')
  _chk(g_normals, f32, 'g_normals')
  M = raw_grad.shape[1]
  out = torch.empty((3, M), dtype=f32, device=raw_grad.device)
  L.check(lib().mnr_density_normals_bwd(M, _ptr(raw_grad), _ptr(g_normals), _ptr(out), _stream()))
  return out


def pred_normals_fwd(small, col):
  """normals_pred = -l2_normalize(small[:, col:col+3]) (models.py:498)."""
  _chk(small, f32, 'small')
  M, ld = small.shape
  out = torch.empty((M, 3), dtype=f32, device=small.device)
  L.check(lib().mnr_pred_normals_fwd(M, _ptr(small), ld, col, _ptr(out), _stream()))
  return out


def pred_normals_bwd(small, col, g_npred, dhb, col_g):
  _chk(small, f32, 'small')
  _chk(g_npred, f32, 'g_npred')
  _chk(dhb, bf16, 'dhb')
  M, ld = small.shape
  L.check(lib().mnr_pred_normals_bwd(M, _ptr(small), ld, col, _ptr(g_npred), _ptr(dhb), dhb.stride(0), col_g, _stream()))


def ref_losses(mult_o, mult_p, target_is_pred, weights, normals, npred, viewdirs, stats, g_w, want_grad, *, B_valid):
  for x, nm in ((weights, 'weights'), (viewdirs, 'viewdirs'), (stats, 'stats')):
    _chk(x, f32, nm)
  _chk(normals, f32, 'normals', allow_none=True)
  _chk(npred, f32, 'npred', allow_none=True)
  B, n = weights.shape
  g_n = torch.zeros_like(normals) if (want_grad and normals is not None) else None
  g_np = torch.zeros_like(npred) if (want_grad and npred is not None) else None
  L.check(lib().mnr_ref_losses(B_valid, n, float(mult_o), float(mult_p), int(target_is_pred), _ptr(weights),
                               _ptr(normals), _ptr(npred), _ptr(viewdirs), _ptr(stats), _ptr(g_w), _ptr(g_n),
                               _ptr(g_np), _stream()))
  return g_n, g_np


def weighted_sum(weights, values):
  _chk(weights, f32, 'weights')
  _chk(values, f32, 'values')
  B, n = weights.shape
  Cn = values.numel() // (B * n)
  out = torch.empty((B, Cn), dtype=f32, device=weights.device)
  L.check(lib().mnr_weighted_sum(B, n, Cn, _ptr(weights), _ptr(values), _ptr(out), _stream()))
  return out


# ----------------------------------------------------------------------------- losses


def lossmult_sum(lossmult, B_valid, out):
  _chk(lossmult, f32, 'lossmult')
  _chk(out, f32, 'out')
  L.check(lib().mnr_lossmult_sum(B_valid, _ptr(lossmult), lossmult.shape[-1], _ptr(out), _stream()))


def render_metrics(B_valid, *, distance_mean=None, disps=None, acc=None, alphas=None, normals=None, normals_gt=None,
                   out_disp=None, out_normal=None):
  for x, nm in ((distance_mean, 'distance_mean'), (disps, 'disps'), (acc, 'acc'), (alphas, 'alphas'),
                (normals, 'normals'), (normals_gt, 'normals_gt'), (out_disp, 'out_disp'), (out_normal, 'out_normal')):
    _chk(x, f32, nm, allow_none=True)
  L.check(lib().mnr_render_metrics(B_valid, _ptr(distance_mean), _ptr(disps), _ptr(acc), _ptr(alphas), _ptr(normals),
                                   _ptr(normals_gt), _ptr(out_disp), _ptr(out_normal), _stream()))


def data_loss(loss_type, charb_padding, loss_mult, rgb, gt, lossmult, denom, stats, *, B_valid, want_grad=True):
  for x, nm in ((rgb, 'rgb'), (gt, 'gt'), (lossmult, 'lossmult'), (denom, 'denom'), (stats, 'stats')):
    _chk(x, f32, nm)
  if loss_type not in L.DATA_LOSS:
    raise ValueError(f'unsupported data_loss_type {loss_type!r}')
  B = rgb.shape[0]
  g = torch.empty_like(rgb) if want_grad else None
  L.check(lib().mnr_data_loss(L.DATA_LOSS[loss_type], float(charb_padding), float(loss_mult), B, B_valid,
                              _ptr(rgb), _ptr(gt), _ptr(lossmult), lossmult.shape[-1], _ptr(denom),
                              _ptr(stats), _ptr(g), _stream()))
  return g


def interlevel_loss(mult, t, w, t_env, w_env, stats, g_w_env, *, B_valid):
  for x, nm in ((t, 't'), (w, 'w'), (t_env, 't_env'), (w_env, 'w_env')):
    _chk(x, f32, nm)
  B, n = w.shape
  ne = w_env.shape[1]
  L.check(lib().mnr_interlevel_loss(float(mult), B, B_valid, n, _ptr(t), _ptr(w), ne, _ptr(t_env), _ptr(w_env),
                                    _ptr(stats), _ptr(g_w_env), _stream()))


def distortion_loss(mult, t, w, stats, g_w, *, B_valid):
  _chk(t, f32, 't')
  _chk(w, f32, 'w')
  B, n = w.shape
  L.check(lib().mnr_distortion_loss(float(mult), B, B_valid, n, _ptr(t), _ptr(w), _ptr(stats), _ptr(g_w),
                                    _stream()))


def lossfun_outer(t, w, t_env, w_env):
  for x, nm in ((t, 't'), (w, 'w'), (t_env, 't_env'), (w_env, 'w_env')):
    _chk(x, f32, nm)
  B, n = w.shape
  out = torch.empty((B, n), dtype=f32, device=w.device)
  L.check(lib().mnr_lossfun_outer(B, n, _ptr(t), _ptr(w), w_env.shape[1], _ptr(t_env), _ptr(w_env), _ptr(out),
                                  _stream()))
  return out


def lossfun_distortion(t, w):
  _chk(t, f32, 't')
  _chk(w, f32, 'w')
  B, n = w.shape
  out = torch.empty((B,), dtype=f32, device=w.device)
  L.check(lib().mnr_lossfun_distortion(B, n, _ptr(t), _ptr(w), _ptr(out), _stream()))
  return out


# ----------------------------------------------------------------------------- optimiser


def weight_decay(params, begin, end, mult, grad, loss_out, sqnorm_out=None):
  _chk(params, f32, 'params')
  L.check(lib().mnr_weight_decay(_ptr(params), begin, end, float(mult), _ptr(grad), _ptr(loss_out), _ptr(sqnorm_out),
                                 _stream()))


def grad_sqnorm(grad, begin, end, max_val, out):
  _chk(grad, f32, 'grad')
  _chk(out, f32, 'out')
  L.check(lib().mnr_grad_sqnorm(_ptr(grad), begin, end, float(max_val), _ptr(out), _stream()))


def clip_adam(grad, params, mu, nu, begin, end, sqnorm, *, lr, b1, b2, eps, step, grad_max_val, grad_max_norm):
  for x, nm in ((grad, 'grad'), (params, 'params'), (mu, 'mu'), (nu, 'nu')):
    _chk(x, f32, nm)
  cfg = L.AdamCfg(float(lr), float(b1), float(b2), float(eps), float(1 - b1**step), float(1 - b2**step),
                  float(grad_max_val), float(grad_max_norm))
  _e = PROFILE.start()
  L.check(lib().mnr_clip_adam(C.byref(cfg), begin, end, _ptr(sqnorm), _ptr(grad), _ptr(params), _ptr(mu),
                              _ptr(nu), _stream()))
  PROFILE.stop(_e, 'clip_adam', 28 * (end - begin))
