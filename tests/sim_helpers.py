"""Loader for tools/hipsim (the host-side functional simulator of the csrc kernels): test infrastructure only."""

import ctypes as C
import importlib.util
import os

import torch

from multinerf_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_sim = None
_sim_f32 = None


def load_sim(f32=False):
  """Build (if stale) and load libmnerf_sim.so (f32: libmnerf_sim_f32.so, the fp32-Dense debug build); binds every prototype of
  multinerf_amd._lib it exports."""
  global _sim, _sim_f32
  if (_sim_f32 if f32 else _sim) is not None:
    return _sim_f32 if f32 else _sim
  spec = importlib.util.spec_from_file_location('hipsim_build', os.path.join(ROOT, 'tools', 'hipsim', 'build.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  lib = C.CDLL(mod.build(verbose=False, f32=f32))
  lib.mnr_last_error.restype = C.c_char_p
  lib.hipsim_error.restype = C.c_char_p
  lib.hipsim_reset.argtypes = [C.c_int, C.c_long]
  lib.hipsim_stats.argtypes = [C.POINTER(C.c_ulonglong)]
  for name, (argtypes, restype) in list(L._PROTOS.items()) + list(L._DEBUG_PROTOS.items()):
    fn = getattr(lib, name, None)
    if fn is not None:
      fn.argtypes = argtypes
      fn.restype = restype
  if f32:
    _sim_f32 = lib
  else:
    _sim = lib
  return lib


def sim_check(lib, status):
  if status != 0:
    raise RuntimeError(f'status {status}: {lib.mnr_last_error().decode()}')
  if lib.hipsim_failed():
    raise RuntimeError('hipsim: ' + lib.hipsim_error().decode())


def ptr(t):
  return None if t is None else C.c_void_p(t.data_ptr())


def sim_gemm_nt(lib, A1, Bt, A2=None, bias=None, relu=False, mask=None, out_bf16=True, out_f32=None, bits_out=False,
                bits_in=None, bits_row_mod=0, nb=None, a1_layout=0, c_layout=0, vcol=None, vcol_bias=None,
                walk_descending=False):
  """C = epi([A1|A2] Bt^T) through the simulated mnr_gemm_nt_bf16; returns (Cb, Cf, bits).  a1_layout / c_layout = 1: A1 is
  given / Cb and the bits come back in MNR_LAYOUT_PANEL storage (the bits in tile order, flat)."""
  M, K1 = A1.shape
  N = Bt.shape[0]
  a = L.GemmNTArgs()
  a.a1_layout, a.c_layout = a1_layout, c_layout
  a.walk_descending = int(walk_descending)
  a.A1, a.lda1, a.K1 = ptr(A1), A1.stride(0), K1
  if A2 is not None:
    a.A2, a.lda2, a.K2 = ptr(A2), A2.stride(0), A2.shape[1]
  a.Bt, a.ldb, a.M, a.N = ptr(Bt), Bt.stride(0), M, N
  if bias is not None:
    a.bias, a.n_bias = ptr(bias), bias.numel()
  a.relu = int(relu)
  if mask is not None:
    a.mask, a.ldmask = ptr(mask), mask.stride(0)
  Cb = Cf = bits = None
  if out_bf16:
    nb = N if nb is None else nb
    Cb = torch.full((M, nb), float('nan'), dtype=torch.bfloat16)
    a.Cb, a.ldcb, a.nb = ptr(Cb), Cb.stride(0), nb
  if out_f32 is not None:
    f0, nf = out_f32
    Cf = torch.full((M, nf), float('nan'), dtype=torch.float32)
    a.Cf, a.ldcf, a.f0, a.nf = ptr(Cf), Cf.stride(0), f0, nf
  if bits_out:
    bits = torch.zeros((M, N // 8), dtype=torch.uint8)
    a.mask_bits_out, a.ld_bits_out = ptr(bits), bits.stride(0)
  if bits_in is not None:
    a.mask_bits_in, a.ld_bits_in, a.bits_row_mod = ptr(bits_in), (bits_in.stride(0) if bits_in.dim() == 2 else 0), bits_row_mod
  vout = None
  if vcol is not None:                                # one more column as a vector -> fp32 [M] (returned in place of Cf)
    vout = torch.full((M,), float('nan'), dtype=torch.float32)
    a.vcol, a.vcol_out = ptr(vcol), ptr(vout)
    if vcol_bias is not None:
      a.vcol_bias = ptr(vcol_bias)
  sim_check(lib, lib.mnr_gemm_nt_bf16(C.byref(a), None))
  return Cb, (vout if vcol is not None else Cf), bits


def sim_gemm_tn(lib, A, B, C_acc, bias_out=None, k_valid=None, n_valid=None, a_layout=0, b_layout=0, gcol=None, gcol_out=None,
                m_interleave=False, max_wgs=0, rank1=None):
  """C_acc[K,N] += A^T B (and bias_out += column sums of B) through the simulated mnr_gemm_tn_bf16.  a_layout / b_layout = 1:
  the operand is given in MNR_LAYOUT_PANEL storage; gcol [M] bf16 / gcol_out [K] fp32: the extra column of B.
  rank1 = (g [M] fp32, w [N] fp32, bits [M, >= N / 8] uint8) with B = None: B is built inside the kernel."""
  M, K = A.shape
  N = B.shape[1] if rank1 is None else rank1[1].numel()
  a = L.GemmTNArgs()
  a.a_layout, a.b_layout = a_layout, b_layout
  a.m_interleave, a.max_wgs = int(m_interleave), int(max_wgs)
  if gcol is not None:
    a.gcol, a.gcol_out = ptr(gcol), ptr(gcol_out)
  a.A, a.lda, a.K = ptr(A), A.stride(0), K
  if rank1 is None:
    a.B, a.ldb, a.N = ptr(B), B.stride(0), N
  else:
    a.B, a.ldb, a.N = None, N, N
    a.rank1_g, a.rank1_w, a.rank1_bits, a.ld_rank1_bits = ptr(rank1[0]), ptr(rank1[1]), ptr(rank1[2]), rank1[2].stride(0)
  a.M, a.C, a.ldc = M, ptr(C_acc), C_acc.stride(0)
  a.k_valid = K if k_valid is None else k_valid
  a.n_valid = N if n_valid is None else n_valid
  if bias_out is not None:
    a.bias_out, a.bias_n_valid = ptr(bias_out), bias_out.numel()
  sim_check(lib, lib.mnr_gemm_tn_bf16(C.byref(a), None))


class simulated_device:
  """Context manager: route the package's C-ABI calls to the simulator build and let it take host tensors.

  Test harness only.  Inside it `multinerf_amd._lib.load()` hands out libmnerf_sim.so (same C ABI, every csrc/*.hip
  compiled for the host against tools/hipsim), launches get a null stream and the "must be a device tensor" checks
  accept CPU tensors, so that Model.build('cpu') / model.apply / train_step run the product's host orchestration and
  kernel SOURCE on the CPU.  Nothing in the package knows about this; outside the context everything is as shipped
  (no CPU fallback)."""

  def __enter__(self):
    from multinerf_amd import ops
    self.lib = load_sim()
    missing = [n for n in L._PROTOS if not hasattr(self.lib, n)]
    if missing:
      raise RuntimeError(f'simulator build lacks {missing}')
    self.lib_f32 = load_sim(f32=True)
    self.saved = (L._lib, ops._stream, ops._on_device, L._lib_f32)
    L._lib = self.lib
    L._lib_f32 = self.lib_f32
    ops._stream = lambda: None
    ops._on_device = lambda t: True
    self.lib.hipsim_reset(0, 0)
    self.lib_f32.hipsim_reset(0, 0)
    return self

  def check(self):
    for lib in (self.lib, self.lib_f32):
      if lib.hipsim_failed():
        raise RuntimeError('hipsim: ' + lib.hipsim_error().decode())

  def __exit__(self, *exc):
    from multinerf_amd import ops
    L._lib, ops._stream, ops._on_device, L._lib_f32 = self.saved
    self.lib.mnr_gemm_nt_set_persistent(1)
    self.lib.mnr_gemm_nt_set_wres(1)
    return False
