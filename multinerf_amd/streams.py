"""Side-by-side HIP streams for the backward pass.

With stop_level_grad (reference models.py:200-201) the sampling levels are independent in the backward pass: the proposal
levels' backward is HBM-bound (activations, gradients and features streamed once each), the NeRF level's is MFMA- and
power-bound.  `BackwardStreams` runs the two on separate HIP streams; with a CU split each stream is created with a CU
mask (mnr_stream_create_cu_mask = hipExtStreamCreateWithCUMask) and the persistent kernels launched on it are sized for
its share (mnr_set_cu_budget), so that neither side's workgroups queue behind the other's.

Environment (A/B switches; the default is what measured best, DESIGN.md section 6: round 3, same box, 360.gin at
16384 rays: one stream 487.6 k rays/s, plain side stream 500.5 k, CU-masked 32 / 64 CUs 466 k / 480 k):
  MNR_SIDE_STREAM = 1 | 0      proposal-level backward on its own stream (default on)
  MNR_SIDE_CUS    = k          k > 0: CU-masked pair, the proposal side gets k CUs (a multiple of 8), the NeRF side the rest;
                               0: two plain streams, every launch sized for the whole chip
"""

import ctypes as C
import os

import torch

from multinerf_amd import _lib as L


def _mask_words(total_cus, lo, hi):
  """Bits [lo, hi) of a `total_cus`-bit mask as uint32 words."""
  words = (total_cus + 31) // 32
  arr = (C.c_uint32 * words)()
  for i in range(lo, hi):
    arr[i // 32] |= 1 << (i % 32)
  return arr, words


def create_masked_stream(device, lo, hi):
  """torch view of a HIP stream restricted to CUs [lo, hi) (driver numbering: consecutive indices go round the XCDs, so a
  range that is a multiple of 8 wide takes the same number of CUs from every XCD)."""
  lib = L.load()
  total = torch.cuda.get_device_properties(device).multi_processor_count
  if not (0 <= lo < hi <= total):
    raise ValueError(f'CU range [{lo}, {hi}) outside the device\'s {total} CUs')
  arr, words = _mask_words(total, lo, hi)
  out = C.c_void_p()
  with torch.cuda.device(device):
    L.check(lib.mnr_stream_create_cu_mask(arr, words, C.byref(out)))
  return torch.cuda.ExternalStream(out.value, device=device)


class BackwardStreams:
  """The pair of streams (and CU budgets) the train step uses for the NeRF level's and the proposal levels' backward."""

  def __init__(self, device, side_cus=0):
    self.device = torch.device(device)
    self.total = torch.cuda.get_device_properties(self.device).multi_processor_count
    self.side_cus = int(side_cus)
    if self.side_cus:
      if self.side_cus % 8 or not 8 <= self.side_cus <= self.total - 8:
        raise ValueError(f'MNR_SIDE_CUS must be a multiple of 8 in [8, {self.total - 8}], is {self.side_cus}')
      self.prop = create_masked_stream(self.device, 0, self.side_cus)
      self.nerf = create_masked_stream(self.device, self.side_cus, self.total)
      self.prop_budget, self.nerf_budget = self.side_cus, self.total - self.side_cus
    else:
      self.prop = torch.cuda.Stream(device=self.device)
      self.nerf = None                                   # the NeRF level stays on the caller's stream
      self.prop_budget = self.nerf_budget = 0
    # MNR_SIDE_STREAMS = 2: every proposal level on a stream of its own (A/B switch; plain streams only)
    n = int(os.environ.get('MNR_SIDE_STREAMS', '1'))
    self.props = [self.prop] + [torch.cuda.Stream(device=self.device) for _ in range(max(0, n - 1) if not self.side_cus else 0)]

  @staticmethod
  def from_env(device):
    if os.environ.get('MNR_SIDE_STREAM', '1') == '0':
      return None
    return BackwardStreams(device, int(os.environ.get('MNR_SIDE_CUS', '0')))

  def describe(self):
    return {'side_stream': True, 'side_cus': self.side_cus, 'nerf_cus': self.total - self.side_cus if self.side_cus else self.total}


def describe_env():
  """What MNR_SIDE_STREAM / MNR_SIDE_CUS ask for (bench.py records it next to its numbers)."""
  on = os.environ.get('MNR_SIDE_STREAM', '1') != '0'
  return {'side_stream': on, 'side_cus': int(os.environ.get('MNR_SIDE_CUS', '0')) if on else 0}


class budget:
  """`with budget(n):` the persistent launches inside are sized for n CUs (0 = the whole chip)."""

  def __init__(self, n):
    self.n = int(n)

  def __enter__(self):
    if self.n:
      L.check(L.load().mnr_set_cu_budget(self.n))

  def __exit__(self, *exc):
    if self.n:
      L.check(L.load().mnr_set_cu_budget(0))
    return False
