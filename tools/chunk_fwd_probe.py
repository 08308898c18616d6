"""Does walking the trunk's forward pass CHUNK by chunk (a block of rows through all layers before the next block) pay?

The 1024-wide trunk's forward GEMMs each write a 1.07 GB activation and the next layer reads it back; whole-batch launches find
only the last ~128-256 MB of it in the 256 MB Infinity Cache (the panel kernel's walk_descending).  A block of `chunk` rows keeps
its activation (chunk x 2 KB) inside the cache between layers.  This probe times 7 layers (K = N = 1024, panel storage, bias +
ReLU + 1-bit masks out, as in training) over 524288 rows: whole-batch launches against chunked launches, on one stream and on
two alternating streams with half the chip each.

    python tools/chunk_fwd_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import ops  # noqa: E402

dev = 'cuda'
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
PAN = ops.LAYOUT_PANEL
M, W, D = 524288, 1024, 7

x0 = ops.to_panel(torch.relu(torch.rand((M, W), generator=g, device=dev) * 2 - 1).to(bf))
Bt = [((torch.rand((W, W), generator=g, device=dev) * 2 - 1) * (6.0 / W) ** 0.5).to(bf) for _ in range(D)]
bias = [0.05 * torch.randn((W,), generator=g, device=dev) for _ in range(D)]
acts = [torch.empty((M, W), dtype=bf, device=dev) for _ in range(D)]
bits = [torch.empty((M * W // 8,), dtype=torch.uint8, device=dev) for _ in range(D)]


def layer(l, r0, r1, max_wgs=0, with_bits=True):
  src = x0 if l == 0 else acts[l - 1]
  ops.gemm_nt(src.view(-1)[r0 * W:r1 * W].view(r1 - r0, W), Bt[l], M=r1 - r0, N=W, K1=W, bias=bias[l], n_bias=W, relu=True,
              Cb=acts[l].view(-1)[r0 * W:r1 * W].view(r1 - r0, W), ldcb=W, nb=W,
              bits_out=bits[l][r0 * W // 8:r1 * W // 8] if with_bits else None, a1_layout=PAN, c_layout=PAN,
              walk_descending=bool(l & 1), max_wgs=max_wgs)


def whole(with_bits=True):
  for l in range(D):
    layer(l, 0, M, with_bits=with_bits)


def chunked(chunk, with_bits=True):
  for r0 in range(0, M, chunk):
    for l in range(D):
      layer(l, r0, r0 + chunk, with_bits=with_bits)


streams = [torch.cuda.Stream() for _ in range(4)]


def chunked_two_streams(chunk, half, ns=2):
  cur = torch.cuda.current_stream()
  ev = torch.cuda.Event()
  ev.record(cur)
  for s in streams:
    s.wait_event(ev)
  for i, r0 in enumerate(range(0, M, chunk)):
    with torch.cuda.stream(streams[i % ns]):
      for l in range(D):
        layer(l, r0, r0 + chunk, max_wgs=half)
  for s in streams:
    e = torch.cuda.Event()
    e.record(s)
    cur.wait_event(e)


def timed(fn, reps=5):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


whole()
ref = [a.clone() for a in acts]
print(f'whole-batch launches, 7 layers:            {timed(whole):9.1f} us', flush=True)
for chunk in (262144, 131072, 65536, 32768, 16384):
  t = timed(lambda: chunked(chunk))
  same = all(torch.equal(a.view(torch.int16), r.view(torch.int16)) for a, r in zip(acts, ref))
  print(f'chunks of {chunk:6d} rows ({chunk * W * 2 >> 20:4d} MiB per activation), one stream: {t:9.1f} us  {"bitwise equal" if same else "MISMATCH"}', flush=True)
print(f'whole-batch launches again:                {timed(whole):9.1f} us', flush=True)
for chunk in (65536, 32768):
  t = timed(lambda: chunked_two_streams(chunk, 128))
  same = all(torch.equal(a.view(torch.int16), r.view(torch.int16)) for a, r in zip(acts, ref))
  print(f'chunks of {chunk:6d} rows, two streams x 128 workgroups: {t:9.1f} us  {"bitwise equal" if same else "MISMATCH"}', flush=True)
for chunk in (131072, 65536, 32768):
  for ns in (2, 3, 4):
    t = timed(lambda: chunked_two_streams(chunk, 0, ns))
    same = all(torch.equal(a.view(torch.int16), r.view(torch.int16)) for a, r in zip(acts, ref))
    print(f'chunks of {chunk:6d} rows, {ns} streams x full grids: {t:9.1f} us  {"bitwise equal" if same else "MISMATCH"}', flush=True)
print(f'whole-batch launches again:                {timed(whole):9.1f} us', flush=True)
print(f'whole-batch, no mask output:               {timed(lambda: whole(False)):9.1f} us', flush=True)
for chunk in (65536, 32768):
  print(f'chunks of {chunk:6d} rows, no mask output:     {timed(lambda: chunked(chunk, False)):9.1f} us', flush=True)

# How much is a just-written A operand worth?  Pairs (layer 0 on block c, layer 1 on block c) against pairs whose second launch
# reads a block written half a batch earlier (cold): same launches, same bytes, only the distance between write and read differs.
for chunk in (131072, 65536, 32768):
  n = M // chunk

  def pairs(shift):
    for c in range(n):
      layer(0, c * chunk, (c + 1) * chunk)
      c2 = (c + shift) % n
      layer(1, c2 * chunk, (c2 + 1) * chunk)
  whole()                                     # (every block of acts[0] exists)
  hot, cold, hot2, cold2 = timed(lambda: pairs(0)), timed(lambda: pairs(n // 2)), timed(lambda: pairs(0)), timed(lambda: pairs(n // 2))
  print(f'chunk {chunk:6d}: layer 1 reads what layer 0 just wrote {hot:8.1f} / {hot2:8.1f} us; reads a block written half a batch ago {cold:8.1f} / {cold2:8.1f} us', flush=True)
