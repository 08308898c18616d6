"""Shader clock and board power under the trunk GEMMs (tools/clock_probe.hip: a sampler kernel that runs NEXT to the GEMMs).

    python tools/clock_probe.py            # the forward stack of tools/chunk_probe.py, whole launches
    MNR_LIB_PATH=.../libmnerf_hip_pn11.so python tools/clock_probe.py    # A operand out of the Infinity Cache (probe build)
"""
import ctypes
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multinerf_amd import ops  # noqa: E402

dev, bf = 'cuda', torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
M, W, L = 524288, 1024, 6
PAN = ops.LAYOUT_PANEL
acts = [torch.relu(torch.rand((M * W,), generator=g, device=dev) * 2 - 1).to(bf)] + [torch.zeros((M * W,), dtype=bf, device=dev) for _ in range(L)]
bits = [torch.zeros((M * W // 8,), dtype=torch.uint8, device=dev) for _ in range(L)]
Bts = [((torch.rand((W, W), generator=g, device=dev) * 2 - 1) * (6.0 / W) ** 0.5).to(bf) for _ in range(L)]
biases = [0.05 * torch.randn((W,), generator=g, device=dev) for _ in range(L)]


def stack():
  for l in range(L):
    ops.gemm_nt(acts[l], Bts[l], M=M, N=W, K1=W, lda1=W, bias=biases[l], n_bias=W, relu=True, Cb=acts[l + 1], ldcb=W, nb=W,
                bits_out=bits[l], a1_layout=PAN, c_layout=PAN)


lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_bin', 'libclock_probe.so'))
lib.clock_sampler_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
WGS, N, PERIOD = 8, 1200, 1000                      # 8 samplers (one per XCD), a pair every 10 us, at most 12 ms
out = torch.zeros((WGS, N, 2), dtype=torch.int64, device=dev)
stop = torch.zeros((64,), dtype=torch.int32, device=dev)
side = torch.cuda.Stream()

for _ in range(3):
  stack()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
  assert lib.clock_sampler_launch(side.cuda_stream, out.data_ptr(), WGS, N, PERIOD, stop.data_ptr()) == 0
time.sleep(0.002)                                   # ~2 ms of idle samples first
e0.record()
stack()
e1.record()
stop.fill_(1)
torch.cuda.synchronize()
t_stack = e0.elapsed_time(e1) * 1e3
o = out.cpu().numpy().astype(np.float64)
print(f'{L} layers in {t_stack:.0f} us = {t_stack / L:.1f} us per layer')
for w in range(WGS):
  rt, st = o[w, :, 0], o[w, :, 1]
  k = int((rt > 0).sum())
  rt, st = rt[:k], st[:k]
  mhz = np.diff(st) / np.diff(rt) * 100.0
  # the GEMMs are the stretch of `t_stack` us that ends with the last samples
  n_g = int(t_stack / (PERIOD / 100.0))
  busy, idle = mhz[max(0, k - 1 - n_g):], mhz[:max(1, k - 1 - n_g - 20)]
  print(f'  sampler {w}: {k} samples; shader clock idle {np.median(idle):7.0f} MHz, under the GEMMs median {np.median(busy):7.0f} (10 % {np.percentile(busy, 10):7.0f}, 90 % {np.percentile(busy, 90):7.0f})')

# board power while the stack runs for ~3 s (rocm-smi samples in a thread)
samples = []
run = [True]


def poll():
  while run[0]:
    try:
      r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--csv'], capture_output=True, text=True, timeout=5)
      samples.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else '')
    except Exception as e:  # noqa: BLE001
      samples.append(repr(e))
    time.sleep(0.05)


th = threading.Thread(target=poll)
th.start()
t0 = time.time()
while time.time() - t0 < 3.0:
  for _ in range(20):
    stack()
  torch.cuda.synchronize()
run[0] = False
th.join()
hdr = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--csv'], capture_output=True, text=True).stdout.strip().splitlines()
print('rocm-smi columns:', hdr[0] if hdr else '?')
for s in samples[1:8]:
  print('  under load:', s)
print('  idle      :', hdr[-1] if hdr else '?')
