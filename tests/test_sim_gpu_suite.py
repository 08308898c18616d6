"""The `-m gpu` tests' own code, run on the kernel-source simulator in a child pytest (MNR_TESTS_ON_SIMULATOR=1).

tests/conftest.py then keeps 'cuda' tensors on the host and hands the package the simulator build of the C ABI, so the
parity tests written for the MI355X (every kernel against the oracle, bit-exact sample indices, Ref-NeRF heads, camera
rays against the reference goldens, composed model cases) execute unchanged against the kernel SOURCE.  The default CPU
run takes the kernel-level files (the composed path is tests/test_sim_model.py's); everything, including all of
tests/test_gpu_model.py (26 cases, ~12 min), runs with

    MNR_TESTS_ON_SIMULATOR=1 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_refnerf.py tests/test_gpu_camera.py tests/test_gpu_model.py -m gpu

A development screen, not evidence about the hardware: host libm instead of the device's transcendental units, no timing,
sequentially consistent memory apart from the LDS-DMA landing modes of tools/hipsim.
"""

import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not (shutil.which('clang++') or os.path.exists('/opt/rocm/lib/llvm/bin/clang++')),
                                reason='needs clang++')


def _child(args, timeout):
  env = dict(os.environ, MNR_TESTS_ON_SIMULATOR='1')
  cmd = [sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', '-n', '4'] + args
  r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
  tail = '\n'.join((r.stdout + r.stderr).splitlines()[-25:])
  assert r.returncode == 0, tail
  return tail


def test_kernel_level_gpu_tests_pass_on_the_simulator():
  # (the trunk-shaped GEMM cases need a real device: their fp64 references are 2 x 10^11 MACs; so does the head-shaped one; the panel kernel's
  # bitwise cases are four minutes of simulator time and tests/test_sim_gemm.py runs that kernel against the tiled one)
  tail = _child(['tests/test_gpu_kernels.py', 'tests/test_gpu_refnerf.py', 'tests/test_gpu_camera.py', '-k', 'not trunk_shapes and not head_shape and not panel_kernel_is_bitwise and not proposal_level_rows'], 1500)
  assert ' passed' in tail and 'failed' not in tail and 'skipped' not in tail, tail


def test_in_kernel_ipe_chain_passes_on_the_simulator():
  """The inference chain with the in-kernel IPE producer (mnr_mlp_chain_fwd_ipe) against mnr_cast_rays_ipe +
  mnr_mlp_chain_fwd, leaf level and through Model.__call__ (tests/test_gpu_chain.py, ~25 s of simulator time); the chain
  kernels' persistent loops over several tiles per workgroup, bit for bit (~60 s)."""
  tail = _child(['tests/test_gpu_chain.py', '-k', 'in_kernel_ipe or small_grids'], 900)
  assert ' passed' in tail and 'failed' not in tail and 'skipped' not in tail, tail


def test_sampling_gradient_kernels_pass_on_the_simulator():
  """Model.stop_level_grad = False (round 5): mnr_resample_level_bwd, mnr_cast_rays_ipe_bwd, mnr_sdist_bwd + the level kernels'
  g_x output against the oracle's autograd, and the composed train steps (at the simulator's reduced widths), with the code of
  tests/test_gpu_sampling_grad.py unchanged."""
  tail = _child(['tests/test_gpu_sampling_grad.py'], 1500)
  assert ' passed' in tail and 'failed' not in tail and 'skipped' not in tail, tail
