"""Build recipe for libmnerf_hip.so (hipcc, gfx950 only; cross-compiles without a GPU).

    python -m multinerf_amd.build            # build if stale
    python -m multinerf_amd.build --force

The shared library is written IN-TREE (multinerf_amd/libmnerf_hip.so) so that it
travels with the source snapshot to the GPU box; it is git-ignored.

A second library, libmnerf_hip_f32.so, is the fp32-Dense DEBUG build (`Model(dense_precision='fp32')`): the same kernel
sources compiled with -DMNR_DENSE_F32 (activation / gradient / packed-weight storage float instead of bf16, csrc/common.h)
and csrc/dense_f32.inc in place of the MFMA GEMMs; the layout-specific MFMA files (gemm_blk.hip, fused_mlp.hip) are not part
of it.  It exists for parity against the plain fp32 oracle and is never on bench.py's path.
"""

import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.path.join(HERE, 'libmnerf_hip.so')
LIB_F32 = os.path.join(HERE, 'libmnerf_hip_f32.so')
OBJ_DIR = os.path.join(HERE, 'build')
SOURCES = ['api.hip', 'gemm.hip', 'gemm_blk.hip', 'fused_mlp.hip', 'resample.hip', 'features.hip', 'render.hip', 'losses.hip', 'optim.hip', 'refnerf.hip', 'camera.hip']
SOURCES_F32 = [s for s in SOURCES if s not in ('gemm_blk.hip', 'fused_mlp.hip')]
F32_DEFINES = ['-DMNR_DENSE_F32=1']
HEADERS = [os.path.join(CSRC, 'dense_f32.inc'), os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'gemm_nt_body.inc'), os.path.join(CSRC, 'gemm_tn_body.inc'), os.path.join(CSRC, 'gemm_nt_side.inc'), os.path.join(CSRC, 'ray_losses.h'), os.path.join(CSRC, 'ipe_math.h'), os.path.join(INCLUDE, 'mnerf.h'), os.path.join(INCLUDE, 'mnerf_debug.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function'] + os.environ.get('MNR_EXTRA_HIPCC_FLAGS', '').split()


def _hipcc():
  for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
    if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
      return c
  raise RuntimeError('hipcc not found')


def _digest():
  h = hashlib.sha256()
  for p in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
    with open(p, 'rb') as f:
      h.update(f.read())
  h.update(' '.join(FLAGS).encode())
  return h.hexdigest()


def is_stale():
  d = _digest()
  for lib in (LIB, LIB_F32):
    stamp = lib + '.stamp'
    if not (os.path.exists(lib) and os.path.exists(stamp)):
      return True
    with open(stamp) as f:
      if f.read().strip() != d:
        return True
  return False


def build(force=False, verbose=True):
  """Compile every HIP translation unit for gfx950 and link the C-ABI library (and the fp32-Dense debug build next to it)."""
  if not force and not is_stale():
    return LIB
  os.makedirs(OBJ_DIR, exist_ok=True)
  hipcc = _hipcc()

  def compile_one(job):
    src, f32 = job
    obj = os.path.join(OBJ_DIR, src.replace('.hip', '_f32.o' if f32 else '.o'))
    cmd = [hipcc] + FLAGS + (F32_DEFINES if f32 else []) + ['-c', os.path.join(CSRC, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError(f'hipcc failed on {src}{" (fp32 build)" if f32 else ""}:\n{r.stdout}\n{r.stderr}')
    return obj

  jobs = [(s, False) for s in SOURCES] + [(s, True) for s in SOURCES_F32]
  with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(jobs), 2 * (os.cpu_count() or 4))) as ex:
    objs = list(ex.map(compile_one, jobs))
  for lib, mine in ((LIB, objs[:len(SOURCES)]), (LIB_F32, objs[len(SOURCES):])):
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + mine
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    with open(lib + '.stamp', 'w') as f:
      f.write(_digest())
    if verbose:
      print(f'built {lib} ({os.path.getsize(lib)} bytes)')
  return LIB


if __name__ == '__main__':
  build(force='--force' in sys.argv)
