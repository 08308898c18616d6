"""Build tools/hipsim/_build/libmnerf_sim.so: csrc/*.hip compiled for the HOST against the hipsim shim.

    python tools/hipsim/build.py [--force]

Plain clang++ (the one hipcc drives), `-I tools/hipsim` first so that <hip/hip_runtime.h> resolves to the shim; the
kernel sources are the product's own files.  The result exports the same C ABI as libmnerf_hip.so for the translation
units listed in SOURCES, taking HOST pointers.  Test infrastructure only (see hip/hip_runtime.h).
"""

import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'multinerf_amd', 'csrc')
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'libmnerf_sim.so')
LIB_F32 = os.path.join(OUT_DIR, 'libmnerf_sim_f32.so')       # the fp32-Dense debug build (multinerf_amd/build.py), simulated
SOURCES = ['api.hip', 'gemm.hip', 'gemm_blk.hip', 'fused_mlp.hip', 'resample.hip', 'features.hip', 'render.hip', 'losses.hip', 'optim.hip', 'refnerf.hip', 'camera.hip']
SOURCES_F32 = [s for s in SOURCES if s not in ('gemm_blk.hip', 'fused_mlp.hip')]
DEPS = [os.path.join(CSRC, 'dense_f32.inc'), os.path.join(HERE, 'hipsim.cpp'), os.path.join(HERE, 'selftest.hip'), os.path.join(HERE, 'hip', 'hip_runtime.h'), os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'gemm_nt_body.inc'), os.path.join(CSRC, 'gemm_tn_body.inc'), os.path.join(CSRC, 'gemm_nt_side.inc'), os.path.join(CSRC, 'ray_losses.h'), os.path.join(CSRC, 'ipe_math.h'),
        os.path.join(ROOT, 'include', 'mnerf.h'), os.path.join(ROOT, 'include', 'mnerf_debug.h')]
FLAGS = ['-std=c++17', '-O0', '-fPIC', '-Wno-psabi', '-I', HERE, '-I', CSRC, '-Wall', '-Wno-unused-function', '-Wno-unused-variable',
         '-Wno-unused-but-set-variable', '-Wno-unknown-pragmas', '-Wno-pass-failed']


def _clang():
  for c in (os.environ.get('HIPSIM_CXX'), '/opt/rocm/lib/llvm/bin/clang++', 'clang++'):
    if c and (not os.path.isabs(c) or os.path.exists(c)):
      return c
  raise RuntimeError('clang++ not found')


def _digest():
  h = hashlib.sha256()
  for p in [os.path.join(CSRC, s) for s in SOURCES] + DEPS:
    with open(p, 'rb') as f:
      h.update(f.read())
  h.update(' '.join(FLAGS + SOURCES).encode())
  return h.hexdigest()


def build(force=False, verbose=True, f32=False):
  lib = LIB_F32 if f32 else LIB
  stamp = lib + '.stamp'
  if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == _digest():
    return lib
  os.makedirs(OUT_DIR, exist_ok=True)
  cxx = _clang()
  sfx = '_f32' if f32 else ''

  def compile_one(path):
    obj = os.path.join(OUT_DIR, os.path.basename(path).rsplit('.', 1)[0] + sfx + '.o')
    cmd = [cxx] + FLAGS + (['-DMNR_DENSE_F32=1'] if f32 else []) + ['-x', 'c++', '-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError(f'{os.path.basename(path)}:\n{r.stdout}\n{r.stderr}')
    return obj

  paths = [os.path.join(CSRC, s) for s in (SOURCES_F32 if f32 else SOURCES)] + [os.path.join(HERE, 'selftest.hip'), os.path.join(HERE, 'hipsim.cpp')]
  with concurrent.futures.ThreadPoolExecutor(max_workers=len(paths)) as ex:
    objs = list(ex.map(compile_one, paths))
  r = subprocess.run([cxx, '-shared', '-fPIC', '-o', lib] + objs, capture_output=True, text=True)
  if r.returncode != 0:
    raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
  with open(stamp, 'w') as f:
    f.write(_digest())
  if verbose:
    print(f'built {lib} ({os.path.getsize(lib)} bytes)')
  return lib


if __name__ == '__main__':
  build(force='--force' in sys.argv)
  build(force='--force' in sys.argv, f32=True)
