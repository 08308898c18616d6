"""Alternate build of ONE translation unit for same-box A/B probes.

    python tools/build_variant.py <tag> <file.hip> [extra hipcc flags...]
      -> multinerf_amd/libmnerf_hip_<tag>.so  (the other objects are the product's; select it with MNR_LIB_PATH)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multinerf_amd import build  # noqa: E402


def main():
  tag, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
  build.build(verbose=False)
  hipcc = build._hipcc()
  var_obj = os.path.join(build.OBJ_DIR, src.replace('.hip', f'.{tag}.o'))
  subprocess.run([hipcc] + build.FLAGS + extra + ['-c', os.path.join(build.CSRC, src), '-o', var_obj], check=True)
  objs = [var_obj if s == src else os.path.join(build.OBJ_DIR, s.replace('.hip', '.o')) for s in build.SOURCES]
  out = os.path.join(os.path.dirname(build.LIB), f'libmnerf_hip_{tag}.so')
  subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, check=True)
  print('built', out)


if __name__ == '__main__':
  main()
