"""Same-box A/B of variant builds of the library (tools/build_variant.py): tools/chain_time.py for the product and every variant,
two rounds, then (--bench) one bench.py line each, product first and last.

    python tools/variant_probe.py [--bench] [--presets 360,blender_256] multinerf_amd/libmnerf_hip_<tag>.so ...
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def env_for(lib):
  e = dict(os.environ, MNR_SKIP_PREFLIGHT='1')
  e.pop('MNR_LIB_PATH', None)
  if lib:
    e['MNR_LIB_PATH'] = os.path.join(ROOT, lib)
  return e


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('libs', nargs='*')
  ap.add_argument('--bench', action='store_true')
  ap.add_argument('--presets', default='360')
  ap.add_argument('--rounds', type=int, default=2)
  ap.add_argument('--no_chain', action='store_true')
  a = ap.parse_args()
  arms = [None] + a.libs
  name = lambda l: os.path.basename(l).replace('libmnerf_hip_', '').replace('.so', '') if l else 'product'
  if not a.no_chain:
    for _ in range(a.rounds):
      for lib in arms:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'chain_time.py')], capture_output=True, text=True, env=env_for(lib))
        lines = [l for l in r.stdout.splitlines() if l.startswith('fwd train')]
        print(f'{name(lib):10s} {lines[-1] if lines else "FAILED " + r.stderr[-400:]}', flush=True)
  if a.bench:
    for preset in a.presets.split(','):
      for lib in arms + [None]:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no_cpu_baseline', '--no_aux', '--preset', preset],
                           capture_output=True, text=True, env=env_for(lib))
        try:
          b = json.loads(r.stdout.strip().splitlines()[-1])
          print(f'bench {preset:16s} {name(lib):10s} {b["value"]:.0f} rays/s {b["ms_per_step"]:.3f} ms  final_loss {b["config"]["final_loss"]:.7f}', flush=True)
        except Exception as e:
          print('bench', preset, name(lib), 'failed', e, r.stderr[-600:], flush=True)


if __name__ == '__main__':
  main()
