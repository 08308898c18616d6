"""Per-preset maxima of the composed parity distances in a `pytest -m gpu -s` log (tests/test_gpu_model.py prints them): the
"measured" columns of DESIGN.md section 2 and of the TOL / TOL32 tables.

    python tools/parity_summary.py profiles/r4f_gpu_suite_s.log
"""
import collections
import re
import sys

mx = collections.defaultdict(lambda: collections.defaultdict(float))
name = None
for line in open(sys.argv[1], errors='replace'):
  m = re.search(r'(\w+) rand=\w+ level (\d+): \|sdist - oracle_bf16\| = (\S+) \(bf16 cost \S+\) FP32DIST sdist (\S+)', line)
  if m:
    name = m.group(1)
    if int(m.group(2)) > 0:
      mx[name]['sdist'] = max(mx[name]['sdist'], float(m.group(3)))
      mx[name]['sdist32'] = max(mx[name]['sdist32'], float(m.group(4)))
    continue
  m = re.search(r'weights err (\S+) \(bf16 cost \S+\) FP32DIST weights (\S+)', line)
  if m and name:
    mx[name]['weights'] = max(mx[name]['weights'], float(m.group(1)))
    mx[name]['weights32'] = max(mx[name]['weights32'], float(m.group(2)))
    continue
  m = re.search(r'(\w+) rand=\w+: rgb \|kernel - oracle_bf16\| = (\S+); bf16 cost .* = (\S+); \|kernel - oracle_fp32\| = (\S+)', line)
  if m:
    mx[m.group(1)]['rgb'] = max(mx[m.group(1)]['rgb'], float(m.group(2)))
    mx[m.group(1)]['rgb32'] = max(mx[m.group(1)]['rgb32'], float(m.group(4)))
    continue
  m = re.search(r'(\w+) (\w+): grad cos (\S+) rel err (\S+) \(bf16 cost \S+\) FP32DIST grad (\S+)', line)
  if m:
    mx[m.group(1)]['grad'] = max(mx[m.group(1)]['grad'], float(m.group(4)))
    mx[m.group(1)]['grad32'] = max(mx[m.group(1)]['grad32'], float(m.group(5)))
print('| preset | vs bf16-emulating oracle: sdist / weights / rgb / grad | vs plain fp32 oracle: sdist / weights / rgb / grad |')
print('|---|---|---|')
for n, d in sorted(mx.items()):
  print(f"| `{n}` | {d['sdist']:.1e} / {d['weights']:.1e} / {d['rgb']:.1e} / {d['grad']:.1e} | "
        f"{d['sdist32']:.1e} / {d['weights32']:.1e} / {d['rgb32']:.1e} / {d['grad32']:.1e} |")

# ---- the fp32-Dense debug mode (tests/test_gpu_fp32_mode.py prints `F32MODE <preset>[...]` lines): maxima per preset against the
# oracle in float64, with the plain fp32 oracle's own distance from float64 next to them
f32 = collections.defaultdict(lambda: collections.defaultdict(float))
sides = [0, 0, 0.0]
for line in open(sys.argv[1], errors='replace'):
  m = re.search(r'F32MODE (sampling )?(\w+).* level (\d+): \|sdist - oracle_fp64\| (\S+), \|weights - oracle_fp64\| (\S+)', line)
  if m:
    k = ('sampling ' if m.group(1) else '') + m.group(2)
    f32[k]['sdist'] = max(f32[k]['sdist'], float(m.group(4)))
    f32[k]['weights'] = max(f32[k]['weights'], float(m.group(5)))
    continue
  m = re.search(r'F32MODE (sampling )?(\w+).*: \|rgb - oracle_fp64\| (\S+)', line)
  if m:
    k = ('sampling ' if m.group(1) else '') + m.group(2)
    f32[k]['rgb'] = max(f32[k]['rgb'], float(m.group(3)))
    continue
  m = re.search(r'F32MODE (sampling )?(\w+).* (\w+): gradient \|kernel_fp32 - oracle_fp64\| (\S+) \(\|oracle_fp32 - oracle_fp64\| (\S+)\)', line)
  if m:
    k = ('sampling ' if m.group(1) else '') + m.group(2)
    f32[k]['grad'] = max(f32[k]['grad'], float(m.group(4)))
    f32[k]['cost32'] = max(f32[k]['cost32'], float(m.group(5)))
    f32[k]['n'] += 1
    continue
  m = re.search(r'RELU_SIDES: (\d+) of (\d+) units .* \|z\| among them (\S+)\)', line)
  if m:
    sides = [sides[0] + int(m.group(1)), sides[1] + int(m.group(2)), max(sides[2], float(m.group(3)))]
if f32:
  print()
  print('| fp32-Dense mode, preset (max over its cases) | sdist / weights / rgb vs float64 oracle | gradient, relative L2 per module vs float64 oracle | the plain fp32 oracle vs its float64 self |')
  print('|---|---|---|---|')
  for n, d in sorted(f32.items()):
    print(f"| `{n}` ({int(d['n'])} module gradients) | {d['sdist']:.1e} / {d['weights']:.1e} / {d['rgb']:.1e} | {d['grad']:.1e} | {d['cost32']:.1e} |")
  print(f'\nReLU kinks: {sides[0]} of {sides[1]} units took the other side in fp32 (largest float64 |z| among them {sides[2]:.1e}).')
