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
