"""Append the arms of tools/ab_bench.sh runs (gpurun_out/<tag>_ab.jsonl) to profiles/r3_ab.md as tables:

    python tools/ab_to_md.py "<section title>" gpurun_out/<tag>_ab.jsonl [more.jsonl ...] >> profiles/r3_ab.md
"""
import json
import sys

title, files = sys.argv[1], sys.argv[2:]
print(f'\n## {title}\n')
print('| arm | environment | rays/s | ms/step (wall / median of per-step HIP events) | MFMA busy ms | frac | whole-step frac |')
print('|---|---|---|---|---|---|---|')
for f in files:
  for line in open(f):
    d = json.loads(line)
    b = d['line']
    if not b:
      print(f"| {d['arm']} | `{d['env'] or '(default)'}` | FAILED | | | | |")
      continue
    r = b['roofline']
    print(f"| {d['arm']} | `{d['env'] or '(default)'}` | {b['value']:,.0f} | {b['ms_per_step']:.3f} / {b.get('ms_per_step_median_hip_event', 0):.3f} | "
          f"{r['gemm_ms_per_step']:.2f} | {r['frac']:.3f} | {r['whole_step_frac']:.3f} |")
