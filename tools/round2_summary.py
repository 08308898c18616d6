"""Table of what tools/round2_first_call.sh left in gpurun_out/: one line per bench A/B and the probe verdicts.

    python tools/round2_summary.py [gpurun_out]
"""

import glob
import json
import os
import re
import sys


def main():
  d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
  rows = []
  for f in sorted(glob.glob(os.path.join(d, 'r2_bench_*.json'))):
    name = os.path.basename(f)[len('r2_bench_'):-len('.json')]
    try:
      lines = [l for l in open(f) if l.strip().startswith('{')]
      j = json.loads(lines[-1])
      rows.append((name, j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['gemm_ms_per_step'], j['config']['final_loss']))
    except Exception as e:          # a variant that crashed or mismatched: say so, keep going
      err = f[:-len('.json')] + '.err'
      tail = open(err).read().strip().splitlines()[-1:] if os.path.exists(err) else []
      rows.append((name, None, None, None, None, f'{type(e).__name__}: {tail}'))
  base = next((r[1] for r in rows if r[0] == 'cfg2' and r[1]), None)
  print(f'{"variant":28s} {"rays/s":>10s} {"vs cfg2":>8s} {"ms/step":>8s} {"GEMM TF/s":>10s} {"GEMM ms":>8s}  loss')
  for name, v, ms, tf, gms, loss in rows:
    if v is None:
      print(f'{name:28s} FAILED  {loss}')
    else:
      rel = f'{v / base:8.3f}' if base else '       -'
      print(f'{name:28s} {v:10.0f} {rel} {ms:8.2f} {tf:10.1f} {gms:8.2f}  {loss:.5f}')
  for f in sorted(glob.glob(os.path.join(d, 'r2_gemm_probe_*.txt'))):
    text = open(f).read()
    bad = re.findall(r'^.*MISMATCH.*$', text, flags=re.M)
    ok = len(re.findall(r'bitwise equal', text))
    print(f'\n{os.path.basename(f)}: {ok} bitwise-equal screens, {len(bad)} mismatches')
    for l in bad[:10]:
      print('   ' + l)
    for l in re.findall(r'^(?:cfg \d+ nt|tn ).*TFLOP/s$', text, flags=re.M):
      print('   ' + l)
  p = os.path.join(d, 'r2_cabi_probe.txt')
  if os.path.exists(p):
    print('\n' + os.path.basename(p))
    for l in open(p):
      if l.startswith(('cfg', 'tn ', 'wres')):
        print('   ' + l.rstrip()[:200])
  p = os.path.join(d, 'r2_ingest_probe.txt')
  if os.path.exists(p):
    print('\n' + os.path.basename(p))
    for l in open(p):
      if 'all CUs' in l or '1/8' in l:
        print('   ' + l.rstrip()[:170])


if __name__ == '__main__':
  main()
