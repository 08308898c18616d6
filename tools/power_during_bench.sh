#!/bin/bash
# Package power and clocks while bench.py's train steps run (rocm-smi polled every ~0.1 s next to a 300-step run):
#   bash tools/power_during_bench.sh <tag> [bench.py args]   -> gpurun_out/<tag>_power.txt
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
( python bench.py --steps 300 --warmup 5 --no_cpu_baseline --no_aux "$@" > $OUT/${TAG}_power_bench.json 2> $OUT/${TAG}_power_bench.err ) &
BP=$!
: > $OUT/${TAG}_power_raw.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --csv 2>/dev/null | grep '^card' | tail -n 1 >> $OUT/${TAG}_power_raw.txt
  sleep 0.1
done
wait $BP
python - "$OUT/${TAG}_power_raw.txt" "$OUT/${TAG}_power_bench.json" > $OUT/${TAG}_power.txt <<'PY'
import json, re, sys
rows = []
for l in open(sys.argv[1]):
  f = l.strip().split(',')
  try:
    rows.append((float(f[-1]), int(re.sub(r'\D', '', f[5]))))
  except Exception:
    pass
b = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
print(f"bench.py --steps 300: {b['value']:.0f} rays/s, {b['ms_per_step']:.3f} ms/step; {len(rows)} rocm-smi samples (columns: package W, sclk MHz)")
busy = [r for r in rows if r[0] > 1000]
print(f"samples above 1000 W: {len(busy)}; package power median {sorted(p for p, _ in busy)[len(busy) // 2] if busy else 0:.0f} W, max {max((p for p, _ in rows), default=0):.0f} W; "
      f"sclk median under load {sorted(c for _, c in busy)[len(busy) // 2] if busy else 0} MHz; idle samples: {[r for r in rows if r[0] < 700][:3]}")
print('trace (W, MHz):', ' '.join(f'{p:.0f}/{c}' for p, c in rows))
PY
cat $OUT/${TAG}_power.txt | cut -c1-1200
