#!/bin/bash
# Same-box A/B of bench.py under different environment switches:
#   bash tools/ab_bench.sh <tag> "name1:VAR=1 VAR2=x" "name2:" ...      (writes gpurun_out/<tag>_ab.jsonl + a summary)
# Every arm is the default `bench.py --no_cpu_baseline --no_aux` line (20 timed steps after 5 warm-up); arms run back to
# back on one box, which is the only comparison DESIGN.md quotes (box-to-box spread is +-3 %).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
: > $OUT/${TAG}_ab.jsonl
EXTRA=${AB_BENCH_ARGS:-}
for arm in "$@"; do
  name=${arm%%:*}
  envs=${arm#*:}
  line=$(cd $R && env $envs timeout 600 python bench.py --no_cpu_baseline --no_aux $EXTRA 2> $OUT/${TAG}_ab_${name}.err | tail -1)
  echo "{\"arm\": \"$name\", \"env\": \"$envs\", \"line\": ${line:-null}}" >> $OUT/${TAG}_ab.jsonl
done
python - "$OUT/${TAG}_ab.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
  d = json.loads(l)
  b = d['line']
  if not b:
    print(f"{d['arm']:24s} FAILED ({d['env']})")
    continue
  r = b['roofline']
  print(f"{d['arm']:24s} {b['value']:10.0f} rays/s  {b['ms_per_step']:7.3f} ms (median {b.get('ms_per_step_median_hip_event', 0):7.3f})  "
        f"mfma busy {r['gemm_ms_per_step']:6.2f} ms  frac {r['frac']:.3f} whole-step {r['whole_step_frac']:.3f}   [{d['env']}]")
PY
