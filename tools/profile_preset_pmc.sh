#!/bin/bash
# HBM traffic of one preset's train step from the PMC counters:  bash tools/profile_preset_pmc.sh <tag> <bench.py args...>
#   -> gpurun_out/<tag>_pmc_FETCH_SIZE.md, <tag>_pmc_WRITE_SIZE.md (separate rocprofv3 --pmc passes, as MI355X_MICROARCH.md's HBM
#      section prescribes: FETCH_SIZE doubled on gfx950, WRITE_SIZE as reported) and <tag>_traffic.txt: GB per step over ALL kernels,
#      and the rate that is over the un-profiled step time of <tag>_bench.json (tools/profile_preset.sh, run first).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp MNR_SKIP_PREFLIGHT=1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -- python $R/bench.py "$@" --steps 2 --warmup 1 --no_cpu_baseline --no_aux > $OUT/${TAG}_pmc_$C.log 2>&1
  python $R/tools/prof_summary.py pmc $OUT/${TAG}_pmc_$C --title "rocprofv3 --pmc $C ($TAG)" --command "rocprofv3 --kernel-trace --pmc $C -- python bench.py $* --steps 2 --warmup 1 --no_cpu_baseline --no_aux" --top 40 > $OUT/${TAG}_pmc_$C.md
  rm -rf $OUT/${TAG}_pmc_$C
done
python - "$OUT/${TAG}_pmc_FETCH_SIZE.md" "$OUT/${TAG}_pmc_WRITE_SIZE.md" "$OUT/${TAG}_bench.json" <<'PY' | tee $OUT/${TAG}_traffic.txt
import json, re, sys
def table(path):
  rows = {}
  for l in open(path):
    m = re.match(r'\| `(.+?)` \| (\d+) \| ([0-9.e+]+) \|', l)
    if m:
      rows[m.group(1)] = (int(m.group(2)), float(m.group(3)))
  return rows
f, w = table(sys.argv[1]), table(sys.argv[2])
bench = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
steps = 1 + 2 + 2          # bench.py --steps 2 --warmup 1: one warm-up, two timed, min(3, steps) = two instrumented steps
fetch = 2.0 * sum(v[1] for v in f.values()) * 1024
write = sum(v[1] for v in w.values()) * 1024
gb = (fetch + write) / steps / 1e9
ms = bench['ms_per_step']
print(f"{bench['config']['workload']}")
print(f"HBM bytes per train step, all kernels (FETCH_SIZE x 2 + WRITE_SIZE, {steps} steps in the profiled command): {gb:.1f} GB  (read {fetch / steps / 1e9:.1f} + written {write / steps / 1e9:.1f})")
print(f"un-profiled step {ms:.3f} ms -> {gb / ms:.2f} TB/s averaged over the step; rays/s {bench['value']:.0f}; MFMA frac {bench['roofline']['frac']:.3f}, whole step {bench['roofline']['whole_step_frac']:.3f}")
PY
