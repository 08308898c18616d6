#!/bin/bash
# Evidence for one BASELINE config other than the headline:  bash tools/profile_preset.sh <tag> <bench.py args...>
#   -> gpurun_out/<tag>_bench.json (the un-profiled bench line, roofline + cpu_baseline) and <tag>_kernel_stats.md
#      (rocprofv3 --kernel-trace --stats of the same command with --steps 5 --warmup 2)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python bench.py "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp
MNR_SKIP_PREFLIGHT=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -- python $R/bench.py "$@" --steps 5 --warmup 2 --no_cpu_baseline --no_aux > $OUT/${TAG}_prof.log 2>&1
python $R/tools/prof_summary.py stats $OUT/${TAG}_prof --title "rocprofv3 --kernel-trace --stats ($TAG)" --command "rocprofv3 --kernel-trace --stats -- python bench.py $* --steps 5 --warmup 2 --no_cpu_baseline --no_aux" > $OUT/${TAG}_kernel_stats.md
rm -rf $OUT/${TAG}_prof
cut -c1-400 $OUT/${TAG}_bench.json
head -14 $OUT/${TAG}_kernel_stats.md | cut -c1-200
