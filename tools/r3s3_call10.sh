#!/bin/bash
# round 3, session 3, call 10: the launches of one train step in order, one stream (clean durations) and the default two streams
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for m in serial default; do
  E=""; [ $m = serial ] && E="MNR_SIDE_STREAM=0"
  env $E timeout 600 rocprofv3 --kernel-trace -d $OUT/r3s3_seq_$m -- python $R/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_aux > $OUT/r3s3_seq_${m}.log 2>&1
  python $R/tools/prof_summary.py seq $OUT/r3s3_seq_$m --title "launches of one 360.gin train step in start order ($m streams)" --command "$E rocprofv3 --kernel-trace -- python bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_aux" > $OUT/r3s3_step_seq_${m}.md
  python $R/tools/prof_summary.py stats $OUT/r3s3_seq_$m --title "kernel stats ($m streams)" > $OUT/r3s3_step_stats_${m}.md
  rm -rf $OUT/r3s3_seq_$m
done
head -5 $OUT/r3s3_step_seq_serial.md
