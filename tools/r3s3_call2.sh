#!/bin/bash
# round 3, session 3, call 2: optimised in-kernel IPE (direction fixed per thread, exact one-FMA wrap): parity, render A/B, kernel stats of both
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_model.py -x -q -m gpu -k "in_kernel_ipe or forward_parity" > $OUT/r3s3_tests2.log 2>&1
tail -3 $OUT/r3s3_tests2.log
grep -E "layer 0:|head:|rgb \|" $OUT/r3s3_tests2.log
timeout 600 python tools/render_probe.py > $OUT/r3s3_render_probe2.txt 2>&1
tail -24 $OUT/r3s3_render_probe2.txt
cd /tmp
for m in on off; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r3s3_render_$m -- python $R/tools/render_probe.py --only $m --reps 10 > $OUT/r3s3_render_${m}_prof.log 2>&1
  python $R/tools/prof_summary.py stats $OUT/r3s3_render_$m --title "rocprofv3 --kernel-trace --stats (render, 16384-ray chunk of 360.gin, in-kernel IPE $m)" --command "rocprofv3 --kernel-trace --stats -- python tools/render_probe.py --only $m --reps 10" > $OUT/r3s3_render_${m}_kernel_stats.md
  rm -rf $OUT/r3s3_render_$m
  head -22 $OUT/r3s3_render_${m}_kernel_stats.md | cut -c1-160
done
