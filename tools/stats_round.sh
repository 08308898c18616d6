#!/bin/bash
# Kernel stats of the headline (two streams and one) and the bench lines + kernel stats of the other BASELINE presets:
#   bash tools/stats_round.sh <tag>     -> gpurun_out/<tag>_kernel_stats.md, <tag>_serial_kernel_stats.md, <tag>_<preset>_*
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export MNR_SKIP_PREFLIGHT=1      # (rocprofv3 follows the preflight child, and its --stats database then holds that process only)
cd /tmp
for arm in "" serial; do
  if [ -z "$arm" ]; then E=""; N=${TAG}; T="$TAG"; else E="MNR_SIDE_STREAM=0"; N=${TAG}_serial; T="$TAG, one stream"; fi
  env $E timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${N}_prof -- python $R/bench.py --steps 10 --warmup 2 --no_cpu_baseline --no_aux > $OUT/${N}_prof.log 2>&1
  python $R/tools/prof_summary.py stats $OUT/${N}_prof --title "rocprofv3 --kernel-trace --stats ($T)" --command "$E rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no_cpu_baseline --no_aux" > $OUT/${N}_kernel_stats.md
  rm -rf $OUT/${N}_prof
  head -16 $OUT/${N}_kernel_stats.md | cut -c1-160
done
cd $R
for p in blender_256 llff_raw blender_refnerf; do
  bash tools/profile_preset.sh ${TAG}_$p --preset $p --no_aux | tail -8 | cut -c1-200
done
