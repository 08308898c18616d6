#!/bin/bash
# Round-end evidence on ONE lease (bash tools/final_round.sh <tag>): the driver's two commands verbatim, then the profile round (+ one-stream kernel stats)
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/${TAG}_driver_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_driver_pytest.log
python3 -c 'import sys; sys.path.insert(0,"."); import __graft_entry__ as e; e.smoke()' > $OUT/${TAG}_driver_smoke.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_driver_smoke.log
tail -3 $OUT/${TAG}_driver_pytest.log; tail -4 $OUT/${TAG}_driver_smoke.log
bash tools/profile_round.sh $TAG
export TMPDIR=/tmp
export MNR_SKIP_PREFLIGHT=1   # (rocprofv3 follows the preflight child, and its --stats database then holds that process only)
cd /tmp
MNR_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_serial_prof -- python $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux > $OUT/${TAG}_serial_prof.log 2>&1
python $R/tools/prof_summary.py stats $OUT/${TAG}_serial_prof --title "rocprofv3 --kernel-trace --stats ($TAG, one stream)" --command "MNR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux" > $OUT/${TAG}_serial_kernel_stats.md
rm -rf $OUT/${TAG}_serial_prof
head -14 $OUT/${TAG}_serial_kernel_stats.md
