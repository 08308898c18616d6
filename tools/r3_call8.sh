#!/bin/bash
# robustness of the documented switches + serial-stream kernel stats of the headline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for envs in "MNR_SIDE_STREAM=0" "MNR_DW_STREAM=1" "MNR_SIDE_STREAMS=2" "MNR_CHAIN_DEFER=0" "MNR_FUSED_CHAIN=0" "MNR_QUAD_LDS_MAX=40960" "MNR_NT_STORES=1" "MNR_SIDE_STREAM=1 MNR_SIDE_CUS=32"; do
  env $envs timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_fullsize.py tests/test_gpu_chain.py -m gpu -q -x -k "train_step or gradient or chain or trunk" > $OUT/r3_switch_test.log 2>&1
  echo "[$envs] rc=$? $(tail -1 $OUT/r3_switch_test.log)"
done
python bench.py --no_cpu_baseline > $OUT/r3_b_bench.json 2> $OUT/r3_b_bench.err; cut -c1-300 $OUT/r3_b_bench.json
cd /tmp
MNR_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r3_a_serial_prof -- python $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux > $OUT/r3_a_serial_prof.log 2>&1
python $R/tools/prof_summary.py stats $OUT/r3_a_serial_prof --title "rocprofv3 --kernel-trace --stats (r3_a, MNR_SIDE_STREAM=0: one stream, launch durations do not overlap)" --command "MNR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux" > $OUT/r3_a_serial_kernel_stats.md
rm -rf $OUT/r3_a_serial_prof
head -16 $OUT/r3_a_serial_kernel_stats.md | cut -c1-160
