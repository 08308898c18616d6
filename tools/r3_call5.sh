#!/bin/bash
# round 3, session 2, call 1: new parity cases + same-box A/B of the one-pass proposal backward
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_chain.py -x -q -m gpu -k "extra13 or one_pass or extra0-40 or chain" > $OUT/r3s2_tests1.log 2>&1
tail -5 $OUT/r3s2_tests1.log
bash tools/ab_bench.sh r3s2_merge "merge0:MNR_MERGE_PROPS=0" "merge1:MNR_MERGE_PROPS=1" "merge0b:MNR_MERGE_PROPS=0" "merge1b:MNR_MERGE_PROPS=1"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3s2_merge_b256 "merge0:MNR_MERGE_PROPS=0" "merge1:MNR_MERGE_PROPS=1"
