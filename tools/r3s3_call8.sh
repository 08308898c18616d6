#!/bin/bash
# round 3, session 3, call 8: minimum reduction steps per dW split (the atomic epilogue of a split against its K loop)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3s3_ms_b256 "ms4:" "ms48:MNR_TN_MIN_STEPS=48" "ms96:MNR_TN_MIN_STEPS=96" "ms160:MNR_TN_MIN_STEPS=160" "ms4b:" "ms96b:MNR_TN_MIN_STEPS=96"
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3s3_ms_raw "ms4:" "ms96:MNR_TN_MIN_STEPS=96" "ms160:MNR_TN_MIN_STEPS=160"
bash tools/ab_bench.sh r3s3_ms_360 "ms4:" "ms48:MNR_TN_MIN_STEPS=48" "ms96:MNR_TN_MIN_STEPS=96" "ms4b:"
