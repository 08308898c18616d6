#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
bash tools/ab_bench.sh r3s3_gcol3 "merged:MNR_HEAD_GCOL=0" "gcol:MNR_HEAD_GCOL=1" "merged_b:MNR_HEAD_GCOL=0" "gcol_b:MNR_HEAD_GCOL=1" "merged_c:MNR_HEAD_GCOL=0" "gcol_c:MNR_HEAD_GCOL=1"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3s3_gcol3_b256 "merged:MNR_HEAD_GCOL=0" "gcol:MNR_HEAD_GCOL=1" "merged_b:MNR_HEAD_GCOL=0" "gcol_b:MNR_HEAD_GCOL=1"
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3s3_gcol3_raw "merged:MNR_HEAD_GCOL=0" "gcol:MNR_HEAD_GCOL=1"
