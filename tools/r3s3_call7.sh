#!/bin/bash
# round 3, session 3, call 7: dW tile / split tuning on the 256-wide presets (they were tuned on 360.gin)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3s3_tn_raw "base:" "small:MNR_TN_BIG_MIN_TILES=2" "big128:MNR_TN_TARGET_WGS=128" "big512:MNR_TN_TARGET_WGS=512" "small512:MNR_TN_BIG_MIN_TILES=2 MNR_TN_SMALL_TARGET_WGS=512" "small1024:MNR_TN_BIG_MIN_TILES=2 MNR_TN_SMALL_TARGET_WGS=1024" "base2:"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3s3_tn_b256 "base:" "small:MNR_TN_BIG_MIN_TILES=2" "big128:MNR_TN_TARGET_WGS=128" "big512:MNR_TN_TARGET_WGS=512" "base2:"
