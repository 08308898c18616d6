#!/bin/bash
# round 3, session 3, call 21: view-direction encoding evaluated once per ray and group of 8 samples (previous build through MNR_LIB_PATH)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k viewdir 2>&1 | tail -1
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3s3_vd_raw "prev:MNR_LIB_PATH=$R/tools/_bin/libmnerf_prev.so" "new:" "prev_b:MNR_LIB_PATH=$R/tools/_bin/libmnerf_prev.so" "new_b:"
bash tools/ab_bench.sh r3s3_vd_360 "prev:MNR_LIB_PATH=$R/tools/_bin/libmnerf_prev.so" "new:" "prev_b:MNR_LIB_PATH=$R/tools/_bin/libmnerf_prev.so" "new_b:"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3s3_vd_b256 "prev:MNR_LIB_PATH=$R/tools/_bin/libmnerf_prev.so" "new:"
