#!/bin/bash
# round 3, session 3, call 23: weights-resident dX kernel with its mask bits through LDS (previous build through MNR_LIB_PATH)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_refnerf.py tests/test_gpu_model.py -q -m gpu -k "gemm_nt or refnerf or tangent or extra3 or extra12" 2>&1 | tail -1
AB_BENCH_ARGS="--preset blender_refnerf --steps 10 --warmup 3" bash tools/ab_bench.sh r3s3_wres_ref "prev:MNR_LIB_PATH=$R/tools/_bin/libmnerf_prev.so" "new:" "prev_b:MNR_LIB_PATH=$R/tools/_bin/libmnerf_prev.so" "new_b:"
bash tools/ab_bench.sh r3s3_wres_360 "prev:MNR_LIB_PATH=$R/tools/_bin/libmnerf_prev.so" "new:"
