#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_refnerf.py tests/test_gpu_model.py -m gpu -q -k "refnerf" > $OUT/r3_gpu_tests6.log 2>&1; echo "refnerf rc=$?"; tail -2 $OUT/r3_gpu_tests6.log
bash tools/ab_bench.sh r3_ab6 "one_side:" "two_side:MNR_SIDE_STREAMS=2" "one_side2:" "two_side2:MNR_SIDE_STREAMS=2"
AB_BENCH_ARGS="--preset blender_refnerf" bash tools/ab_bench.sh r3_ab6_ref "vec8:"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3_ab6_b256 "one_side:" "two_side:MNR_SIDE_STREAMS=2"
