#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/r3_gpu_tests3.log 2>&1; echo "gpu suite rc=$?"; tail -4 $OUT/r3_gpu_tests3.log
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3_ab3_raw "base:" "quad80k:MNR_QUAD_LDS_MAX=81920" "quad52k:MNR_QUAD_LDS_MAX=53248"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3_ab3_b256 "base:" "quad80k:MNR_QUAD_LDS_MAX=81920" "quad52k:MNR_QUAD_LDS_MAX=53248"
bash tools/profile_preset.sh r3b_llff_raw --preset llff_raw --no_cpu_baseline
