#!/bin/bash
# round 3, session 3, call 5: llff_raw / blender_refnerf with the four-lanes-per-ray level kernels allowed up to 104 KiB of LDS
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3s3_quad_raw "q80:" "q104:MNR_QUAD_LDS_MAX=106496" "q80b:" "q104b:MNR_QUAD_LDS_MAX=106496"
AB_BENCH_ARGS="--preset blender_refnerf --steps 10 --warmup 3" bash tools/ab_bench.sh r3s3_quad_ref "q80:" "q104:MNR_QUAD_LDS_MAX=106496"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3s3_quad_b256 "q80:" "q104:MNR_QUAD_LDS_MAX=106496"
