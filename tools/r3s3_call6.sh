#!/bin/bash
# round 3, session 3, call 6: which level wants the four-lanes-per-ray kernels (82 KiB: NeRF level with rgb; 99 KiB: + interlevel scratch)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3s3_quad2_b256 "q80:" "q104:MNR_QUAD_LDS_MAX=106496" "q80b:" "q104b:MNR_QUAD_LDS_MAX=106496" "q80c:" "q104c:MNR_QUAD_LDS_MAX=106496"
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3s3_quad2_raw "q80:" "q88:MNR_QUAD_LDS_MAX=90112" "q104:MNR_QUAD_LDS_MAX=106496" "q160:MNR_QUAD_LDS_MAX=163840"
