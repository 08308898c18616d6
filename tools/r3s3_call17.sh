#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
bash tools/ab_bench.sh r3s3_prio "base:" "prio:MNR_NERF_PRIORITY=1" "base_b:" "prio_b:MNR_NERF_PRIORITY=1"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3s3_prio_b256 "base:" "prio:MNR_NERF_PRIORITY=1" "base_b:" "prio_b:MNR_NERF_PRIORITY=1"
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3s3_prio_raw "base:" "prio:MNR_NERF_PRIORITY=1"
