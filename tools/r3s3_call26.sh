#!/bin/bash
# round 3, session 3, call 26: constant level-loop inputs cached on the device (no per-step host -> device copies, which synchronise the stream)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/ab_bench.sh r3s3_const "percall:MNR_CONST_CACHE=0" "cached:" "percall_b:MNR_CONST_CACHE=0" "cached_b:" "percall_c:MNR_CONST_CACHE=0" "cached_c:"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3s3_const_b256 "percall:MNR_CONST_CACHE=0" "cached:" "percall_b:MNR_CONST_CACHE=0" "cached_b:"
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3s3_const_raw "percall:MNR_CONST_CACHE=0" "cached:"
