#!/bin/bash
# round 3, session 3, call 24: evidence refresh on the committed build: GPU suite, headline profile (traffic.json), blender_refnerf
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/ -q -m gpu > $OUT/r3g_gpu_suite.log 2>&1; tail -1 $OUT/r3g_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r3g > $OUT/r3g_round.log 2>&1; grep -E '^\{"metric' $OUT/r3g_round.log | cut -c1-160
bash tools/profile_preset.sh r3g_blender_refnerf --preset blender_refnerf | head -1 | cut -c1-160
