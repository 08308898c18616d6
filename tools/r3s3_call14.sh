#!/bin/bash
# round 3, session 3, call 14: the whole GPU suite + smoke on the session's final kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1700 python -m pytest tests/ -q -m gpu --durations=8 > $OUT/r3s3_gpu_suite.log 2>&1
tail -16 $OUT/r3s3_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r3s3_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/r3s3_smoke.log
