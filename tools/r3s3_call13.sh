#!/bin/bash
# round 3, session 3, call 13: density head as a row-dot in layer 7's store loop (MNR_HEAD_ROWDOT) + vector column in the dW GEMM (MNR_HEAD_GCOL)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_zz_fullsize.py -x -q -m gpu -k "gemm_nt or gemm_tn or extra15 or extra0 or fullsize or full_width or zz" > $OUT/r3s3_tests13.log 2>&1
tail -3 $OUT/r3s3_tests13.log
bash tools/ab_bench.sh r3s3_rowdot "old:MNR_HEAD_ROWDOT=0 MNR_HEAD_GCOL=0" "gcol:MNR_HEAD_ROWDOT=0" "both:" "old_b:MNR_HEAD_ROWDOT=0 MNR_HEAD_GCOL=0" "gcol_b:MNR_HEAD_ROWDOT=0" "both_b:"
