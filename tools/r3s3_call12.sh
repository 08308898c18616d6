#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 300 python tools/tn_head_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r3s3_tn_head_probe2.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "gemm_tn or (train_step_parity and (extra0 or extra4 or extra8))" > $OUT/r3s3_tests12.log 2>&1
tail -2 $OUT/r3s3_tests12.log
bash tools/ab_bench.sh r3s3_gcol2 "merged:MNR_HEAD_GCOL=0" "gcol:MNR_HEAD_GCOL=1" "merged_b:MNR_HEAD_GCOL=0" "gcol_b:MNR_HEAD_GCOL=1"
