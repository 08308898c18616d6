#!/bin/bash
# round 3, session 3, call 9: merged head dW as N = 256 on the 256x256 tile + the density column as a vector (MNR_HEAD_GCOL)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_chain.py -x -q -m gpu -k "gemm_tn or in_kernel_ipe or (train_step_parity and (extra0 or extra4 or extra8 or extra14)) or (forward_parity and extra14)" > $OUT/r3s3_tests9.log 2>&1
tail -3 $OUT/r3s3_tests9.log
bash tools/ab_bench.sh r3s3_gcol "merged:MNR_HEAD_GCOL=0" "gcol:MNR_HEAD_GCOL=1" "merged_b:MNR_HEAD_GCOL=0" "gcol_b:MNR_HEAD_GCOL=1"
timeout 600 python tools/render_probe.py > $OUT/r3s3_render_probe9.txt 2>&1
grep -E "render,|proposal level|layer 0 \(|tile total" $OUT/r3s3_render_probe9.txt
