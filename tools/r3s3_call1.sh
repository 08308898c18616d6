#!/bin/bash
# round 3, session 3, call 1: the inference chain with the in-kernel IPE producer: parity on the GPU, render A/B, timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_model.py -x -q -m gpu -k "in_kernel_ipe or forward_parity" > $OUT/r3s3_tests1.log 2>&1
tail -5 $OUT/r3s3_tests1.log
grep -E "layer 0:|head:|rgb \|" $OUT/r3s3_tests1.log
timeout 600 python tools/render_probe.py > $OUT/r3s3_render_probe.txt 2>&1
cat $OUT/r3s3_render_probe.txt | tail -30
