#!/bin/bash
# round 3, session 3, call 3: half-tile staged layer 0 (MFMAs of one wave under the encoding of its SIMD mate)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_model.py -x -q -m gpu -k "in_kernel_ipe or forward_parity" > $OUT/r3s3_tests3.log 2>&1
tail -3 $OUT/r3s3_tests3.log
grep -E "layer 0:|head:|rgb \|" $OUT/r3s3_tests3.log
timeout 600 python tools/render_probe.py > $OUT/r3s3_render_probe3.txt 2>&1
tail -24 $OUT/r3s3_render_probe3.txt
