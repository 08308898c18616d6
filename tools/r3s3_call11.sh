#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for e in "" "MNR_TN_TARGET_WGS=128" "MNR_TN_TARGET_WGS=512" "MNR_TN_TARGET_WGS=1024"; do
  echo "== $e"
  env $e timeout 300 python tools/tn_head_probe.py 2>&1 | grep -v amdgpu.ids
done > $OUT/r3s3_tn_head_probe.txt
cat $OUT/r3s3_tn_head_probe.txt
