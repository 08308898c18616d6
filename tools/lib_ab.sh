#!/bin/bash
# same-box A/B of the product library against another build of the same ABI (MNR_LIB_PATH): a probe script, then bench.py lines, alternating
#   bash tools/lib_ab.sh <tag> <other .so> [probe.py]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
TAG=$1; OTHER=$R/$2; PROBE=${3:-}
export MNR_SKIP_PREFLIGHT=1
if [ -n "$PROBE" ]; then
  for arm in other product other product; do
    if [ $arm = other ]; then export MNR_LIB_PATH=$OTHER; else unset MNR_LIB_PATH; fi
    echo "== $arm"; timeout 200 python $PROBE 2>&1 | grep -v amdgpu.ids
  done > gpurun_out/${TAG}_probe.txt 2>&1
fi
for arm in other product other product; do
  if [ $arm = other ]; then export MNR_LIB_PATH=$OTHER; else unset MNR_LIB_PATH; fi
  timeout 300 python bench.py --no_cpu_baseline --no_aux 2>/dev/null | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read()); print('$arm', round(b['value']), 'rays/s', round(b['ms_per_step'],3), 'ms  mfma union', round(b['roofline']['gemm_ms_per_step'],2), 'final_loss', round(b['config']['final_loss'],7), b['library']['path'])"
done > gpurun_out/${TAG}_bench.txt 2>&1
unset MNR_LIB_PATH
cat gpurun_out/${TAG}_probe.txt gpurun_out/${TAG}_bench.txt 2>/dev/null
