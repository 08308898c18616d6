#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/chain_probe.py > gpurun_out/r2_chain_probe2.txt 2>&1; cat gpurun_out/r2_chain_probe2.txt
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_model.py tests/test_gpu_zz_fullsize.py -m gpu -q -x > gpurun_out/r2_gpu_tests3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_gpu_tests3.log
tail -4 gpurun_out/r2_gpu_tests3.log
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2c_bench_$name.json 2> gpurun_out/r2c_bench_$name.err
  python - <<PY
import json
try:
  j = json.loads([l for l in open('gpurun_out/r2c_bench_$name.json') if l.startswith('{')][-1])
  print('$name', round(j['value']), round(j['ms_per_step'], 2), 'gemm ms', round(j['roofline']['gemm_ms_per_step'], 2), 'loss', j['config']['final_loss'])
except Exception as e:
  print('$name FAILED', e)
PY
}
run chain_on MNR_FUSED_CHAIN=1
run chain_off MNR_FUSED_CHAIN=0
