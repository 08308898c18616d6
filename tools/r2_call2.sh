#!/bin/bash
# Round 2, GPU call 2: fused per-level Dense chain (csrc/fused_mlp.hip): its own test, the whole GPU gate, A/B bench.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chain.py -m gpu -q -s -x > gpurun_out/r2_chain_test.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_chain_test.log
tail -15 gpurun_out/r2_chain_test.log
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2_gpu_tests2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_gpu_tests2.log
tail -5 gpurun_out/r2_gpu_tests2.log
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2b_bench_$name.json 2> gpurun_out/r2b_bench_$name.err
  tail -c 400 gpurun_out/r2b_bench_$name.err
  python - <<PY
import json
try:
  j = json.loads([l for l in open('gpurun_out/r2b_bench_$name.json') if l.startswith('{')][-1])
  print('$name', round(j['value']), round(j['ms_per_step'], 2), 'gemm ms', round(j['roofline']['gemm_ms_per_step'], 2), 'loss', j['config']['final_loss'])
except Exception as e:
  print('$name FAILED', e)
PY
}
run chain_on MNR_FUSED_CHAIN=1
run chain_off MNR_FUSED_CHAIN=0
run chain_off_wres MNR_FUSED_CHAIN=0 MNR_NT_WRES=1
run chain_on_again MNR_FUSED_CHAIN=1
run old_defaults MNR_FUSED_CHAIN=0 MNR_NT_CFG=2,0 MNR_TN_SPLIT=0
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2_b_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux > $GRAFT_REPO_ROOT/gpurun_out/r2_b_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py stats gpurun_out/r2_b_prof --title "rocprofv3 --kernel-trace --stats (r2_b)" --command "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux" > gpurun_out/r2_b_kernel_stats.md
rm -rf gpurun_out/r2_b_prof
head -40 gpurun_out/r2_b_kernel_stats.md
