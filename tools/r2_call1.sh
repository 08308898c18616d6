#!/bin/bash
# Round 2, GPU call 1: the GPU gate (all -m gpu tests, output kept for the tolerance table), then the A/B of the GEMM
# configurations prepared in round 1 (tools/round2_first_call.sh minus the Python probes the C-ABI probe duplicates).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s -x > gpurun_out/r2_gpu_tests1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_gpu_tests1.log
tail -5 gpurun_out/r2_gpu_tests1.log
hipcc -O2 -std=c++17 -o /tmp/cabi_probe tools/cabi_probe.cpp -Iinclude -Lmultinerf_amd -lmnerf_hip -Wl,-rpath,$PWD/multinerf_amd
for cfg in 2 43 44 45 18 41 40 42 35 36 37 38 39; do
  timeout 90 /tmp/cabi_probe $cfg >> gpurun_out/r2_cabi_probe.txt 2>&1 || echo "cfg $cfg: probe exited with $?" >> gpurun_out/r2_cabi_probe.txt
done
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_$name.json 2> gpurun_out/r2_bench_$name.err
}
run cfg2 MNR_NT_CFG=2,0
run cfg36 MNR_NT_CFG=36,0
run cfg37 MNR_NT_CFG=37,0
run cfg40 MNR_NT_CFG=40,0
run tnsplit MNR_TN_SPLIT=1
run tnimm MNR_TN_SPLIT=2
run cfg40_tnsplit MNR_NT_CFG=40,0 MNR_TN_SPLIT=1
run cfg41 MNR_NT_CFG=41,0
run cfg43 MNR_NT_CFG=43,0
run cfg44 MNR_NT_CFG=44,0
run cfg45 MNR_NT_CFG=45,0
run cfg43_phased1024 MNR_NT_CFG=43,0 MNR_NT_PHASED_MIN_K=1024
run cfg2_phased1024 MNR_NT_CFG=2,0 MNR_NT_PHASED_MIN_K=1024
run cfg42_tnsplit MNR_NT_CFG=42,0 MNR_TN_SPLIT=1
run cfg43_tnimm MNR_NT_CFG=43,0 MNR_TN_SPLIT=2
for sk in 38,512 39,512 38,256 36,512 40,512 43,512; do
  run shortk_${sk/,/_} MNR_NT_SHORTK_CFG=$sk
done
run wres MNR_NT_WRES=1
run wres_cfg43 MNR_NT_WRES=1 MNR_NT_CFG=43,0
run cfg2_again MNR_NT_CFG=2,0
python tools/round2_summary.py gpurun_out > gpurun_out/r2_summary.txt 2>&1; cat gpurun_out/r2_summary.txt
