#!/bin/bash
# First GPU call of the next round (everything here was prepared without a GPU):
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/round2_first_call.sh'
# 1. operand-ingest probe: which path / pattern bounds the 64 KiB-per-step operand stream of the GEMM loops
# 2. direct-weights NT loop (NtC36 / NtC37): bitwise screen against NtC2, then timing next to NtC2 / NtC35
# 3. the same A/B end to end (MNR_NT_CFG selects the 256x256 configuration for the whole step)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
# 0. the C-ABI probe (no Python: seconds): bitwise screens + timings of every prepared NT configuration and the TN pair.
#    Each configuration in its own process under its own timeout: a variant that misbehaves on hardware costs one line.
hipcc -O2 -std=c++17 -o /tmp/cabi_probe tools/cabi_probe.cpp -Iinclude -Lmultinerf_amd -lmnerf_hip -Wl,-rpath,$PWD/multinerf_amd
for cfg in 2 43 44 45 18 41 40 42 35 36 37 38 39; do
  timeout 90 /tmp/cabi_probe $cfg >> gpurun_out/r2_cabi_probe.txt 2>&1 || echo "cfg $cfg: probe exited with $?" >> gpurun_out/r2_cabi_probe.txt
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/ingest_probe tools/ingest_probe.hip && timeout 300 /tmp/ingest_probe > gpurun_out/r2_ingest_probe.txt 2>&1
timeout 600 python tools/gemm_probe.py --cfgs 2,35,36,37 > gpurun_out/r2_gemm_probe_direct.txt 2>&1
for cfg in 2 36 37; do
  MNR_NT_CFG=$cfg,0 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_cfg$cfg.json 2> gpurun_out/r2_bench_cfg$cfg.err
done
# 3b. split operand paths: NT (NtC40) and TN (TnBigSplit) probes, then end to end (alone and together)
timeout 600 python tools/gemm_probe.py --cfgs 2,40,41,42,43 > gpurun_out/r2_gemm_probe_split_nt.txt 2>&1
timeout 600 python tools/gemm_probe.py --which tn > gpurun_out/r2_gemm_probe_split_tn.txt 2>&1
MNR_NT_CFG=40,0 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_cfg40.json 2> gpurun_out/r2_bench_cfg40.err
MNR_TN_SPLIT=1 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_tnsplit.json 2> gpurun_out/r2_bench_tnsplit.err
MNR_TN_SPLIT=2 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_tnimm.json 2> gpurun_out/r2_bench_tnimm.err
MNR_NT_CFG=40,0 MNR_TN_SPLIT=1 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_cfg40_tnsplit.json 2> gpurun_out/r2_bench_cfg40_tnsplit.err
# 3c. the epilogue with its 16 LDS reads per thread issued together: on the default loop (41) and on the split-path loop (42)
MNR_NT_CFG=41,0 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_cfg41.json 2> gpurun_out/r2_bench_cfg41.err
MNR_NT_CFG=43,0 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_cfg43.json 2> gpurun_out/r2_bench_cfg43.err
MNR_NT_CFG=44,0 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_cfg44.json 2> gpurun_out/r2_bench_cfg44.err
MNR_NT_CFG=45,0 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_cfg45.json 2> gpurun_out/r2_bench_cfg45.err
MNR_NT_CFG=43,0 MNR_NT_PHASED_MIN_K=1024 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_cfg43_phased1024.json 2> gpurun_out/r2_bench_cfg43_phased1024.err
MNR_NT_CFG=42,0 MNR_TN_SPLIT=1 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_cfg42_tnsplit.json 2> gpurun_out/r2_bench_cfg42_tnsplit.err
# 4. short-K (proposal) GEMMs on 256x128 tiles, two workgroups per CU (one's epilogue under the other's K loop)
for sk in 38,512 39,512 38,256 36,512 40,512 43,512; do
  MNR_NT_SHORTK_CFG=$sk timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_shortk_${sk/,/_}.json 2> gpurun_out/r2_bench_shortk_${sk/,/_}.err
done

# 5. weights-resident persistent kernel for the short-K (proposal) layers, alone and with the overhead-trimmed trunk configuration
MNR_NT_WRES=1 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_wres.json 2> gpurun_out/r2_bench_wres.err
MNR_NT_WRES=1 MNR_NT_CFG=43,0 timeout 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_aux > gpurun_out/r2_bench_wres_cfg43.json 2> gpurun_out/r2_bench_wres_cfg43.err
python tools/round2_summary.py gpurun_out > gpurun_out/r2_summary.txt 2>&1; cat gpurun_out/r2_summary.txt
