#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_model.py -m gpu -q > $OUT/r3_gpu_tests5.log 2>&1; echo "chain+model rc=$?"; tail -3 $OUT/r3_gpu_tests5.log
bash tools/ab_bench.sh r3_ab5 "defer:" "nodefer:MNR_CHAIN_DEFER=0" "defer2:"
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3_ab5_raw "defer:" "nodefer:MNR_CHAIN_DEFER=0" "perlayer:MNR_FUSED_CHAIN=0"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3_ab5_b256 "defer:" "nodefer:MNR_CHAIN_DEFER=0"
AB_BENCH_ARGS="--preset blender_refnerf" bash tools/ab_bench.sh r3_ab5_ref "defer:" "nodefer:MNR_CHAIN_DEFER=0"
python tools/chain_probe.py --M 2097152 --K0 128 --depth 8 --timeline > $OUT/r3_chain_probe_d8_defer.txt 2>&1; head -12 $OUT/r3_chain_probe_d8_defer.txt; grep -A28 "fwd train:" $OUT/r3_chain_probe_d8_defer.txt | head -30
python tools/chain_probe.py --timeline > $OUT/r3_chain_probe_d4_defer.txt 2>&1; head -12 $OUT/r3_chain_probe_d4_defer.txt
