#!/bin/bash
# round 3, GPU call 2: full GPU suite on the new build (chain trunk with skip, activations, use_viewdirs=False, bottleneck
# noise), A/B of streaming stores / side stream on the headline, fused trunk on/off on the 256-wide presets, NT probes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/r3_gpu_tests2.log 2>&1; echo "gpu suite rc=$?"; tail -6 $OUT/r3_gpu_tests2.log
bash tools/ab_bench.sh r3_ab2 "base:" "noside:MNR_SIDE_STREAM=0" "ntst:MNR_NT_STORES=1" "ntst_noside:MNR_NT_STORES=1 MNR_SIDE_STREAM=0" "base2:"
AB_BENCH_ARGS="--preset llff_raw" bash tools/ab_bench.sh r3_ab2_raw "chain:" "perlayer:MNR_FUSED_CHAIN=0" "chain_ntst:MNR_NT_STORES=1"
AB_BENCH_ARGS="--preset blender_256" bash tools/ab_bench.sh r3_ab2_b256 "chain:" "perlayer:MNR_FUSED_CHAIN=0" "noside:MNR_SIDE_STREAM=0"
AB_BENCH_ARGS="--preset blender_refnerf" bash tools/ab_bench.sh r3_ab2_ref "chain:" "perlayer:MNR_FUSED_CHAIN=0"
echo "== NT probe default"; PIPES=1 timeout 300 python tools/nt_pipe_probe.py 2>&1 | grep -v "bitwise" | head -30
echo "== NT probe 128 workgroups"; PIPES=1 MNR_NT_PERSIST=-128 timeout 300 python tools/nt_pipe_probe.py 2>&1 | grep -v "bitwise" | head -30
echo "== NT probe streaming stores"; PIPES=1 MNR_NT_STORES=1 timeout 300 python tools/nt_pipe_probe.py 2>&1 | grep -v "bitwise" | head -30
