#!/bin/bash
# round 3, GPU call 1: full GPU suite, stream-mode parity, same-box A/B of the stream switches, evidence for the other presets
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -s > $OUT/r3_gpu_tests1.log 2>&1; echo "gpu suite rc=$?"; tail -4 $OUT/r3_gpu_tests1.log
MNR_DW_STREAM=1 timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_fullsize.py -m gpu -q -k "train_step or gradient" > $OUT/r3_gpu_tests1_dw.log 2>&1; echo "dw-stream parity rc=$?"; tail -2 $OUT/r3_gpu_tests1_dw.log
MNR_SIDE_STREAM=1 MNR_SIDE_CUS=32 MNR_DW_STREAM=1 timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_fullsize.py -m gpu -q -k "train_step or gradient" > $OUT/r3_gpu_tests1_side.log 2>&1; echo "side-stream parity rc=$?"; tail -2 $OUT/r3_gpu_tests1_side.log
bash tools/ab_bench.sh r3_ab1 "base:" "dw:MNR_DW_STREAM=1" "side:MNR_SIDE_STREAM=1" "side32:MNR_SIDE_STREAM=1 MNR_SIDE_CUS=32" "side64:MNR_SIDE_STREAM=1 MNR_SIDE_CUS=64" "dw_side:MNR_DW_STREAM=1 MNR_SIDE_STREAM=1" "dw_side32:MNR_DW_STREAM=1 MNR_SIDE_STREAM=1 MNR_SIDE_CUS=32" "base2:"
bash tools/profile_preset.sh r3_blender_256 --preset blender_256
bash tools/profile_preset.sh r3_llff_raw --preset llff_raw
bash tools/profile_preset.sh r3_blender_refnerf --preset blender_refnerf
bash tools/profile_preset.sh r3_360_4096x192 --gin_bindings "Model.num_nerf_samples = 64" --batch_size 4096
