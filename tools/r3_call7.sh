#!/bin/bash
# round 3, final evidence call: full GPU suite (3-seed equal-step PSNR log), the headline round profile (bench line with
# cpu_baseline, kernel stats, SQ / FETCH / WRITE PMC passes -> traffic.json), bench line + kernel stats of the other configs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
MNR_PSNR_LOG=$OUT/r3_psnr360_equal_step.jsonl timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/r3_gpu_tests7.log 2>&1; echo "gpu suite rc=$?"; tail -4 $OUT/r3_gpu_tests7.log; grep "equal-step PSNR" $OUT/r3_gpu_tests7.log
bash tools/profile_round.sh r3_a
bash tools/profile_preset.sh r3c_blender_256 --preset blender_256
bash tools/profile_preset.sh r3c_llff_raw --preset llff_raw
bash tools/profile_preset.sh r3c_blender_refnerf --preset blender_refnerf
bash tools/profile_preset.sh r3c_360_4096x192 --gin_bindings "Model.num_nerf_samples = 64" --batch_size 4096
