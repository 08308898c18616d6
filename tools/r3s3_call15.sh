#!/bin/bash
# round 3, session 3, call 15: evidence of the session's final build: headline profile (bench line, kernel stats, SQ / FETCH / WRITE PMC,
# traffic.json), serial-stream kernel stats, the other BASELINE configs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd $R
bash tools/profile_round.sh r3d
cd /tmp
MNR_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r3d_serial_prof -- python $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux > $OUT/r3d_serial_prof.log 2>&1
python $R/tools/prof_summary.py stats $OUT/r3d_serial_prof --title "rocprofv3 --kernel-trace --stats (r3d, MNR_SIDE_STREAM=0: one stream, launch durations do not overlap)" --command "MNR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux" > $OUT/r3d_serial_kernel_stats.md
rm -rf $OUT/r3d_serial_prof
cd $R
bash tools/profile_preset.sh r3d_blender_256 --preset blender_256
bash tools/profile_preset.sh r3d_llff_raw --preset llff_raw
bash tools/profile_preset.sh r3d_blender_refnerf --preset blender_refnerf
bash tools/profile_preset.sh r3d_360_4096x192 --gin_bindings "Model.num_nerf_samples = 64" --batch_size 4096
