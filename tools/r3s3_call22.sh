#!/bin/bash
# round 3, session 3, call 22: final evidence on the final build: GPU suite + smoke, headline profile (traffic.json), the other BASELINE configs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/ -q -m gpu > $OUT/r3f_gpu_suite.log 2>&1; tail -2 $OUT/r3f_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r3f > $OUT/r3f_round.log 2>&1; tail -3 $OUT/r3f_round.log | cut -c1-300
cd /tmp
MNR_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r3f_serial_prof -- python $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux > $OUT/r3f_serial_prof.log 2>&1
python $R/tools/prof_summary.py stats $OUT/r3f_serial_prof --title "rocprofv3 --kernel-trace --stats (r3f, MNR_SIDE_STREAM=0: one stream, launch durations do not overlap)" --command "MNR_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux" > $OUT/r3f_serial_kernel_stats.md
rm -rf $OUT/r3f_serial_prof
cd $R
bash tools/profile_preset.sh r3f_blender_256 --preset blender_256 | head -1 | cut -c1-200
bash tools/profile_preset.sh r3f_llff_raw --preset llff_raw | head -1 | cut -c1-200
bash tools/profile_preset.sh r3f_blender_refnerf --preset blender_refnerf | head -1 | cut -c1-200
bash tools/profile_preset.sh r3f_360_4096x192 --gin_bindings "Model.num_nerf_samples = 64" --batch_size 4096 | head -1 | cut -c1-200
