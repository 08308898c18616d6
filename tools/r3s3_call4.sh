#!/bin/bash
# round 3, session 3, call 4: headline A/B of the head-dX K granule; kernel stats (IPE kernel with the one-FMA wrap)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "cast_rays_ipe or (train_step_parity and (extra0 or extra1-))" > $OUT/r3s3_tests4.log 2>&1
tail -3 $OUT/r3s3_tests4.log
bash tools/ab_bench.sh r3s3_k64 "k384:MNR_HEAD_K64=0" "k320:MNR_HEAD_K64=1" "k384b:MNR_HEAD_K64=0" "k320b:MNR_HEAD_K64=1"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r3s3_d_prof -- python $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux > $OUT/r3s3_d_prof.log 2>&1
python $R/tools/prof_summary.py stats $OUT/r3s3_d_prof --title "rocprofv3 --kernel-trace --stats (r3s3_d)" --command "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux" > $OUT/r3s3_d_kernel_stats.md
rm -rf $OUT/r3s3_d_prof
head -24 $OUT/r3s3_d_kernel_stats.md | cut -c1-150
