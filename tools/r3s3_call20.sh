#!/bin/bash
# round 3, session 3, call 20: SQ counters of the render path's kernels with the in-kernel IPE (what bounds mlp_chain_fwd_ipe_kernel)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/r3s3_render_pmc -- python $R/tools/render_probe.py --only on --reps 4 > $OUT/r3s3_render_pmc.log 2>&1
python $R/tools/prof_summary.py pmc $OUT/r3s3_render_pmc --title "rocprofv3 --pmc SQ counters (render, in-kernel IPE on)" --command "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- python tools/render_probe.py --only on --reps 4" --top 8 > $OUT/r3s3_render_pmc_SQ.md
rm -rf $OUT/r3s3_render_pmc
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/r3s3_render_pmc2 -- python $R/tools/render_probe.py --only on --reps 4 > $OUT/r3s3_render_pmc2.log 2>&1
python $R/tools/prof_summary.py pmc $OUT/r3s3_render_pmc2 --title "rocprofv3 --pmc instruction counters (render, in-kernel IPE on)" --top 8 > $OUT/r3s3_render_pmc_INST.md
rm -rf $OUT/r3s3_render_pmc2
cat $OUT/r3s3_render_pmc_SQ.md | cut -c1-220; cat $OUT/r3s3_render_pmc_INST.md | cut -c1-220
