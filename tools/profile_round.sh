#!/bin/bash
# Round profile on the GPU box:  bash tools/profile_round.sh <tag>   (writes gpurun_out/<tag>_*)
# 1 un-profiled bench line, 1 rocprofv3 kernel-trace/stats run, 3 separate PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE)
# and the traffic.json bench.py reads (copy <tag>_traffic.json to profiles/traffic.json together with the summaries).
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export MNR_SKIP_PREFLIGHT=1   # (rocprofv3 follows the preflight child, and its --stats database then holds that process only)
cd $R
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
CMD="python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -- python $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux > $OUT/${TAG}_prof.log 2>&1
python $R/tools/prof_summary.py stats $OUT/${TAG}_prof --title "rocprofv3 --kernel-trace --stats ($TAG)" --command "rocprofv3 --kernel-trace --stats -- $CMD" > $OUT/${TAG}_kernel_stats.md
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_SQ -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_aux > $OUT/${TAG}_pmc_SQ.log 2>&1
python $R/tools/prof_summary.py pmc $OUT/${TAG}_pmc_SQ --title "rocprofv3 --pmc SQ counters ($TAG)" --command "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_aux" --top 12 > $OUT/${TAG}_pmc_SQ.md
rm -rf $OUT/${TAG}_pmc_SQ
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_aux > $OUT/${TAG}_pmc_$C.log 2>&1
  python $R/tools/prof_summary.py pmc $OUT/${TAG}_pmc_$C --title "rocprofv3 --pmc $C ($TAG)" --command "rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_aux" --top 40 > $OUT/${TAG}_pmc_$C.md
  rm -rf $OUT/${TAG}_pmc_$C      # raw CSVs are large (kernel names); the summary is what is kept
done
rm -rf $OUT/${TAG}_prof
python $R/tools/traffic_json.py $OUT/${TAG}_pmc_FETCH_SIZE.md $OUT/${TAG}_pmc_WRITE_SIZE.md > $OUT/${TAG}_traffic.json
cat $OUT/${TAG}_traffic.json
cat $OUT/${TAG}_bench.json | cut -c1-600
head -12 $OUT/${TAG}_kernel_stats.md
