#!/bin/bash
# Round-end evidence on a short lease (bash tools/final_lean.sh <tag>): the driver's two commands verbatim, the two PMC passes
# bench.py's roofline.traffic needs for this build's digest, one bench.py line, kernel stats: most important first.
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python3 -m pytest tests/ -x -q -s -m gpu -p no:cacheprovider > $OUT/${TAG}_driver_pytest.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_driver_pytest.log
python3 -c 'import sys; sys.path.insert(0,"."); import __graft_entry__ as e; e.smoke()' > $OUT/${TAG}_driver_smoke.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_driver_smoke.log
tail -3 $OUT/${TAG}_driver_pytest.log; tail -4 $OUT/${TAG}_driver_smoke.log
export TMPDIR=/tmp
export MNR_SKIP_PREFLIGHT=1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_aux > $OUT/${TAG}_pmc_$C.log 2>&1
  python $R/tools/prof_summary.py pmc $OUT/${TAG}_pmc_$C --title "rocprofv3 --pmc $C ($TAG)" --command "rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_aux" --top 40 > $OUT/${TAG}_pmc_$C.md
  rm -rf $OUT/${TAG}_pmc_$C
done
python $R/tools/traffic_json.py $OUT/${TAG}_pmc_FETCH_SIZE.md $OUT/${TAG}_pmc_WRITE_SIZE.md > $OUT/${TAG}_traffic.json
cp $OUT/${TAG}_traffic.json $R/profiles/traffic.json
cd $R
unset MNR_SKIP_PREFLIGHT
timeout 300 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cut -c1-400 $OUT/${TAG}_bench.json
export MNR_SKIP_PREFLIGHT=1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -- python $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux > $OUT/${TAG}_prof.log 2>&1
python $R/tools/prof_summary.py stats $OUT/${TAG}_prof --title "rocprofv3 --kernel-trace --stats ($TAG)" --command "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_aux" > $OUT/${TAG}_kernel_stats.md
rm -rf $OUT/${TAG}_prof
head -12 $OUT/${TAG}_kernel_stats.md
