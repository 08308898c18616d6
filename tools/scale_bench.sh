#!/bin/bash
# Scaling bench on ONE node with up to 8 MI355X (BASELINE.json configs 2 / 3; reference train_utils.py:319-321,341-345:
# one pmean of the gradient per step over the devices of a pmap).  Never run so far: gpurun exposes one GPU.
#
#   bash tools/scale_bench.sh [tag] [max_gpus]        -> gpurun_out/<tag>_scale.jsonl + a table on stdout
#
# What it does, so that first contact with N > 1 cannot fail on plumbing:
#   1. `python bench.py` (N = 1, the driver's command) and the SAME workload through torch.distributed.run at N = 1: the two
#      values must agree within 2 % (the launcher, the process group and the collective check cost nothing in the timed region);
#   2. weak scaling: `bench.py --gpus N` for N in {1, 2, 4, 8} (16384 rays per GPU), one rank per GPU over RCCL;
#   3. strong scaling: `--gpus 8 --global_batch 65536` (configs/360.gin 'garden', 8192 rays per GPU = BASELINE config 3);
#   every N > 1 line prints distributed.rank_devices (one device uuid per rank, asserted distinct inside bench.py) and
#   distributed.collective_check (all-reduce / all-gather / packed all-gather of known patterns, asserted, + a timed 36 MB
#   all-reduce) before its timed region.
# Efficiency is the reader's to compute from the per-N values; this script only reports value(N) / (N * value(1)).
set -u
TAG=${1:-scale}
MAXG=${2:-8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
JL=$OUT/${TAG}_scale.jsonl
: > "$JL"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NDEV=$(python -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)
echo "visible GPUs: $NDEV (asked for up to $MAXG)"
if [ "$NDEV" -lt 1 ]; then echo "no GPU visible"; exit 2; fi
STEPS=${SCALE_STEPS:-20}; WARM=${SCALE_WARMUP:-5}; PORT=${SCALE_PORT:-29541}

run() {  # name, n_gpus, extra args...
  local name=$1 n=$2; shift 2
  local line
  if [ "$n" = "plain" ]; then
    line=$(cd "$R" && timeout 1200 python bench.py --steps "$STEPS" --warmup "$WARM" --no_cpu_baseline --no_aux "$@" 2> "$OUT/${TAG}_scale_${name}.err" | tail -1)
  else
    line=$(cd "$R" && timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 \
             --master-port "$PORT" bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARM" --no_cpu_baseline --no_aux --check_collectives "$@" \
             2> "$OUT/${TAG}_scale_${name}.err" | tail -1)
    PORT=$((PORT + 1))
  fi
  case "$line" in "{"*) ;; *) line=null ;; esac
  echo "{\"arm\": \"$name\", \"line\": $line}" >> "$JL"
}

run plain plain
run launcher_n1 1
for n in 2 4 8; do
  if [ "$n" -le "$NDEV" ] && [ "$n" -le "$MAXG" ]; then run "weak_n$n" "$n"; fi
done
if [ 8 -le "$NDEV" ] && [ 8 -le "$MAXG" ]; then run strong_n8_b65536 8 --global_batch 65536; fi

python - "$JL" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
by = {r['arm']: r['line'] for r in rows}
base = by.get('plain')
ok = True
for r in rows:
  b = r['line']
  if not b:
    print(f"{r['arm']:20s} FAILED (see the .err file)")
    ok = False
    continue
  d = b.get('distributed') or {}
  eff = b['value'] / (b['n_gpus'] * base['value']) if base else float('nan')
  cc = d.get('collective_check') or {}
  print(f"{r['arm']:20s} N={b['n_gpus']} {b['scaling']:6s} {b['value']:12.0f} rays/s {b['ms_per_step']:8.3f} ms/step  value/(N*value(1)) {eff:.3f}  "
        f"backend {d.get('backend')} rccl {d.get('rccl_version')}  all-reduce 36 MB {cc.get('allreduce_ms')} ms")
  if b['n_gpus'] > 1:
    print(f"{'':20s} rank_devices {d.get('rank_devices')}")
    print(f"{'':20s} collective_check {cc}")
if base and by.get('launcher_n1'):
  rel = abs(by['launcher_n1']['value'] / base['value'] - 1)
  print(f"N = 1 through the launcher vs plain bench.py: {100 * rel:.2f} % apart ({'ok' if rel <= 0.02 else 'MORE THAN 2 %'})")
  ok = ok and rel <= 0.02
sys.exit(0 if ok else 1)
PY
