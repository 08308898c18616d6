"""profiles/traffic.json from the two PMC summaries of tools/profile_round.sh: HBM bytes of the MFMA kernels per train step.

    python tools/traffic_json.py <tag>_pmc_FETCH_SIZE.md <tag>_pmc_WRITE_SIZE.md > profiles/traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section).  The
number of train steps inside the profiled command is read off the dispatch count of clip_adam_kernel (one per top-level
module and step: two at 360.gin), or given as a fourth argument after the module count.
"""
import json
import os
import re
import sys

MFMA = ('gemm_nt_kernel', 'gemm_nt_panel_kernel', 'gemm_tn_kernel', 'gemm_tn_gcol_kernel', 'gemm_tn_rank1_kernel', 'gemm_nt_wres_kernel', 'mlp_chain_fwd_kernel', 'mlp_chain_bwd_kernel',
        'mlp_chain_fwd_ipe_kernel')


def table(path):
  rows = {}
  for l in open(path):
    m = re.match(r'\| `(.+?)` \| (\d+) \| ([0-9.e+]+) \|', l)
    if m:
      rows[m.group(1)] = (int(m.group(2)), float(m.group(3)))
  return rows


def main():
  fetch, write = table(sys.argv[1]), table(sys.argv[2])
  adam = [v for k, v in fetch.items() if 'clip_adam_kernel' in k]
  modules = int(sys.argv[3]) if len(sys.argv) > 3 else 2
  steps = adam[0][0] // modules if adam else None
  if len(sys.argv) > 4:                      # explicit step count (a summary cut off above clip_adam_kernel)
    steps = int(sys.argv[4])
  if steps is None:
    raise SystemExit('traffic_json.py: clip_adam_kernel is not in the summary: pass <modules> <steps> explicitly')
  f = sum(v[1] for k, v in fetch.items() if any(n in k for n in MFMA))
  w = sum(v[1] for k, v in write.items() if any(n in k for n in MFMA))
  total_f = sum(v[1] for v in fetch.values())
  total_w = sum(v[1] for v in write.values())
  import os
  stamp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multinerf_amd', 'libmnerf_hip.so.stamp')
  out = {
      # the build these counters were read on (bench.py reports the figure only for that build)
      'lib_digest': open(stamp).read().strip() if os.path.exists(stamp) else None,
      # (the tracked copies: gpurun_out/ is scratch and git-ignored, the same files are committed under profiles/)
      'source': f'profiles/{os.path.basename(sys.argv[1])} + profiles/{os.path.basename(sys.argv[2])} (separate rocprofv3 --pmc passes of bench.py --steps 2 --warmup 1; tools/final_lean.sh)',
      'train_steps_in_profile': steps,
      'mfma_kernels': list(MFMA),
      'gemm_fetch_kib_reported': f,
      'gemm_write_kib_reported': w,
      'correction': 'FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported',
      'gemm_hbm_bytes_per_step': (2 * f + w) * 1024 / steps,
      'all_kernels_hbm_bytes_per_step': (2 * total_f + total_w) * 1024 / steps,
  }
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main()
