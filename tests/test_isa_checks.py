"""Static checks on the device ISA hipcc produces for the hand-scheduled MFMA loops (no GPU needed, ~30 s of hipcc).

What cannot be seen by any functional test on the host but costs the most on the device: hipcc adding its own vmcnt waits
inside a software-pipelined loop (it drains to 0 when register loads and LDS-DMA are mixed), spills inside the MFMA loops
(a reload is a vector-memory operation whose wait also waits for every LDS-DMA and store issued before it), and a shipped
kernel that is not the code the GPU suite validated.  tools/isa_report.py reads all three off the generated assembly.
"""

import importlib.util
import json
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VALIDATED = os.path.join(ROOT, 'profiles', 'r6_validated_isa.json')

pytestmark = pytest.mark.skipif(not os.path.exists('/opt/rocm/bin/hipcc') and not shutil.which('hipcc'), reason='needs hipcc')


@pytest.fixture(scope='module')
def report():
  spec = importlib.util.spec_from_file_location('isa_report', os.path.join(ROOT, 'tools', 'isa_report.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  bodies = {}
  mod.all_digests()                                  # compiles every file once, in parallel; compile_file is cached from here on
  for f in ('gemm.hip', 'gemm_blk.hip', 'fused_mlp.hip'):
    bodies.update({n: b for n, b in mod.all_kernel_bodies(mod.compile_file(f)).items() if any('v_mfma' in l for l in b)})
  return mod, bodies


def _mfma_span(body):
  mf = [k for k, l in enumerate(body) if 'v_mfma' in l.split(';')[0]]
  return body[mf[0]:mf[-1] + 1]


def test_the_shipped_mfma_kernels_are_all_there(report):
  _, bodies = report
  names = sorted(bodies)
  assert sum('gemm_nt_kernel' in n for n in names) == 8          # NtBigP (pipelined) / NtBig / NtSmall, with / without bit-mask input; NtBigP reading a panel A1, the same with a vector column
  assert sum('gemm_nt_panel_kernel' in n for n in names) == 4    # panel result: forward / dX, A1 row-major / panel
  assert sum('gemm_nt_wres_kernel' in n for n in names) == 2
  assert sum('gemm_tn_kernel' in n for n in names) == 5          # TnBig / TnSmall; TnBig with panel A, panel B, both
  assert sum('mlp_chain_fwd_kernel' in n for n in names) == 2 and sum('mlp_chain_bwd_kernel' in n for n in names) == 2   # W = 128 / 256
  assert sum('mlp_chain_fwd_ipe_kernel' in n for n in names) == 2      # the inference chain with the in-kernel IPE producer
  assert sum('gemm_tn_gcol_kernel' in n for n in names) == 2           # TnBig with one more B column from a vector; the same with a panel A
  assert sum('gemm_tn_rank1_kernel' in n for n in names) == 1          # TnBig with B built in LDS from its rank-1 factors and mask bits
  assert len(names) == 28, names


def test_tiled_gemm_k_loops_carry_only_the_hand_counted_vmcnt_waits(report):
  """NT / TN: between the first and the last MFMA of the K loop (one straight-line block per K step) there is no wait
  hipcc added: the counted `s_waitcnt vmcnt(N)` of the pipeline are inline asm, the step's barrier sits outside."""
  mod, bodies = report
  for name, body in bodies.items():
    if 'gemm_nt_kernel' in name or 'gemm_tn_kernel' in name or 'gemm_tn_gcol_kernel' in name:
      span = _mfma_span(body)
      assert sum('v_mfma' in l for l in span) >= 16, name
      assert mod.compiler_vmcnt_waits(span) == [], (name, mod.compiler_vmcnt_waits(span))


def test_pipelined_nt_loop_spreads_its_dma_between_the_mfmas(report):
  """NtBigP's steady-state K-tile (the loop block with the counted vmcnt(6) wait): 16 MFMAs, the tile's 4 LDS-DMA pieces
  issued between them (as one burst behind the barrier, every wave waits in the address unit's queue instead of in its MFMAs:
  K-tile time = MFMA time + DMA time, profiles/r2_nt_pipe_probe.txt), and no lgkmcnt wait between the barrier and the
  first MFMA after it (the fragments of that k-step were waited for in front of the barrier; a wait there is for the NEXT
  tile's reads)."""
  _, bodies = report
  pipe = {n: b for n, b in bodies.items() if 'gemm_nt_kernel' in n and 'Li32ELi4EE' in n}
  assert len(pipe) == 4                                          # forward / dX with mask bits / forward from a panel A1 / + a vector column
  for name, body in pipe.items():
    vcol = name.endswith('Lb0ELb1ELb1EEv16mnr_gemm_nt_argsix')    # (one extra MFMA per k-step: mnr_gemm_nt_args.vcol)
    code = [l.split(';')[0].rstrip() for l in body]
    w = next(k for k, l in enumerate(code) if 's_waitcnt vmcnt(6)' in l)
    start = max(k for k in range(w) if code[k].startswith('.LBB'))
    end = next(k for k in range(w, len(code)) if code[k].startswith('\ts_cbranch'))
    blk = code[start:end]
    assert sum('v_mfma' in l for l in blk) == (18 if vcol else 16), name
    assert sum('global_load_lds' in l for l in blk) == 4, name
    assert not any('scratch_' in l for l in blk), name
    ops = [l for l in blk if 'v_mfma' in l or 'global_load_lds' in l]
    run = longest = 0
    for l in ops:
      run = run + 1 if 'global_load_lds' in l else 0
      longest = max(longest, run)
    assert longest <= 2, (name, longest)
    b = next(k for k, l in enumerate(blk) if 's_barrier' in l)
    m = next(k for k in range(b, len(blk)) if 'v_mfma' in blk[k])
    assert not any('lgkmcnt' in l for l in blk[b:m]), (name, blk[b:m])
  # the TN kernels' loop has the same structure (its bias MFMAs sit in a branch, so the check is over the whole MFMA span)
  for name, body in bodies.items():
    if 'gemm_tn_kernel' not in name:
      continue
    ops = [l for l in _mfma_span(body) if 'v_mfma' in l or 'global_load_lds' in l]
    assert sum('global_load_lds' in l for l in ops) >= 4, name
    run = longest = 0
    for l in ops:
      run = run + 1 if 'global_load_lds' in l else 0
      longest = max(longest, run)
    assert longest <= 1, (name, longest)


def test_panel_kernel_walk_has_no_spills_and_only_counted_waits(report):
  """gemm_nt_panel_kernel (csrc/gemm_blk.hip): the whole walk over the tiles is one MFMA stream, so NOTHING may spill (with two
  sets of tail flavours hipcc spilled per-lane constants and reloaded them, behind a vmcnt(0), in every K-tile; with 64-bit
  integer divisions in the tile decode it kept six tile pointers in VGPR pairs); no wait of hipcc's own between the first and
  the last MFMA; the steady-state K-tile block holds 16 MFMAs with its 4 LDS-DMA pieces spread between them; the interleaved
  epilogue block holds the tile's 16 (+ 1 mask) stores and 32 half-wave exchanges between MFMAs."""
  mod, bodies = report
  pan = {n: b for n, b in bodies.items() if 'gemm_nt_panel_kernel' in n}
  assert len(pan) == 4
  for name, body in pan.items():
    assert not any('scratch_' in l for l in body), name
    assert mod.compiler_vmcnt_waits(_mfma_span(body)) == [], (name, mod.compiler_vmcnt_waits(_mfma_span(body)))
    code = [l.split(';')[0].rstrip() for l in body]
    starts = [k for k, l in enumerate(code) if l.startswith('.LBB')] + [len(code)]
    blocks = [code[a:b] for a, b in zip(starts[:-1], starts[1:])]
    steady = [b for b in blocks if sum('v_mfma' in l for l in b) == 16 and sum('global_load_lds' in l for l in b) == 4]
    assert steady, name
    for blk in steady:
      ops = [l for l in blk if 'v_mfma' in l or 'global_load_lds' in l]
      run = longest = 0
      for l in ops:
        run = run + 1 if 'global_load_lds' in l else 0
        longest = max(longest, run)
      assert longest <= 1, (name, longest)
    epi = [b for b in blocks if sum('v_permlane32_swap' in l for l in b) == 32 and any('v_mfma' in l for l in b)]
    assert len(epi) == 1, (name, len(epi))
    assert sum('global_store_dwordx4' in l for l in epi[0]) in (16, 17), name


def test_no_spills_inside_the_mfma_loops(report):
  """(The dX flavour of the 256x256 NT kernel spills ~25 registers around its prologue and epilogue; none may sit between
  the MFMAs.)  The fused chain kernels must not spill at all: their copy-out and stage phases issue stores / LDS-DMA that a
  reload's vmcnt(0) would wait for (seen while building them: 13k cycles per layer)."""
  _, bodies = report
  for name, body in bodies.items():
    if 'mlp_chain' in name:
      assert not any('scratch_' in l for l in body), name
    else:
      assert not any('scratch_' in l for l in _mfma_span(body)), name


def test_weights_resident_kernel_keeps_its_prefetch_in_flight(report):
  """gemm_nt_wres_kernel: no spills (a spilled weight fragment is reloaded with scratch_load inside the K step, whose vmcnt
  wait also drains the activation prefetch: seen while building it), and no hipcc vmcnt wait between the LDS-DMA issue of a
  step and its last MFMA (weight loads left pending into the loop made hipcc re-wait vmcnt(0) at every use)."""
  mod, bodies = report
  wres = {n: b for n, b in bodies.items() if 'gemm_nt_wres_kernel' in n}
  assert len(wres) == 2
  for name, body in wres.items():
    assert not any('scratch_' in l for l in body), name
    mf = [k for k, l in enumerate(body) if 'v_mfma' in l]
    dma = [k for k, l in enumerate(body) if 'global_load_lds' in l and k < mf[0]]
    start = max(k for k in dma if mf[0] - k < 400)        # the step's DMA issue in front of its MFMAs
    assert mod.compiler_vmcnt_waits(body[start:mf[-1] + 1]) == [], name


def test_fused_chain_weight_chunks_are_prefetched_not_waited_for_on_the_spot(report):
  """mlp_chain_*_kernel<256>: the resident layers' weight fragments arrive in chunks of four 16-byte loads issued together
  (pinned with sched_barrier: left alone, hipcc sank every load to its first use, one L2 round trip per k-step), and
  the waits in the MFMA stream are counted (vmcnt(N), N > 0) except where a chunk is consumed to its end."""
  import re
  _, bodies = report
  for name, body in bodies.items():
    if 'mlp_chain' not in name or 'Li256' not in name:
      continue
    code = [l.split(';')[0] for l in body]
    loads = [k for k, l in enumerate(code) if l.startswith('\tglobal_load_dwordx4')]
    groups = sum(1 for i, k in enumerate(loads) if i == 0 or k - loads[i - 1] > 8)
    assert len(loads) >= 16 and groups <= len(loads) // 2, (name, len(loads), groups)      # (sunk loads: one group per load)
    counted = [int(m.group(1)) for l in code for m in [re.search(r's_waitcnt vmcnt\((\d+)\)', l)] if m]
    assert any(c >= 4 for c in counted), (name, counted)


def test_shipped_kernels_are_the_ones_validated_on_the_gpu(report):
  """profiles/r6_validated_isa.json holds digests of the device code of every kernel as it ran the round-6 GPU suite and
  bench (every kernel of round 5 unchanged, profiles/r5_validated_isa.json, plus cast_rays_ipe_tangent_bwd_kernel).  Host-side or simulator work must not change them; an intended kernel change re-validates on the GPU and rewrites
  the file (python tools/isa_report.py --write-digests profiles/r6_validated_isa.json)."""
  mod, _ = report
  want = json.load(open(VALIDATED))['kernels']
  got = mod.all_digests()
  assert set(want) == set(got), sorted(set(want) ^ set(got))
  changed = [n for n in want if got[n] != want[n]]
  assert not changed, changed
