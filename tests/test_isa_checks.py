"""Static checks on the device ISA hipcc produces for the hand-scheduled GEMM loops (no GPU needed, ~10 s of hipcc).

The direct-weights NT configurations (NtC36 / NtC37) load weight fragments with inline asm into registers they reserve
with amdgpu_num_vgpr(224).  Two things would silently corrupt results and cannot be seen by any functional test on the
host: hipcc touching a reserved register while those loads are in flight, and hipcc adding its own vmcnt waits (it drains
to 0 when register loads and LDS-DMA are mixed, which is why the loads are asm in the first place).  tools/isa_report.py
reads both off the generated assembly.
"""

import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.exists('/opt/rocm/bin/hipcc') and not shutil.which('hipcc'), reason='needs hipcc')


@pytest.fixture(scope='module')
def report():
  spec = importlib.util.spec_from_file_location('isa_report', os.path.join(ROOT, 'tools', 'isa_report.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  asm, tmp = mod.compile_selected([2, 36, 37, 40])
  yield mod, {n: b for n, b in mod.all_kernel_bodies(asm).items() if 'gemm_nt_kernel' in n or 'gemm_tn_kernel' in n}
  shutil.rmtree(tmp, ignore_errors=True)


def test_reserved_registers_are_left_alone_while_weight_loads_fly(report):
  import re
  mod, bodies = report
  res = {n: b for n, b in bodies.items() if re.search(r'kernel_r2\d\d', n)}
  assert len(res) == 7                                   # NtC36 / NtC37 (v224+), NtC40 (v240+) with / without bit-mask input; TnBigSplit
  for name, body in res.items():
    lo = int(re.search(r'kernel_r(2\d\d)', name).group(1))
    loads = [l for l in body if l.startswith('\tglobal_load_dwordx4') and min(mod._vregs(l.split(',')[0])) >= lo]
    assert len(loads) >= 4, name                         # the asm loads are there
    assert mod.reserved_register_violations(body, lo) == [], name


def test_the_in_flight_analysis_sees_a_seeded_violation():
  """The checker itself, on a hand-written listing: a copy out of a reserved register between the load and its wait
  (what hipcc did with ordinary asm outputs), a fragment read INTO one inside the loop (what it did when the kernel
  needed more registers than amdgpu_num_vgpr left it), and the clean pattern."""
  import importlib.util
  spec = importlib.util.spec_from_file_location('isa_report', os.path.join(ROOT, 'tools', 'isa_report.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  def listing(middle, after):
    return ['_Zk:', '\ts_mov_b32 s0, 0', '.LBB0_1:                  ; =>This Inner Loop Header: Depth=1', '\t;;#ASMSTART',
            '\tglobal_load_dwordx4 v[240:243], v[2:3], off', '\t;;#ASMEND'] + middle + [
                '\t;;#ASMSTART', '\ts_waitcnt vmcnt(4)', '\t;;#ASMEND'] + after + ['\ts_cbranch_scc1 .LBB0_1', '\ts_endpgm']
  clean = listing(['\tv_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]'], ['\tds_write_b128 v9, v[240:243]'])
  assert mod.reserved_register_violations(clean, 240) == []
  copied = listing(['\tv_mov_b64_e32 v[10:11], v[240:241]'], ['\tds_write_b128 v9, v[10:13]'])
  assert len(mod.reserved_register_violations(copied, 240)) == 1
  reused = listing(['\tds_read_b128 v[240:243], v201 offset:8192'], [])
  assert len(mod.reserved_register_violations(reused, 240)) == 1
  # in flight around the back-edge: a use at the top of the next iteration, before the wait, is seen through the loop
  late = listing([], [])
  late.insert(3, '\tv_add_u32_e32 v5, v241, v6')
  assert len(mod.reserved_register_violations(late, 240)) == 0     # the wait precedes the back-edge here ...
  nowait = [l for l in late if 's_waitcnt' not in l]
  assert len(mod.reserved_register_violations(nowait, 240)) == 1     # ... without it the use is in flight


def test_k_loops_carry_only_the_hand_counted_vmcnt_waits(report):
  mod, bodies = report
  for name, body in bodies.items():
    seg = mod.k_loop_lines(body)
    assert sum('v_mfma' in l for l in seg) >= 16, name
    own = mod.compiler_vmcnt_waits(seg)
    if 'gemm_tn_kernel' in name:                         # one per step by design: the __syncthreads() that ends it
      assert own == ['s_waitcnt vmcnt(0) lgkmcnt(0)'], (name, own)
    else:
      assert own == [], (name, own)


def test_no_spills_inside_the_k_loops(report):
  """(The default forward kernel spills two registers around its prologue; none may sit between the MFMAs.)"""
  mod, bodies = report
  for name, body in bodies.items():
    assert not any('scratch_' in l for l in mod.k_loop_lines(body)), name


def test_default_gemm_kernels_are_the_ones_validated_on_the_gpu(report):
  """profiles/r1_validated_isa.json holds digests of the device code of every kernel of the default path (gemm.hip: NtC0,
  NtC2, both TN tiles, the small kernels; all kernels of the other csrc files) as they ran the round-1 GPU suite and bench.  Work on the optional configurations, the
  simulator seams or the host side must not change them; an intended change re-validates on the GPU and rewrites the file
  (tools/isa_report.py: normalized_digest)."""
  import json
  mod, _ = report
  want = json.load(open(os.path.join(ROOT, 'profiles', 'r1_validated_isa.json')))['kernels']
  asm, tmp = mod.compile_selected([0, 2])
  try:
    got = {mod.canonical_kernel_name(n): mod.normalized_digest(b) for n, b in mod.all_kernel_bodies(asm).items()}
  finally:
    shutil.rmtree(tmp, ignore_errors=True)
  for f in ('resample.hip', 'features.hip', 'render.hip', 'losses.hip', 'optim.hip', 'refnerf.hip', 'camera.hip'):
    got.update({mod.canonical_kernel_name(n): mod.normalized_digest(b) for n, b in mod.all_kernel_bodies(mod.compile_file(f)).items()})
  assert set(want) <= set(got), sorted(set(want) - set(got))
  changed = [n for n in want if got[n] != want[n]]
  assert not changed, changed


def test_weights_resident_kernel_keeps_its_prefetch_in_flight(report):
  """gemm_nt_wres_kernel: no spills (a spilled weight fragment is reloaded with scratch_load inside the K step, whose vmcnt
  wait also drains the activation prefetch: seen while building it), and no hipcc vmcnt wait between the LDS-DMA issue of a
  step and its last MFMA (weight loads left pending into the loop made hipcc re-wait vmcnt(0) at every use)."""
  mod, _ = report
  asm, tmp = mod.compile_selected([2])
  try:
    bodies = {n: b for n, b in mod.all_kernel_bodies(asm).items() if 'gemm_nt_wres_kernel' in n}
  finally:
    shutil.rmtree(tmp, ignore_errors=True)
  assert len(bodies) == 2
  for name, body in bodies.items():
    assert not any('scratch_' in l for l in body), name
    mf = [k for k, l in enumerate(body) if 'v_mfma' in l]
    dma = [k for k, l in enumerate(body) if 'global_load_lds' in l and k < mf[0]]
    start = max(k for k in dma if mf[0] - k < 400)        # the step's DMA issue in front of its MFMAs
    assert mod.compiler_vmcnt_waits(body[start:mf[-1] + 1]) == [], name
