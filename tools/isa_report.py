"""Device ISA of selected NT configurations of csrc/gemm.hip, without a GPU: registers, spills, and the K loop.

    python tools/isa_report.py 36 37 [--loop] [--keep]

Compiles a temporary copy of gemm.hip in which nt_dispatch() only instantiates the requested configurations (device
only, gfx950), then prints per kernel: VGPR / AGPR / SGPR counts, scratch bytes, spilled registers, and the counts of
MFMA, LDS-DMA, ds_read, global_load and s_waitcnt instructions inside the innermost loops; --loop dumps those loops.
This is how the waits hipcc inserts around LDS-DMA and register loads are checked before a GPU run (see the notes at
the top of gemm.hip).
"""

import os
import re
import subprocess
import sys
import tempfile

RESERVED_LO = 224          # gemm_nt_kernel_r224: v224-v255 belong to the inline-asm weight loads


def _vregs(text):
  out = set()
  for m in re.finditer(r'\bv\[(\d+):(\d+)\]', text):
    out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
  for m in re.finditer(r'\bv(\d+)\b', text):
    out.add(int(m.group(1)))
  return out


def _basic_blocks(body):
  """[(label, first line, last line + 1, successor labels, falls through)] of a kernel listing."""
  starts = [k for k, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)]
  bounds = [0] + starts + [len(body)]
  blocks = []
  for a, b in zip(bounds[:-1], bounds[1:]):
    if a == b:
      continue
    label = body[a].split(':')[0] if body[a].startswith('.LBB') else '<entry>'
    succ, fall = [], True
    for l in body[a:b]:
      code = l.split(';')[0]
      m = re.match(r'^\ts_(c?branch\w*)\s+(\.LBB\d+_\d+)', code)
      if m:
        succ.append(m.group(2))
        if m.group(1) == 'branch':
          fall = False
      if re.match(r'^\ts_endpgm', code):
        fall = False
    blocks.append((label, a, b, succ, fall))
  return blocks


def reserved_register_violations(body, RESERVED_LO=RESERVED_LO):
  """Instructions that touch a reserved register (v224+ / v240+) while an inline-asm load into it may still be in flight.

  Forward may-analysis over the kernel's basic blocks.  State: the reserved registers with a load in flight.  An asm
  `global_load_dwordx4` (between ASMSTART / ASMEND) into reserved registers adds them; an asm `s_waitcnt vmcnt(..)` lands
  them all (the counted waits of these loops only ever leave younger LDS-DMAs outstanding); anything else that reads or
  writes a register of the state is reported.  amdgpu_num_vgpr(N) is a request, not a cap: hipcc does hand out the
  reserved registers when a kernel needs more than N (seen: fragment ds_reads into v240+ inside the K loop of a variant
  that held the bias row in registers), which no functional test on the host can notice."""
  codes = [l.split(';')[0] for l in body]
  in_asm, flag = False, []
  for l in body:
    if 'ASMSTART' in l:
      in_asm = True
    elif 'ASMEND' in l:
      in_asm = False
    flag.append(in_asm)

  def load_dst(k):
    c = codes[k]
    if flag[k] and c.startswith('\tglobal_load_dwordx4'):
      d = _vregs(c.split(None, 1)[1].split(',', 1)[0])
      if d and all(r >= RESERVED_LO for r in d):
        return d
    return None

  def is_wait(k):
    return flag[k] and 's_waitcnt' in codes[k] and 'vmcnt' in codes[k]

  blocks = _basic_blocks(body)
  index = {lab: i for i, (lab, *_r) in enumerate(blocks)}
  succ = []
  for i, (lab, a, b, targets, fall) in enumerate(blocks):
    out = [index[t] for t in targets if t in index]
    if fall and i + 1 < len(blocks):
      out.append(i + 1)
    succ.append(out)

  def transfer(i, state, report=None):
    _, a, b, *_r = blocks[i]
    state = set(state)
    for k in range(a, b):
      c = codes[k]
      if not c.startswith('\t'):
        continue
      d = load_dst(k)
      if d is not None:
        state |= d
      elif is_wait(k):
        state.clear()
      elif state and report is not None and (_vregs(c) & state):
        report.append(body[k].strip())
    return state

  entry = [set() for _ in blocks]
  todo = list(range(len(blocks)))
  while todo:
    i = todo.pop()
    out = transfer(i, entry[i])
    for j in succ[i]:
      if not out <= entry[j]:
        entry[j] |= out
        todo.append(j)
  bad = []
  for i in range(len(blocks)):
    transfer(i, entry[i], bad)
  return bad


def k_loop_lines(body):
  """Lines of the basic blocks that lie on a cycle-or-path between MFMAs (reachable from a block with MFMAs and reaching
  one): the K loops, without prologue and epilogue, however hipcc lays the blocks out or duplicates the loop."""
  codes = [l.split(';')[0] for l in body]
  blocks = _basic_blocks(body)
  index = {lab: i for i, (lab, *_r) in enumerate(blocks)}
  succ = []
  for i, (lab, a, b, targets, fall) in enumerate(blocks):
    out = [index[t] for t in targets if t in index]
    if fall and i + 1 < len(blocks):
      out.append(i + 1)
    succ.append(out)
  pred = [[] for _ in blocks]
  for i, out in enumerate(succ):
    for j in out:
      pred[j].append(i)

  def closure(seeds, edges):
    seen, todo = set(seeds), list(seeds)
    while todo:
      i = todo.pop()
      for j in edges[i]:
        if j not in seen:
          seen.add(j)
          todo.append(j)
    return seen

  mf = [i for i, (_, a, b, *_r) in enumerate(blocks) if any(c.startswith('\tv_mfma') for c in codes[a:b])]
  hot = closure(mf, succ) & closure(mf, pred)
  out = []
  for i in sorted(hot):
    _, a, b, *_r = blocks[i]
    out.extend(body[a:b])
  return out


def compiler_vmcnt_waits(seg):
  """s_waitcnt with a vmcnt field that hipcc inserted itself (i.e. outside ASMSTART / ASMEND) in a code segment."""
  out, in_asm = [], False
  for l in seg:
    if 'ASMSTART' in l:
      in_asm = True
    elif 'ASMEND' in l:
      in_asm = False
    elif not in_asm and 's_waitcnt' in l and 'vmcnt' in l:
      out.append(l.strip())
  return out


def kernel_bodies(asm_text):
  """{kernel name: list of lines} for every gemm_nt kernel in a device assembly listing."""
  out = {}
  for m in re.finditer(r'^(_Z\d+gemm_nt_kernel\w*):', asm_text, flags=re.M):
    i = m.start()
    out[m.group(1)] = asm_text[i:asm_text.index('.Lfunc_end', i)].split('\n')
  return out


def normalized_digest(body):
  """sha1 of a kernel's instructions with labels, symbol names and comments normalised: equal digests = same device code."""
  import hashlib
  h = hashlib.sha1()
  for l in body[1:]:
    c = re.sub(r';.*', '', l).rstrip()
    if not c.strip():
      continue
    c = re.sub(r'\.LBB\d+_', '.LBB_', c)
    c = re.sub(r'_Z\w+', 'SYM', c)
    h.update(c.encode() + b'\n')
  return h.hexdigest()


def canonical_kernel_name(name):
  """Kernel name without trailing zero-valued template parameters of NtCfg / TnCfg (parameters added with a default of 0
  do not rename the configurations that do not use them)."""
  name = re.sub(r'(NtCfgI(?:Li\d+E)+?)(?:Li0E)*(ELb[01]E)', r'\1\2', name)
  name = re.sub(r'(TnCfgI(?:Li\d+E)+?)(?:Li0E)*(EEv16mnr_gemm_tn)', r'\1\2', name)
  return name


def all_kernel_bodies(asm_text):
  """{kernel name: lines} for EVERY kernel of a device assembly listing."""
  out = {}
  for m in re.finditer(r'^(_Z\w+):', asm_text, flags=re.M):
    i = m.start()
    j = asm_text.find('.Lfunc_end', i)
    if j > 0:
      out[m.group(1)] = asm_text[i:j].split('\n')
  return out


def mfma_loops(body):
  """Innermost loops (label .. back-edge) that contain MFMAs."""
  labels = {l.split(':')[0]: k for k, l in enumerate(body) if l.startswith('.LBB')}
  loops = []
  for lab, k0 in labels.items():
    if 'Loop Header' not in body[k0]:
      continue
    k1 = max((k for k, l in enumerate(body) if k > k0 and re.search(r's_c?branch\w*\s+%s\b' % re.escape(lab), l)), default=None)
    if k1 is not None and any('v_mfma' in l for l in body[k0:k1 + 1]):
      loops.append((lab, body[k0:k1 + 1]))
  return loops


def compile_file(name):
  """Device-only assembly text of csrc/<name> (any kernel file but gemm.hip, which compile_selected handles)."""
  tmp = tempfile.mkdtemp(prefix='isa_')
  out = os.path.join(tmp, name + '.s')
  cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '--cuda-device-only', '-S',
         os.path.join(CSRC, name), '-o', out]
  r = subprocess.run(cmd, capture_output=True, text=True)
  if r.returncode != 0:
    raise RuntimeError(r.stderr)
  text = open(out).read()
  import shutil
  shutil.rmtree(tmp, ignore_errors=True)
  return text


def compile_selected(cfgs):
  """Device-only assembly of gemm.hip with nt_dispatch() restricted to `cfgs`; returns (assembly text, temp dir)."""
  src = open(os.path.join(CSRC, 'gemm.hip')).read()
  keep = lambda m: m.group(0) if int(m.group(1)) in cfgs else ''
  src = re.sub(r'^\s*case (\d+): return nt_launch<NtC\d+>\(a, fast_epi, stream\);\n', keep, src, flags=re.M)
  tmp = tempfile.mkdtemp(prefix='isa_')
  path = os.path.join(tmp, 'gemm_sel.hip')
  open(path, 'w').write(src)
  out = os.path.join(tmp, 'gemm_sel.s')
  cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '--cuda-device-only', '-S', '-I', CSRC,
         path, '-o', out]
  r = subprocess.run(cmd, capture_output=True, text=True)
  if r.returncode != 0:
    raise RuntimeError(r.stderr)
  return open(out).read(), tmp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'multinerf_amd', 'csrc')


def main():
  cfgs = [int(a) for a in sys.argv[1:] if a.isdigit()]
  show_loop = '--loop' in sys.argv
  s, tmp = compile_selected(cfgs)
  meta = {}
  for m in re.finditer(r'- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size', s, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
    meta[g('name')] = (g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('vgpr_spill_count'))
  for name, body in kernel_bodies(s).items():
    v, a, sg, scr, sp = meta[name]
    print(f'{name}\n  vgpr {v} agpr {a} sgpr {sg} scratch {scr} B spilled {sp}')
    m_res = re.search(r'kernel_r(2\d\d)', name)
    if m_res:
      lo = int(m_res.group(1))
      bad = reserved_register_violations(body, lo)
      print(f'  v{lo}+ touched while an asm load into it may be in flight: {len(bad)}' + ''.join('\n    ' + b for b in bad[:8]))
    kl = k_loop_lines(body)
    cnt = lambda pat: sum(1 for l in kl if re.search(pat, l.split(';')[0]))
    print(f'  K-loop blocks: {len(kl)} lines, mfma {cnt("v_mfma")}, lds-dma {cnt("global_load_lds")}, ds_read {cnt("ds_read")}, '
          f'ds_write {cnt("ds_write")}, global_load {cnt(r"global_load_dwordx")}, scratch {cnt("scratch_")}, barriers {cnt("s_barrier")}, '
          f'vmcnt waits by hipcc {len(compiler_vmcnt_waits(kl))}')
    for lab, seg in mfma_loops(body):
      n = lambda pat: sum(1 for l in seg if re.search(pat, l))
      waits = [l.strip() for l in seg if 's_waitcnt' in l and 'vmcnt' in l]
      print(f'  loop {lab}: {len(seg)} lines, mfma {n("v_mfma")}, lds-dma {n("global_load_lds")}, ds_read {n("ds_read")}, '
            f'global_load {n(r"global_load_dwordx")}, scratch {n("scratch_")}, barriers {n("s_barrier")}')
      print('    vmcnt waits: ' + ' | '.join(waits) + f'   (inserted by hipcc: {len(compiler_vmcnt_waits(seg))})')
      if show_loop:
        print('\n'.join(seg))
  if '--keep' in sys.argv:
    print('kept', tmp)


if __name__ == '__main__':
  main()
