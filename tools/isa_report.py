"""Device ISA of the MFMA kernels (csrc/gemm.hip, csrc/gemm_blk.hip, csrc/fused_mlp.hip), without a GPU: registers, spills, and the hot loops.

    python tools/isa_report.py [--loop]                           # report
    python tools/isa_report.py --write-digests profiles/rN_validated_isa.json

Compiles the kernel files device-only for gfx950 and prints per MFMA kernel: VGPR / AGPR / SGPR counts, scratch bytes,
spilled registers, and the counts of MFMA, LDS-DMA, ds_read, global_load and s_waitcnt instructions inside the loops that
carry MFMAs; --loop dumps those loops.  This is how the waits hipcc inserts around LDS-DMA and register loads are checked
before a GPU run (see the notes at the top of gemm.hip and fused_mlp.hip).  --write-digests records the normalised device
code of every kernel: run it on the build that passed the GPU suite (tests/test_isa_checks.py compares against it).
"""

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'multinerf_amd', 'csrc')


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


_ASM_CACHE = {}


def compile_file(name):
  """Device-only assembly text of csrc/<name> (cached per process: the tests read gemm.hip's three times)."""
  if name in _ASM_CACHE:
    return _ASM_CACHE[name]
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
  _ASM_CACHE[name] = text
  return text


KERNEL_FILES = ('gemm.hip', 'gemm_blk.hip', 'fused_mlp.hip', 'resample.hip', 'features.hip', 'render.hip', 'losses.hip', 'optim.hip', 'refnerf.hip',
                'camera.hip')


def all_digests():
  """{kernel name: normalized_digest} over every csrc file."""
  out = {}
  import concurrent.futures
  with concurrent.futures.ThreadPoolExecutor(max_workers=len(KERNEL_FILES)) as ex:      # (hipcc runs in child processes)
    texts = list(ex.map(compile_file, KERNEL_FILES))
  for text in texts:
    out.update({n: normalized_digest(b) for n, b in all_kernel_bodies(text).items()})
  return out


def main():
  if '--write-digests' in sys.argv:
    import json
    path = sys.argv[sys.argv.index('--write-digests') + 1]
    note = ('normalized sha1 (tools/isa_report.normalized_digest) of the device code of every kernel as validated on MI355X '
            '(the GPU suite and bench.py of the round this file is named after); tests/test_isa_checks.py compares the current '
            'build against it')
    json.dump({'note': note, 'kernels': all_digests()}, open(path, 'w'), indent=1)
    print('wrote', path)
    return
  show_loop = '--loop' in sys.argv
  for f in ('gemm.hip', 'gemm_blk.hip', 'fused_mlp.hip'):
    s = compile_file(f)
    meta = {}
    for m in re.finditer(r'- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size', s, re.S):
      blk = m.group(0)
      g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
      meta[g('name')] = (g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('vgpr_spill_count'))
    for name, body in all_kernel_bodies(s).items():
      if not any('v_mfma' in l for l in body):
        continue
      v, a, sg, scr, sp = meta[name]
      print(f'{name}\n  vgpr {v} agpr {a} sgpr {sg} scratch {scr} B spilled {sp}')
      kl = k_loop_lines(body)
      cnt = lambda pat: sum(1 for l in kl if re.search(pat, l.split(';')[0]))
      print(f'  MFMA-loop blocks: {len(kl)} lines, mfma {cnt("v_mfma")}, lds-dma {cnt("global_load_lds")}, ds_read {cnt("ds_read")}, '
            f'ds_write {cnt("ds_write")}, global_load {cnt(r"global_load_dwordx")}, scratch {cnt("scratch_")}, barriers {cnt("s_barrier")}, '
            f'vmcnt waits by hipcc {len(compiler_vmcnt_waits(kl))}')
      if show_loop:
        for lab, seg in mfma_loops(body):
          print(f'  loop {lab}: {len(seg)} lines')
          print('\n'.join(seg))


if __name__ == '__main__':
  main()
