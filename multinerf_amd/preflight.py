"""Torch-only GPU preflight, run in a SUBPROCESS before anything of this package touches the device.

Round 4's driver record died with `Memory access fault by GPU node-2` at the first host -> device copy of the first test,
before libmnerf_hip.so had been loaded (GPUTEST_r04.json): nothing in the record could tell a faulty box from a faulty
product.  `check()` makes a record say which it is: a child python process that imports NOTHING but torch does a pageable
1 MiB host -> device copy, a matmul, a device -> host copy and compares.  When the child dies, one line

    BOX_FAULT: torch-only preflight aborted before libmnerf_hip.so was loaded (...)

goes to stdout with the ROCm / driver versions and HSA_XNACK.  A pageable H2D copy runs on the SDMA engines; the runtime's
documented way around a broken engine is `HSA_ENABLE_SDMA=0` (copies as shader blits), so the child is retried once with it,
and if THAT passes the variable is set in the calling process (which must not have initialised HIP yet) so that the run can go
on, with the fault on record.

Callers: tests/conftest.py (before the first `-m gpu` test), __graft_entry__.smoke(), bench.py.  Not on the compute path.
"""

import os
import subprocess
import sys

_CHILD = r'''
import os, sys
import torch
assert torch.cuda.is_available(), 'torch.cuda.is_available() is False'
torch.cuda.set_device(0)
h = torch.arange(1 << 18, dtype=torch.float32)            # 1 MiB, pageable
d = h.cuda()                                              # the copy that aborted in GPUTEST_r04
torch.cuda.synchronize()
a = (d[:65536].reshape(256, 256) % 7.0)
c = (a @ a).cpu()
ref = (h[:65536].reshape(256, 256) % 7.0) @ (h[:65536].reshape(256, 256) % 7.0)
assert torch.equal(d.cpu(), h), 'D2H(H2D(x)) != x'
assert torch.allclose(c, ref, rtol=1e-5, atol=1e-3), 'device matmul != host matmul'
p = torch.empty(1 << 18, dtype=torch.float32).pin_memory()
p.copy_(h)
assert torch.equal(p.cuda(non_blocking=True).cpu(), h), 'pinned H2D'
print('PREFLIGHT_OK', torch.cuda.get_device_name(0), torch.version.hip)
'''


def _versions():
  out = {'HSA_XNACK': os.environ.get('HSA_XNACK', '(unset)'), 'HSA_ENABLE_SDMA': os.environ.get('HSA_ENABLE_SDMA', '(unset)')}
  for key, path in (('rocm', '/opt/rocm/.info/version'), ('amdgpu_driver', '/sys/module/amdgpu/version')):
    try:
      with open(path) as f:
        out[key] = f.read().strip()
    except OSError:
      out[key] = '(unreadable)'
  try:
    import torch
    out['torch'] = torch.__version__
    out['torch_hip'] = str(torch.version.hip)
  except Exception as e:                                    # pragma: no cover
    out['torch'] = f'(import failed: {e})'
  return out


def _run_child(extra_env=None, timeout=300):
  env = dict(os.environ)
  env.update(extra_env or {})
  try:
    r = subprocess.run([sys.executable, '-c', _CHILD], env=env, capture_output=True, text=True, timeout=timeout)
    return r.returncode, (r.stdout + r.stderr)[-2000:]
  except subprocess.TimeoutExpired as e:
    return -999, f'timeout after {timeout} s: {(e.stdout or b"")[-500:]!r}'


def check(verbose=True):
  """Returns {'ok', 'rc', 'workaround', 'versions', 'tail'}.  Never raises: the caller decides."""
  rc, tail = _run_child()
  res = {'ok': rc == 0 and 'PREFLIGHT_OK' in tail, 'rc': rc, 'workaround': None, 'versions': _versions(), 'tail': tail}
  if res['ok']:
    if verbose:
      print('preflight: ' + tail.strip().splitlines()[-1], flush=True)
    return res
  v = ', '.join(f'{k}={x}' for k, x in res['versions'].items())
  print(f'BOX_FAULT: torch-only preflight aborted before libmnerf_hip.so was loaded (child rc {rc}; {v})', flush=True)
  print('BOX_FAULT: child output tail: ' + ' | '.join(tail.strip().splitlines()[-4:]), flush=True)
  rc2, tail2 = _run_child({'HSA_ENABLE_SDMA': '0'})
  if rc2 == 0 and 'PREFLIGHT_OK' in tail2:
    os.environ['HSA_ENABLE_SDMA'] = '0'
    res.update(ok=True, workaround='HSA_ENABLE_SDMA=0')
    print('BOX_FAULT: the same child passes with HSA_ENABLE_SDMA=0 (copies as shader blits instead of the SDMA engines); '
          'continuing with it set in this process', flush=True)
  else:
    print(f'BOX_FAULT: the child fails with HSA_ENABLE_SDMA=0 as well (rc {rc2}); the device of this lease is unusable from torch alone',
          flush=True)
  return res


def hip_initialised():
  """True when this process has already created a HIP context (too late to change HSA_* variables)."""
  try:
    import torch
    return torch.cuda.is_initialized()
  except Exception:
    return False


if __name__ == '__main__':
  r = check()
  sys.exit(0 if r['ok'] else 1)
