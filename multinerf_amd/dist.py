"""Data-parallel plumbing: one process per GPU, RCCL over xGMI via torch.distributed.

Replaces the reference's three collectives:
  jax.lax.pmean(grad / stats, 'batch')   train_utils.py:319-321 -> all_reduce_mean_
  jax.lax.all_gather(render, 'batch')    train_utils.py:380-388 -> all_gather_cat (pixel buffers only)
  the TPU keep-alive psum                eval.py:244-247         -> barrier()
The ray batch is sharded contiguously B/N per rank (utils.shard, utils.py:125-128);
parameters and Adam moments are replicated; ONE all-reduce of the flat fp32
gradient vector (36 MB at 360.gin) per step.  Backend 'nccl' is RCCL on ROCm; the
same code runs under 'gloo' on CPU tensors in the tests.
"""

import os

import torch
import torch.distributed as td


def is_initialized():
  return td.is_available() and td.is_initialized()


def world_size():
  return td.get_world_size() if is_initialized() else 1


def rank():
  return td.get_rank() if is_initialized() else 0


def init_from_env(backend=None):
  """Join the job torch.distributed.run launched (RANK / WORLD_SIZE / MASTER_* in the env)."""
  ws = int(os.environ.get('WORLD_SIZE', '1'))
  if ws <= 1 or is_initialized():
    return
  if backend is None:
    backend = 'nccl' if torch.cuda.is_available() else 'gloo'
  if backend == 'nccl':
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  td.init_process_group(backend=backend)


def all_reduce_mean_(t):
  """In-place mean over ranks (jax.lax.pmean)."""
  if world_size() > 1:
    td.all_reduce(t, op=td.ReduceOp.SUM)
    t.div_(world_size())
  return t


def all_reduce_sum_async(t):
  """Start summing `t` over ranks; returns a handle for finish_mean_ (None when there is one rank).  Under RCCL the
  reduction runs on the collective stream behind everything already queued on the current stream, and the kernels
  queued after this call overlap with it."""
  if world_size() == 1:
    return None
  return td.all_reduce(t, op=td.ReduceOp.SUM, async_op=True)


def finish_mean_(t, handle):
  """Wait for all_reduce_sum_async (the current stream waits, not the host) and turn the sum into the mean."""
  if handle is not None:
    handle.wait()
    t.div_(world_size())
  return t


def all_gather_cat(t):
  """Concatenate every rank's [n_local, ...] block along dim 0 (all_gather + unshard)."""
  if world_size() == 1:
    return t
  out = [torch.empty_like(t) for _ in range(world_size())]
  td.all_gather(out, t.contiguous())
  return torch.cat(out, 0)


def all_gather_packed(tensors):
  """all_gather_cat of several per-ray tensors ([n_local, ...] each, same n_local, same dtype) with ONE collective: the
  tensors are packed side by side into an [n_local, sum of widths] buffer, gathered, and split again.  (Image rendering
  gathers ~8 pixel buffers per level and chunk: one RCCL call instead of 24.)"""
  if world_size() == 1 or not tensors:
    return list(tensors)
  n = tensors[0].shape[0]
  flat = [t.reshape(n, -1) for t in tensors]
  assert all(f.shape[0] == n and f.dtype == flat[0].dtype for f in flat)
  widths = [f.shape[1] for f in flat]
  packed = all_gather_cat(torch.cat(flat, 1))
  outs, c = [], 0
  for t, w in zip(tensors, widths):
    outs.append(packed[:, c:c + w].reshape((packed.shape[0],) + tuple(t.shape[1:])).contiguous())
    c += w
  return outs


def barrier():
  if world_size() > 1:
    td.barrier()


def shard_batch(batch, ws=None, rk=None):
  """This rank's contiguous slice of a Batch/Rays whose leading dim is the global batch."""
  ws = world_size() if ws is None else ws
  rk = rank() if rk is None else rk
  if ws == 1:
    return batch

  def fn(x):
    n = x.shape[0]
    assert n % ws == 0, f'batch {n} not divisible by world size {ws}'
    per = n // ws
    return x[rk * per:(rk + 1) * per].contiguous()

  return batch.map(fn)


def describe(device=None):
  """What the bench line records about the process group: backend, world size, the RCCL version torch was built
  against, this rank's device."""
  info = {'world_size': world_size(), 'rank': rank(), 'backend': td.get_backend() if is_initialized() else None}
  try:
    info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
  except Exception as e:                                 # CPU-only build / no RCCL
    info['rccl_version'] = f'unavailable ({type(e).__name__})'
  if device is not None and torch.cuda.is_available() and torch.device(device).type == 'cuda':
    p = torch.cuda.get_device_properties(device)
    info['device'] = f'{torch.device(device)}: {p.name}, {p.multi_processor_count} CUs, {p.total_memory >> 30} GiB'
    info['device_uuid'] = str(getattr(p, 'uuid', '')) or f'index {torch.device(device).index}'
  else:
    info['device'], info['device_uuid'] = str(device), f'host pid {os.getpid()}'
  return info


def all_gather_objects(obj):
  """Every rank's picklable `obj`, in rank order."""
  if world_size() == 1:
    return [obj]
  out = [None] * world_size()
  td.all_gather_object(out, obj)
  return out


def check_collectives(device, nbytes=36 * 1024 * 1024):
  """All-reduce / all-gather known patterns and assert the results; time an all-reduce of `nbytes` of fp32 (the flat
  gradient of 360.gin is 36 MB).  Raises on a wrong result: a mis-wired group must not reach the timed region."""
  import time
  ws, rk = world_size(), rank()
  n = 1 << 16
  i = torch.arange(n, device=device, dtype=torch.float32)
  t = (i % 251) * (rk + 1)                               # rank r holds (r + 1) * pattern
  all_reduce_mean_(t)
  want = (i % 251) * (ws + 1) / 2
  if not torch.allclose(t, want, rtol=1e-6, atol=1e-6):
    raise RuntimeError(f'all-reduce check failed on rank {rk}: max |diff| {(t - want).abs().max().item():.3e}')
  g = all_gather_cat(torch.full((3, 2), float(rk), device=device))
  want_g = torch.arange(ws, device=device, dtype=torch.float32).repeat_interleave(3)[:, None].expand(3 * ws, 2)
  if not torch.equal(g, want_g):
    raise RuntimeError(f'all-gather check failed on rank {rk}')
  parts = all_gather_packed([torch.full((2, 3), float(rk), device=device), torch.full((2,), float(10 + rk), device=device)])
  if not (torch.equal(parts[0][:, 0], torch.arange(ws, device=device, dtype=torch.float32).repeat_interleave(2)) and
          torch.equal(parts[1], 10 + torch.arange(ws, device=device, dtype=torch.float32).repeat_interleave(2))):
    raise RuntimeError(f'packed all-gather check failed on rank {rk}')
  buf = torch.ones(nbytes // 4, device=device, dtype=torch.float32)
  sync = torch.cuda.synchronize if torch.device(device).type == 'cuda' else (lambda: None)
  for _ in range(2):
    all_reduce_mean_(buf)
  sync()
  barrier()
  reps = 5
  t0 = time.perf_counter()
  for _ in range(reps):
    h = all_reduce_sum_async(buf)
    finish_mean_(buf, h)
  sync()
  barrier()
  dt = (time.perf_counter() - t0) / reps
  if not torch.allclose(buf, torch.ones_like(buf)):
    raise RuntimeError(f'timed all-reduce corrupted its buffer on rank {rk}')
  return {'ok': True, 'allreduce_bytes': nbytes, 'allreduce_ms': 1e3 * dt,
          'allreduce_busbw_GBps': (2 * (ws - 1) / ws * nbytes / dt / 1e9) if ws > 1 else None}
