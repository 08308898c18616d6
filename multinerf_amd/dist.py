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
