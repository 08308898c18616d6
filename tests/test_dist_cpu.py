"""N>1 data-parallel plumbing under gloo on CPU, world_size 2 (the GPU job uses the same code under
'nccl' = RCCL): gradient mean, pixel all-gather + unshard, batch sharding, render_image padding."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, q):
  sys.path.insert(0, ROOT)
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank))
  from multinerf_amd import configs, dist as mdist, models, utils
  mdist.init_from_env(backend='gloo')
  assert mdist.world_size() == world and mdist.rank() == rank
  # 1. pmean of a flat gradient vector
  g = torch.arange(10, dtype=torch.float32) * (rank + 1)
  mdist.all_reduce_mean_(g)
  ok1 = torch.allclose(g, torch.arange(10, dtype=torch.float32) * 1.5)
  # 2. contiguous batch sharding + all-gather restores the global order
  glob = torch.arange(12, dtype=torch.float32).reshape(6, 2)
  rays = utils.Rays(*[glob.clone() for _ in range(9)])
  local = mdist.shard_batch(rays)
  ok2 = torch.equal(local.origins, glob[rank * 3:(rank + 1) * 3])
  back = mdist.all_gather_cat(local.origins * 2)
  ok3 = torch.equal(back, glob * 2)
  # 2b. several pixel buffers in one collective: packed all-gather = per-tensor all-gathers
  a, b3, c = local.origins[:, 0] + 100 * rank, local.origins.repeat(1, 2)[:, :3] * 3, local.origins.reshape(3, 2, 1) + 0.5
  pa, pb, pc = mdist.all_gather_packed([a, b3, c])
  ok3 = ok3 and torch.equal(pa, mdist.all_gather_cat(a)) and torch.equal(pb, mdist.all_gather_cat(b3)) and \
      torch.equal(pc, mdist.all_gather_cat(c)) and pc.shape == (6, 2, 1)
  # 3. render_image: chunking, edge padding to a multiple of the world size, per-rank slice, unpad
  cfg = configs.Config()
  cfg.render_chunk_size = 7          # 5x3 = 15 rays -> chunks 7,7,1 -> each padded to an even count
  H, W = 5, 3
  base = torch.arange(H * W, dtype=torch.float32).reshape(H, W, 1)
  img_rays = utils.Rays(*[base.clone() for _ in range(9)])

  def render_fn(rng, chunk):
    val = chunk.origins[:, 0]
    rend = {'rgb': torch.stack([val, val, val], -1), 'acc': val}
    rend = {k: mdist.all_gather_cat(v) for k, v in rend.items()}
    rend['ray_sdist'] = torch.zeros((2, 4))
    return [dict(rend), dict(rend)], None

  out = models.render_image(render_fn, img_rays, None, cfg, verbose=False, world_size=world, rank=rank)
  ok4 = out['rgb'].shape == (H, W, 3) and torch.equal(out['acc'], base[..., 0])
  ok5 = len(out['ray_sdist']) == 2
  # 4. the bench's first-contact check (bench.py --check_collectives): known patterns through every collective the
  # product uses, a timed all-reduce, and the rank / device roll call
  chk = mdist.check_collectives('cpu', nbytes=1 << 20)
  info = mdist.describe('cpu')
  ids = mdist.all_gather_objects((rank, info['device_uuid']))
  ok5 = ok5 and chk['ok'] and chk['allreduce_ms'] > 0 and info['world_size'] == world and info['backend'] == 'gloo' \
      and [i[0] for i in ids] == list(range(world)) and len(set(ids)) == world
  mdist.barrier()
  q.put((rank, ok1, ok2, ok3, ok4, ok5))
  td.destroy_process_group()


def test_world_size_2_gloo():
  world = 2
  port = _free_port()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = [q.get(timeout=120) for _ in range(world)]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for r in res:
    assert all(r[1:]), r


def test_single_process_is_identity():
  from multinerf_amd import dist as mdist
  t = torch.ones(3)
  assert mdist.world_size() == 1 and mdist.rank() == 0
  assert torch.equal(mdist.all_reduce_mean_(t.clone()), t)
  assert torch.equal(mdist.all_gather_cat(t), t)
  assert torch.equal(mdist.all_gather_packed([t, t * 2])[1], t * 2)
