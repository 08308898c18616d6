"""Flax-compatible checkpoints for TrainState (reference call sites train.py:84,219-223,286,
eval.py:73, render.py:112: `flax.training.checkpoints.{save,restore}_checkpoint`).

On-disk format restated from flax's published `flax.serialization` (flax is un-pinned in the reference's
requirements.txt and not installable here, so this has NOT been checked against a checkpoint written by flax):
a file `<ckpt_dir>/checkpoint_<step>` holding msgpack of the state dict

  {'step': <int32 scalar>, 'params': {'params': {<module>: {<Dense_k>: {'kernel', 'bias'}} ...}},
   'opt_state': {'0': {'count', 'mu': <params tree>, 'nu': <params tree>}, '1': {'count'}}}

(optax.adam = chain(scale_by_adam, scale_by_schedule): tuples become dicts keyed '0', '1'; namedtuples
become dicts of their fields).  ndarrays are msgpack ExtType(1, packb((shape, dtype.name, raw bytes)));
NumPy scalars ExtType(3, same payload).  Arrays above 2**30 bytes would be chunked by flax; none of the
BASELINE models come near that, writing refuses them and reading understands the chunked form.
"""

import os
import re

import msgpack
import numpy as np
import torch

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
_MAX_CHUNK = 2**30


def _nd_to_bytes(a):
  a = np.asarray(a)
  if a.nbytes > _MAX_CHUNK:
    raise ValueError('array larger than 2**30 bytes: flax would chunk it; not needed for these models')
  return msgpack.packb((a.shape, a.dtype.name, a.tobytes('C')), use_bin_type=True)


def _pack_default(x):
  if isinstance(x, np.ndarray):
    return msgpack.ExtType(_EXT_NDARRAY, _nd_to_bytes(x))
  if isinstance(x, np.generic):
    return msgpack.ExtType(_EXT_NPSCALAR, _nd_to_bytes(x))
  if isinstance(x, complex):
    return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((x.real, x.imag)))
  raise TypeError(f'cannot serialise {type(x)}')


def _ext_hook(code, data):
  if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
    shape, dtype, buf = msgpack.unpackb(data, raw=False)
    a = np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape)
    return a if code == _EXT_NDARRAY else a[()]
  if code == _EXT_COMPLEX:
    re_, im = msgpack.unpackb(data)
    return complex(re_, im)
  return msgpack.ExtType(code, data)


def _unchunk(tree):
  """flax.serialization._unchunk_array_leaves_in_place."""
  if isinstance(tree, dict):
    if '__msgpack_chunked_array__' in tree:
      shape = tree['shape']
      chunks = [tree['chunks'][str(i)] for i in range(len(tree['chunks']))]
      return np.concatenate([np.asarray(c).reshape(-1) for c in chunks]).reshape(shape)
    return {k: _unchunk(v) for k, v in tree.items()}
  return tree


def msgpack_serialize(state_dict):
  return msgpack.packb(state_dict, default=_pack_default, strict_types=True)


def msgpack_restore(blob):
  return _unchunk(msgpack.unpackb(blob, ext_hook=_ext_hook, raw=False, strict_map_key=False))


def _to_numpy_tree(tree):
  if isinstance(tree, dict):
    return {k: _to_numpy_tree(v) for k, v in tree.items()}
  return tree.detach().cpu().numpy().copy()


def state_dict(model, state):
  """TrainState -> the nested dict flax would serialise (names as in train.py:194-195 / SURVEY section 5)."""
  count = np.asarray(state.step, dtype=np.int32)
  tree = lambda flat: _to_numpy_tree(model.params_tree(flat))
  return {
      'step': np.asarray(state.step, dtype=np.int32),
      'params': {'params': tree(state.params['flat'])},
      'opt_state': {'0': {'count': count, 'mu': {'params': tree(state.mu)}, 'nu': {'params': tree(state.nu)}},
                    '1': {'count': count}},
  }


def checkpoint_path(ckpt_dir, step, prefix='checkpoint_'):
  return os.path.join(ckpt_dir, f'{prefix}{int(step)}')


def latest_checkpoint(ckpt_dir, prefix='checkpoint_'):
  """flax.training.checkpoints.latest_checkpoint: highest step (natural sort) or None."""
  if not os.path.isdir(ckpt_dir):
    return None
  best = None
  for name in os.listdir(ckpt_dir):
    m = re.fullmatch(re.escape(prefix) + r'(\d+)', name)
    if m and (best is None or int(m.group(1)) > best[0]):
      best = (int(m.group(1)), os.path.join(ckpt_dir, name))
  return best[1] if best else None


def save_checkpoint(ckpt_dir, model, state, step, keep=1, prefix='checkpoint_', overwrite=False):
  """flax.training.checkpoints.save_checkpoint(ckpt_dir, target, step, keep=...) (train.py:219-223)."""
  os.makedirs(ckpt_dir, exist_ok=True)
  path = checkpoint_path(ckpt_dir, step, prefix)
  latest = latest_checkpoint(ckpt_dir, prefix)
  if latest is not None and not overwrite:
    last_step = int(latest[len(os.path.join(ckpt_dir, prefix)):])
    if last_step >= int(step):
      raise ValueError(f'Trying to save an outdated checkpoint at step: "{step}" and overwrite=False. '
                       f'Latest checkpoint: {latest}')          # flax's InvalidCheckpointError case
  tmp = path + '.tmp'
  with open(tmp, 'wb') as f:
    f.write(msgpack_serialize(state_dict(model, state)))
  os.replace(tmp, path)
  ckpts = sorted((int(n[len(prefix):]), n) for n in os.listdir(ckpt_dir) if re.fullmatch(re.escape(prefix) + r'\d+', n))
  for _, n in ckpts[:-keep] if keep > 0 else []:
    os.remove(os.path.join(ckpt_dir, n))
  return path


def restore_checkpoint(ckpt_dir, model, state, step=None, prefix='checkpoint_'):
  """flax.training.checkpoints.restore_checkpoint(ckpt_dir, target): returns `state` unchanged when the
  directory holds no checkpoint (train.py:84 relies on that), else a TrainState filled from the file."""
  path = checkpoint_path(ckpt_dir, step, prefix) if step is not None else latest_checkpoint(ckpt_dir, prefix)
  if path is None or not os.path.exists(path):
    return state
  with open(path, 'rb') as f:
    sd = msgpack_restore(f.read())
  dev = state.params['flat'].device
  to_t = lambda tree: {k: (to_t(v) if isinstance(v, dict) else torch.as_tensor(np.array(v))) for k, v in tree.items()}
  flat = model.flat_from_tree(to_t(sd['params']['params']), device=dev)
  adam = sd['opt_state']['0']
  mu = model.flat_from_tree(to_t(adam['mu']['params']), device=dev)
  nu = model.flat_from_tree(to_t(adam['nu']['params']), device=dev)
  # validate everything BEFORE touching the live state: a failed restore must leave it as it was
  step_ = int(np.asarray(sd['step']))
  counts = [int(np.asarray(adam['count']))]
  sched = sd['opt_state'].get('1')                      # optax.scale_by_schedule state: its own copy of the count
  if isinstance(sched, dict) and 'count' in sched:
    counts.append(int(np.asarray(sched['count'])))
  if any(c != step_ for c in counts):
    raise ValueError(f'checkpoint step {step_} and optimiser counts {counts} disagree')
  want = state.params['flat'].shape
  for name, t in (('params', flat), ('mu', mu), ('nu', nu)):
    if t.shape != want or not torch.isfinite(t).all():
      raise ValueError(f'checkpoint {name}: shape {tuple(t.shape)} vs model {tuple(want)}, or non-finite values')
  state.params['flat'].copy_(flat)
  state.mu.copy_(mu)
  state.nu.copy_(nu)
  return type(state)(step=step_, params=state.params, mu=state.mu, nu=state.nu)
