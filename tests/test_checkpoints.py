"""multinerf_amd/checkpoints.py: flax-msgpack layout and TrainState round trip (CPU)."""

import os

import msgpack
import numpy as np
import pytest
import torch

from multinerf_amd import checkpoints, configs, models, train_utils


def _state(model, seed):
  g = torch.Generator().manual_seed(seed)
  mk = lambda: torch.randn(model.num_params, generator=g)
  return train_utils.TrainState(step=1234, params={'flat': mk(), 'params': None}, mu=mk(), nu=mk().abs())


@pytest.mark.parametrize('name,extra', [('blender_256', []), ('360', ['Model.num_glo_features = 4',
                                                                       'NerfMLP.net_width = 128', 'PropMLP.net_width = 128']),
                                        ('llff_raw', ['NerfMLP.net_width = 128'])])
def test_round_trip(tmp_path, name, extra):
  cfg = configs.load_preset(name, extra)
  model = models.Model(config=cfg).build('cpu')
  st = _state(model, 1)
  path = checkpoints.save_checkpoint(str(tmp_path), model, st, 1234)
  assert os.path.basename(path) == 'checkpoint_1234'
  blank = train_utils.TrainState(step=0, params={'flat': torch.zeros(model.num_params), 'params': None},
                                 mu=torch.zeros(model.num_params), nu=torch.zeros(model.num_params))
  got = checkpoints.restore_checkpoint(str(tmp_path), model, blank)
  assert got.step == 1234
  for a, b in ((got.params['flat'], st.params['flat']), (got.mu, st.mu), (got.nu, st.nu)):
    assert torch.equal(a, b)


def test_file_layout_is_flax_state_dict(tmp_path):
  cfg = configs.load_preset('blender_256')
  model = models.Model(config=cfg).build('cpu')
  st = _state(model, 2)
  path = checkpoints.save_checkpoint(str(tmp_path), model, st, 7)
  raw = msgpack.unpackb(open(path, 'rb').read(), raw=False, strict_map_key=False)     # no ext hook: see the ExtTypes
  assert sorted(raw) == ['opt_state', 'params', 'step']
  assert sorted(raw['opt_state']) == ['0', '1'] and sorted(raw['opt_state']['0']) == ['count', 'mu', 'nu']
  assert list(raw['params']) == ['params'] and list(raw['params']['params']) == ['NerfMLP_0', 'PropMLP_0']
  k = raw['params']['params']['NerfMLP_0']['Dense_0']['kernel']
  assert isinstance(k, msgpack.ExtType) and k.code == 1
  shape, dtype, buf = msgpack.unpackb(k.data, raw=False)
  assert tuple(shape) == (96, 256) and dtype == 'float32' and len(buf) == 96 * 256 * 4
  want = model.params_tree(st.params['flat'])['NerfMLP_0']['Dense_0']['kernel'].numpy()
  np.testing.assert_array_equal(np.frombuffer(buf, np.float32).reshape(shape), want)
  step = raw['step']
  assert isinstance(step, msgpack.ExtType) and step.code == 1          # 0-d int32 ndarray, like jnp scalars
  assert msgpack.unpackb(step.data, raw=False)[:2] == [[], 'int32']


def test_reads_a_blob_built_the_flax_way():
  """Bytes assembled exactly as flax.serialization.msgpack_serialize does (restated) parse back."""
  def nd(a):
    a = np.asarray(a)
    return msgpack.ExtType(1, msgpack.packb((a.shape, a.dtype.name, a.tobytes('C')), use_bin_type=True))
  tree = {'step': nd(np.int32(5)), 'x': {'0': nd(np.arange(6, dtype=np.float32).reshape(2, 3)), '1': nd(np.float32(2.5))},
          'chunked': {'__msgpack_chunked_array__': True, 'shape': [4], 'chunks': {'0': nd(np.array([1., 2.], np.float32)),
                                                                                '1': nd(np.array([3., 4.], np.float32))}}}
  out = checkpoints.msgpack_restore(msgpack.packb(tree, strict_types=True))
  assert int(out['step']) == 5 and out['x']['0'].shape == (2, 3) and float(out['x']['1']) == 2.5
  np.testing.assert_array_equal(out['chunked'], [1., 2., 3., 4.])


def test_restore_without_checkpoint_and_outdated_save(tmp_path):
  cfg = configs.load_preset('blender_256')
  model = models.Model(config=cfg).build('cpu')
  st = _state(model, 3)
  assert checkpoints.restore_checkpoint(str(tmp_path / 'none'), model, st) is st       # train.py:84 behaviour
  checkpoints.save_checkpoint(str(tmp_path), model, st, 10, keep=2)
  checkpoints.save_checkpoint(str(tmp_path), model, st, 20, keep=2)
  checkpoints.save_checkpoint(str(tmp_path), model, st, 30, keep=2)
  assert sorted(os.listdir(tmp_path)) == ['checkpoint_20', 'checkpoint_30']
  with pytest.raises(ValueError, match='outdated'):
    checkpoints.save_checkpoint(str(tmp_path), model, st, 25)
