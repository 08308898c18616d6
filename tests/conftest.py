import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
  if os.environ.get('MNR_TESTS_ON_SIMULATOR') == '1':
    _route_gpu_tests_to_the_simulator()


def _route_gpu_tests_to_the_simulator():
  """MNR_TESTS_ON_SIMULATOR=1 (set only by tests/test_sim_gpu_suite.py for a child pytest): run the `gpu` tests'
  own code on the kernel-source simulator (tools/hipsim).  `.cuda()` / `.to('cuda')` keep tensors on the host, the
  package's library handle is the simulator build, torch.cuda.* queries the tests make are answered as a one-GPU box.
  A development screen for the test code and the kernel source; never active in the driver's `-m gpu` run."""
  import torch
  from torch.overrides import TorchFunctionMode
  from tests import sim_helpers
  ctx = sim_helpers.simulated_device()
  ctx.__enter__()                                  # for the whole (child) session

  def is_cuda(x):
    return (isinstance(x, str) and x.startswith('cuda')) or (isinstance(x, torch.device) and x.type == 'cuda')

  class CudaIsHost(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
      kwargs = dict(kwargs or {})
      if func is torch.Tensor.cuda:
        return args[0]
      if is_cuda(kwargs.get('device')):
        kwargs['device'] = 'cpu'
      args = tuple('cpu' if is_cuda(a) else a for a in args)
      return func(*args, **kwargs)

  mode = CudaIsHost()
  mode.__enter__()
  torch.cuda.is_available = lambda: True
  torch.cuda.synchronize = lambda *a, **k: None


@pytest.fixture(scope='session')
def golden():
  import numpy as np
  return np.load(os.path.join(ROOT, 'tests', 'golden', 'leaves.npz'))
