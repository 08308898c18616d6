"""Run a GPU tool script on the kernel-source simulator to screen its LOGIC before GPU minutes are spent on it:

    python tools/hipsim/run_on_simulator.py tools/gemm_probe.py --M 512 --reps 1 --cfgs 2,36,40,42

('cuda' tensors stay on the host, the package gets the simulator build, CUDA events become wall-clock stamps; the numbers
it prints mean nothing.)"""
import os, sys, time, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['MNR_TESTS_ON_SIMULATOR'] = '1'
import torch
from tests import conftest
conftest._route_gpu_tests_to_the_simulator()
class Ev:
  def __init__(self, enable_timing=True): self.t = 0
  def record(self, *a): self.t = time.perf_counter()
  def elapsed_time(self, o): return (o.t - self.t) * 1e3 + 1e-6
torch.cuda.Event = Ev
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.current_stream = lambda *a, **k: None
_G = torch.Generator


class _HostGenerator(_G):                      # torch.Generator(device='cuda') -> a host generator, still a torch.Generator
  def __new__(cls, device=None):
    return _G.__new__(cls)

  def __init__(self, device=None):
    pass


torch.Generator = _HostGenerator
script = sys.argv[1]
sys.argv = [script] + sys.argv[2:]
runpy.run_path(script, run_name='__main__')
