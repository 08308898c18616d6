"""Side-by-side HIP streams for the backward pass.

With stop_level_grad (reference models.py:200-201) the sampling levels are independent in the backward pass.  `BackwardStreams`
runs the proposal levels' backward (HBM-bound) on a HIP stream of its own next to the NeRF level's (MFMA- and power-bound) on
the caller's stream; both launch full-size grids and the hardware fills the CUs one kernel's tail leaves idle with the other's
workgroups (round 3, same box: 487.6 k -> 500.5 k rays/s at 360.gin).  The arms that measured worse (CU-masked stream pairs, a
stream per proposal level, the weight-gradient GEMMs on a third stream) are recorded in profiles/HISTORY.md and gone.

Environment: MNR_SIDE_STREAM = 1 | 0 (A/B switch, default on).
"""

import os

import torch


class BackwardStreams:
  """The side stream the train step runs the proposal levels' backward on."""

  def __init__(self, device):
    self.device = torch.device(device)
    self.prop = torch.cuda.Stream(device=self.device)

  @staticmethod
  def from_env(device):
    if os.environ.get('MNR_SIDE_STREAM', '1') == '0':
      return None
    return BackwardStreams(device)


def describe_env():
  """What MNR_SIDE_STREAM asks for (bench.py records it next to its numbers)."""
  return {'side_stream': os.environ.get('MNR_SIDE_STREAM', '1') != '0'}
