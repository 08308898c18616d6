"""cast_rays_ipe at the 360.gin shapes (proposal level: 16384 rays x 64 samples, NeRF level: x 32; 21 directions x 12 degrees -> 504 of
512 bf16 columns): us per launch and TB/s of the feature rows.  With MNR_LIB_PATH = a -DFE_DBG=1 / 2 build of csrc/features.hip:
the same launch without its encoding loop / without its write-out (timing only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multinerf_amd import configs, geopoly, ops  # noqa: E402

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


B = 16384
basis = torch.tensor(geopoly.generate_basis('icosahedron', 2), dtype=torch.float32).contiguous().to(dev)        # [21, 3]
K = basis.shape[0]
o = torch.randn((B, 3), generator=g, device=dev)
d = torch.nn.functional.normalize(torch.randn((B, 3), generator=g, device=dev), dim=-1)
radii = torch.full((B,), 1e-3, device=dev)
for n in (64, 32):
  t = torch.sort(torch.rand((B, n + 1), generator=g, device=dev) * 6.0 + 0.2, dim=-1).values
  feat = torch.empty((B * n, 512), dtype=torch.bfloat16, device=dev)
  fn = lambda: ops.cast_rays_ipe(t, o, d, radii, basis, ray_shape='cone', warp_contract=True, min_deg=0, max_deg=12, ld_feat=512, out=feat)
  us = timed(fn)
  print(f'{os.environ.get("MNR_LIB_PATH", "product"):60s} n = {n}: {us:7.1f} us  {feat.numel() * 2 / us / 1e6:5.2f} TB/s of rows', flush=True)
