"""models._RANK1_LAST on against off (the proposal MLP's last dY built inside its weight-gradient GEMM instead of stored):
the two kernels at the 360.gin proposal shape (both levels in one pass: 2^21 rows x 256), one train step at full width on 2048
rays (gradients must agree up to the order of the weight-gradient atomics), then bench.py lines of both arms, twice, on one box."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multinerf_amd import configs, models, ops, synthetic, train_utils  # noqa: E402

dev, bf = 'cuda', torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=10):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


M, W, D = 1 << 21, 256, 4
x = torch.relu(torch.randn((M, W), generator=g, device=dev)).to(bf)
gh = 0.1 * torch.randn((M,), generator=g, device=dev)
wh = torch.randn((W,), generator=g, device=dev)
bits = [torch.randint(0, 256, (M, W // 8), generator=g, device=dev, dtype=torch.uint8) for _ in range(D)]
Bws = [None] + [(torch.randn((W, W), generator=g, device=dev) * 0.06).to(bf) for _ in range(1, D)]
dYs = [torch.zeros((M, W), dtype=bf, device=dev) for _ in range(D)]
t_store = timed(lambda: ops.mlp_chain_bwd(gh, wh, bits, Bws, dYs, M=M, W=W))
t_skip = timed(lambda: ops.mlp_chain_bwd(gh, wh, bits, Bws, dYs[:-1] + [None], M=M, W=W))
print(f'mlp_chain_bwd, {M} rows x {W}, depth {D}: last dY stored {t_store:.1f} us, not stored {t_skip:.1f} us', flush=True)
C0, b0 = torch.zeros((W, W), device=dev), torch.zeros((W,), device=dev)
C1, b1 = torch.zeros((W, W), device=dev), torch.zeros((W,), device=dev)
ops.gemm_tn(x, dYs[-1], C0, M=M, K=W, N=W, bias_out=b0, bias_n_valid=W)
ops.gemm_tn(x, None, C1, M=M, K=W, N=W, bias_out=b1, bias_n_valid=W, rank1=(gh, wh, bits[-1]))
torch.cuda.synchronize()
rel = ((C1 - C0).norm() / C0.norm()).item()
relb = ((b1 - b0).norm() / b0.norm()).item()
print(f'rank1 dW against the stored dY: |dC| / |C| = {rel:.3e}, bias {relb:.3e}', flush=True)
assert rel < 1e-5 and relb < 1e-5
t_b = timed(lambda: ops.gemm_tn(x, dYs[-1], C0, M=M, K=W, N=W, bias_out=b0, bias_n_valid=W))
t_r = timed(lambda: ops.gemm_tn(x, None, C1, M=M, K=W, N=W, bias_out=b1, bias_n_valid=W, rank1=(gh, wh, bits[-1])))
t_b2 = timed(lambda: ops.gemm_tn(x, dYs[-1], C0, M=M, K=W, N=W, bias_out=b0, bias_n_valid=W))
t_r2 = timed(lambda: ops.gemm_tn(x, None, C1, M=M, K=W, N=W, bias_out=b1, bias_n_valid=W, rank1=(gh, wh, bits[-1])))
print(f'gemm_tn {M} x {W} x {W}: B stored {t_b:.1f} / {t_b2:.1f} us, B built in the kernel {t_r:.1f} / {t_r2:.1f} us', flush=True)
del x, dYs, bits, Bws
torch.cuda.empty_cache()

cfg = configs.load_preset('360')
model = models.Model(config=cfg).build('cuda')
flat = model.init_flat_params(seed=3)
B = 2048
batch = synthetic.synthetic_rays(B, near=cfg.near, far=cfg.far).map(lambda t: t.cuda())
gs = {}
for on in (False, True, False):
  models._RANK1_LAST = on
  state, _ = train_utils.create_optimizer(cfg, {'flat': flat.clone(), 'params': None})
  gen = torch.Generator(device='cuda').manual_seed(5)
  _, stats, _ = train_utils.create_train_step(model, cfg)(gen, state, batch, None, 0.5, 0.0, return_grads=True)
  torch.cuda.synchronize()
  gr = stats['_grads'].double()
  if on in gs:
    print(f'off vs off (atomics noise): {((gr - gs[on]).norm() / gr.norm()).item():.3e}')
  gs.setdefault(on, gr)
rel = ((gs[True] - gs[False]).norm() / gs[False].norm()).item()
print(f'rank1 on vs off: |dg| / |g| = {rel:.3e}, finite {bool(torch.isfinite(gs[True]).all())}', flush=True)
assert rel < 1e-3
del model, flat, batch, gs
torch.cuda.empty_cache()
for arm in (0, 1, 0, 1):
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no_cpu_baseline', '--no_aux', '--rank1_last', str(arm)],
                     capture_output=True, text=True, env=dict(os.environ, MNR_SKIP_PREFLIGHT='1'))
  try:
    b = json.loads(r.stdout.strip().splitlines()[-1])
    print(f'rank1_last={arm}: {b["value"]:.0f} rays/s {b["ms_per_step"]:.3f} ms  mfma union {b["roofline"]["gemm_ms_per_step"]:.2f} ms  final_loss {b["config"]["final_loss"]:.7f}', flush=True)
  except Exception as e:
    print('arm', arm, 'failed', e, r.stderr[-500:])
