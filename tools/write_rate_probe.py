import torch
def timed(fn, reps=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps
for gb in (0.5, 1.0, 2.0):
  n = int(gb * (1 << 30)) // 2
  x = torch.empty(n, dtype=torch.bfloat16, device='cuda'); y = torch.empty_like(x)
  t = timed(lambda: x.zero_()); print(f'zero_ {gb} GiB: {t:.1f} us = {n*2/t/1e6:.2f} TB/s')
  t = timed(lambda: x.fill_(1.5)); print(f'fill_ {gb} GiB: {t:.1f} us = {n*2/t/1e6:.2f} TB/s')
  t = timed(lambda: y.copy_(x)); print(f'copy_ {gb} GiB: {t:.1f} us = {2*n*2/t/1e6:.2f} TB/s (read + write)')
  t = timed(lambda: x.sum()); print(f'sum   {gb} GiB: {t:.1f} us = {n*2/t/1e6:.2f} TB/s (read)')
  del x, y
