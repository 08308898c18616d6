"""Summarise rocprofv3 output into the markdown kept under profiles/.

  python tools/prof_summary.py stats <dir with *_results.db> [--title ...]    # --kernel-trace --stats run
  python tools/prof_summary.py pmc   <dir with *counter_collection.csv>       # --pmc run (--output-format csv)

Runs on the GPU box right after rocprofv3 (no dependency beyond sqlite3 / csv).
"""

import argparse
import collections
import csv
import glob
import os
import sqlite3
import sys


def find(d, pat):
  out = sorted(glob.glob(os.path.join(d, '**', pat), recursive=True))
  if not out:
    sys.exit(f'no {pat} under {d}')
  return out


def short(name, n=70):
  name = name.replace('(anonymous namespace)::', '')
  cut = name.find('(')
  if cut > 0:
    name = name[:cut]
  return name[:n]


def stats(args):
  rows = collections.OrderedDict()
  shapes = collections.defaultdict(list)
  for db in find(args.dir, '*_results.db'):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kt = [t for t in tabs if t == 'kernels' or t.startswith('kernels')]
    if not kt:
      sys.exit(f'{db}: no kernels view; tables: {tabs[:20]}')
    cols = [r[1] for r in cur.execute(f'pragma table_info({kt[0]})')]
    namecol = 'name' if 'name' in cols else 'kernel_name'
    q = f'select {namecol}, start, end, grid_x, grid_y, grid_z from {kt[0]}' if 'grid_x' in cols else \
        f'select {namecol}, start, end, grid_size_x, grid_size_y, grid_size_z from {kt[0]}'
    for name, st, en, gx, gy, gz in cur.execute(q):
      k = short(name)
      d = (en - st) / 1e3
      r = rows.setdefault(k, [0, 0.0, 1e30, 0.0])
      r[0] += 1
      r[1] += d
      r[2] = min(r[2], d)
      r[3] = max(r[3], d)
      if 'gemm' in k:
        shapes[(k, gx * max(gy, 1) * max(gz, 1))].append(d)
  total = sum(r[1] for r in rows.values())
  print(f'# {args.title}\n')
  if args.command:
    print(f'Command: `{args.command}`\n')
  print(f'Total kernel time {total / 1e3:.2f} ms over {sum(r[0] for r in rows.values())} launches.\n')
  print('| kernel | calls | total us | avg us | min us | max us | % |')
  print('|---|---|---|---|---|---|---|')
  for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:args.top]:
    print(f'| `{k}` | {r[0]} | {r[1]:.0f} | {r[1] / r[0]:.1f} | {r[2]:.1f} | {r[3]:.1f} | {100 * r[1] / total:.2f} |')
  if shapes:
    print('\nPer-shape GEMM launches (grid threads -> avg / min us):\n')
    print('| kernel | grid threads | calls | avg us | min us | total us |')
    print('|---|---|---|---|---|---|')
    for (k, g), ds in sorted(shapes.items(), key=lambda kv: -sum(kv[1])):
      print(f'| `{k}` | {g} | {len(ds)} | {sum(ds) / len(ds):.1f} | {min(ds):.1f} | {sum(ds):.0f} |')


def seq(args):
  """The launches of the LAST train step in start order (a step = the launches between two clip_adam_kernel groups), with
  start offsets: which kernel is which layer's, and what overlaps what."""
  rowsq = []
  for db in find(args.dir, '*_results.db'):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kt = [t for t in tabs if t == 'kernels' or t.startswith('kernels')]
    cols = [r[1] for r in cur.execute(f'pragma table_info({kt[0]})')]
    namecol = 'name' if 'name' in cols else 'kernel_name'
    gcols = 'grid_x, grid_y, grid_z' if 'grid_x' in cols else 'grid_size_x, grid_size_y, grid_size_z'
    for name, st, en, gx, gy, gz in cur.execute(f'select {namecol}, start, end, {gcols} from {kt[0]}'):
      rowsq.append((st, en, short(name, 90), gx * max(gy, 1) * max(gz, 1)))
  rowsq.sort()
  ends = [i for i, r in enumerate(rowsq) if 'clip_adam_kernel' in r[2]]
  # the last clip_adam launch closes the last step; the one before the previous group closes the step before it
  groups = [i for k, i in enumerate(ends) if k == 0 or i - ends[k - 1] > 3]
  lo = groups[-2] if len(groups) >= 2 else 0
  step = [r for r in rowsq[lo:ends[-1] + 1]]
  # drop the previous step's trailing optimizer launches
  first = next(i for i, r in enumerate(step) if 'clip_adam' not in r[2] and 'grad_sqnorm' not in r[2])
  step = step[first:]
  t0 = step[0][0]
  print(f'# {args.title}\n')
  if args.command:
    print(f'Command: `{args.command}`\n')
  print(f'{len(step)} launches, {(step[-1][1] - t0) / 1e6:.3f} ms from the first start to the last end.\n')
  print('| # | start us | duration us | grid threads | kernel |')
  print('|---|---|---|---|---|')
  for i, (st, en, k, g) in enumerate(step):
    print(f'| {i} | {(st - t0) / 1e3:.1f} | {(en - st) / 1e3:.1f} | {g} | `{k}` |')


def pmc(args):
  acc = collections.defaultdict(lambda: collections.defaultdict(float))
  calls = collections.defaultdict(lambda: collections.defaultdict(int))
  for f in find(args.dir, '*counter_collection.csv'):
    with open(f) as fh:
      for row in csv.DictReader(fh):
        k = short(row['Kernel_Name'])
        c = row['Counter_Name']
        acc[k][c] += float(row['Counter_Value'])
        calls[k][c] += 1
  counters = sorted({c for k in acc for c in acc[k]})
  print(f'# {args.title}\n')
  if args.command:
    print(f'Command: `{args.command}`\n')
  print('Sums over all dispatches of each kernel (counter value as reported; FETCH_SIZE / WRITE_SIZE are in KiB '
        'and FETCH_SIZE must be doubled on gfx950 per MI355X_MICROARCH.md).\n')
  print('| kernel | dispatches | ' + ' | '.join(counters) + ' |')
  print('|---|---|' + '---|' * len(counters))
  order = sorted(acc, key=lambda k: -max(acc[k].values()))
  for k in order[:args.top]:
    n = max(calls[k].values())
    print(f'| `{k}` | {n} | ' + ' | '.join(f'{acc[k].get(c, 0):.4g}' for c in counters) + ' |')


if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('mode', choices=['stats', 'pmc', 'seq'])
  ap.add_argument('dir')
  ap.add_argument('--title', default='rocprofv3 summary')
  ap.add_argument('--command', default='')
  ap.add_argument('--top', type=int, default=30)
  a = ap.parse_args()
  {'stats': stats, 'pmc': pmc, 'seq': seq}[a.mode](a)
