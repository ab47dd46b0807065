"""usage: python tools/trace_gaps.py <dir with *kernel_trace.csv>   -- per kernel: calls, mean duration; and the idle time between consecutive
dispatches on the GPU (launch gaps + host synchronisations) of the LAST batch in the trace (batches are split at k_counts, the last kernel of wfs_run)"""
import csv, glob, os, sys, collections
f = max(glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('void ', '').split('(')[0].split('<')[0]
ends = [i for i, r in enumerate(rows) if name(r) == 'k_counts']
if len(ends) < 2:
    raise SystemExit('fewer than two batches in the trace')
batch = rows[ends[-2] + 1:ends[-1] + 1]
t0, t1 = int(batch[0]['Start_Timestamp']), int(batch[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in batch)
gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp']), name(a), name(b)) for a, b in zip(batch[:-1], batch[1:])]
print(f'last batch: {len(batch)} dispatches, span {(t1 - t0) / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms')
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f'  gap {g / 1e3:8.1f} us  after {a}  before {b}')
per = collections.defaultdict(list)
for r in batch:
    per[name(r)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f'  {k:24s} {len(v):3d} x {sum(v) / len(v) / 1e3:9.1f} us = {sum(v) / 1e6:7.3f} ms')
