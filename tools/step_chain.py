"""Kernel-by-kernel listing of one replayed train_ae step from a rocprofv3 kernel trace (start offset, duration, queue,
how many other kernels overlap it) -- what sits on the dependency chain and where the gaps are.
  python tools/step_chain.py <x_kernel_trace.csv> [step_index] [min_us]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0][:40]
    r['q'] = r.get('Queue_Id', r.get('Stream_Id', '?'))
rows.sort(key=lambda r: r['s'])
ad = [i for i, r in enumerate(rows) if r['n'].startswith('adam_kernel')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, len(ad) // 4)
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
a, b = rows[ad[2 * k + 1]]['e'], rows[ad[2 * k + 3]]['e']
sel = [r for r in rows if r['s'] >= a and r['e'] <= b]
qs = {}
for r in sel:
    qs.setdefault(r['q'], len(qs))
print('step %d: %.3f ms, %d kernels, %d queues' % (k, (b - a) / 1e6, len(sel), len(qs)))
print('%9s %8s %3s %3s %8s  %s' % ('start_us', 'dur_us', 'q', 'ov', 'gap_us', 'kernel (grid)'))
last_end = a
for i, r in enumerate(sel):
    ov = sum(1 for o in sel if o is not r and o['s'] < r['e'] and o['e'] > r['s'])
    gap = (r['s'] - last_end) / 1e3          # idle time on the whole device before this kernel (negative: overlapped)
    last_end = max(last_end, r['e'])
    d = (r['e'] - r['s']) / 1e3
    if d >= min_us:
        grid = r.get('Grid_Size', r.get('Grid_Size_X', ''))
        wg = r.get('Workgroup_Size', r.get('Workgroup_Size_X', ''))
        print('%9.1f %8.1f %3d %3d %8.1f  %s (%s/%s)' % ((r['s'] - a) / 1e3, d, qs[r['q']], ov, gap, r['n'], grid, wg))
