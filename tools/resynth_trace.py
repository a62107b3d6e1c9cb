"""One batch of `bench.py --mode resynth` from a rocprofv3 kernel trace: GPU-busy time against wall time, kernel time by name,
and the idle gaps (host-bound stretches) with the kernels on either side.
  python tools/resynth_trace.py <x_kernel_trace.csv> [min_gap_us]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0][:48]
rows.sort(key=lambda r: r['s'])
ends = [i for i, r in enumerate(rows) if r['n'].startswith('gl_deemph_scan_kernel')]
a, b = rows[ends[-2]]['e'], rows[ends[-1]]['e']
sel = [r for r in rows if r['s'] >= a and r['e'] <= b]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
busy, last = 0, a
gaps = []
for i, r in enumerate(sel):
    if r['s'] > last:
        gaps.append(((r['s'] - last) / 1e3, sel[i - 1]['n'] if i else '(batch start)', r['n'], (r['s'] - a) / 1e3))
    busy += max(0, r['e'] - max(last, r['s']))
    last = max(last, r['e'])
print('batch: %.2f ms wall, %.2f ms GPU-busy (union of kernel intervals), %d kernels' % ((b - a) / 1e6, busy / 1e6, len(sel)))
by = collections.defaultdict(lambda: [0, 0])
for r in sel:
    by[r['n']][0] += 1; by[r['n']][1] += r['e'] - r['s']
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:18]:
    print('  %-50s %5d x  %9.3f ms' % (n, c, t / 1e6))
print('idle gaps >= %.0f us (at ms, length, between):' % min_gap)
for g, p, n, at in gaps:
    if g >= min_gap:
        print('  %8.2f ms  %8.1f us  %s -> %s' % (at / 1e3, g, p, n))
print('sum of all gaps: %.2f ms' % (sum(g for g, _, _, _ in gaps) / 1e3))
