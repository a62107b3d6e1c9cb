"""Where the in-graph host fetch runs inside a replayed step (rocprofv3 kernel trace of `bench.py --host-input`):
  python tools/fetch_trace.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0][:40]
rows.sort(key=lambda r: r['s'])
fetch = [r for r in rows if r['n'].startswith('host_fetch') and r['e'] - r['s'] > 200000]
print('%d big host_fetch launches' % len(fetch))
for f in fetch[-4:]:
    conc = [r for r in rows if r['e'] > f['s'] and r['s'] < f['e'] and r is not f]
    print('fetch %.3f ms: %d kernels overlap it' % ((f['e'] - f['s']) / 1e6, len(conc)))
    for r in conc[:12]:
        print('    %-40s start %+8.3f ms  dur %7.3f ms' % (r['n'], (r['s'] - f['s']) / 1e6, (r['e'] - r['s']) / 1e6))
    prev = [r for r in rows if r['e'] <= f['s']][-2:]
    nxt = [r for r in rows if r['s'] >= f['e']][:2]
    for r in prev:
        print('  before: %-36s ended %+8.3f ms' % (r['n'], (r['e'] - f['s']) / 1e6))
    for r in nxt:
        print('  after : %-36s starts %+8.3f ms after the fetch ends' % (r['n'], (r['s'] - f['e']) / 1e6))
ad = [r for r in rows if r['n'].startswith('adam_kernel')]
steps = [(ad[i + 2]['e'] - ad[i]['e']) / 1e6 for i in range(len(ad) - 2)][-8::2]
print('step times (ms):', ['%.2f' % s for s in steps])
