"""Wall-clock split of one replayed train_ae step from a rocprofv3 kernel trace: for how long which kernel classes are active.
  python tools/step_timeline.py gpurun_out/prof/x_kernel_trace.csv [step_index]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0][:44]
rows.sort(key=lambda r: r['s'])
# a step ends with the encoder's Adam launch (the decoder's runs earlier on the optimizer stream): every second adam_kernel
ad = [i for i, r in enumerate(rows) if r['n'].startswith('adam_kernel')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, len(ad) // 4)      # default: a step inside the timed (graph replay) region of bench.py
a, b = rows[ad[2 * k + 1]]['e'], rows[ad[2 * k + 3]]['e']
sel = [r for r in rows if r['s'] >= a and r['e'] <= b]


def cls(n):
    if n.startswith('rowblock') or n.startswith('gru_'):
        return 'gru'
    if 'wgrad' in n:
        return 'wgrad'
    if n.startswith('gemm_conv'):
        return 'conv'
    return 'other'


ev = []
for r in sel:
    c = cls(r['n']); ev.append((r['s'], 1, c)); ev.append((r['e'], -1, c))
ev.sort()
act = collections.Counter(); last = ev[0][0]; tm = collections.Counter()
for t, d, c in ev:
    if t > last:
        tm['+'.join(sorted(k for k, v in act.items() if v > 0)) or 'idle'] += t - last
        last = t
    act[c] += d
print('step %d: %.2f ms, %d kernels' % (k, (b - a) / 1e6, len(sel)))
for key, v in sorted(tm.items(), key=lambda kv: -kv[1]):
    if v > 5000:
        print('  %-26s %6.2f ms' % (key, v / 1e6))
per = collections.Counter()
for r in sel:
    per[r['n']] += r['e'] - r['s']
print('kernel time by name:')
for key, v in per.most_common(16):
    print('  %-46s %6.3f ms' % (key, v / 1e6))
