"""Union of the kernel intervals of one train_ae step in a rocprofv3 kernel trace (GPU-busy time: what the step would take if the
host fed it without gaps) and the span of the step.  python tools/busy_time.py <kernel_trace.csv> [step]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0]
rows.sort(key=lambda r: r['s'])
ad = [i for i, r in enumerate(rows) if r['n'].startswith('adam_kernel')]
for k in ([int(sys.argv[2])] if len(sys.argv) > 2 else range(2, min(8, len(ad) // 2 - 2))):
    a, b = rows[ad[2 * k + 1]]['e'], rows[ad[2 * k + 3]]['e']
    sel = [r for r in rows if r['s'] >= a and r['e'] <= b]
    busy, cur_s, cur_e = 0, None, None
    for r in sel:
        if cur_e is None or r['s'] > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = r['s'], r['e']
        else:
            cur_e = max(cur_e, r['e'])
    busy += (cur_e - cur_s) if cur_e is not None else 0
    print('step %d: span %.3f ms, busy (union of kernels) %.3f ms, kernel time (sum) %.3f ms, %d kernels' %
          (k, (b - a) / 1e6, busy / 1e6, sum(r['e'] - r['s'] for r in sel) / 1e6, len(sel)))
