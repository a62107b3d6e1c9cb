"""Per (kernel, grid) HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE): launches of one kernel with
different shapes kept apart.   python tools/pmc_by_launch.py fetch_counter_collection.csv write_counter_collection.csv [kernel substring]"""
import collections
import csv
import re
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'\(.*$', '', name).replace('void ', '')
        k = (name, int(r['Grid_Size']))
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
    return agg


f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
sub = sys.argv[3] if len(sys.argv) > 3 else ''
print('%-44s %10s %5s %12s %12s' % ('kernel', 'grid', 'n', 'fetch MB x2', 'write MB'))
for k in sorted(f, key=lambda k: -f[k][1]):
    if sub not in k[0]:
        continue
    n = f[k][0]
    print('%-44s %10d %5d %12.1f %12.1f' % (k[0][:44], k[1], n, f[k][1] * 1024 * 2 / n / 1e6, w.get(k, [1, 0.0])[1] * 1024 / max(1, w.get(k, [1])[0]) / 1e6))
