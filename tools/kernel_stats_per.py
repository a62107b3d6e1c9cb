"""Kernel time by name from a rocprofv3 *_kernel_stats.csv, divided by a unit count (steps).  python tools/kernel_stats_per.py <csv> <n_units> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per unit: %.2f ms' % (tot / n / 1e6))
for r in rows[:top]:
    name = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:64]
    print('%-64s %7.1f calls %8.3f ms  avg %8.1f us' % (name, float(r['Calls']) / n, float(r['TotalDurationNs']) / n / 1e6, float(r['AverageNs']) / 1e3))
