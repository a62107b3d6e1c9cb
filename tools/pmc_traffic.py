"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes).  FETCH_SIZE/WRITE_SIZE are in KiB-ish units of 1 KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for
wide coalesced reads, so fetched bytes are doubled (MI355X_MICROARCH.md, HBM section).
  python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv out.json"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'\(.*$', '', name).replace('void ', '')
        a = agg[name]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return agg


f = per_kernel(sys.argv[1], 'FETCH_SIZE')
w = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate runs) -- python3 bench.py --steps 3 --warmup 1 '
                  '--no-cpu-baseline --no-kernel-events',
       'correction': 'gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads: fetched bytes x2 '
                     '(MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is; both counters are in KB',
       'kernels': {}}
tot_f = tot_w = 0.0
for k in sorted(f, key=lambda k: -f[k][1]):
    n = f[k][0]
    fb = f[k][1] * 1024.0 * 2.0
    wb = w.get(k, [0, 0.0])[1] * 1024.0
    tot_f += fb
    tot_w += wb
    out['kernels'][k] = {'launches': n, 'fetch_bytes_per_launch_corrected': fb / n, 'write_bytes_per_launch': wb / max(1, w.get(k, [1])[0]),
                         'traffic_bytes_per_launch': fb / n + wb / max(1, w.get(k, [1])[0])}
out['whole_run_gb'] = {'fetch_corrected': tot_f / 1e9, 'write': tot_w / 1e9}
dom = 'gemm_conv_p8m16_kernel<unsigned short>'
if dom in out['kernels']:
    out['kernel'] = 'gemm_conv_p8m16_kernel<bf16>'
    out['traffic_bytes_per_launch'] = out['kernels'][dom]['traffic_bytes_per_launch']
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k, v in list(out['kernels'].items())[:12]:
    print('%-50s n=%5d fetch %8.1f MB write %8.1f MB per launch' % (k[:50], v['launches'], v['fetch_bytes_per_launch_corrected'] / 1e6, v['write_bytes_per_launch'] / 1e6))
print('whole run GB', out['whole_run_gb'])
