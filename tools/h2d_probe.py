"""Host->device bandwidth of this box for the batch of BASELINE config 2 (256 x 128 x 513 fp32 = 67.2 MB): hipMemcpyAsync from
pinned memory, from pageable memory, and the zs_host_fetch kernel (PCIe reads issued by the GPU)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zs_amd  # noqa: E402,F401
from zs_amd import _lib as L  # noqa: E402

dev = torch.device('cuda:0')
n = 256 * 128 * 513
hp = torch.zeros(n).pin_memory()
hq = torch.zeros(n)
d = torch.zeros(n, device=dev)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for name, fn in (('memcpy pinned -> device', lambda: d.copy_(hp, non_blocking=True)),
                 ('memcpy pageable -> device', lambda: d.copy_(hq)),
                 ('memcpy device -> pinned', lambda: hp.copy_(d, non_blocking=True))):
    dt = timeit(fn)
    print('%-34s %7.2f ms  %6.1f GB/s' % (name, dt * 1e3, n * 4 / dt / 1e9), flush=True)
st = torch.cuda.current_stream().cuda_stream
for wg in (8, 32, 128, 512):
    dt = timeit(lambda: L.check(L.lib().zs_host_fetch(L.ptr(hp), L.ptr(d), n * 4, wg, st), 'zs_host_fetch'))
    print('%-34s %7.2f ms  %6.1f GB/s' % ('zs_host_fetch, %d workgroups' % wg, dt * 1e3, n * 4 / dt / 1e9), flush=True)
