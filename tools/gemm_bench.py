"""Microbenchmark of gemm_conv_kernel on one shape (used under rocprofv3 for PMC counters).
  python tools/gemm_bench.py [--B 256 --T 64 --cin 1024 --cout 2048 --k 3 --dtype bf16 --iters 20 --mode fwd|dgrad|wgrad]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zs_amd  # noqa: E402,F401
from zs_amd import _lib as L, layers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=256); ap.add_argument('--T', type=int, default=64)
ap.add_argument('--cin', type=int, default=1024); ap.add_argument('--cout', type=int, default=2048)
ap.add_argument('--k', type=int, default=3); ap.add_argument('--dtype', default='bf16')
ap.add_argument('--iters', type=int, default=20); ap.add_argument('--mode', default='fwd')
a = ap.parse_args()
ctx = layers.Ctx('cuda:0', a.dtype)
dev = ctx.device
w = torch.randn(a.cout, a.cin, a.k, device=dev) * 0.02
b = torch.zeros(a.cout, device=dev)
l = layers.ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b))
l.pack()
X = ctx.act('x', a.B, a.T, a.cin)
X.t.normal_()
Y = ctx.act('y', a.B, a.T, a.cout)
Y.t.normal_()
GP = ctx.act('gp', a.B, a.T + l.pad_l + l.pad_r, a.cin)


def run():
    if a.mode == 'fwd':
        l.fwd(X, out=Y, act=L.ZS_ACT_LRELU, slope=0.01)
    elif a.mode == 'dgrad':
        l.dgrad(Y, a.T, GP)
    else:
        l.wgrad(Y, X)


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.iters):
    run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / a.iters
fl = 2.0 * a.B * a.T * a.cout * a.cin * a.k
print('%s %s B%d T%d cin%d cout%d k%d: %.3f ms  %.1f TFLOP/s' % (a.mode, a.dtype, a.B, a.T, a.cin, a.cout, a.k, ms, fl / ms / 1e9))
