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
ap.add_argument('--variants', default='p8m16,ring,dma')
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--fill', default='normal', help='normal | zeros | ones: operand values (MFMA power depends on bit toggling)')
a = ap.parse_args()
os.environ['ZS_OVERLAP_WGRAD'] = '0'      # time the weight gradient on the stream the events are recorded on
ctx = layers.Ctx('cuda:0', a.dtype)
dev = ctx.device
w = torch.randn(a.cout, a.cin, a.k, device=dev) * 0.02
if a.fill in ('zeros', 'ones'):
    w.fill_(0.0 if a.fill == 'zeros' else 1.0)
elif a.fill == 'uniform':
    w.uniform_(-1, 1)
b = torch.zeros(a.cout, device=dev)
l = layers.ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b))
l.pack()
X = ctx.act('x', a.B, a.T, a.cin)
X.t.normal_()
if a.fill in ('zeros', 'ones'):
    X.t.fill_(0.0 if a.fill == 'zeros' else 1.0)
elif a.fill == 'uniform':
    X.t.uniform_(-1, 1)
elif a.fill == 'lrelu':
    X.t.copy_(torch.nn.functional.leaky_relu(X.t.float(), 0.01))
elif a.fill == 'drop':
    X.t.mul_((torch.rand(X.t.shape, device=dev) < 0.5).to(X.t.dtype) * 2)
Y = ctx.act('y', a.B, a.T, a.cout)
Y.t.normal_()
if a.fill in ('zeros', 'ones'):
    Y.t.fill_(0.0 if a.fill == 'zeros' else 1.0)
GP = ctx.act('gp', a.B, a.T + l.pad_l + l.pad_r, a.cin)


def run():
    if a.mode == 'fwd':
        l.fwd(X, out=Y, act=L.ZS_ACT_LRELU, slope=0.01)
    elif a.mode == 'dgrad':
        l.dgrad(Y, a.T, GP)
    else:
        l.wgrad(Y, X)


def timed():
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters


fl = 2.0 * a.B * a.T * a.cout * a.cin * a.k
VARIANTS = {'p8m16': (1, 1, 1, 1, 1, 1), 'pp': (1, 1, 1, 1, 0, 200), 'ring': (1, 1, 1, 0, 0, 200), 'dma': (1, 0, 256, 0, 0, 200), 'reg': (0, 0, 256, 0, 0, 200)}
names = a.variants.split(',') if a.mode != 'wgrad' else ['p8', 't128']


def select(name):
    if a.mode == 'wgrad':
        L.set_option('wgrad_p8', 1 if name == 'p8' else 0)
    else:
        for k, v in zip(('gemm_dma', 'gemm_ring', 'gemm_ring_min_tiles', 'gemm_pp', 'gemm_p8', 'gemm_p8_min_tiles'), VARIANTS[name]):
            L.set_option(k, v)


# interleaved rounds (A B C A B C ...): the chip is power-capped, so a variant measured first / on a cool chip reads high
res = {n: [] for n in names}
for rnd in range(a.rounds):
    for n in names:
        select(n)
        res[n].append(timed())
for n in names:
    ms = sorted(res[n])[len(res[n]) // 2]
    print('%s %s %-5s %s B%d T%d cin%d cout%d k%d: %.3f ms  %.1f TFLOP/s  (median of %d interleaved rounds)' % (a.mode, a.dtype, n, a.fill, a.B, a.T, a.cin, a.cout, a.k, ms, fl / ms / 1e9, a.rounds), flush=True)
