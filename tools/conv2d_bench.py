"""The stage-2 critic's Conv2d layers alone (B = 128, bf16): forward, data gradient and weight gradient of layers 2-5 through
layers.Conv2dLayer (2-D taps in the GEMM row pointers), each timed on an otherwise idle GPU.   python tools/conv2d_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zs_amd  # noqa: E402,F401
from zs_amd import _lib as L, layers  # noqa: E402

os.environ['ZS_OVERLAP_WGRAD'] = '0'
ctx = layers.Ctx('cuda:0', 'bf16')
ctx.overlap_wgrad = False
dev = ctx.device
B = 128


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, H, W, C, Cout in (('layer 2', 64, 257, 64, 128), ('layer 3', 32, 129, 128, 256), ('layer 4', 16, 65, 256, 512), ('layer 5', 8, 33, 512, 512)):
    w = torch.randn(Cout, C, 5, 5, device=dev) * 0.02
    b = torch.zeros(Cout, device=dev)
    lay = layers.Conv2dLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), stride=2, name='cb' + name)
    lay.pack()
    Ho, Wo = lay.out_hw(H, W)
    X = ctx.act('x' + name, B, H * W, C); X.valid().normal_()
    Y = ctx.act('y' + name, B, Ho * Wo, Cout)
    dY = ctx.act('dy' + name, B, Ho * Wo, Cout); dY.valid().normal_()
    dX = ctx.act('dx' + name, B, H * W, C)
    fl = 2.0 * B * Ho * Wo * Cout * C * 25
    tf = timeit(lambda: lay.fwd(X, H, W, Y, act=L.ZS_ACT_LRELU, slope=0.01))
    td = timeit(lambda: lay.dgrad(dY, H, W, dX, 'g' + name))
    tw = timeit(lambda: lay.wgrad(dY, X, H, W, accumulate=True))
    print('%s [%d x %d x %d -> %d]: %.0f GFLOP per pass; forward %.0f us (%.0f TFLOP/s), data gradient %.0f us (%.0f), weight gradient %.0f us (%.0f)' %
          (name, H, W, C, Cout, fl / 1e9, tf, fl / tf / 1e6, td, fl / td / 1e6, tw, fl / tw / 1e6), flush=True)
