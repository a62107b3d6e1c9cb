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

# the first layer (one input channel) straight from the image: zs_conv1_fwd / zs_conv1_wgrad against im2col + GEMM
H, W, Cout, k = 128, 513, 64, 5
Ho, Wo = (H + 4 - 5) // 2 + 1, (W + 4 - 5) // 2 + 1
x = torch.rand(B, H, W, device=dev)
wv = torch.randn(Cout, 25, device=dev) * 0.1
lay = layers.ConvLayer(ctx, wv, torch.zeros(Cout, device=dev), torch.zeros_like(wv), torch.zeros(Cout, device=dev), name='c1')
lay.pack()
out = ctx.act('c1out', B * Ho, Wo, Cout)
gz = ctx.act('c1gz', B * Ho, Wo, Cout); gz.valid().normal_()
xh = ctx.act('c1xh', B * Ho, Wo, 25, ld=lay.cin_pad)
ws = torch.empty(L.lib().zs_conv1_wgrad_workspace() // 4, device=dev)
t_d = timeit(lambda: L.call('zs_conv1_fwd', 'ZsConv1Fwd', ctx.stream, x=L.ptr(x), W=L.ptr(lay.wf), ldw=lay.ldw, bias=L.ptr(lay.b), act=L.ZS_ACT_LRELU,
                            slope=0.01, out=out.ptr(), ldo=out.ld, B=B, H=H, Wd=W, Cout=Cout, k=k, pad_mode=L.ZS_PAD_REFLECT))
t_g = timeit(lambda: L.call('zs_conv2d_gather', 'ZsConv2dGather', ctx.stream, dtype=ctx.dt, x=L.ptr(x), ldx=1, x_f32=1, out=xh.ptr(), ldo=xh.ld, B=B,
                            H_in=H, H_out=Ho, Wd=W, C=1, k=k, stride=2, pad=2, pad_mode=L.ZS_PAD_REFLECT, full=1))
t_f = timeit(lambda: lay.fwd(xh, out=out, act=L.ZS_ACT_LRELU, slope=0.01))
t_w = timeit(lambda: L.call('zs_conv1_wgrad', 'ZsConv1Wgrad', ctx.stream, x=L.ptr(x), gz=gz.ptr(), ldg=gz.ld, dW=L.ptr(lay.gw), lddw=25, db=L.ptr(lay.gb),
                            accumulate=1, B=B, H=H, Wd=W, Cout=Cout, k=k, pad_mode=L.ZS_PAD_REFLECT, workspace=L.ptr(ws), workspace_bytes=ws.numel() * 4))
t_wo = timeit(lambda: lay.wgrad(gz, xh, accumulate=True))
mb = B * Ho * Wo * Cout * 2 / 1e6
print('layer 1 [%d x %d x 1 -> %d], %.0f MB of output: forward direct %.0f us; im2col %.0f + GEMM %.0f us.  weight gradient direct %.0f us; GEMM over the im2col rows %.0f us' %
      (H, W, Cout, mb, t_d, t_g, t_f, t_w, t_wo), flush=True)
