"""Epilogue microbenchmark of the conv GEMM at the decoder's shapes (default dispatch = 256x256 kernel): the same K loop with
each epilogue the step uses, so that the cost of an epilogue variant is the difference to `plain`.
  python tools/epi_bench.py [--iters 20 --rounds 3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zs_amd  # noqa: E402,F401
from zs_amd import _lib as L, layers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=20); ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--only', default='')
a = ap.parse_args()
ctx = layers.Ctx('cuda:0', 'bf16')
dev = ctx.device
B, T, C = 256, 128, 1024


def conv(cout, cin, k, **kw):
    w = torch.randn(cout, cin, k, device=dev) * 0.02 if k > 1 else torch.randn(cout, cin, device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    l = layers.ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), **kw)
    l.pack()
    return l


def act(name, B_, T_, C_):
    x = ctx.act(name, B_, T_, C_)
    x.t.normal_()
    return x


lin = conv(C, C, 1)
lin3 = conv(3 * C, C, 1)
c3 = conv(2 * C, C, 3, split2=True)
X, Y, Y2, D, R = act('x', B, T, C), act('y', B, T, C), act('y2', B, T, C), act('d', B, T, C), act('r', B, T, C)
Y3 = act('y3', B, T, 3 * C)
X64, YA, S = act('x64', B, 64, C), act('ya', B, 64, 2 * C), act('s', B, 128, C)
emb = torch.randn(102, C, device=dev)
idx = torch.randint(0, 102, (B,), device=dev)
cs = torch.zeros(B, C, device=dev)
col = (cs.data_ptr(), C, 0)
CASES = {
    'plain  M32768 N1024 K1024 (bias+lrelu)': (lambda: lin.fwd(X, out=Y, act=L.ZS_ACT_LRELU, slope=0.01), 2.0 * B * T * C * C),
    'dual   + out2 = out + emb[idx]': (lambda: lin.fwd(X, out=Y, act=L.ZS_ACT_LRELU, slope=0.01, out2=Y2, vec2=emb, idx=idx), 2.0 * B * T * C * C),
    'plain  N3072 (GRU input projection)': (lambda: lin3.fwd(X, out=Y3), 6.0 * B * T * C * C),
    'split2 conv k3 1024->2048 T64 + out2 shuffled + emb': (lambda: c3.fwd(X64, out=YA, act=L.ZS_ACT_LRELU, slope=0.01, out2=S, vec2=emb, idx=idx,
                                                                              store_mode2=L.ZS_STORE_SPLIT2), 2.0 * B * 64 * 2 * C * C * 3),
    'dgrad  plain': (lambda: lin.dgrad(Y, T, D), 2.0 * B * T * C * C),
    'dgrad  * lrelu\'(dact)': (lambda: lin.dgrad(Y, T, D, dact_src=X, slope=0.01), 2.0 * B * T * C * C),
    'dgrad  + add_src': (lambda: lin.dgrad(Y, T, D, add_src=R), 2.0 * B * T * C * C),
    'dgrad  mask + colsum': (lambda: lin.dgrad(Y, T, D, dact_src=X, slope=0.01, colsum=col), 2.0 * B * T * C * C),
    'dgrad  add + colsum': (lambda: lin.dgrad(Y, T, D, add_src=R, colsum=col), 2.0 * B * T * C * C),
}


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters


res = {k: [] for k in CASES}
for r in range(a.rounds):
    for k, (fn, fl) in CASES.items():
        if a.only and a.only not in k:
            continue
        res[k].append(timed(fn))
for k, (fn, fl) in CASES.items():
    if res[k]:
        t = min(res[k])
        print('%-56s %7.1f us  %6.0f TFLOP/s   (rounds: %s)' % (k, t * 1e3, fl / t / 1e9, ' '.join('%.1f' % (x * 1e3) for x in res[k])))
