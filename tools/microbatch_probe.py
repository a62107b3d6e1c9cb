"""Upper-bound probe for micro-batch pipelining: two independent half-batch train_ae steps (separate weights, separate
hipGraphs) replayed on two streams at once, against one full-batch step.  The pair does the optimizer twice, so the pair's
time minus one optimizer pass (~1 ms) bounds what a real two-micro-batch schedule could reach."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zs_amd  # noqa: E402,F401
from zs_amd.model import Decoder, Encoder  # noqa: E402
from zs_amd.trainer import AEStep  # noqa: E402

dev = torch.device('cuda', 0)
seg_len, F, E, ch, nspk = 128, 513, 1024, 1024, 102


def make(B, seed):
    torch.manual_seed(seed)
    enc = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=seg_len, enc_mode='multilabel_binary', dtype='bf16').to(dev)
    dec = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=seg_len, dtype='bf16').to(dev)
    ae = AEStep(enc, dec, use_graph=True)
    x = (torch.rand(B, seg_len, F) * (1 - 1e-8) + 1e-8).to(dev)
    c = torch.randint(0, nspk, (B,)).to(dev)
    return ae, x, c


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full, xf, cf = make(256, 1)
for _ in range(4):
    full.step(xf, cf)
torch.cuda.synchronize()
print('one full-batch step (B=256): %.2f ms' % timed(lambda: full.step(xf, cf)), flush=True)

h1, x1, c1 = make(128, 2)
h2, x2, c2 = make(128, 3)
for _ in range(4):
    h1.step(x1, c1); h2.step(x2, c2)
torch.cuda.synchronize()
print('one half-batch step alone (B=128): %.2f ms' % timed(lambda: h1.step(x1, c1)), flush=True)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def pair():
    with torch.cuda.stream(s1):
        h1.step(x1, c1)
    with torch.cuda.stream(s2):
        h2.step(x2, c2)


print('two half-batch steps on two streams: %.2f ms per pair' % timed(pair), flush=True)
