"""Per-call table of every zs_gemm_conv / zs_gemm_wgrad launch in one --train_ae step at the bench shape: rows, N, K, isolated
duration (each call bracketed by a device synchronise) and TFLOP/s, grouped by shape.
  python tools/gemm_shapes.py [--dtype bf16] [--batch 256]"""
import argparse
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zs_amd  # noqa: E402,F401
from zs_amd import _lib as L, layers  # noqa: E402
from zs_amd.model import Decoder, Encoder  # noqa: E402
from zs_amd.trainer import AEStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16'); ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--other', action='store_true', help='also time every non-GEMM entry point')
a = ap.parse_args()
os.environ['ZS_OVERLAP_WGRAD'] = '0'
dev = torch.device('cuda', 0)
seg_len, F, E, ch, nspk, B = 128, 513, 1024, 1024, 102, a.batch
torch.manual_seed(1234)
enc = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=seg_len, enc_mode='multilabel_binary', dtype=a.dtype).to(dev)
dec = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=seg_len, dtype=a.dtype).to(dev)
ae = AEStep(enc, dec, lr=1e-4, max_grad_norm=5.0, use_graph=False)
x = (torch.rand(B, seg_len, F) * (1 - 1e-8) + 1e-8).to(dev)
c = torch.randint(0, nspk, (B,)).to(dev)
for _ in range(2):
    ae.step(x, c)
torch.cuda.synchronize()

rec = collections.OrderedDict()
orig = L.call


def hooked(fname, sname, stream, **kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = orig(fname, sname, stream, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if fname == 'zs_gemm_conv':
        g = max(1, kw.get('groups') or 1)
        rows = kw['B'] * kw['T_out'] * g
        key = ('conv', rows, kw['N'], kw['taps'] * kw['cin_pad'], kw['taps'], kw.get('gather') or 0, kw.get('stride') or 1)
        fl = 2.0 * rows * kw['n_pad'] * kw['taps'] * kw['cin_pad']
    else:
        if not a.other:
            return s
        key = (fname, 0, 0, 0, 0, 0, 0)
        fl = 0.0
    r = rec.setdefault(key, [0, 0.0, 0.0])
    r[0] += 1; r[1] += dt; r[2] += fl
    return s


orig_w = layers.wgrad_call


def hooked_w(ctx, kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    orig_w(ctx, kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows = kw['B'] * kw['T_out']
    key = ('wgrad', rows, kw['Cout'], kw['taps'] * kw['Cin'], kw['taps'], 0, kw['stride'])
    r = rec.setdefault(key, [0, 0.0, 0.0])
    r[0] += 1; r[1] += dt; r[2] += 2.0 * rows * kw['Cout'] * kw['taps'] * kw['Cin']


layers.wgrad_call = hooked_w
L.call = hooked
for m in list(sys.modules.values()):
    if m is not None and getattr(m, '__name__', '').startswith('zs_amd') and getattr(m, 'call', None) is orig:
        m.call = hooked
ae.step(x, c)
torch.cuda.synchronize()
L.call = orig
layers.wgrad_call = orig_w
tot = sum(r[1] for r in rec.values())
print('%-28s %7s %6s %6s %4s %3s %3s %5s %9s %8s %6s' % ('kind', 'rows', 'N', 'K', 'taps', 'g', 's', 'calls', 'ms_total', 'TFLOP/s', '%'))
for k, r in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    print('%-28s %7d %6d %6d %4d %3d %3d %5d %9.3f %8.1f %6.1f' % (k + (r[0], r[1] * 1e3, r[2] / r[1] / 1e12 if r[2] else 0.0, 100 * r[1] / tot)))
print('sum of isolated call times: %.2f ms (includes ~10-15 us of host sync per call)' % (tot * 1e3))
