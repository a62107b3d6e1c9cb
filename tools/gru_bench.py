"""Decoder-size biGRU (B=256, T=128, Cin=1024, H=512, bf16): time of the persistent forward / BPTT launches alone.
  python tools/gru_bench.py [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zs_amd  # noqa: E402,F401
from zs_amd import _lib as L, layers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--B', type=int, default=256); ap.add_argument('--T', type=int, default=128); ap.add_argument('--H', type=int, default=512)
a = ap.parse_args()
ctx = layers.Ctx('cuda:0', 'bf16')
dev = ctx.device
B, T, H, Cin = a.B, a.T, a.H, 2 * a.H
names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
P, G = {}, {}
for sfx in ('', '_reverse'):
    for n, shp in zip(names, [(3 * H, Cin), (3 * H, H), (3 * H,), (3 * H,)]):
        P['RNN.' + n + sfx] = torch.randn(*shp, device=dev) * 0.03
        G['RNN.' + n + sfx] = torch.zeros(*shp, device=dev)
g = layers.GruLayer(ctx, P, G, 'RNN.', name='bench')
with L.pack_batch(ctx.stream):
    g.pack()
X = ctx.act('x', B, T, Cin); X.t.normal_()
out = ctx.act('out', B, T, 3 * Cin)
gi = ctx.act('gi', B, T, 6 * H)
gates = ctx.raw('gates', B * T * 8 * H, ctx.tdt)
dout = ctx.act('dout', B, T, 3 * Cin); dout.t.normal_(); dout.t.mul_(0.01)
dgi, dgh, dX = ctx.act('dgi', B, T, 6 * H), ctx.act('dgh', B, T, 6 * H), ctx.act('dX', B, T, Cin)
os.environ['ZS_OVERLAP_WGRAD'] = '0'


def fwd_only():
    work = g._work(B)
    L.call('zs_gru_fwd', 'ZsGruFwd', ctx.stream, dtype=ctx.dt, B=B, T=T, H=H, gi=gi.ptr(), ldgi=gi.ld, whh=L.ptr(g.whh_f), ldw=g.hh_ldw,
           n_pad=g.hh_npad, w_gstride=g.hh_npad * g.hh_ldw, bhh=L.ptr(g.bhh), bhh_gstride=3 * H, out=out.ptr(), ldo=out.ld, out_col=Cin,
           gates=L.ptr(gates), work=L.ptr(work), work_bytes=work.numel() * 4, whh_interleaved=int(g.fast), status=L.ptr(ctx.status))


def bwd_only():
    work = g._work(B)
    L.call('zs_gru_bwd', 'ZsGruBwd', ctx.stream, dtype=ctx.dt, B=B, T=T, H=H, dout=dout.ptr(), ldd=dout.ld, dout_col=Cin, out=out.ptr(),
           ldo=out.ld, out_col=Cin, gates=L.ptr(gates), whh_t=L.ptr(g.whh_t), ldw=g.hh_ldw_t, n_pad=g.hh_npad_t,
           w_gstride=g.hh_npad_t * g.hh_ldw_t, dgi=dgi.ptr(), ldgi=dgi.ld, dgh=dgh.ptr(), ldgh=dgh.ld, work=L.ptr(work),
           work_bytes=work.numel() * 4, status=L.ptr(ctx.status))


def timed(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters


g.fwd(X, out, Cin, gi, gates)            # fills gi
torch.cuda.synchronize()
for r in range(3):
    tf, tb = timed(fwd_only), timed(bwd_only)
    print('round %d: forward %.1f us (%.2f us/step)   BPTT %.1f us (%.2f us/step)' % (r, tf * 1e3, tf * 1e3 / T, tb * 1e3, tb * 1e3 / T))
layers.check_status(dev)
if os.environ.get('ZS_GRU_PROFILE_DUMP'):
    w = g._work(B)
    hdr = w[:64].view(torch.int64).cpu()
    names = ['sweep', 'barrier', 'mfma', 'gate math', 'hand-off + staging']
    tot = float(sum(int(hdr[8 + k]) for k in range(5)))
    print('BPTT phases of one wave (cycle counter units per step, share):')
    for k, n in enumerate(names):
        print('  %-20s %8.0f  %4.1f %%' % (n, int(hdr[8 + k]) / T, 100.0 * int(hdr[8 + k]) / max(tot, 1.0)))
