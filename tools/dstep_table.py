"""Every GEMM launch of ONE stage-2 discriminator step (BASELINE config 5 shapes) with its shape, HIP-event time and rate:
zs_gemm_conv (forward, data-gradient and adjoint convolutions) and zs_gemm_wgrad launches, timed on the stream they are launched on.
  python tools/dstep_table.py"""
import collections
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zs_amd  # noqa: E402,F401
from zs_amd import _lib as L  # noqa: E402
from zs_amd.hps import make_hps  # noqa: E402
from zs_amd.trainer import Trainer  # noqa: E402

dev = torch.device('cuda:0')
B = 128
hps = make_hps(enc_size=1024, emb_size=1024, batch_size=B)
tr = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=tempfile.mkdtemp(), dtype='bf16', device=dev)
s2 = tr.stage2()
g = torch.Generator().manual_seed(0)
x_s, x_t = torch.rand(B, 128, 513, generator=g).to(dev), torch.rand(B, 128, 513, generator=g).to(dev)
c_t = torch.randint(hps.n_speakers - hps.n_target_speakers, hps.n_speakers, (B,), generator=g).to(dev)
x_gen = s2.gen_forward(x_s, c_t, False).clone()
for _ in range(2):
    s2.d_step(None, x_t, c_t, x_gen=x_gen)
torch.cuda.synchronize()
rec = []
orig = L.call


def call(name, sname, stream, **kw):
    if name not in ('zs_gemm_conv', 'zs_gemm_wgrad'):
        return orig(name, sname, stream, **kw)
    st = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(st)
    r = orig(name, sname, stream, **kw)
    e.record(st)
    if name == 'zs_gemm_conv':
        M, N, K = kw['B'] * kw['T_out'], kw['N'], kw['taps'] * kw.get('cin_pad', 0)
        tag = 'conv gather=%d stride=%d taps=%d' % (kw.get('gather', 0), kw.get('stride', 1), kw['taps'])
    else:
        M, N, K = kw['Cout'], kw['Cin'] * kw['taps'], kw['B'] * kw['T_out']
        tag = 'wgrad taps=%d' % kw['taps']
    rec.append((s, e, tag, M, N, K))
    return r


L.call = call
s2.d_step(None, x_t, c_t, x_gen=x_gen)
torch.cuda.synchronize()
L.call = orig
agg = collections.OrderedDict()
for s, e, tag, M, N, K in rec:
    k = (tag, M, N, K)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += s.elapsed_time(e) * 1e3
tot = sum(a[1] for a in agg.values())
print('%d GEMM launches, %.2f ms of launch-stream time' % (len(rec), tot / 1e3))
print('%-34s %9s %6s %7s %4s %9s %8s %8s' % ('launch', 'M', 'N', 'K', 'n', 'us each', 'TFLOP/s', 'ms total'))
for (tag, M, N, K), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * M * N * K
    print('%-34s %9d %6d %7d %4d %9.1f %8.0f %8.2f' % (tag, M, N, K, n, us / n, fl / (us / n) / 1e6, us / 1e3))
