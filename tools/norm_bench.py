"""Isolated HBM rate of the InstanceNorm forward / backward and grad_combine kernels at the decoder's largest site
([B=256, T=128, C=1024] bf16).   python tools/norm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zs_amd  # noqa: E402,F401
from zs_amd import _lib as L, layers  # noqa: E402

B, T, C = 256, 128, 1024
ctx = layers.Ctx('cuda:0', 'bf16')
x = ctx.act('x', B, T, C); x.t.normal_()
res = ctx.act('res', B, T, C); res.t.normal_()
out = ctx.act('out', B, T, C)
dout = ctx.act('dout', B, T, C); dout.t.normal_()
dz = ctx.act('dz', B, T, C)
gp = ctx.act('gp', B, T + 2, C); gp.t.normal_()
mean = ctx.f32('mean', B * C); rstd = ctx.f32('rstd', B * C)
st = ctx.stream
MB = B * T * C * 2 / 1e6


def fwd():
    L.call('zs_instnorm_fwd', 'ZsInstNormFwd', st, dtype=ctx.dt, x=x.ptr(), ldx=x.ld, out=out.ptr(), ldo=out.ld, mean=L.ptr(mean), rstd=L.ptr(rstd),
           B=B, T=T, C=C, eps=1e-5, drop_p=0.0, seed=1, stream_id=1, res_mode=L.ZS_RES_IDENTITY, res=res.ptr(), ldres=res.ld, T_res=T)


def bwd():
    L.call('zs_instnorm_bwd', 'ZsInstNormBwd', st, dtype=ctx.dt, dout=dout.ptr(), ldd=dout.ld, x=x.ptr(), ldx=x.ld, mean=L.ptr(mean), rstd=L.ptr(rstd),
           dz=dz.ptr(), ldz=dz.ld, B=B, T=T, C=C, drop_p=0.0, slope=0.01)


def comb():
    L.call('zs_grad_combine', 'ZsGradCombine', st, dtype=ctx.dt, gp=gp.ptr(), ldg=gp.ld, pad_left=1, pad_right=1, pad_mode=L.ZS_PAD_REFLECT,
           B=B, T=T, C=C, res_mode=L.ZS_RES_IDENTITY, res=res.ptr(), ldres=res.ld, dact_src=x.ptr(), dact_ld=x.ld, slope=0.01, out=out.ptr(), ldo=out.ld)


def timed(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, f, tensors in (('instnorm_fwd (x + res -> out)', fwd, 3), ('instnorm_bwd (dout + x -> dz)', bwd, 3), ('grad_combine (gp + res + dact -> out)', comb, 4)):
    us = timed(f)
    print('%-40s %6.1f us  %5.2f TB/s (%d tensors of %.0f MB)' % (name, us, tensors * MB / us, tensors, MB))
y = torch.empty(B * T * C, dtype=torch.bfloat16, device='cuda'); z = torch.randn(B * T * C, device='cuda').bfloat16()
us = timed(lambda: torch.add(z, z, out=y))
print('%-40s %6.1f us  %5.2f TB/s' % ('torch add (2 reads + 1 write, reference)', us, 3 * MB / us))
