"""VERDICT r2 item 6, measured: what would fusing the InstanceNorm statistics into the producing GEMM's epilogue buy?
Side A (the norm): zs_instnorm_fwd on the three decoder conv-block shapes of the train_ae step (B = 256, C = 1024, T = 32 / 64 / 128,
bf16, upsampled residual + second output, statistics saved for the backward) as it runs today, against the same call with the
statistics GIVEN (ZsInstNormFwd.stats_given: both reductions over T skipped) -- the best the norm could do after a fusion.
Side B (the GEMM): the 256x256 kernel's epilogue with and without the per-sample column sums it already knows how to emit
(ZsGemmConv.colsum; a sum of squares beside it reads the same staged tile), on the conv that feeds the T = 128 norm.
  python tools/instnorm_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zs_amd  # noqa: E402,F401
from zs_amd import _lib as L, layers  # noqa: E402

dev = torch.device('cuda', 0)
ctx = layers.Ctx(dev, 'bf16')
B, C, nspk = 256, 1024, 102


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


tot = [0.0, 0.0]
for T in (32, 64, 128):
    X, R = ctx.act('x%d' % T, B, T, C), ctx.act('r%d' % T, B, T // 2, C)
    X.valid().copy_(torch.randn(B, T, C, device=dev)); R.valid().copy_(torch.randn(B, T // 2, C, device=dev))
    o1, o2 = ctx.act('o1%d' % T, B, T, C), ctx.act('o2%d' % T, B, T, C)
    vec2 = torch.randn(nspk, C, device=dev)
    idx = torch.randint(0, nspk, (B,), device=dev)
    mean_b, rstd_b = ctx.f32('mean%d' % T, B * C), ctx.f32('rstd%d' % T, B * C)
    kw = dict(dtype=ctx.dt, x=X.ptr(), ldx=X.ld, out=o1.ptr(), ldo=o1.ld, out2=o2.ptr(), ldo2=o2.ld, vec2=L.ptr(vec2), vec2_ld=C, vec2_cols=C,
              idx=L.ptr(idx), mean=L.ptr(mean_b), rstd=L.ptr(rstd_b), B=B, T=T, C=C, eps=1e-5, drop_p=0.0, res_mode=L.ZS_RES_UPSAMPLE2,
              res=R.ptr(), ldres=R.ld, T_res=T // 2, res_pad_mode=L.ZS_PAD_REFLECT)
    a = timeit(lambda: L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, **kw))
    b = timeit(lambda: L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, stats_given=1, **kw))
    mb = B * T * C * 2 * (1 + 0.5 + 2) / 1e6
    print('instnorm_fwd B=%d T=%3d C=%d (%.0f MB moved): computes its statistics %.1f us (%.2f TB/s), statistics given %.1f us (%.2f TB/s)' %
          (B, T, C, mb, a, mb / a, b, mb / b), flush=True)
    tot[0] += a; tot[1] += b
print('three decoder conv blocks per step: %.1f us -> %.1f us with the statistics given (-%.1f us of a 10.9 ms step)' % (tot[0], tot[1], tot[0] - tot[1]))

# the decomposition for short samples (zs_set_option 'norm_wide'): 8 row groups x 256 channels against 32 x 64, T' = 16 layers included
for T in (16, 32, 64):
    X, R = ctx.act('wx%d' % T, B, T, C), ctx.act('wr%d' % T, B, T, C)
    X.valid().copy_(torch.randn(B, T, C, device=dev)); R.valid().copy_(torch.randn(B, T, C, device=dev))
    o1, dz = ctx.act('wo%d' % T, B, T, C), ctx.act('wdz%d' % T, B, T, C)
    mean_b, rstd_b = ctx.f32('wmean%d' % T, B * C), ctx.f32('wrstd%d' % T, B * C)
    fw = dict(dtype=ctx.dt, x=X.ptr(), ldx=X.ld, out=o1.ptr(), ldo=o1.ld, mean=L.ptr(mean_b), rstd=L.ptr(rstd_b), B=B, T=T, C=C, eps=1e-5,
              drop_p=0.0, res_mode=L.ZS_RES_IDENTITY, res=R.ptr(), ldres=R.ld, T_res=T, res_pad_mode=L.ZS_PAD_REFLECT)
    bw = dict(dtype=ctx.dt, dout=R.ptr(), ldd=R.ld, x=X.ptr(), ldx=X.ld, mean=L.ptr(mean_b), rstd=L.ptr(rstd_b), dz=dz.ptr(), ldz=dz.ld,
              B=B, T=T, C=C, drop_p=0.0, slope=0.01)
    res = {}
    for wide in (0, 1):
        L.set_option('norm_wide', wide)
        res[wide] = (timeit(lambda: L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, **fw)),
                     timeit(lambda: L.call('zs_instnorm_bwd', 'ZsInstNormBwd', ctx.stream, **bw)))
    print('B=%d T=%3d C=%d  instnorm_fwd 32x64: %.1f us, 8x256: %.1f us;  instnorm_bwd 32x64: %.1f us, 8x256: %.1f us' %
          (B, T, C, res[0][0], res[1][0], res[0][1], res[1][1]), flush=True)
