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

# the decompositions (workgroup = row groups x channels): 32x64 (Narrow) and 8x256 (T <= 64) for the three kernels that share them.
# grad_combine as the decoder's conv blocks call it: reflect-pad fold of a k = 3 data gradient, un-pixel-shuffle, lrelu'.
def shapes(T):
    out = [('32x64', dict(norm_wide=0))]
    if T <= 64:
        out.append(('8x256', dict(norm_wide=1, norm_lim0=64, norm_lim1=64, norm_lim2=64)))
    return out


for T in (16, 32, 64, 128):
    X, R = ctx.act('wx%d' % T, B, T, C), ctx.act('wr%d' % T, B, T, C)
    X.valid().copy_(torch.randn(B, T, C, device=dev)); R.valid().copy_(torch.randn(B, T, C, device=dev))
    o1, dz = ctx.act('wo%d' % T, B, T, C), ctx.act('wdz%d' % T, B, T, C)
    gp, da, oc = ctx.act('wgp%d' % T, B, T + 2, C), ctx.act('wda%d' % T, B, T // 2, 2 * C), ctx.act('woc%d' % T, B, T // 2, 2 * C)
    gp.valid().copy_(torch.randn(B, T + 2, C, device=dev)); da.valid().copy_(torch.randn(B, T // 2, 2 * C, device=dev))
    mean_b, rstd_b = ctx.f32('wmean%d' % T, B * C), ctx.f32('wrstd%d' % T, B * C)
    fw = dict(dtype=ctx.dt, x=X.ptr(), ldx=X.ld, out=o1.ptr(), ldo=o1.ld, mean=L.ptr(mean_b), rstd=L.ptr(rstd_b), B=B, T=T, C=C, eps=1e-5,
              drop_p=0.0, res_mode=L.ZS_RES_IDENTITY, res=R.ptr(), ldres=R.ld, T_res=T, res_pad_mode=L.ZS_PAD_REFLECT)
    bw = dict(dtype=ctx.dt, dout=R.ptr(), ldd=R.ld, x=X.ptr(), ldx=X.ld, mean=L.ptr(mean_b), rstd=L.ptr(rstd_b), dz=dz.ptr(), ldz=dz.ld,
              B=B, T=T, C=C, drop_p=0.0, slope=0.01)
    gc = dict(dtype=ctx.dt, gp=gp.ptr(), ldg=gp.ld, pad_left=1, pad_right=1, pad_mode=L.ZS_PAD_REFLECT, B=B, T=T, C=C, res_mode=L.ZS_RES_NONE,
              dact_src=da.ptr(), dact_ld=da.ld, slope=0.01, out=oc.ptr(), ldo=oc.ld, unshuffle=1)
    line = 'B=%d T=%3d C=%d ' % (B, T, C)
    for name, opts in shapes(T):
        for k, v in opts.items():
            L.set_option(k, v)
        line += ' | %s: fwd %.1f bwd %.1f combine %.1f us' % (
            name, timeit(lambda: L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, **fw)),
            timeit(lambda: L.call('zs_instnorm_bwd', 'ZsInstNormBwd', ctx.stream, **bw)),
            timeit(lambda: L.call('zs_grad_combine', 'ZsGradCombine', ctx.stream, **gc)))
    print(line, flush=True)
