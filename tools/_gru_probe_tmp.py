import os, sys, math, torch
sys.path.insert(0, '/root/repo')
import zs_amd
from zs_amd import _lib as L, layers
B, T, Cin, H = 256, 128, 1024, 512
ctx = layers.Ctx('cuda:0', 'bf16'); dev = ctx.device
g = torch.Generator().manual_seed(2)
P = {}
for sfx in ('', '_reverse'):
    P['RNN.weight_ih_l0' + sfx] = torch.randn(3 * H, Cin, generator=g) / math.sqrt(Cin)
    P['RNN.weight_hh_l0' + sfx] = torch.randn(3 * H, H, generator=g) / math.sqrt(H)
    P['RNN.bias_ih_l0' + sfx] = torch.randn(3 * H, generator=g) * 0.1
    P['RNN.bias_hh_l0' + sfx] = torch.randn(3 * H, generator=g) * 0.1
Pd = {k: v.to(dev).contiguous() for k, v in P.items()}
Gd = {k: torch.zeros_like(v) for k, v in Pd.items()}
gru = layers.GruLayer(ctx, Pd, Gd, 'RNN.', name='t'); gru.pack()
X = ctx.act('x', B, T, Cin); X.t.normal_()
cat = ctx.act('cat', B, T, Cin + 2 * H); gi = ctx.act('gi', B, T, 6 * H)
gates = ctx.raw('gates', B * T * 8 * H, ctx.tdt)
dcat = ctx.act('dcat', B, T, Cin + 2 * H); dcat.t.normal_()
dgi, dgh = ctx.act('dgi', B, T, 6 * H), ctx.act('dgh', B, T, 6 * H); dX = ctx.act('dX', B, T, Cin)
work = gru._work(B)
nb = L.lib().zs_gru_work_bytes(B, H)
def dbg():
    torch.cuda.synchronize()
    w = work[:(nb + 3) // 4].view(torch.uint8)[nb - 256 + 16: nb - 256 + 64].clone().view(torch.int64).cpu().tolist()
    return w
for it in range(3):
    gru.fwd(X, cat, Cin, gi, gates); f = dbg()
    gru.bwd(dcat, Cin, cat, Cin, gates, X, dgi, dgh, dX); b = dbg()
print("fwd ticks per step (cycles): gi-issue %d | sweep+mfma %d | part write %d | barrier1 %d | epilogue %d | barrier2 %d" % tuple(v // T for v in f))

