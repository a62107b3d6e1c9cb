"""GPU parity tests, kernel by kernel, through the C ABI (libzs_amd.so) against plain PyTorch fp32 on
the CPU.  Tolerances: fp32 path 2e-5 relative to the output scale (different summation order only);
bf16 path: reference computed from bf16-rounded operands, 1.5e-2 relative (bf16 output rounding 2^-8)."""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = ['fp32', 'bf16']


@pytest.fixture(scope='module')
def zs():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import zs_amd  # noqa: F401
    from zs_amd import _lib, layers
    _lib.lib()
    return _lib, layers


def _ctx(layers, dtype):
    return layers.Ctx('cuda:0', dtype)


def _tol(dtype):
    return 2e-5 if dtype == 'fp32' else 1.5e-2


def _round(t, dtype):
    return t if dtype == 'fp32' else t.to(torch.bfloat16).float()


def _close(name, got, ref, tol):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), name + ': non-finite output'
    scale = max(1e-6, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    if err > tol * scale:
        idx = np.unravel_index(int((got - ref).abs().argmax()), got.shape)
        raise AssertionError('%s: max err %.4g > %.4g (scale %.3g) at %s got %.6g ref %.6g' %
                             (name, err, tol * scale, scale, idx, got[idx].item(), ref[idx].item()))


def _to_act(layers, ctx, name, x_btc):
    """[B,T,C] fp32 cpu -> Act on device in the compute dtype (zero padded)."""
    B, T, C = x_btc.shape
    a = ctx.act(name, B, T, C)
    v = a.t[:B * T * a.ld].view(B * T, a.ld)
    v.zero_()
    v[:, :C] = x_btc.reshape(B * T, C).to(ctx.device, a.t.dtype)
    return a


def _mk_conv(layers, ctx, w, b, **kw):
    dev = ctx.device
    w_d, b_d = w.to(dev).contiguous(), (b.to(dev).contiguous() if b is not None else None)
    gw, gb = torch.zeros_like(w_d), (torch.zeros_like(b_d) if b is not None else None)
    l = layers.ConvLayer(ctx, w_d, b_d, gw, gb, **kw)
    l.pack()
    return l


def _ref_conv(x_btc, w, b, stride, reflect):
    k = w.shape[2] if w.dim() == 3 else 1
    w3 = w if w.dim() == 3 else w.unsqueeze(2)
    pl, pr = k // 2, k - 1 - k // 2
    xp = F.pad(x_btc.permute(0, 2, 1), (pl, pr), mode='reflect' if reflect else 'constant')
    return F.conv1d(xp, w3, b, stride=stride).permute(0, 2, 1)


GEMM_KEYS = ('gemm_dma', 'gemm_ring', 'gemm_ring_min_tiles', 'gemm_pp', 'gemm_p8', 'gemm_p8_min_tiles', 'wgrad_p8')
GEMM_VARIANTS = {'p8m16': (1, 1, 1, 1, 1, 1, 2), 'pp': (1, 1, 1, 1, 0, 200, 0), 'ring': (1, 1, 1, 0, 0, 200, 0),
                 'dma': (1, 0, 256, 0, 0, 200, 0), 'reg': (0, 0, 256, 0, 0, 200, 0)}


@pytest.fixture(params=list(GEMM_VARIANTS))
def gemm_variant(request, zs):
    """Run a test with each conv-GEMM kernel: 256x256 quadrant ping-pong (where the packed weight has a multiple of 256 rows,
    else it falls through to the ring), 256x128 3-stage ring with the ping-pong schedule, the same ring in lock-step,
    128x128 LDS-DMA, 128x128 register-staged.  The 'p8m16' variant also forces the 256x256 ping-pong weight-gradient kernel
    (bf16; wgrad_p8=2 overrides its size heuristics), the others use the 128x128 one."""
    L, _ = zs
    old = [L.set_option(k, v) for k, v in zip(GEMM_KEYS, GEMM_VARIANTS[request.param])]
    yield request.param
    for k, v in zip(GEMM_KEYS, old):
        L.set_option(k, v)


CONV_CASES = [
    # B, T, Cin, Cout, k, stride, reflect
    (2, 24, 80, 16, 1, 1, True), (2, 24, 80, 16, 2, 1, True), (2, 24, 80, 16, 3, 1, True), (2, 24, 80, 16, 4, 1, True),
    (2, 24, 80, 16, 5, 1, True), (2, 24, 80, 16, 6, 1, True), (2, 24, 80, 16, 7, 1, True),
    (3, 50, 513, 130, 7, 1, True), (2, 9, 32, 32, 5, 2, True), (2, 33, 64, 48, 5, 2, True), (5, 64, 96, 513, 1, 1, True),
    (2, 20, 32, 40, 3, 1, False), (2, 21, 32, 40, 5, 2, False), (1, 300, 160, 256, 3, 1, True), (3, 200, 64, 200, 3, 1, True),
    (4, 130, 192, 512, 3, 1, True), (3, 90, 136, 250, 5, 2, False),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd(zs, dtype, case, gemm_variant):
    L, layers = zs
    B, T, Cin, Cout, k, stride, reflect = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, T, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g)
    ctx = _ctx(layers, dtype)
    l = _mk_conv(layers, ctx, w, b, stride=stride, pad_mode=L.ZS_PAD_REFLECT if reflect else L.ZS_PAD_ZERO)
    A = _to_act(layers, ctx, 'x', x)
    To = l.t_out(T)
    out = ctx.act('y', B, To, Cout)
    out.t.fill_(float('nan'))
    l.fwd(A, out=out, act=L.ZS_ACT_LRELU, slope=0.01)
    torch.cuda.synchronize()
    ref = F.leaky_relu(_ref_conv(_round(x, dtype), _round(w, dtype), b, stride, reflect), 0.01)
    assert ref.shape[1] == To
    _close('conv', out.valid(), ref, _tol(dtype))
    full = out.t[:B * To * out.ld].view(B * To, out.ld).float().cpu()
    assert (full[:, Cout:] == 0).all(), 'pad columns must be written as zeros'


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_epilogue_split2_vec(zs, dtype, gemm_variant):
    """pixel_shuffle + speaker-embedding adds + dual output (decoder conv_block first conv)."""
    L, layers = zs
    B, T, C = 3, 10, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, C, generator=g)
    w = torch.randn(2 * C, C, 3, generator=g) / math.sqrt(3 * C)
    b = torch.randn(2 * C, generator=g)
    emb = torch.randn(4, C, generator=g)
    cidx = torch.tensor([2, 0, 3])
    ctx = _ctx(layers, dtype)
    l = _mk_conv(layers, ctx, w, b, split2=True)
    A = _to_act(layers, ctx, 'x', x)
    ya = ctx.act('ya', B, T, 2 * C)
    s = ctx.act('s', B, 2 * T, C)
    l.fwd(A, out=ya, act=L.ZS_ACT_LRELU, slope=0.01, out2=s, vec2=emb.to(ctx.device), idx=cidx.to(ctx.device),
          store_mode2=L.ZS_STORE_SPLIT2)
    torch.cuda.synchronize()
    y = F.leaky_relu(_ref_conv(_round(x, dtype), _round(w, dtype), b, 1, True), 0.01)       # [B,T,2C] reference channel order
    y_bct = y.permute(0, 2, 1)
    shuf = y_bct.contiguous().view(B, C, 2, T).permute(0, 1, 3, 2).contiguous().view(B, C, 2 * T)   # model/model.py:43-51
    ref_s = (shuf + emb[cidx].unsqueeze(2)).permute(0, 2, 1)
    _close('shuffle+emb', s.valid(), ref_s, _tol(dtype))
    perm = torch.cat([torch.arange(0, 2 * C, 2), torch.arange(1, 2 * C, 2)])
    _close('ya (packed order)', ya.valid(), y[:, :, perm], _tol(dtype))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(2, 16, 32, 48, 5, 1), (2, 16, 32, 48, 5, 2), (3, 12, 64, 32, 3, 1), (2, 10, 40, 24, 1, 1),
                                  (2, 128, 96, 80, 5, 2), (3, 64, 64, 256, 5, 2), (4, 200, 300, 520, 3, 1), (3, 130, 136, 250, 1, 1)])
def test_conv_backward(zs, dtype, case, gemm_variant):
    """dgrad (padded domain) + reflect fold + lrelu' ; wgrad ; bias grad -- against autograd."""
    L, layers = zs
    B, T, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(11)
    x = _round(torch.randn(B, T, Cin, generator=g), dtype).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k))
    wr = _round(w, dtype).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    y = _ref_conv(x, wr, b, stride, True)
    dy = _round(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    ctx = _ctx(layers, dtype)
    l = _mk_conv(layers, ctx, w, b.detach(), stride=stride)
    X = _to_act(layers, ctx, 'x', x.detach())
    dY = _to_act(layers, ctx, 'dy', dy)
    gp = ctx.act('gp', B, T + l.pad_l + l.pad_r, Cin)
    l.dgrad(dY, T, gp)
    dx = ctx.act('dx', B, T, Cin)
    L.call('zs_grad_combine', 'ZsGradCombine', ctx.stream, dtype=ctx.dt, gp=gp.ptr(), ldg=gp.ld, pad_left=l.pad_l,
           pad_right=l.pad_r, pad_mode=L.ZS_PAD_REFLECT, B=B, T=T, C=gp.ld, res_mode=L.ZS_RES_NONE, out=dx.ptr(), ldo=dx.ld)
    l.wgrad(dY, X)
    torch.cuda.synchronize()
    _close('dx', dx.valid(), x.grad, _tol(dtype))
    _close('dW', l.gw, wr.grad, _tol(dtype))
    _close('db', l.gb, b.grad, _tol(dtype))


@pytest.mark.parametrize('dtype', DTYPES)
def test_wgrad_split2_and_shift(zs, dtype, gemm_variant):
    L, layers = zs
    B, T, Cin, C = 2, 12, 32, 16
    g = torch.Generator().manual_seed(5)
    x = _round(torch.randn(B, T, Cin, generator=g), dtype)
    w = torch.randn(2 * C, Cin, 3, generator=g)
    dy_ref = _round(torch.randn(B, T, 2 * C, generator=g), dtype)      # reference channel order
    perm = torch.cat([torch.arange(0, 2 * C, 2), torch.arange(1, 2 * C, 2)])
    ctx = _ctx(layers, dtype)
    l = _mk_conv(layers, ctx, w, torch.zeros(2 * C), split2=True)
    l.wgrad(_to_act(layers, ctx, 'dy', dy_ref[:, :, perm]), _to_act(layers, ctx, 'x', x))
    wr = w.clone().requires_grad_(True)
    b = torch.zeros(2 * C, requires_grad=True)
    _ref_conv(x, wr, b, 1, True).backward(dy_ref)
    torch.cuda.synchronize()
    _close('dW split2', l.gw, wr.grad, _tol(dtype))
    _close('db split2', l.gb, b.grad, _tol(dtype))
    # shifted zero-padded gather (GRU dW_hh: h_{t-1} / h_{t+1})
    H = 16
    hbuf = _round(torch.randn(B, T, H, generator=g), dtype)
    dgh = _round(torch.randn(B, T, 3 * H, generator=g), dtype)
    Hh, D = _to_act(layers, ctx, 'h', hbuf), _to_act(layers, ctx, 'dgh', dgh)
    for shift, pad_left in ((-1, 1), (1, -1)):
        dW = torch.zeros(3 * H, H, device=ctx.device)
        layers.wgrad_call(ctx, dict(dtype=ctx.dt, dY=D.ptr(), ldy=D.ld, y_cols=D.ld, X=Hh.ptr(), ldx=Hh.ld, x_batch_stride=T * Hh.ld,
                                    x_cols=Hh.ld, B=B, T_in=T, T_out=T, taps=1, stride=1, pad_left=pad_left, pad_mode=L.ZS_PAD_ZERO,
                                    Cout=3 * H, Cin=H, dW=L.ptr(dW), so=H, si=1, sj=0, co_split2=0, accumulate=0, splits=0))
        hs = torch.zeros_like(hbuf)
        if shift == -1:
            hs[:, 1:] = hbuf[:, :-1]
        else:
            hs[:, :-1] = hbuf[:, 1:]
        ref = torch.einsum('btn,bth->nh', dgh, hs)
        torch.cuda.synchronize()
        _close('dW_hh shift %d' % shift, dW, ref, _tol(dtype))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(3, 16, 96, 128), (2, 64, 160, 256), (5, 128, 64, 192)])
def test_conv_epilogue_wide_split2_vec_prevec(zs, dtype, shape, gemm_variant):
    """The decoder's first conv of a block at widths that reach the 256x256 kernel's register epilogue (n_pad a multiple of
    256): per-sample pre_vec, lrelu, packed output, pixel-shuffled second output + speaker embedding."""
    L, layers = zs
    B, T, Cin, C = shape
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, T, Cin, generator=g)
    w = torch.randn(2 * C, Cin, 3, generator=g) / math.sqrt(3 * Cin)
    b = torch.randn(2 * C, generator=g)
    emb = torch.randn(6, C, generator=g)
    pre = torch.randn(6, 2 * C, generator=g)                     # per-sample bias in the PACKED channel order
    cidx = torch.randint(0, 6, (B,), generator=g)
    ctx = _ctx(layers, dtype)
    l = _mk_conv(layers, ctx, w, b, split2=True)
    A = _to_act(layers, ctx, 'x', x)
    ya = ctx.act('ya', B, T, 2 * C)
    s = ctx.act('s', B, 2 * T, C)
    ya.t.fill_(float('nan')); s.t.fill_(float('nan'))
    l.fwd(A, out=ya, act=L.ZS_ACT_LRELU, slope=0.01, out2=s, vec2=emb.to(ctx.device), idx=cidx.to(ctx.device),
          store_mode2=L.ZS_STORE_SPLIT2, pre_vec=pre.to(ctx.device))
    torch.cuda.synchronize()
    perm = torch.cat([torch.arange(0, 2 * C, 2), torch.arange(1, 2 * C, 2)])
    inv = torch.empty_like(perm); inv[perm] = torch.arange(2 * C)
    pre_ref = pre[:, inv]                                        # reference channel order
    y = F.leaky_relu(_ref_conv(_round(x, dtype), _round(w, dtype), b, 1, True) + pre_ref[cidx].unsqueeze(1), 0.01)
    shuf = y.permute(0, 2, 1).contiguous().view(B, C, 2, T).permute(0, 1, 3, 2).contiguous().view(B, C, 2 * T)
    ref_s = (shuf + emb[cidx].unsqueeze(2)).permute(0, 2, 1)
    _close('ya (packed order)', ya.valid(), y[:, :, perm], _tol(dtype))
    _close('shuffle+emb', s.valid(), ref_s, _tol(dtype))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(5, 64, 136, 256), (3, 128, 96, 384), (9, 32, 64, 250), (4, 16, 48, 256)])
def test_dgrad_epilogue_mask_add_colsum(zs, dtype, shape, gemm_variant):
    """k = 1 data gradient with everything the decoder's backward hangs on it: lrelu' mask, residual add, per-sample column
    sums of the raw gradient (nn.Embedding backward) from a column offset, stores limited to out_cols."""
    L, layers = zs
    B, T, Cout, Cin = shape
    g = torch.Generator().manual_seed(41)
    w = torch.randn(Cout, Cin, generator=g) / math.sqrt(Cout)
    dy = _round(torch.randn(B, T, Cout, generator=g), dtype)
    ysrc = _round(torch.randn(B, T, Cin, generator=g), dtype)    # forward activation whose sign gives the lrelu' mask
    add = _round(torch.randn(B, T, Cin, generator=g), dtype)
    ctx = _ctx(layers, dtype)
    l = _mk_conv(layers, ctx, w, None)
    dY = _to_act(layers, ctx, 'dy', dy)
    Ys, Ad = _to_act(layers, ctx, 'ys', ysrc), _to_act(layers, ctx, 'ad', add)
    raw = torch.einsum('btn,nc->btc', dy, _round(w, dtype))
    mask = torch.where(ysrc > 0, torch.ones(()), torch.full((), 0.01))
    col0 = (Cin // 2) // 8 * 8
    for mode in ('mask', 'add', 'plain'):
        out = ctx.act('o_' + mode, B, T, Cin)
        out.t.fill_(float('nan'))
        cs = torch.full((B, Cin), 0.5, device=ctx.device)
        kw = dict(colsum=(cs.data_ptr(), Cin, col0 if mode == 'plain' else 0))
        if mode == 'mask':
            l.dgrad(dY, T, out, dact_src=Ys, slope=0.01, **kw)
            ref = raw * mask
        elif mode == 'add':
            l.dgrad(dY, T, out, add_src=Ad, **kw)
            ref = raw + add
        else:
            l.dgrad(dY, T, out, out_cols=col0, **kw)
            ref = raw
        torch.cuda.synchronize()
        got = out.valid().float().cpu()
        c0 = col0 if mode == 'plain' else 0
        if mode == 'plain':
            _close('dgrad ' + mode, got[:, :, :col0], ref[:, :, :col0], _tol(dtype))
            assert torch.isnan(got[:, :, col0:]).all(), 'columns past out_cols must not be stored'
        else:
            _close('dgrad ' + mode, got, ref, _tol(dtype))
        want = 0.5 + raw.sum(1)[:, c0:]
        _close('colsum ' + mode, cs[:, :Cin - c0], want, _tol(dtype))
        assert (cs[:, Cin - c0:] == 0.5).all()


# ---- ragged batches (per-sample lengths): every sample must come out exactly as if it had been launched alone ---------------

RAGGED_LENS = [40, 9, 17, 33, 24, 40, 12]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(80, 48, 3, 1), (64, 256, 5, 1), (64, 48, 5, 2), (32, 40, 4, 1), (513, 128, 7, 1), (96, 256, 2, 1)])
def test_conv_ragged_batch_equals_each_sample_alone(zs, dtype, case, gemm_variant):
    """ZsGemmConv.lengths: reflect padding at each sample's own end (the reference forwards the 128..254-frame tail fragments one
    by one, convert.py:154-165).  The valid rows of the ragged launch equal the launch of that sample alone BIT FOR BIT (same
    kernel variant, same K order), and the fp32 torch reference to the usual tolerance."""
    L, layers = zs
    Cin, Cout, k, stride = case
    lens = RAGGED_LENS
    B, Tm = len(lens), max(lens)
    g = torch.Generator().manual_seed(11 + k)
    x = torch.randn(B, Tm, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g)
    ctx = _ctx(layers, dtype)
    l = _mk_conv(layers, ctx, w, b, stride=stride, pad_mode=L.ZS_PAD_REFLECT)
    A = _to_act(layers, ctx, 'rx', x)
    To = l.t_out(Tm)
    out = ctx.act('ry', B, To, Cout)
    out.t.fill_(float('nan'))
    lens_d = torch.tensor(lens, dtype=torch.int32, device=ctx.device)
    l.fwd(A, out=out, act=L.ZS_ACT_LRELU, slope=0.01, lengths=lens_d)
    torch.cuda.synchronize()
    got = out.valid().float().cpu()
    assert torch.isfinite(got).all(), 'rows past a length must be finite (computed from zero rows)'
    for i, n in enumerate(lens):
        Ai = _to_act(layers, ctx, 'rx1_%d' % n, x[i:i + 1, :n])
        to = l.t_out(n)
        oi = ctx.act('ry1_%d' % n, 1, to, Cout)
        l.fwd(Ai, out=oi, act=L.ZS_ACT_LRELU, slope=0.01)
        torch.cuda.synchronize()
        alone = oi.valid().float().cpu()[0]
        assert torch.equal(got[i, :to], alone), 'sample %d (length %d): ragged launch differs from the sample alone' % (i, n)
        ref = F.leaky_relu(_ref_conv(_round(x[i:i + 1, :n], dtype), _round(w, dtype), b, stride, True), 0.01)[0]
        _close('ragged conv vs torch', got[i, :to], ref, _tol(dtype))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('res', ['none', 'identity', 'avgpool', 'upsample'])
def test_instnorm_ragged_batch_equals_each_sample_alone(zs, dtype, res):
    """ZsInstNormFwd.lengths / res_lengths: statistics over each sample's own rows; the avg-pool residual's odd-length pad at
    the sample's own end.  Bit-identical to the sample alone; rows past a length are zeros."""
    L, layers = zs
    lens = [20, 9, 13, 20, 2, 7]
    B, Tm, C = len(lens), max(lens), 40
    g = torch.Generator().manual_seed(5)
    x = _round(torch.randn(B, Tm, C, generator=g), dtype)
    rlens = {'none': lens, 'identity': lens, 'avgpool': [2 * n - (i % 2) for i, n in enumerate(lens)], 'upsample': None}[res]
    if res == 'upsample':
        lens = [2 * (n // 2 + 1) for n in lens]                        # even lengths; residual rows = length / 2
        Tm = max(lens)
        x = _round(torch.randn(B, Tm, C, generator=g), dtype)
        rlens = [n // 2 for n in lens]
    Trm = {'none': Tm, 'identity': Tm, 'avgpool': 2 * Tm, 'upsample': Tm // 2}[res]
    r = _round(torch.randn(B, Trm, C, generator=g), dtype)
    mode = {'none': L.ZS_RES_NONE, 'identity': L.ZS_RES_IDENTITY, 'avgpool': L.ZS_RES_AVGPOOL2, 'upsample': L.ZS_RES_UPSAMPLE2}[res]
    ctx = _ctx(layers, dtype)
    X, R = _to_act(layers, ctx, 'nx', x), _to_act(layers, ctx, 'nr', r)
    out = ctx.act('no', B, Tm, C)
    out.t.fill_(float('nan'))
    ld_, rl_ = torch.tensor(lens, dtype=torch.int32, device=ctx.device), torch.tensor(rlens, dtype=torch.int32, device=ctx.device)
    L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, dtype=ctx.dt, x=X.ptr(), ldx=X.ld, out=out.ptr(), ldo=out.ld, B=B, T=Tm, C=X.ld,
           eps=1e-5, drop_p=0.0, res_mode=mode, res=R.ptr(), ldres=R.ld, T_res=Trm, res_pad_mode=L.ZS_PAD_REFLECT,
           lengths=L.ptr(ld_), res_lengths=(L.ptr(rl_) if res == 'avgpool' else None))
    torch.cuda.synchronize()
    got = out.valid().float().cpu()
    for i, n in enumerate(lens):
        Xi = _to_act(layers, ctx, 'nx1_%d' % n, x[i:i + 1, :n])
        Ri = _to_act(layers, ctx, 'nr1_%d' % rlens[i], r[i:i + 1, :rlens[i]])
        oi = ctx.act('no1_%d' % n, 1, n, C)
        L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, dtype=ctx.dt, x=Xi.ptr(), ldx=Xi.ld, out=oi.ptr(), ldo=oi.ld, B=1, T=n, C=Xi.ld,
               eps=1e-5, drop_p=0.0, res_mode=mode, res=Ri.ptr(), ldres=Ri.ld, T_res=rlens[i], res_pad_mode=L.ZS_PAD_REFLECT)
        torch.cuda.synchronize()
        assert torch.equal(got[i, :n], oi.valid().float().cpu()[0]), (res, i, n)
        assert (got[i, n:] == 0).all()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(24, 16), (64, 128), (96, 512)])
def test_gru_ragged_batch_equals_each_sample_alone(zs, dtype, shape):
    """GruLayer.fwd(lengths=...): zs_rows_reverse on the reverse direction's gate inputs, both directions as forward recurrences
    (ZsGruFwd.dir1_forward), outputs reversed back -- the valid rows of every sample equal the bidirectional GRU run on that
    sample alone with T = its length (per-step, narrow and wide persistent kernels, whichever the shape dispatches to)."""
    L, layers = zs
    Cin, H = shape
    if dtype == 'fp32' and H == 512:
        pytest.skip('covered in bf16 (the fp32 H = 512 recurrence runs one launch per step: slow)')
    lens = [12, 3, 7, 12, 2, 9, 5, 11, 4]
    B, Tm = len(lens), max(lens)
    g = torch.Generator().manual_seed(3)
    gru = torch.nn.GRU(Cin, H, bidirectional=True)
    P = {'RNN.' + k: v.detach().clone().to('cuda:0') for k, v in gru.state_dict().items()}
    G = {k: torch.zeros_like(v) for k, v in P.items()}
    ctx = _ctx(layers, dtype)
    layer = layers.GruLayer(ctx, P, G, 'RNN.', name='rag')
    layer.pack()
    x = torch.randn(B, Tm, Cin, generator=g)
    X = _to_act(layers, ctx, 'gx', x)
    out = ctx.act('gout', B, Tm, 2 * H)
    out.t.zero_()
    gi = ctx.act('ggi', B, Tm, 6 * H)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=ctx.device)
    layer.fwd(X, out, 0, gi, None, lengths=lens_d)
    layer.check(B)
    got = out.valid().float().cpu()
    for i, n in enumerate(lens):
        Xi = _to_act(layers, ctx, 'gx1_%d' % n, x[i:i + 1, :n])
        oi = ctx.act('gout1_%d' % n, 1, n, 2 * H)
        gii = ctx.act('ggi1_%d' % n, 1, n, 6 * H)
        layer.fwd(Xi, oi, 0, gii, None)
        torch.cuda.synchronize()
        alone = oi.valid().float().cpu()[0]
        # the ragged launch and the single-sample launch may take different kernels (batch 9 vs 1): compare to rounding
        _close('ragged GRU sample %d (length %d)' % (i, n), got[i, :n], alone, 2e-5 if dtype == 'fp32' else 2e-2)
        with torch.no_grad():
            ref = gru(_round(x[i, :n], dtype).unsqueeze(1))[0][:, 0]
        _close('ragged GRU vs nn.GRU sample %d' % i, got[i, :n], ref, 1e-4 if dtype == 'fp32' else 4e-2)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('res', ['none', 'identity', 'avgpool', 'avgpool_odd', 'upsample'])
def test_instnorm(zs, dtype, res):
    L, layers = zs
    B, T, C, p = 3, 20, 40, 0.5
    g = torch.Generator().manual_seed(7)
    x = _round(F.leaky_relu(torch.randn(B, T, C, generator=g), 0.01), dtype).requires_grad_(True)
    mask = (torch.rand(B, T, C, generator=g) >= p)
    Tr = {'none': T, 'identity': T, 'avgpool': 2 * T, 'avgpool_odd': 2 * T - 1, 'upsample': T // 2}[res]
    r = _round(torch.randn(B, Tr, C, generator=g), dtype)
    xb = x.permute(0, 2, 1)
    mean = xb.mean(2, keepdim=True)
    xh = (xb - mean) / torch.sqrt(((xb - mean) ** 2).mean(2, keepdim=True) + 1e-5)
    y = xh * mask.permute(0, 2, 1) / (1 - p)
    rb = r.permute(0, 2, 1)
    if res == 'identity':
        y = y + rb
    elif res.startswith('avgpool'):
        y = y + F.avg_pool1d(F.pad(rb, (0, Tr % 2), mode='reflect'), 2)
    elif res == 'upsample':
        y = y + rb.repeat_interleave(2, dim=2)
    y = y.permute(0, 2, 1)
    dout = _round(torch.randn(B, T, C, generator=g), dtype)
    y.backward(dout)
    # dz = dx * lrelu'(x): autograd through x = lrelu(z) is emulated by multiplying with lrelu'(x)
    ref_dz = x.grad * torch.where(x.detach() > 0, torch.ones(()), torch.tensor(0.01))
    ctx = _ctx(layers, dtype)
    X, R, D = _to_act(layers, ctx, 'x', x.detach()), _to_act(layers, ctx, 'r', r), _to_act(layers, ctx, 'd', dout)
    out, dz = ctx.act('o', B, T, C), ctx.act('dz', B, T, C)
    Cp = X.ld
    mean_b, rstd_b = ctx.f32('mean', B * Cp), ctx.f32('rstd', B * Cp)
    m8 = mask.to(torch.uint8).contiguous().to(ctx.device)
    mode = {'none': L.ZS_RES_NONE, 'identity': L.ZS_RES_IDENTITY, 'avgpool': L.ZS_RES_AVGPOOL2, 'avgpool_odd': L.ZS_RES_AVGPOOL2,
            'upsample': L.ZS_RES_UPSAMPLE2}[res]
    L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, dtype=ctx.dt, x=X.ptr(), ldx=X.ld, out=out.ptr(), ldo=out.ld,
           mean=L.ptr(mean_b), rstd=L.ptr(rstd_b), B=B, T=T, C=Cp, eps=1e-5, drop_p=p, mask=L.ptr(m8), mask_ld=C,
           res_mode=mode, res=R.ptr(), ldres=R.ld, T_res=Tr, res_pad_mode=L.ZS_PAD_REFLECT)
    L.call('zs_instnorm_bwd', 'ZsInstNormBwd', ctx.stream, dtype=ctx.dt, dout=D.ptr(), ldd=D.ld, x=X.ptr(), ldx=X.ld,
           mean=L.ptr(mean_b), rstd=L.ptr(rstd_b), dz=dz.ptr(), ldz=dz.ld, B=B, T=T, C=Cp, drop_p=p, mask=L.ptr(m8), mask_ld=C,
           slope=0.01)
    torch.cuda.synchronize()
    _close('instnorm fwd ' + res, out.valid(), y, _tol(dtype))
    _close('instnorm bwd ' + res, dz.valid(), ref_dz, max(_tol(dtype), 1e-4))


@pytest.mark.parametrize('dtype', DTYPES)
def test_instnorm_rng_dropout_consistency(zs, dtype):
    """Counter-hash dropout: fwd and bwd regenerate the same mask; keep rate ~ 1-p."""
    L, layers = zs
    B, T, C, p = 4, 64, 64, 0.5
    ctx = _ctx(layers, dtype)
    x = torch.randn(B, T, C)
    X = _to_act(layers, ctx, 'x', x)
    out, dz = ctx.act('o', B, T, C), ctx.act('dz', B, T, C)
    ones = _to_act(layers, ctx, 'ones', torch.ones(B, T, C))
    mean_b, rstd_b = ctx.f32('mean', B * C), ctx.f32('rstd', B * C)
    kw = dict(dtype=ctx.dt, B=B, T=T, C=C, drop_p=p, seed=1234, stream_id=3)
    L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, x=X.ptr(), ldx=X.ld, out=out.ptr(), ldo=out.ld, mean=L.ptr(mean_b),
           rstd=L.ptr(rstd_b), eps=1e-5, res_mode=L.ZS_RES_NONE, **kw)
    torch.cuda.synchronize()
    o = out.valid().float().cpu()
    keep = (o != 0)
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02
    xh = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    _close('kept values', o[keep], (xh / (1 - p))[keep], _tol(dtype))


def test_mbv_bit_exact_and_grad(zs):
    L, layers = zs
    import zs_oracle as O
    torch.manual_seed(0)
    rows, E = 37, 24
    logits = torch.randn(rows, 2 * E) * 3
    U = torch.rand(rows, E, 2)
    G = O.gumbel_from_uniform(U)
    lg = logits.view(rows, E, 2).clone().requires_grad_(True)
    out, y = O.gumbel_softmax_hard(lg, G)
    dbits = torch.randn(rows, E)
    out[..., 0].backward(dbits)
    ctx = _ctx(layers, 'fp32')
    dev = ctx.device
    bits_f = torch.zeros(rows, E, device=dev)
    y0 = torch.zeros(rows, E, device=dev)
    lgd = logits.to(dev)
    for kind, noise in ((0, G), (1, U)):
        noise_d = noise.to(dev).contiguous()           # keep alive: the call only takes raw pointers
        L.call('zs_mbv_fwd', 'ZsMbvFwd', ctx.stream, dtype=ctx.dt, logits=L.ptr(lgd), ld=2 * E, logits_f32=1, noise=L.ptr(noise_d),
               noise_kind=kind, rows=rows, E=E, tau=0.1, bits_f32=L.ptr(bits_f), y0=L.ptr(y0))
        torch.cuda.synchronize()
        flips = (bits_f.cpu() != out[..., 0].detach())
        if kind == 0:
            assert not flips.any(), 'MBV bits must be bit-exact given (logits, G)'
            assert torch.equal(bits_f.cpu(), out[..., 0].detach())          # values are exactly 0.0 / 1.0
        else:
            s = (logits.view(rows, E, 2) + G)
            assert (not flips.any()) or (s[..., 0] - s[..., 1]).abs()[flips].max() < 1e-4
    _close('y0', y0, y[..., 0].detach(), 1e-5)
    db = torch.zeros(rows, 32, device=dev)
    db[:, :E] = dbits.to(dev)
    dl = torch.zeros(rows, 64, device=dev)
    L.call('zs_mbv_bwd', 'ZsMbvBwd', ctx.stream, dtype=ctx.dt, dbits=L.ptr(db), ld_dbits=32, y0=L.ptr(y0), rows=rows, E=E, tau=0.1,
           dlogits=L.ptr(dl), ld=64, fill_cols=64)
    torch.cuda.synchronize()
    _close('dlogits', dl[:, :2 * E], lg.grad.reshape(rows, 2 * E), 1e-4)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('persist', ['wide', 'narrow', 0])
@pytest.mark.parametrize('shape', [(3, 5, 24, 16), (2, 16, 32, 32), (70, 6, 64, 64), (5, 9, 40, 96), (40, 12, 48, 128), (33, 7, 32, 256),
                                   (64, 10, 64, 512)])
def test_gru(zs, dtype, shape, persist):
    """Bidirectional GRU forward + BPTT + all parameter gradients vs. the oracle GRU under autograd; with the persistent
    time-loop kernels (where H and the grid size allow them) and with one launch per step."""
    L, layers = zs
    old_persist = L.set_option('gru_persist', 1 if persist else 0)
    old_wide = L.set_option('gru_wide', 1 if persist == 'wide' else 0)      # bf16, H in {128, 256, 512}: 16 rows x 64 units tiling
    try:
        _gru_case(L, layers, dtype, shape)
    finally:
        L.set_option('gru_persist', old_persist)
        L.set_option('gru_wide', old_wide)


def _gru_case(L, layers, dtype, shape):
    import zs_oracle as O
    B, T, Cin, H = shape
    g = torch.Generator().manual_seed(2)
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    P = {}
    for sfx in ('', '_reverse'):
        P['RNN.weight_ih_l0' + sfx] = torch.randn(3 * H, Cin, generator=g) / math.sqrt(Cin)
        P['RNN.weight_hh_l0' + sfx] = torch.randn(3 * H, H, generator=g) / math.sqrt(H)
        P['RNN.bias_ih_l0' + sfx] = torch.randn(3 * H, generator=g) * 0.1
        P['RNN.bias_hh_l0' + sfx] = torch.randn(3 * H, generator=g) * 0.1
    x = _round(torch.randn(B, T, Cin, generator=g), dtype)
    Pr = {k: (_round(v, dtype) if 'weight' in k else v).clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    out = O.bigru(xr.permute(0, 2, 1), Pr, 'RNN.').permute(0, 2, 1)          # [B,T,2H]
    dout = _round(torch.randn(B, T, 2 * H, generator=g), dtype)
    out.backward(dout)
    ctx = _ctx(layers, dtype)
    dev = ctx.device
    Pd = {k: v.to(dev).contiguous() for k, v in P.items()}
    Gd = {k: torch.zeros_like(v) for k, v in Pd.items()}
    gru = layers.GruLayer(ctx, Pd, Gd, 'RNN.', name='t')
    gru.pack()
    X = _to_act(layers, ctx, 'x', x)
    cat = ctx.act('cat', B, T, Cin + 2 * H)
    gi = ctx.act('gi', B, T, 6 * H)
    gates = ctx.raw('gates', B * T * 8 * H, ctx.tdt)
    gru.fwd(X, cat, Cin, gi, gates)
    gru.check(B)
    torch.cuda.synchronize()
    got = cat.valid()[:, :, Cin:]
    _close('gru fwd', got, out, 4 * _tol(dtype))
    if H % 4 == 0:
        # append_emb riding on the recurrence (ZsGruFwd.bcast_*): a third block of the output rows = vec[idx[b]] at every t
        vec = torch.randn(5, 2 * H, generator=g)
        idx = torch.randint(0, 5, (B,), generator=g)
        cat3 = ctx.act('cat3', B, T, Cin + 4 * H)
        gru.fwd(X, cat3, Cin, gi, None, bcast=(vec.to(dev), idx.to(dev), Cin + 2 * H))
        torch.cuda.synchronize()
        v3 = cat3.valid()
        _close('gru fwd (with broadcast)', v3[:, :, Cin:Cin + 2 * H], out, 4 * _tol(dtype))
        _close('broadcast block', v3[:, :, Cin + 2 * H:], _round(vec, dtype)[idx].unsqueeze(1).expand(B, T, 2 * H), 1e-6)
    dcat = _to_act(layers, ctx, 'dcat', torch.cat([torch.zeros(B, T, Cin), dout], dim=2))
    dgi, dgh = ctx.act('dgi', B, T, 6 * H), ctx.act('dgh', B, T, 6 * H)
    dX = ctx.act('dX', B, T, Cin)
    gru.bwd(dcat, Cin, cat, Cin, gates, X, dgi, dgh, dX)
    gru.check(B)
    torch.cuda.synchronize()
    tol = 6 * _tol(dtype)
    _close('gru dX', dX.valid(), xr.grad, tol)
    for k in P:
        _close('gru grad ' + k, Gd[k], Pr[k].grad, tol)


def test_gru_full_size_persistent_equals_stepwise_and_is_deterministic(zs):
    """Decoder GRU at the bench size (B=256, T=128, H=512, bf16): the persistent kernels (granule hand-off between 256
    workgroups) against the one-launch-per-step path, and two persistent runs against each other bit for bit -- a race in the
    hand-off would show up here as a difference."""
    L, layers = zs
    B, T, Cin, H = 256, 128, 1024, 512
    ctx = _ctx(layers, 'bf16')
    dev = ctx.device
    g = torch.Generator().manual_seed(21)
    P = {}
    for sfx in ('', '_reverse'):
        P['RNN.weight_ih_l0' + sfx] = torch.randn(3 * H, Cin, generator=g) / math.sqrt(Cin)
        P['RNN.weight_hh_l0' + sfx] = torch.randn(3 * H, H, generator=g) / math.sqrt(H)
        P['RNN.bias_ih_l0' + sfx] = torch.randn(3 * H, generator=g) * 0.1
        P['RNN.bias_hh_l0' + sfx] = torch.randn(3 * H, generator=g) * 0.1
    Pd = {k: v.to(dev).contiguous() for k, v in P.items()}
    Gd = {k: torch.zeros_like(v) for k, v in Pd.items()}
    gru = layers.GruLayer(ctx, Pd, Gd, 'RNN.', name='full')
    gru.pack()
    X = ctx.act('fx', B, T, Cin)
    X.t[:B * T * X.ld].copy_(torch.randn(B * T * X.ld, generator=g).to(dev))
    dcat = ctx.act('fdcat', B, T, Cin + 2 * H)
    dcat.t[:B * T * dcat.ld].copy_((torch.randn(B * T * dcat.ld, generator=g) * 0.1).to(dev))
    res = {}
    for tag, persist in (('p1', 1), ('p2', 1), ('step', 0)):
        old = L.set_option('gru_persist', persist)
        try:
            cat = ctx.act('fcat' + tag, B, T, Cin + 2 * H)
            gi = ctx.act('fgi' + tag, B, T, 6 * H)
            gates = ctx.raw('fgates' + tag, B * T * 8 * H, ctx.tdt)
            dgi, dgh = ctx.act('fdgi' + tag, B, T, 6 * H), ctx.act('fdgh' + tag, B, T, 6 * H)
            dX = ctx.act('fdX' + tag, B, T, Cin)
            gru.fwd(X, cat, Cin, gi, gates)
            gru.check(B)
            gru.bwd(dcat, Cin, cat, Cin, gates, X, dgi, dgh, dX)
            gru.check(B)
            torch.cuda.synchronize()
            res[tag] = (cat.valid()[:, :, Cin:].float().clone(), dX.valid().float().clone(), dgh.valid().float().clone(),
                        Gd['RNN.weight_hh_l0'].clone())
        finally:
            L.set_option('gru_persist', old)
    for a, b in zip(res['p1'], res['p2']):
        assert torch.equal(a, b), 'persistent GRU is not deterministic'
    for name, a, b in zip(('h', 'dX', 'dgh', 'dW_hh'), res['p1'], res['step']):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-2 * scale, (name, float((a - b).abs().max()), scale)


def test_loss_norm_adam(zs):
    L, layers = zs
    torch.manual_seed(1)
    ctx = _ctx(layers, 'fp32')
    dev = ctx.device
    rows, Fv = 50, 80
    xd = torch.rand(rows, 96)
    x = torch.rand(rows, Fv)
    z = torch.logit(xd[:, :Fv].clamp(1e-4, 1 - 1e-4)).requires_grad_(True)
    loss = torch.mean(torch.abs(torch.sigmoid(z) - x))
    loss.backward()
    xdd, xx = torch.sigmoid(z.detach()), x
    xd[:, :Fv] = xdd
    dl = torch.full((rows, 96), float('nan'), device=dev)
    part, lo = torch.zeros(1024, device=dev), torch.zeros(1, device=dev)
    xd_d, xx_d = xd.to(dev), xx.to(dev)                # keep alive: the call only takes raw pointers
    L.call('zs_l1_loss', 'ZsL1Loss', ctx.stream, dtype=ctx.dt, x_dec=L.ptr(xd_d), ld_dec=96, x=L.ptr(xx_d), ldx=Fv, rows=rows,
           F=Fv, dlogits=L.ptr(dl), ldg=96, fill_cols=96, partial=L.ptr(part), loss_out=L.ptr(lo), grad_scale=1.0)
    torch.cuda.synchronize()
    assert abs(lo.item() - loss.item()) < 1e-6
    _close('dlogit', dl[:, :Fv], z.grad, 1e-5)
    assert (dl[:, Fv:] == 0).all()
    # sqnorm + clip + Adam against torch.optim.Adam / clip_grad_norm_
    n = 10007
    p0, g0 = torch.randn(n), torch.randn(n) * 3
    for max_norm in (5.0, 1e9):
        p = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([p], lr=1e-3, betas=(0.5, 0.9))
        pd, gd = p0.clone().to(dev), None
        m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        sq, dpart = torch.zeros(1, device=dev), torch.zeros(1024, dtype=torch.float64, device=dev)
        for step in range(1, 4):
            g = g0 * step
            p.grad = g.clone()
            tn = torch.nn.utils.clip_grad_norm_([p], max_norm)
            opt.step()
            gd = g.clone().to(dev)
            L.check(L.lib().zs_sqnorm(L.ptr(gd), n, L.ptr(dpart), L.ptr(sq), ctx.stream), 'sqnorm')
            L.call('zs_adam_clip', 'ZsAdam', ctx.stream, p=L.ptr(pd), g=L.ptr(gd), m=L.ptr(m), v=L.ptr(v), n=n, lr=1e-3, beta1=0.5,
                   beta2=0.9, eps=1e-8, bc1=1 - 0.5 ** step, bc2=1 - 0.9 ** step, sumsq=L.ptr(sq), max_norm=max_norm)
            torch.cuda.synchronize()
            assert abs(math.sqrt(sq.item()) - tn.item()) < 1e-4 * tn.item()
            _close('adam p step %d' % step, pd, p.detach(), 2e-6)


def test_pack_weight_known_answer(zs):
    L, layers = zs
    ctx = _ctx(layers, 'fp32')
    w = torch.arange(4 * 3 * 2, dtype=torch.float32).view(4, 3, 2)       # [Cout=4, Cin=3, k=2]
    l = _mk_conv(layers, ctx, w, None)
    torch.cuda.synchronize()
    wf = l.wf[:l.n_pad * l.ldw].view(l.n_pad, l.ldw).cpu()
    for n in range(4):
        for ci in range(3):
            for j in range(2):
                assert wf[n, j * l.cin_pad + ci] == w[n, ci, j]
    assert wf.sum() == w.sum()
    wd = l.wd[:l.n_pad_d * l.ldw_d].view(l.n_pad_d, l.ldw_d).cpu()
    for n in range(4):
        for ci in range(3):
            for j in range(2):
                assert wd[ci, j * l.cout_pad + n] == w[n, ci, j]
    assert wd.sum() == w.sum()


@pytest.mark.parametrize('dtype', DTYPES)
def test_pack_weight_batch_equals_single_calls(zs, dtype):
    """zs_pack_weight_batch (one launch per 32 jobs) writes exactly what the per-job zs_pack_weight calls write; 40 jobs
    of different shapes, kernel sizes and split2 flags span two launches."""
    L, layers = zs
    ctx = _ctx(layers, dtype)
    g = torch.Generator().manual_seed(9)
    specs = [(8 + 4 * (i % 5), 16 + 8 * (i % 3), 1 + (i % 4), bool(i % 2)) for i in range(20)]
    convs_a, convs_b = [], []
    for (co, ci, k, sp) in specs:
        w = torch.randn(co, ci, k, generator=g)
        convs_a.append(layers.ConvLayer(ctx, w.to(ctx.device), None, torch.zeros_like(w).to(ctx.device), None, split2=sp))
        convs_b.append(layers.ConvLayer(ctx, w.to(ctx.device), None, torch.zeros_like(w).to(ctx.device), None, split2=sp))
    for l in convs_a:
        l.pack()                                          # 40 separate launches
    with L.pack_batch(ctx.stream):
        for l in convs_b:
            l.pack()                                      # 2 launches
    torch.cuda.synchronize()
    for la, lb in zip(convs_a, convs_b):
        assert torch.equal(la.wf, lb.wf) and torch.equal(la.wd, lb.wd)
        assert float(la.wf.float().abs().sum()) > 0


def test_errors_are_loud(zs):
    L, layers = zs
    ctx = _ctx(layers, 'fp32')
    with pytest.raises(L.ZsError):
        L.call('zs_gemm_conv', 'ZsGemmConv', ctx.stream, dtype=0)            # null operands
    w = torch.randn(8, 32, 5)
    l = _mk_conv(layers, ctx, w, torch.zeros(8))
    A = _to_act(layers, ctx, 'x', torch.randn(1, 2, 32))
    out = ctx.act('y', 1, 2, 8)
    with pytest.raises(L.ZsError, match='Padding size should be less'):     # same failure the reference hits for T' = 2
        l.fwd(A, out=out)


@pytest.mark.parametrize('dtype', DTYPES)
def test_instnorm_fwd_stats_given(zs, dtype):
    """ZsInstNormFwd.stats_given: mean / rstd as inputs (the reductions over T skipped) -> bit for bit the output of the call that
    computed them, with the upsampled residual and the second output of the decoder's conv blocks."""
    L, layers = zs
    torch.manual_seed(3)
    B, T, C, nspk = 6, 64, 128, 3
    ctx = _ctx(layers, dtype)
    X = _to_act(layers, ctx, 'x', torch.randn(B, T, C) * 2 + 0.3)
    R = _to_act(layers, ctx, 'r', torch.randn(B, T // 2, C))
    o1, o2, p1, p2 = ctx.act('o1', B, T, C), ctx.act('o2', B, T, C), ctx.act('p1', B, T, C), ctx.act('p2', B, T, C)
    vec2 = torch.randn(nspk, C, device=ctx.device)
    idx = torch.tensor([0, 2, 1, 1, 0, 2], device=ctx.device)
    mean_b, rstd_b = ctx.f32('mean', B * C), ctx.f32('rstd', B * C)
    kw = dict(dtype=ctx.dt, x=X.ptr(), ldx=X.ld, vec2=L.ptr(vec2), vec2_ld=C, vec2_cols=C, idx=L.ptr(idx), mean=L.ptr(mean_b), rstd=L.ptr(rstd_b),
              B=B, T=T, C=C, eps=1e-5, drop_p=0.0, res_mode=L.ZS_RES_UPSAMPLE2, res=R.ptr(), ldres=R.ld, T_res=T // 2, res_pad_mode=L.ZS_PAD_REFLECT)
    L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, out=o1.ptr(), ldo=o1.ld, out2=o2.ptr(), ldo2=o2.ld, **kw)
    torch.cuda.synchronize()
    m0, r0 = mean_b.clone(), rstd_b.clone()
    L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, out=p1.ptr(), ldo=p1.ld, out2=p2.ptr(), ldo2=p2.ld, stats_given=1, **kw)
    torch.cuda.synchronize()
    assert torch.equal(o1.valid(), p1.valid()) and torch.equal(o2.valid(), p2.valid())
    assert torch.equal(m0, mean_b) and torch.equal(r0, rstd_b)              # inputs, not rewritten
    with pytest.raises(L.ZsError):
        L.call('zs_instnorm_fwd', 'ZsInstNormFwd', ctx.stream, dtype=ctx.dt, x=X.ptr(), ldx=X.ld, out=o1.ptr(), ldo=o1.ld, B=B, T=T, C=C, eps=1e-5,
               res_mode=L.ZS_RES_NONE, stats_given=1)


CONV2D_CASES = [
    # B, H, W, Cin, Cout, k, reflect
    (2, 12, 17, 16, 24, 5, True), (3, 16, 33, 64, 128, 5, True), (2, 8, 17, 128, 96, 5, True), (2, 12, 17, 32, 40, 5, False),
    (2, 9, 10, 64, 64, 3, True), (1, 64, 65, 64, 128, 5, True),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CONV2D_CASES)
def test_conv2d_layer_vs_torch(zs, dtype, case, gemm_variant):
    """layers.Conv2dLayer (2-D taps in the GEMM kernels' row pointers: ZsGemmConv.w_in / ZsGemmWgrad.w_in, stride-2 data gradient by
    (h, w) output parity + zs_conv2d_unpad) against torch: nn.Conv2d on the reflect- / zero-padded input, stride 2, its autograd
    data, weight and bias gradients.  Rows layout [B, H*W, C]; the weight's kernel dims run along (W, H)."""
    L, layers = zs
    B, H, W, Cin, Cout, k, reflect = case
    g = torch.Generator().manual_seed(5)
    x = _round(torch.randn(B, H, W, Cin, generator=g), dtype).requires_grad_(True)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)        # [co, ci, kw, kh]
    wr = _round(w, dtype).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    p = k // 2
    xp = F.pad(x.permute(0, 3, 2, 1), (p, p, p, p), mode='reflect' if reflect else 'constant')       # [B, C, W, H]: dims (2, 3) = (kw, kh)
    y = F.conv2d(xp, wr, b, stride=2).permute(0, 3, 2, 1)                                               # [B, Ho, Wo, Cout]
    dy = _round(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    ctx = _ctx(layers, dtype)
    dev = ctx.device
    w_d, b_d = w.to(dev).contiguous(), b.detach().to(dev).contiguous()
    lay = layers.Conv2dLayer(ctx, w_d, b_d, torch.zeros_like(w_d), torch.zeros_like(b_d), stride=2,
                             pad_mode=(L.ZS_PAD_REFLECT if reflect else L.ZS_PAD_ZERO), name='t2d')
    lay.pack()
    Ho, Wo = lay.out_hw(H, W)
    assert (Ho, Wo) == (y.shape[1], y.shape[2])
    X = _to_act(layers, ctx, 'x2', x.detach().reshape(B, H * W, Cin))
    out = ctx.act('y2', B, Ho * Wo, Cout)
    lay.fwd(X, H, W, out)
    dY = _to_act(layers, ctx, 'dy2', dy.reshape(B, Ho * Wo, Cout))
    dx = ctx.act('dx2', B, H * W, Cin)
    lay.dgrad(dY, H, W, dx, 't2d_g')
    lay.wgrad(dY, X, H, W)
    torch.cuda.synchronize()
    tol = _tol(dtype)
    _close('conv2d fwd', out.valid(), y.detach().reshape(B, Ho * Wo, Cout), tol)
    _close('conv2d dx', dx.valid(), x.grad.reshape(B, H * W, Cin), tol)
    _close('conv2d dW', lay.gw, wr.grad, tol)
    _close('conv2d db', lay.gb, b.grad, tol)
    # accumulate + add operand
    add = _to_act(layers, ctx, 'add2', torch.ones(B, H * W, Cin))
    lay.dgrad(dY, H, W, dx, 't2d_g', add=add)
    lay.wgrad(dY, X, H, W, accumulate=True)
    torch.cuda.synchronize()
    _close('conv2d dx + add', dx.valid(), x.grad.reshape(B, H * W, Cin) + 1.0, tol)
    _close('conv2d dW accumulated', lay.gw, 2 * wr.grad, tol)


@pytest.mark.parametrize('case', [(2, 12, 17, 64, 5, True), (3, 128, 513, 64, 5, True), (2, 9, 20, 32, 3, False), (1, 16, 33, 16, 5, True)])
def test_conv1_fwd_direct_vs_torch_and_im2col_path(zs, case):
    """zs_conv1_fwd (the critic's first Conv2d straight from the fp32 image, bf16 MFMA) against torch's conv2d on bf16-rounded
    operands (fp32 accumulation: 1e-5 of scale before the output's own bf16 rounding) and against the im2col + GEMM path it
    replaces (same rounding of the operands: equal up to the accumulation order, well inside one bf16 ulp of the output scale)."""
    L, layers = zs
    B, H, W, Cout, k, reflect = case
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, H, W, generator=g)
    w = torch.randn(Cout, 1, k, k, generator=g) / k                                # [co, 1, kw, kh]
    b = torch.randn(Cout, generator=g)
    p = k // 2
    xr, wr = x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()
    xp = F.pad(xr[:, None].permute(0, 1, 3, 2), (p, p, p, p), mode='reflect' if reflect else 'constant')      # [B, 1, W, H]
    y = F.leaky_relu(F.conv2d(xp, wr, b, stride=2), 0.01).permute(0, 3, 2, 1)                                   # [B, Ho, Wo, Cout]
    ctx = _ctx(layers, 'bf16')
    dev = ctx.device
    Ho, Wo = y.shape[1], y.shape[2]
    # the layer's virtual Linear weight [Cout, kh*k + kw] and its packed forward operand
    wv = w[:, 0].permute(0, 2, 1).reshape(Cout, k * k).contiguous().to(dev)
    lay = layers.ConvLayer(ctx, wv, b.to(dev), torch.zeros_like(wv), torch.zeros(Cout, device=dev))
    lay.pack()
    xd = x.to(dev).contiguous()
    out = ctx.act('c1o', B * Ho, Wo, Cout)
    pm = L.ZS_PAD_REFLECT if reflect else L.ZS_PAD_ZERO
    L.call('zs_conv1_fwd', 'ZsConv1Fwd', ctx.stream, x=L.ptr(xd), W=L.ptr(lay.wf), ldw=lay.ldw, bias=L.ptr(lay.b), act=L.ZS_ACT_LRELU, slope=0.01,
           out=out.ptr(), ldo=out.ld, B=B, H=H, Wd=W, Cout=Cout, k=k, pad_mode=pm)
    # the path it replaces
    xh = ctx.act('c1xh', B * Ho, Wo, k * k, ld=lay.cin_pad)
    L.call('zs_conv2d_gather', 'ZsConv2dGather', ctx.stream, dtype=ctx.dt, x=L.ptr(xd), ldx=1, x_f32=1, out=xh.ptr(), ldo=xh.ld, B=B, H_in=H,
           H_out=Ho, Wd=W, C=1, k=k, stride=2, pad=p, pad_mode=pm, full=1)
    ref = ctx.act('c1r', B * Ho, Wo, Cout)
    lay.fwd(xh, out=ref, act=L.ZS_ACT_LRELU, slope=0.01)
    torch.cuda.synchronize()
    got = out.valid().float().cpu().reshape(B, Ho, Wo, Cout)
    _close('conv1 direct vs torch', got, y, 1e-2)
    _close('conv1 direct vs im2col path', got, ref.valid().float().cpu().reshape(B, Ho, Wo, Cout), 8e-3)
    assert (got == ref.valid().float().cpu().reshape(B, Ho, Wo, Cout)).float().mean().item() > 0.98


@pytest.mark.parametrize('case', [(2, 12, 17, 64, 5, True), (3, 64, 257, 64, 5, True), (2, 9, 20, 32, 3, False), (1, 16, 33, 16, 5, True)])
def test_conv1_wgrad_direct_vs_torch(zs, case):
    """zs_conv1_wgrad (weight + bias gradient of the critic's first Conv2d straight from the fp32 image) against autograd on the
    bf16-rounded operands; accumulate adds; two runs are bitwise equal (fixed tile -> wave assignment, fixed-order reduction)."""
    L, layers = zs
    B, H, W, Cout, k, reflect = case
    g = torch.Generator().manual_seed(10)
    x = torch.rand(B, H, W, generator=g)
    p = k // 2
    xr = x.to(torch.bfloat16).float()
    w = (torch.randn(Cout, 1, k, k, generator=g) / k).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    xp = F.pad(xr[:, None].permute(0, 1, 3, 2), (p, p, p, p), mode='reflect' if reflect else 'constant')
    y = F.conv2d(xp, w, b, stride=2).permute(0, 3, 2, 1)                           # [B, Ho, Wo, Cout]
    dy = torch.randn(y.shape, generator=g).to(torch.bfloat16).float()
    y.backward(dy)
    ref_w = w.grad[:, 0].permute(0, 2, 1).reshape(Cout, k * k)                   # [co][kh*k + kw]
    ctx = _ctx(layers, 'bf16')
    dev = ctx.device
    Ho, Wo = y.shape[1], y.shape[2]
    gz = _to_act(layers, ctx, 'c1gz', dy.reshape(B * Ho, Wo, Cout))
    xd = x.to(dev).contiguous()
    dW, db = torch.zeros(Cout, k * k, device=dev), torch.zeros(Cout, device=dev)
    ws = torch.empty(L.lib().zs_conv1_wgrad_workspace() // 4, device=dev)
    kw = dict(x=L.ptr(xd), gz=gz.ptr(), ldg=gz.ld, dW=L.ptr(dW), lddw=k * k, db=L.ptr(db), B=B, H=H, Wd=W, Cout=Cout, k=k,
              pad_mode=(L.ZS_PAD_REFLECT if reflect else L.ZS_PAD_ZERO), workspace=L.ptr(ws), workspace_bytes=ws.numel() * 4)
    L.call('zs_conv1_wgrad', 'ZsConv1Wgrad', ctx.stream, accumulate=0, **kw)
    torch.cuda.synchronize()
    first = (dW.clone(), db.clone())
    _close('conv1 dW', dW, ref_w, 2e-3)
    _close('conv1 db', db, b.grad, 2e-3)
    L.call('zs_conv1_wgrad', 'ZsConv1Wgrad', ctx.stream, accumulate=0, **kw)
    torch.cuda.synchronize()
    assert torch.equal(dW, first[0]) and torch.equal(db, first[1])
    L.call('zs_conv1_wgrad', 'ZsConv1Wgrad', ctx.stream, accumulate=1, **kw)
    torch.cuda.synchronize()
    _close('conv1 dW accumulated', dW, 2 * ref_w, 2e-3)
    _close('conv1 db accumulated', db, 2 * b.grad, 2e-3)
