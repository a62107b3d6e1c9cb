"""Stage 2 nets (reference model/model.py:113-228): PatchDiscriminator and TargetClassifier as drop-in modules, and the
engine that runs them -- forward, backward and the WGAN-GP double backward (utils.py:58-77) -- on libzs_amd.so.

Layout: activations are channels-last [B, H, W, C] with H = time, W = frequency, so the loader's [B, T, F] batch is the
input without a transpose (the reference's Conv2d sees [B, 1, F, T]; its kernel index along F is the GEMM's tap axis here
and the one along T the im2col axis).  A 5x5 / stride-2 Conv2d = zs_conv2d_gather (im2col along H) + the implicit-GEMM conv
over W on the weight viewed as a Conv1d weight [Cout, 5*C, 5] ("virtual" weight: a permuted copy refreshed after every
optimizer step; its gradient is accumulated in the same view and permuted back once per step).

The gradient penalty needs d/dtheta of || d D(x^) / d x^ ||: `gp_backward` runs the first backward (data gradients only),
then its reverse: an adjoint pass in forward order (every conv-transpose of the first backward becomes a forward conv, every
InstanceNorm backward its own adjoint, zs_in2d_adj) and a second sweep through the forward graph that carries the adjoints
of xhat and rstd back to the weights.  LeakyReLU is piecewise linear (no second-order term), Dropout2d a constant mask.
"""
import torch
import torch.nn as nn

from . import _lib as L
import os

from .layers import Act, ConvLayer, Ctx, join_side, rup, side_launch
from .model import ZsModule

LRELU = L.ZS_ACT_LRELU
DIRECT_CONV1 = os.environ.get('ZS_PATCH_CONV1_DIRECT', '1') == '1'
EPS_IN = 1e-5
_UID = [0]


def _half(n):
    """Output length of a k5 / stride-2 conv behind pad 2 + 2."""
    return (n + 4 - 5) // 2 + 1


class _Net2d(ZsModule):
    """Shared body of PatchDiscriminator / TargetClassifier: conv1..conv6 (+ conv7) + conv_classify (model/model.py:114-131)."""

    def __init__(self, n_class, ns, dp, seg_len, with_val, dtype=None):
        super(_Net2d, self).__init__(dtype)
        self.ns, self.dp, self.seg_len, self.n_class, self.with_val = ns, dp, seg_len, n_class, with_val
        self.conv1 = nn.Conv2d(1, 64, kernel_size=5, stride=2)
        self.conv2 = nn.Conv2d(64, 128, kernel_size=5, stride=2)
        self.conv3 = nn.Conv2d(128, 256, kernel_size=5, stride=2)
        self.conv4 = nn.Conv2d(256, 512, kernel_size=5, stride=2)
        self.conv5 = nn.Conv2d(512, 512, kernel_size=5, stride=2)
        self.conv6 = nn.Conv2d(512, 32, kernel_size=1)
        if seg_len not in (128, 64, 32):
            raise NotImplementedError('Segement length {} is not supported!'.format(seg_len))
        kt = {128: 4, 64: 2, 32: 1}[seg_len]
        self.conv7 = nn.Conv2d(32, 1, kernel_size=(17, kt))          # TargetClassifier owns (and checkpoints) one too
        self.conv_classify = nn.Conv2d(32, n_class, kernel_size=(17, kt))

    def _make_engine(self, ctx, P, G):
        return PatchEngine(ctx, P, G, self.n_class, self.ns, self.dp, self.seg_len)

    def _run(self, x, classify, drop_masks=None):
        """x [B, 513, T] (reference layout) -> (mean_val [B], logits [B, n_class])."""
        eng = self._engine()
        xb = x.detach().permute(0, 2, 1).contiguous().float()
        val, logits = eng.forward(xb, 'api', self.training, masks=drop_masks, classify=classify)
        return val.clone(), (logits.clone() if logits is not None else None)


class PatchDiscriminator(_Net2d):
    """model/model.py:113-173."""

    def __init__(self, n_class=33, ns=0.2, dp=0.1, seg_len=128, dtype=None):
        super(PatchDiscriminator, self).__init__(n_class, ns, dp, seg_len, True, dtype)

    def forward(self, x, classify=False, drop_masks=None):
        val, logits = self._run(x, classify, drop_masks)
        return (val, logits) if classify else val


class TargetClassifier(_Net2d):
    """model/model.py:176-228."""

    def __init__(self, n_class=2, ns=0.2, dp=0.8, seg_len=128, dtype=None):
        super(TargetClassifier, self).__init__(n_class, ns, dp, seg_len, False, dtype)

    def forward(self, x, drop_masks=None):
        return self._run(x, True, drop_masks)[1]


class _Conv2dLayer(object):
    """One k5 / stride-2 Conv2d: real parameter [Cout, C, kf, kt] <-> virtual Conv1d weight [Cout, kt*C, kf]."""

    def __init__(self, ctx, W, b, gW, gb, pad_mode, name):
        self.W, self.gW = W, gW
        self.Cout, self.C, self.kf, self.kt = W.shape
        dev = W.device
        self.Wv = torch.zeros(self.Cout, self.kt * self.C, self.kf, dtype=torch.float32, device=dev)
        self.gWv = torch.zeros_like(self.Wv)
        self.layer = ConvLayer(ctx, self.Wv, b, self.gWv, gb, stride=2, pad_mode=pad_mode, name=name)
        self.pad_mode = pad_mode

    def sync(self):
        self.Wv.view(self.Cout, self.kt, self.C, self.kf).copy_(self.W.permute(0, 3, 1, 2))

    def flush(self):
        self.gW.copy_(self.gWv.view(self.Cout, self.kt, self.C, self.kf).permute(0, 2, 3, 1))


class _Conv2dFirstLayer(object):
    """conv1 (C = 1): im2col along both axes -- all 25 taps of a position in ONE 128-byte K chunk -- and a Linear over those rows:
    real parameter [Cout, 1, kf, kt] <-> virtual Linear weight [Cout, kt*kf].  (The H-only gather pads the 5 columns of every
    W tap to a whole chunk: 12.8 x the input bytes and MFMA work, 16 % of a discriminator step.)"""

    def __init__(self, ctx, W, b, gW, gb, pad_mode, name):
        self.W, self.gW = W, gW
        self.Cout, self.C, self.kf, self.kt = W.shape
        self.Wv = torch.zeros(self.Cout, self.kt * self.kf * self.C, dtype=torch.float32, device=W.device)
        self.gWv = torch.zeros_like(self.Wv)
        self.layer = ConvLayer(ctx, self.Wv, b, self.gWv, gb, name=name)
        self.pad_mode = pad_mode
        self.full = True

    def sync(self):      # virtual column (kt*kf + kf_i)*C + c  (zs_conv2d_gather(full): kh = time tap, kw = frequency tap)
        self.Wv.view(self.Cout, self.kt, self.kf, self.C).copy_(self.W.permute(0, 3, 2, 1))

    def flush(self):
        self.gW.copy_(self.gWv.view(self.Cout, self.kt, self.kf, self.C).permute(0, 3, 2, 1))


class _HeadLayer(object):
    """conv7 / conv_classify: an un-padded Conv2d whose kernel spans the whole [17, kt] map = a Linear on the flattened rows."""

    def __init__(self, ctx, W, b, gW, gb, name):
        self.W, self.gW = W, gW
        self.n, self.C, self.kf, self.kt = W.shape
        self.Wv = torch.zeros(self.n, self.kt * self.kf * self.C, dtype=torch.float32, device=W.device)
        self.gWv = torch.zeros_like(self.Wv)
        self.layer = ConvLayer(ctx, self.Wv, b, self.gWv, gb, name=name)

    def sync(self):
        self.Wv.view(self.n, self.kt, self.kf, self.C).copy_(self.W.permute(0, 3, 2, 1))

    def flush(self):
        self.gW.copy_(self.gWv.view(self.n, self.kt, self.kf, self.C).permute(0, 3, 2, 1))


class PatchEngine(object):
    def __init__(self, ctx, P, G, n_class, ns, dp, seg_len):
        _UID[0] += 1
        self.uid = _UID[0]
        self.ctx = ctx
        # several weight gradients accumulate into one buffer per step (real / fake / penalty passes): every layer's weight
        # gradients are pinned to ONE side stream, so stream order is the summation order (ZS_PATCH_OVERLAP=0: all on the main stream)
        ctx.overlap_wgrad = os.environ.get('ZS_PATCH_OVERLAP', '1') == '1'
        ctx.pin_wgrad_streams = True
        self.ns, self.dp, self.n_class = float(ns), float(dp), n_class
        self.pad_mode = L.ZS_PAD_REFLECT if seg_len >= 64 else L.ZS_PAD_ZERO
        self.G = G
        self.convs = [(_Conv2dFirstLayer if (i == 1 and P['conv1.weight'].shape[1] * 25 <= ctx.kc) else _Conv2dLayer)(
            ctx, P['conv%d.weight' % i], P['conv%d.bias' % i], G['conv%d.weight' % i], G['conv%d.bias' % i], self.pad_mode, 'pd_conv%d' % i)
            for i in range(1, 6)]
        w6 = P['conv6.weight']
        self.conv6 = ConvLayer(ctx, w6.view(w6.shape[0], w6.shape[1]), P['conv6.bias'], G['conv6.weight'].view(w6.shape[0], w6.shape[1]),
                               G['conv6.bias'], name='pd_conv6')
        self.head_val = _HeadLayer(ctx, P['conv7.weight'], P['conv7.bias'], G['conv7.weight'], G['conv7.bias'], 'pd_conv7')
        self.head_clf = _HeadLayer(ctx, P['conv_classify.weight'], P['conv_classify.bias'], G['conv_classify.weight'],
                                   G['conv_classify.bias'], 'pd_clf')
        self.tapes = {}
        self._zeroed = False

    # ---- parameters ---------------------------------------------------------------------------------------------------------
    def pack(self):
        with L.pack_batch(self.ctx.stream):
            for c in self.convs:
                c.sync()
                c.layer.pack()
            self.conv6.pack()
            for h in (self.head_val, self.head_clf):
                h.sync()
                h.layer.pack()

    def zero_grads(self):
        """Start of a step: every weight gradient of the step ACCUMULATES (real / fake / penalty passes)."""
        for c in self.convs:
            c.gWv.zero_()
        for h in (self.head_val, self.head_clf):
            h.gWv.zero_()
        bases = {id(g._base): g._base for g in self.G.values()}
        if len(bases) == 1 and None not in [g._base for g in self.G.values()]:
            next(iter(bases.values())).zero_()                     # the views tile the net's flat gradient buffer: one fill
        else:
            for k, g in self.G.items():
                g.zero_()

    def flush_grads(self):
        """End of a step: virtual weight gradients -> the parameters' gradient views (reference layouts)."""
        join_side(self.ctx.device)                      # the side-stream weight gradients of this step
        for c in self.convs:
            c.flush()
        for h in (self.head_val, self.head_clf):
            h.flush()

    # ---- helpers ------------------------------------------------------------------------------------------------------------
    def _name(self, key, what, *dims):
        return 'p%d_%s_%s_%s' % (self.uid, key, what, '_'.join(str(d) for d in dims))

    def _ws(self, B, T, C):
        n = L.lib().zs_row_moments_workspace(B, T, C)
        return self.ctx.f32('p%d_momws' % self.uid, (n + 3) // 4)

    def _moments(self, u, B, T, C, s1, s2=None, v=None, y=None, w=None, s3=None, center_sum=None):
        c = self.ctx
        ws = self._ws(B, T, C)
        L.call('zs_row_moments', 'ZsRowMoments', c.stream, dtype=c.dt, u=u.ptr(), ldu=u.ld, v=(v.ptr() if v is not None else None),
               ldv=(v.ld if v is not None else 0), y=(y.ptr() if y is not None else None), ldy=(y.ld if y is not None else 0),
               slope=self.ns, w=(w.ptr() if w is not None else None), ldw=(w.ld if w is not None else 0),
               center_sum=L.ptr(center_sum), center_scale=1.0 / T, B=B, T=T, C=C, s1=L.ptr(s1), s2=L.ptr(s2), s3=L.ptr(s3),
               partial=L.ptr(ws), partial_bytes=ws.numel() * 4)

    def _stat(self, key, what, B, C):
        return self.ctx.f32(self._name(key, what, B, C), B * C)

    def _gather(self, x_ptr, ldx, x_f32, B, Hin, Hout, Wd, C, out, full=False):
        c = self.ctx
        L.call('zs_conv2d_gather', 'ZsConv2dGather', c.stream, dtype=c.dt, x=x_ptr, ldx=ldx, x_f32=int(x_f32), out=out.ptr(), ldo=out.ld,
               B=B, H_in=Hin, H_out=Hout, Wd=Wd, C=C, k=5, stride=2, pad=2, pad_mode=self.pad_mode, full=int(full))

    def _conv1_ok(self, cl, src_f32):
        c = self.ctx
        return (DIRECT_CONV1 and getattr(cl, 'full', False) and src_f32 and c.dtype_name == 'bf16' and cl.C == 1 and cl.Cout % 16 == 0
                and cl.Cout <= 64 and cl.kf == cl.kt and cl.kf * cl.kt < 32)

    def _conv1_wgrad(self, cl, img_ptr, gz, B, H, W, bias=True):
        """Accumulating weight (+ bias) gradient of the first layer straight from the image (zs_conv1_wgrad), on the layer's side stream."""
        c, lay = self.ctx, cl.layer
        stream, ws = side_launch(c, lay.sid, L.lib().zs_conv1_wgrad_workspace())
        L.call('zs_conv1_wgrad', 'ZsConv1Wgrad', stream, x=img_ptr, gz=gz.ptr(), ldg=gz.ld, dW=L.ptr(lay.gw), lddw=lay.gw.shape[1],
               db=(L.ptr(lay.gb) if bias else None), accumulate=1, B=B, H=H, Wd=W, Cout=cl.Cout, k=cl.kf, pad_mode=self.pad_mode,
               workspace=ws.data_ptr(), workspace_bytes=ws.numel())

    def _conv1_direct(self, cl, src_ptr, src_f32, B, H, W, out, act, slope, bias):
        """First layer straight from the fp32 image (zs_conv1_fwd: bf16, one input channel, <= 64 output channels): the 25-column
        im2col buffer is then only the weight gradient's operand.  Returns False where the GEMM over the im2col rows has to run."""
        c = self.ctx
        if not self._conv1_ok(cl, src_f32):
            return False
        lay = cl.layer
        L.call('zs_conv1_fwd', 'ZsConv1Fwd', c.stream, x=src_ptr, W=L.ptr(lay.wf), ldw=lay.ldw, bias=(L.ptr(lay.b) if bias else None), act=act,
               slope=slope, out=out.ptr(), ldo=out.ld, B=B, H=H, Wd=W, Cout=cl.Cout, k=cl.kf, pad_mode=self.pad_mode)
        return True

    def _gathered(self, cl, name, B, H, W, C):
        """The im2col buffer of layer `cl` for an input of [B, H, W, C]: rows (b, ho) x W with 5C columns, or -- first layer --
        rows (b, ho) x W_out with all 25 C columns."""
        Ho, Wo = _half(H), _half(W)
        if getattr(cl, 'full', False):
            return self.ctx.act(name, B * Ho, Wo, 25 * C, ld=cl.layer.cin_pad)
        return self.ctx.act(name, B * Ho, W, 5 * C, ld=cl.layer.cin_pad)

    def _fold(self, gp, B, Hin, Hout, Wd, C, out_ptr, ldo, out_f32, fill_cols, add=None, full=False):
        c = self.ctx
        L.call('zs_conv2d_fold', 'ZsConv2dFold', c.stream, dtype=c.dt, gp=gp.ptr(), ldg=gp.ld, out=out_ptr, ldo=ldo, out_f32=int(out_f32),
               fill_cols=fill_cols, add=(add.ptr() if add is not None else None), ldadd=(add.ld if add is not None else 0),
               B=B, H_in=Hin, H_out=Hout, Wd=Wd, C=C, k=5, stride=2, pad=2, pad_mode=self.pad_mode, gp_rows=(0 if full else gp.T),
               full=int(full))

    def _f32_act(self, t, name, B, n):
        """fp32 tensor [B, n] -> Act [B, 1, n] in the compute dtype."""
        c = self.ctx
        a = c.act(name, B, 1, n)
        L.call('zs_cast_rows', 'ZsCastRows', c.stream, dtype=c.dt, src=L.ptr(t), ld_src=t.stride(0), src_f32=1, dst=a.ptr(), ld_dst=a.ld,
               dst_f32=0, col_off=0, rows=B, cols=n, fill_cols=a.ld, act=L.ZS_ACT_NONE)
        return a

    # ---- forward ------------------------------------------------------------------------------------------------------------
    def forward(self, x_btf, key, training, masks=None, classify=True, need_val=True):
        """x_btf: fp32 [B, T, F] contiguous on the device.  masks: optional six [B, C] keep masks (Dropout2d); otherwise drawn with
        torch.rand when training.  Returns (val fp32 [B] or None, logits fp32 [B, n_class] or None); keeps the tape under `key`."""
        c, ns = self.ctx, self.ns
        B, T, F = x_btf.shape
        assert x_btf.dtype == torch.float32 and x_btf.is_contiguous()
        dp = self.dp if training else 0.0
        tp = {'B': B, 'x': x_btf, 'layers': [], 'dp': dp}
        H, W, C = T, F, 1
        src_ptr, src_ld, src_f32 = L.ptr(x_btf), 1, True
        for i in range(6):
            img = None
            if i < 5:
                cl = self.convs[i]
                Ho, Wo = _half(H), _half(W)
                Cout = cl.Cout
                y = c.act(self._name(key, 'y%d' % i, B, H, W), B * Ho, Wo, Cout)
                if i == 0 and self._conv1_direct(cl, src_ptr, src_f32, B, H, W, y, LRELU, ns, True):
                    xh, img = None, src_ptr              # no im2col buffer: forward and weight gradient read the image itself
                else:
                    xh = self._gathered(cl, self._name(key, 'xh%d' % i, B, H, W), B, H, W, C)
                    self._gather(src_ptr, src_ld, src_f32, B, H, Ho, W, C, xh, full=getattr(cl, 'full', False))
                    cl.layer.fwd(xh, out=y, act=LRELU, slope=ns)
                layer, xin = cl.layer, xh
            else:
                Ho, Wo, Cout = H, W, self.conv6.Cout
                y = c.act(self._name(key, 'y%d' % i, B, H, W), B * H, W, Cout)
                self.conv6.fwd(prev_a, out=y, act=LRELU, slope=ns)
                layer, xin = self.conv6, prev_a
            Tn = Ho * Wo
            yv = Act(y.t, B, Tn, Cout, y.ld)
            mean, rstd = self._stat(key, 'mean_%d' % i, B, Cout), self._stat(key, 'rstd_%d' % i, B, Cout)
            ws = self._ws(B, Tn, Cout)
            L.check(L.lib().zs_in2d_stats(c.dt, yv.ptr(), yv.ld, B, Tn, Cout, EPS_IN, L.ptr(mean), L.ptr(rstd), L.ptr(ws), ws.numel() * 4, c.stream),
                    'zs_in2d_stats')                                    # one pass over y (was: two moment passes + finalize)
            dm = None
            if dp > 0.0:
                if masks is not None and masks[i] is not None:
                    keep = masks[i].to(c.device, torch.float32)
                else:
                    keep = (torch.rand(B, Cout, device=c.device) >= dp).float()
                dm = (keep / (1.0 - dp)).contiguous()
            a = c.act(self._name(key, 'a%d' % i, B, H, W), B, Tn, Cout)
            L.call('zs_in2d_fwd', 'ZsIn2dFwd', c.stream, dtype=c.dt, y=yv.ptr(), ldy=yv.ld, a=a.ptr(), lda=a.ld, mean=L.ptr(mean),
                   rstd=L.ptr(rstd), dm=L.ptr(dm), B=B, T=Tn, C=Cout)
            tp['layers'].append(dict(layer=layer, xin=xin, img=img, y=yv, a=a, rstd=rstd, dm=dm, H=H, W=W, C=C, Ho=Ho, Wo=Wo, Cout=Cout, Tn=Tn))
            prev_a = Act(a.t, B * Ho, Wo, Cout, a.ld)             # rows view for the next conv / linear
            src_ptr, src_ld, src_f32 = a.ptr(), a.ld, False
            H, W, C = Ho, Wo, Cout
        a6 = tp['layers'][5]['a']
        if a6.ld != a6.C:
            raise L.ZsError('PatchEngine: the head expects unpadded 32-channel rows')
        flat = Act(a6.t, B, 1, a6.T * a6.C, a6.T * a6.C)
        tp['flat'] = flat
        val = logits = None
        if need_val:
            vo = c.act(self._name(key, 'val', B), B, 1, 1, dtype=torch.float32)
            self.head_val.layer.fwd(flat, out=vo, out_f32=True)
            val = vo.valid()[:, 0, 0]
        if classify:
            lo = c.act(self._name(key, 'logits', B), B, 1, self.n_class, dtype=torch.float32)
            self.head_clf.layer.fwd(flat, out=lo, out_f32=True)
            logits = lo.valid()[:, 0, :]
        self.tapes[key] = tp
        return val, logits

    # ---- backward (also the first backward of the gradient penalty) -----------------------------------------------------------
    def backward(self, key, dval=None, dlogits=None, need_dx=False, param_grads=True, keep=False):
        """dval fp32 [B] / dlogits fp32 [B, n_class]: gradients of the loss w.r.t. the two heads.  Weight gradients ACCUMULATE.
        keep=True stores what the double backward needs (incoming gradients, gz, S2).  Returns dx fp32 [B, T, F] or None."""
        c, ns, tp = self.ctx, self.ns, self.tapes[key]
        B, flat = tp['B'], tp['flat']
        n6 = flat.C
        ga = c.act(self._name(key, 'ga5', B), B, 1, n6)
        first = True
        for head, d, nm in ((self.head_val, dval, 'dv'), (self.head_clf, dlogits, 'dl')):
            if d is None:
                continue
            d2 = d.reshape(B, -1).contiguous().float()
            da = self._f32_act(d2, self._name(key, nm, B), B, d2.shape[1])
            if param_grads:
                head.layer.wgrad(da, flat, accumulate=True)
            head.layer.dgrad(da, 1, ga, add_src=(None if first else ga))
            first = False
        assert not first, 'backward needs dval or dlogits'
        L6 = tp['layers'][5]
        g_in = Act(ga.t, B, L6['Tn'], L6['Cout'], L6['Cout'])
        dx = None
        for i in (5, 4, 3, 2, 1, 0):
            Ld = tp['layers'][i]
            g_in = self._in_bwd(key, 'b', i, Ld, g_in, None, keep)
            gz = g_in
            gzr = Act(gz.t, B * Ld['Ho'], Ld['Wo'], Ld['Cout'], gz.ld)
            if param_grads:
                self._wgrad(i, Ld, gz, gzr, Ld['xin'], Ld['img'])
            if i == 0 and not need_dx:
                break
            g_in, dx = self._dgrad(key, 'b', i, Ld, gzr, None)
        return dx

    def _wgrad(self, i, Ld, gz, gz_rows, xin, img, bias=True):
        """Accumulating weight gradient of layer i: from the gathered rows `xin`, or (first layer, direct) from the image `img`."""
        if img is not None:
            self._conv1_wgrad(self.convs[i], img, gz, gz.B, Ld['H'], Ld['W'], bias=bias)
        else:
            Ld['layer'].wgrad(gz_rows, xin, accumulate=True, bias=bias)

    def _in_bwd(self, key, tag, i, Ld, ga, S2x, keep):
        """InstanceNorm2d + Dropout2d + LeakyReLU backward of layer i: ga (gradient w.r.t. a) -> gz (w.r.t. the conv output)."""
        c = self.ctx
        B, Tn, C = Ld['a'].B, Ld['Tn'], Ld['Cout']
        S1, S2 = self._stat(key, tag + 'S1_%d' % i, B, C), self._stat(key, tag + 'S2_%d' % i, B, C)
        self._moments(ga, B, Tn, C, S1, s2=S2, v=Ld['a'])
        gz = c.act(self._name(key, tag + 'gz%d' % i, B), B, Tn, C)
        L.call('zs_in2d_bwd', 'ZsIn2dBwd', c.stream, dtype=c.dt, ga=ga.ptr(), ldga=ga.ld, a=Ld['a'].ptr(), lda=Ld['a'].ld, y=Ld['y'].ptr(),
               ldy=Ld['y'].ld, S1=L.ptr(S1), S2=L.ptr(S2), S2x=L.ptr(S2x), rstd=L.ptr(Ld['rstd']), dm=L.ptr(Ld['dm']), slope=self.ns,
               gz=gz.ptr(), ldgz=gz.ld, B=B, T=Tn, C=C)
        if keep:
            Ld['ga'], Ld['gz'], Ld['S2'] = ga, gz, S2
        return gz

    def _dgrad(self, key, tag, i, Ld, gzr, add):
        """Data gradient of layer i's conv: gz rows -> (gradient w.r.t. the layer input as Act [B, H*W, C], fp32 dx for layer 0)."""
        c = self.ctx
        B = Ld['a'].B
        H, W, C = Ld['H'], Ld['W'], Ld['C']
        if i == 5:
            out = c.act(self._name(key, tag + 'gin%d' % i, B), B * H, W, C)
            Ld['layer'].dgrad(gzr, W, out, add_src=(Act(add.t, B * H, W, C, add.ld) if add is not None else None))
            return Act(out.t, B, H * W, C, out.ld), None
        lay = Ld['layer']
        if i == 0 and getattr(self.convs[0], 'full', False):          # first layer: Linear over the 25-tap rows, then col2im
            dcol = c.act(self._name(key, tag + 'dcol', B), B * Ld['Ho'], Ld['Wo'], 25 * C)
            lay.dgrad(gzr, Ld['Wo'], dcol)
            dx = c.f32(self._name(key, tag + 'dx', B), B * H * W)
            self._fold(dcol, B, H, Ld['Ho'], W, C, L.ptr(dx), 1, True, 0, full=True)
            return None, dx[:B * H * W].view(B, H, W)
        Wp = W + 4
        gp = c.act(self._name(key, tag + 'gp%d' % i, B), B * Ld['Ho'], Wp + (Wp & 1), 5 * C)    # (even rows: stride-2 dgrad by parity)
        lay.dgrad(gzr, W, gp)
        if i == 0:
            dx = c.f32(self._name(key, tag + 'dx', B), B * H * W)
            self._fold(gp, B, H, Ld['Ho'], W, C, L.ptr(dx), 1, True, 0)
            return None, dx[:B * H * W].view(B, H, W)
        out = c.act(self._name(key, tag + 'gin%d' % i, B), B, H * W, C)
        self._fold(gp, B, H, Ld['Ho'], W, C, out.ptr(), out.ld, False, out.ld, add=add)
        return out, None

    # ---- gradient penalty (utils.py:58-77) ---------------------------------------------------------------------------------------
    def gp_backward(self, key, scale):
        """On the tape of the interpolated batch x^: gp = mean_b (1 - ||dD/dx^_b||)^2 and the accumulation of scale * d gp / d theta
        into the weight gradients.  Returns the device scalar gp."""
        c, ns, tp = self.ctx, self.ns, self.tapes[key]
        B = tp['B']
        ones = torch.ones(B, dtype=torch.float32, device=c.device)
        g = self.backward(key, dval=ones, need_dx=True, param_grads=False, keep=True)         # create_graph=True backward
        n = g[0].numel()
        s = c.f32(self._name(key, 'gp_s', B), B)
        gp = c.f32(self._name(key, 'gp_v', B), 1)
        gbar = c.f32(self._name(key, 'gp_gbar', B), B * n)
        L.check(L.lib().zs_gp_penalty(L.ptr(g), B, n, float(scale), L.ptr(s), L.ptr(gp), L.ptr(gbar), c.stream), 'zs_gp_penalty')
        # ---- adjoint pass (forward order): gbar_a[l-1] -> gbar_z[l] -> gbar_a[l]; weight adjoints of the conv-transposes
        src_ptr, src_ld, src_f32 = L.ptr(gbar), 1, True
        prev_rows = None
        for i in range(6):
            Ld = tp['layers'][i]
            H, W, C, Ho, Wo, Cout, Tn = Ld['H'], Ld['W'], Ld['C'], Ld['Ho'], Ld['Wo'], Ld['Cout'], Ld['Tn']
            gbz = c.act(self._name(key, 'gbz%d' % i, B), B * Ho, Wo, Cout)
            gz_rows = Act(Ld['gz'].t, B * Ho, Wo, Cout, Ld['gz'].ld)
            if i == 0 and self._conv1_direct(self.convs[i], src_ptr, src_f32, B, H, W, gbz, L.ZS_ACT_NONE, 0.0, False):
                self._conv1_wgrad(self.convs[i], src_ptr, Ld['gz'], B, H, W, bias=False)
            elif i < 5:
                xh = self._gathered(self.convs[i], self._name(key, 'gxh%d' % i, B), B, H, W, C)
                self._gather(src_ptr, src_ld, src_f32, B, H, Ho, W, C, xh, full=getattr(self.convs[i], 'full', False))
                Ld['layer'].fwd(xh, out=gbz, bias=False)
                Ld['layer'].wgrad(gz_rows, xh, accumulate=True, bias=False)
            else:
                Ld['layer'].fwd(prev_rows, out=gbz, bias=False)
                Ld['layer'].wgrad(gz_rows, prev_rows, accumulate=True, bias=False)
            gbzv = Act(gbz.t, B, Tn, Cout, gbz.ld)
            A1, A2, A3 = (self._stat(key, 'A%d_%d' % (k, i), B, Cout) for k in (1, 2, 3))
            self._moments(gbzv, B, Tn, Cout, A1, s2=A2, v=Ld['a'], y=Ld['y'], w=Ld['gz'], s3=A3)
            gba = c.act(self._name(key, 'gba%d' % i, B), B, Tn, Cout)
            xba = c.act(self._name(key, 'xba%d' % i, B), B, Tn, Cout)
            L.call('zs_in2d_adj', 'ZsIn2dAdj', c.stream, dtype=c.dt, gbz=gbzv.ptr(), ldgbz=gbzv.ld, y=Ld['y'].ptr(), ldy=Ld['y'].ld,
                   a=Ld['a'].ptr(), lda=Ld['a'].ld, ga=Ld['ga'].ptr(), ldga=Ld['ga'].ld, A1=L.ptr(A1), A2=L.ptr(A2), S2=L.ptr(Ld['S2']),
                   rstd=L.ptr(Ld['rstd']), dm=L.ptr(Ld['dm']), slope=ns, gba=gba.ptr(), ldgba=gba.ld, xba=xba.ptr(), ldxba=xba.ld,
                   B=B, T=Tn, C=Cout)
            Ld['xba'], Ld['A3'] = xba, A3
            prev_rows = Act(gba.t, B * Ho, Wo, Cout, gba.ld)
            src_ptr, src_ld, src_f32 = gba.ptr(), gba.ld, False
        # the first backward started from ga5 = W7 (dval = 1): adjoint of W7 = sum_b gbar_a5[b]
        gba5 = Act(prev_rows.t, B, 1, tp['flat'].C, tp['flat'].C)
        ones_a = self._f32_act(ones.view(B, 1), self._name(key, 'ones', B), B, 1)
        self.head_val.layer.wgrad(ones_a, gba5, accumulate=True, bias=False)
        # ---- reverse sweep through the forward graph of x^ with the adjoints of xhat (xba) and rstd (A3)
        incoming = None
        for i in (5, 4, 3, 2, 1, 0):
            Ld = tp['layers'][i]
            ga = Ld['xba'] if incoming is None else incoming           # incoming already holds dgrad + xba (fold / dgrad epilogue)
            zb = self._in_bwd(key, 'r', i, Ld, ga, Ld['A3'], False)
            zbr = Act(zb.t, B * Ld['Ho'], Ld['Wo'], Ld['Cout'], zb.ld)
            self._wgrad(i, Ld, zb, zbr, Ld['xin'], Ld['img'])
            if i == 0:
                break
            incoming, _ = self._dgrad(key, 'r', i, Ld, zbr, tp['layers'][i - 1]['xba'])
        return gp
