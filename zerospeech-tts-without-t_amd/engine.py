"""Forward / backward schedules of the autoencoder on the HIP kernels (channels-last, explicit tape).

EncoderEngine  <->  Encoder.forward          (reference model/model.py:440-489) and its autograd backward
DecoderEngine  <->  Decoder.forward          (reference model/model.py:344-365) and its autograd backward
Every arithmetic step is a libzs_amd.so kernel; torch only owns the memory.
"""
import os
import torch

from . import _lib as L
from .layers import Act, ConvLayer, GruLayer, colsum_ok, fork_side, join_side, rup

LRELU = L.ZS_ACT_LRELU
EPS_IN = 1e-5
_UID = [0]


def _uid():
    _UID[0] += 1
    return _UID[0]


def _half_up(t):
    return (t + 1) // 2


class EncoderEngine(object):
    def __init__(self, ctx, P, G, c_in, c_h1, c_h2, c_h3, enc_size, ns, dp, seg_len):
        """P / G: dicts name -> fp32 parameter / gradient tensors (reference state_dict names)."""
        self.ctx = ctx
        self.uid = _uid()
        self.c_in, self.c1, self.c2, self.H, self.E = c_in, c_h1, c_h2, c_h3, enc_size
        self.ns, self.dp = float(ns), float(dp)
        self.pad_mode = L.ZS_PAD_REFLECT if seg_len >= 64 else L.ZS_PAD_ZERO     # model/model.py:38
        if c_h1 % 8 or c_h2 % 8:
            raise ValueError('c_h1 and c_h2 must be multiples of 8 (16-byte rows)')
        mk = lambda n, **kw: ConvLayer(ctx, P[n + '.weight'], P[n + '.bias'], G[n + '.weight'], G[n + '.bias'],
                                       pad_mode=self.pad_mode, name=n, **kw)
        self.conv1s = [mk('conv1s.%d' % i) for i in range(7)]
        self.conv2 = mk('conv2')
        self.convs = [(mk('conv%d' % a), mk('conv%d' % b, stride=2)) for a, b in ((3, 4), (5, 6), (7, 8))]
        self.dense = [(mk('dense%d' % a), mk('dense%d' % b)) for a, b in ((1, 2), (3, 4))]
        self.gru = GruLayer(ctx, P, G, 'RNN.', name='enc%d' % self.uid)
        self.linear = mk('linear')
        self.layers = self.conv1s + [self.conv2] + [l for p in self.convs for l in p] + [l for p in self.dense for l in p] + \
            [self.linear]
        self.ncat = 7 * c_h1 + c_in
        self.tape = None

    def pack(self):
        with L.pack_batch(self.ctx.stream):
            for l in self.layers:
                l.pack()
            self.gru.pack()

    # ------------------------------------------------------------------------------------------
    def forward(self, x, training, noise=None, noise_kind=2, seed=0, drop_masks=None, seed_ptr=None, lengths=None):
        """x: fp32 [B, T, c_in] contiguous on device.  Returns (bits Act [B,T',E] (T dtype, feeds the decoder),
        bits_f32 [B,T',E], logits_f32 [B,T',ld]).  noise: fp32 [B,T',E,2] (Gumbel if noise_kind 0, uniform if 1).
        lengths: host int sequence [B], frames per sample (ragged batch, inference only): sample b is computed exactly as if it ran
        alone with T = lengths[b] (padding, InstanceNorm statistics and the reverse GRU at its own end); its first
        ceil(ceil(ceil(lengths[b]/2)/2)/2) output rows are valid, the rest is garbage."""
        c, ns = self.ctx, self.ns
        B, T, F = x.shape
        Ls = [None] * 4
        if lengths is not None:
            if training:
                raise L.ZsError('EncoderEngine: ragged batches are an inference feature')
            l0 = torch.as_tensor(lengths, dtype=torch.int32).reshape(-1)
            if l0.numel() != B or int(l0.max()) > T or int(l0.min()) < 9:
                raise L.ZsError('EncoderEngine: lengths must be %d values in [9, %d]' % (B, T))     # MIN_LEN (convert.py:36)
            l1 = (l0 + 1) // 2; l2 = (l1 + 1) // 2; l3 = (l2 + 1) // 2
            self.last_lengths = l3
            ld = torch.stack([l0, l1, l2, l3]).to(c.device, non_blocking=True)
            Ls = [ld[i] for i in range(4)]
        assert F == self.c_in and x.dtype == torch.float32 and x.is_contiguous()
        st = c.stream
        c1, c2, H, E = self.c1, self.c2, self.H, self.E
        dp = self.dp if training else 0.0
        masks = drop_masks if drop_masks is not None else [None] * 6
        Ts = [T, _half_up(T), _half_up(_half_up(T)), _half_up(_half_up(_half_up(T)))]
        T4 = Ts[3]
        tag = '_%d_%d_%d' % (self.uid, B, T)
        tp = {'B': B, 'T': T, 'Ts': Ts, 'training': training, 'seed': seed, 'masks': masks, 'dp': dp, 'seed_ptr': seed_ptr}

        xin = c.act('e_xin' + tag, B, T, F)
        cat = c.act('e_cat' + tag, B, T, self.ncat)
        L.call('zs_cast_rows', 'ZsCastRows', st, dtype=c.dt, src=L.ptr(x), ld_src=F, src_f32=1, dst=xin.ptr(), ld_dst=xin.ld,
               dst_f32=0, col_off=0, rows=B * T, cols=F, fill_cols=xin.ld, act=L.ZS_ACT_NONE,
               dst2=cat.ptr(), ld_dst2=cat.ld, col_off2=7 * c1, fill_cols2=cat.ld - 7 * c1, act2=LRELU, slope2=ns)    # :445-446
        if c.overlap_wgrad and B * T >= 4096:
            # the seven bank convs are independent and each fills half the chip at most (N = 128): run them side by side
            sts = fork_side(c.device)
            for i, l in enumerate(self.conv1s):                                                               # :441-444
                with torch.cuda.stream(sts[i % len(sts)]):
                    l.fwd(xin, out=cat.sub(i * c1, c1), act=LRELU, slope=ns, lengths=Ls[0])
            join_side(c.device)
        else:
            for i, l in enumerate(self.conv1s):
                l.fwd(xin, out=cat.sub(i * c1, c1), act=LRELU, slope=ns, lengths=Ls[0])
        y2 = c.act('e_y2' + tag, B, T, c2)
        self.conv2.fwd(cat, out=y2, act=LRELU, slope=ns)                                                      # :447
        a = c.act('e_a0' + tag, B, T, c2)
        st0 = self._in(y2, a, tp, 0, res_mode=L.ZS_RES_NONE, lengths=Ls[0])
        tp.update(xin=xin, cat=cat, y2=y2, a0=a, st0=st0, blocks=[], dense=[])
        for i, (la, lb) in enumerate(self.convs):                                                            # :448-450
            ya = c.act('e_ya%d' % i + tag, B, Ts[i], c2)
            la.fwd(a, out=ya, act=LRELU, slope=ns, lengths=Ls[i])
            yb = c.act('e_yb%d' % i + tag, B, Ts[i + 1], c2)
            lb.fwd(ya, out=yb, act=LRELU, slope=ns, lengths=Ls[i])
            an = c.act('e_a%d' % (i + 1) + tag, B, Ts[i + 1], c2)
            stt = self._in(yb, an, tp, 1 + i, res_mode=L.ZS_RES_AVGPOOL2, res=a, lengths=Ls[i + 1], res_lengths=Ls[i])
            tp['blocks'].append((a, ya, yb, stt))
            a = an
        cat2 = c.act('e_cat2' + tag, B, T4, c2 + 2 * H)
        for j, (la, lb) in enumerate(self.dense):                                                            # :452-453
            d1 = c.act('e_d1%d' % j + tag, B, T4, c2)
            la.fwd(a, out=d1, act=LRELU, slope=ns)
            d2 = c.act('e_d2%d' % j + tag, B, T4, c2)
            lb.fwd(d1, out=d2, act=LRELU, slope=ns)
            out = c.act('e_do%d' % j + tag, B, T4, c2) if j == 0 else Act(cat2.t, B, T4, c2, cat2.ld, 0, rup(c2, 32))
            stt = self._in(d2, out, tp, 4 + j, res_mode=L.ZS_RES_IDENTITY, res=a, lengths=Ls[3])
            tp['dense'].append((a, d1, d2, stt))
            a = out
        gi = c.act('e_gi' + tag, B, T4, 6 * H)
        gates = c.raw('e_gates' + tag, B * T4 * 8 * H, c.tdt) if training else None
        self.gru.fwd(a, cat2, c2, gi, gates, lengths=Ls[3])                                                 # :454-455
        logits = c.act('e_logits' + tag, B, T4, 2 * E, dtype=torch.float32)
        self.linear.fwd(cat2, out=logits, out_f32=True)                                                      # :475
        bits = c.act('e_bits' + tag, B, T4, E)
        bits_f32 = c.f32('e_bitsf' + tag, B * T4 * E)
        y0 = c.f32('e_y0' + tag, B * T4 * E)
        L.call('zs_mbv_fwd', 'ZsMbvFwd', st, dtype=c.dt, logits=logits.ptr(), ld=logits.ld, logits_f32=1,
               noise=L.ptr(noise), noise_kind=noise_kind, seed=seed, seed_ptr=seed_ptr, rows=B * T4, E=E, tau=0.1, bits=bits.ptr(),
               ld_bits=bits.ld, bits_fill_cols=bits.ld, bits_f32=L.ptr(bits_f32), y0=L.ptr(y0))               # :476-480
        tp.update(cat2=cat2, gates=gates, y0=y0, T4=T4, gru_in=a, logits=logits)
        self.tape = tp
        return bits, bits_f32[:B * T4 * E].view(B, T4, E), logits

    def _in(self, x, out, tp, k, res_mode, res=None, lengths=None, res_lengths=None):
        """InstanceNorm + Dropout + residual (model/model.py:421-427, 434-437)."""
        c = self.ctx
        B, T, C = x.B, x.T, x.ld
        mean = rstd = None
        if tp['training']:
            mean = c.f32('e_mean%d_%d_%d_%d' % (self.uid, k, B, T), B * C)
            rstd = c.f32('e_rstd%d_%d_%d_%d' % (self.uid, k, B, T), B * C)
        m = tp['masks'][k]
        L.call('zs_instnorm_fwd', 'ZsInstNormFwd', c.stream, dtype=c.dt, x=x.ptr(), ldx=x.ld, out=out.ptr(), ldo=out.ld,
               mean=L.ptr(mean), rstd=L.ptr(rstd), B=B, T=T, C=C, eps=EPS_IN, drop_p=tp['dp'], seed=tp['seed'],
               stream_id=k + 1, mask=L.ptr(m), mask_ld=(m.shape[-1] if m is not None else 0), res_mode=res_mode,
               res=(res.ptr() if res is not None else None), ldres=(res.ld if res is not None else 0),
               T_res=(res.T if res is not None else 0), res_pad_mode=self.pad_mode, seed_ptr=tp.get('seed_ptr'),
               lengths=L.ptr(lengths), res_lengths=L.ptr(res_lengths))
        return (mean, rstd, k)

    def _in_bwd(self, dout, x, stt, dz, tp):
        c = self.ctx
        mean, rstd, k = stt
        m = tp['masks'][k]
        L.call('zs_instnorm_bwd', 'ZsInstNormBwd', c.stream, dtype=c.dt, dout=dout.ptr(), ldd=dout.ld, x=x.ptr(), ldx=x.ld,
               mean=L.ptr(mean), rstd=L.ptr(rstd), dz=dz.ptr(), ldz=dz.ld, B=x.B, T=x.T, C=x.ld, drop_p=tp['dp'],
               seed=tp['seed'], stream_id=k + 1, mask=L.ptr(m), mask_ld=(m.shape[-1] if m is not None else 0), slope=self.ns,
               seed_ptr=tp.get('seed_ptr'))

    # ------------------------------------------------------------------------------------------
    def backward(self, dbits, dlogits_extra=None):
        """dbits: Act [B,T',E] gradient w.r.t. enc_act (may be None); dlogits_extra: Act [B,T',2E] gradient w.r.t. the
        pre-activation `enc` (speaker-classifier term of trainer.py:444).  Fills every encoder parameter gradient."""
        self.backward_main(dbits, dlogits_extra)
        self.backward_bank()

    def backward_main(self, dbits, dlogits_extra=None):
        """Everything but the conv bank's seven weight gradients (the last gradients of the step: data parallel training starts
        the all-reduce of the rest of the encoder before them, trainer.AEStep)."""
        c, ns, tp = self.ctx, self.ns, self.tape
        assert tp is not None and tp['training']
        B, T, Ts, T4 = tp['B'], tp['T'], tp['Ts'], tp['T4']
        c1, c2, H, E = self.c1, self.c2, self.H, self.E
        st = c.stream
        tag = '_%d_%d_%d' % (self.uid, B, T)
        cat2 = tp['cat2']
        dlog = c.act('e_dlog' + tag, B, T4, 2 * E)
        if dbits is not None:
            L.call('zs_mbv_bwd', 'ZsMbvBwd', st, dtype=c.dt, dbits=dbits.ptr(), ld_dbits=dbits.ld, y0=L.ptr(tp['y0']),
                   rows=B * T4, E=E, tau=0.1, dlogits=dlog.ptr(), ld=dlog.ld, fill_cols=dlog.ld)
            if dlogits_extra is not None:
                dlog.t[:B * T4 * dlog.ld].add_(dlogits_extra.t[:B * T4 * dlogits_extra.ld])
        else:
            dlog = dlogits_extra
        self.linear.wgrad(dlog, cat2)
        dcat2 = c.act('e_dcat2' + tag, B, T4, cat2.C)
        self.linear.dgrad(dlog, T4, dcat2)
        dgi = c.act('e_dgi' + tag, B, T4, 6 * H)
        dgh = c.act('e_dgh' + tag, B, T4, 6 * H)
        gin = tp['gru_in']
        da = c.act('e_da_d' + tag, B, T4, c2)
        self.gru.bwd(dcat2, c2, cat2, c2, tp['gates'], gin, dgi, dgh, da, add_src=dcat2)
        for j in (1, 0):
            la, lb = self.dense[j]
            xin, d1, d2, stt = tp['dense'][j]
            dz2 = c.act('e_dz2%d' % j + tag, B, T4, c2)          # unique per block: read later by the side-stream wgrad
            self._in_bwd(da, d2, stt, dz2, tp)
            lb.wgrad(dz2, d1)
            dz1 = c.act('e_dz1%d' % j + tag, B, T4, c2)
            lb.dgrad(dz2, T4, dz1, dact_src=d1, slope=ns)
            la.wgrad(dz1, xin)
            dn = c.act('e_da_d%d' % j + tag, B, T4, c2)
            la.dgrad(dz1, T4, dn, add_src=da)
            da = dn
        for i in (2, 1, 0):
            la, lb = self.convs[i]
            a, ya, yb, stt = tp['blocks'][i]
            dzb = c.act('e_dzb%d' % i + tag, B, Ts[i + 1], c2)
            self._in_bwd(da, yb, stt, dzb, tp)
            lb.wgrad(dzb, ya)
            gp = c.act('e_gp%d' % i + tag, B, Ts[i] + 4, c2)
            lb.dgrad(dzb, Ts[i], gp)
            dza = c.act('e_dza%d' % i + tag, B, Ts[i], c2)
            self._combine(gp, Ts[i], 2, 2, dza, dact=ya)
            la.wgrad(dza, a)
            la.dgrad(dza, Ts[i], gp)
            dn = c.act('e_da%d' % i + tag, B, Ts[i], c2)
            self._combine(gp, Ts[i], 2, 2, dn, res_mode=L.ZS_RES_AVGPOOL2, res=da)
            da = dn
        y2, cat, xin = tp['y2'], tp['cat'], tp['xin']
        dz = c.act('e_dzy2' + tag, B, T, c2)
        self._in_bwd(da, y2, tp['st0'], dz, tp)
        self.conv2.wgrad(dz, cat)
        dcat = c.act('e_dcat' + tag, B, T, self.ncat)
        # only the conv bank's 7*c_h1 columns are consumed: the pass-through columns of the concatenation are the gradient w.r.t.
        # the input spectrogram, which the reference computes and drops (utils.py:43-45) -- a third of this GEMM
        self.conv2.dgrad(dz, T, dcat, dact_src=cat, slope=ns, n_cols=7 * c1)
        self._bank = (dcat, xin)

    def backward_bank(self):
        dcat, xin = self._bank
        for i, l in enumerate(self.conv1s):
            l.wgrad(dcat.sub(i * self.c1, self.c1), xin)

    def _combine(self, gp, T, pl, pr, out, res_mode=L.ZS_RES_NONE, res=None, dact=None):
        c = self.ctx
        L.call('zs_grad_combine', 'ZsGradCombine', c.stream, dtype=c.dt, gp=gp.ptr(), ldg=gp.ld, pad_left=pl, pad_right=pr,
               pad_mode=self.pad_mode, B=gp.B, T=T, C=gp.ld, res_mode=res_mode, res=(res.ptr() if res is not None else None),
               ldres=(res.ld if res is not None else 0), dact_src=(dact.ptr() if dact is not None else None),
               dact_ld=(dact.ld if dact is not None else 0), slope=self.ns, out=out.ptr(), ldo=out.ld, unshuffle=0)


class DecoderEngine(object):
    def __init__(self, ctx, P, G, c_in, c_out, c_h, c_a, ns, seg_len, output_mask=False):
        self.ctx = ctx
        self.uid = _uid()
        self.E, self.F, self.ch, self.n_spk = c_in, c_out, c_h, c_a
        self.ns = float(ns)
        self.pad_mode = L.ZS_PAD_REFLECT if seg_len >= 64 else L.ZS_PAD_ZERO
        self.out_act = L.ZS_ACT_TANH if output_mask else L.ZS_ACT_SIGMOID
        if c_h % 32:
            raise ValueError('Decoder c_h must be a multiple of 32 (got %d)' % c_h)
        mk = lambda n, **kw: ConvLayer(ctx, P[n + '.weight'], P[n + '.bias'], G[n + '.weight'], G[n + '.bias'],
                                       pad_mode=self.pad_mode, name=n, **kw)
        self.input_emb = mk('input_emb')
        self.convs = [(mk('conv%d' % a, split2=True), mk('conv%d' % b)) for a, b in ((1, 2), (3, 4), (5, 6))]
        self.dense = [(mk('dense%d' % a), mk('dense%d' % b)) for a, b in ((1, 2), (3, 4))]
        self.gru = GruLayer(ctx, P, G, 'RNN.', name='dec%d' % self.uid)
        self.dense5 = mk('dense5')                        # the literal cat[out, rnn, emb5 x T] GEMM (K = 3 c_h), model/model.py:357-358
        self.linear = mk('linear')
        self.emb = [P['emb%d.weight' % i] for i in range(1, 6)]
        self.gemb = [G['emb%d.weight' % i] for i in range(1, 6)]
        self.layers = [self.input_emb] + [l for p in self.convs for l in p] + [l for p in self.dense for l in p] + \
            [self.dense5, self.linear]
        self.tape = None

    def pack(self):
        with L.pack_batch(self.ctx.stream):
            for l in self.layers:
                l.pack()
            self.gru.pack()

    def forward(self, bits, cidx, training, lengths=None):
        """bits: Act [B,T',E] (T dtype, zero padded); cidx int64 [B].  Returns x_dec Act fp32 [B, 8T', F].
        lengths: host int sequence [B], encoded frames per sample (ragged batch, inference only): sample b is computed as if it
        ran alone with T' = lengths[b]; its first 8 * lengths[b] output rows are valid."""
        c, ns, ch = self.ctx, self.ns, self.ch
        B, T0 = bits.B, bits.T
        Ls = [None] * 4
        if lengths is not None:
            if training:
                raise L.ZsError('DecoderEngine: ragged batches are an inference feature')
            l0 = torch.as_tensor(lengths, dtype=torch.int32).reshape(-1)
            if l0.numel() != B or int(l0.max()) > T0 or int(l0.min()) < 2:
                raise L.ZsError('DecoderEngine: lengths must be %d values in [2, %d]' % (B, T0))    # reflect pad 1 of conv k3 needs 2 rows
            ld = torch.stack([l0, 2 * l0, 4 * l0, 8 * l0]).to(c.device, non_blocking=True)
            Ls = [ld[i] for i in range(4)]
        tag = '_%d_%d_%d' % (self.uid, B, T0)
        st = c.stream
        emb = self.emb
        tp = {'B': B, 'T0': T0, 'training': training, 'cidx': cidx, 'bits': bits, 'blocks': [], 'dense': []}
        x = c.act('d_x0' + tag, B, T0, ch)
        xe = c.act('d_xe0' + tag, B, T0, ch)
        self.input_emb.fwd(bits, out=x, out2=xe, vec2=emb[0], idx=cidx)                                   # :346, :319
        T = T0
        for i, (la, lb) in enumerate(self.convs):                                                        # :317-331
            ya = c.act('d_ya%d' % i + tag, B, T, 2 * ch) if training else None
            s = c.act('d_s%d' % i + tag, B, 2 * T, ch)
            la.fwd(xe, out=ya, act=LRELU, slope=ns, out2=s, vec2=emb[i], idx=cidx, store_mode2=L.ZS_STORE_SPLIT2, lengths=Ls[i])
            yb = c.act('d_yb%d' % i + tag, B, 2 * T, ch)
            lb.fwd(s, out=yb, act=LRELU, slope=ns, lengths=Ls[i + 1])
            xn = c.act('d_x%d' % (i + 1) + tag, B, 2 * T, ch)
            xen = c.act('d_xe%d' % (i + 1) + tag, B, 2 * T, ch)
            nxt = emb[i + 1] if i < 2 else emb[3]                                                        # emb2, emb3, then emb4 (:350)
            stt = self._in(yb, xn, xen, nxt, cidx, L.ZS_RES_UPSAMPLE2, x, training, 'c%d' % i, lengths=Ls[i + 1])
            tp['blocks'].append((x, xe, ya, s, yb, stt, T))
            x, xe, T = xn, xen, 2 * T
        cat3 = c.act('d_cat3' + tag, B, T, 3 * ch)
        for j, (la, lb) in enumerate(self.dense):                                                        # :333-342, :350-351
            y1 = c.act('d_y1%d' % j + tag, B, T, ch)
            y1e = c.act('d_y1e%d' % j + tag, B, T, ch)
            la.fwd(xe, out=y1, act=LRELU, slope=ns, out2=y1e, vec2=emb[3], idx=cidx)
            y2 = c.act('d_y2%d' % j + tag, B, T, ch)
            lb.fwd(y1e, out=y2, act=LRELU, slope=ns)
            xn = c.act('d_dx%d' % j + tag, B, T, ch) if j == 0 else Act(cat3.t, B, T, ch, cat3.ld, 0, ch)
            xen = c.act('d_dxe%d' % j + tag, B, T, ch)
            stt = self._in(y2, xn, xen, emb[3] if j == 0 else emb[4], cidx, L.ZS_RES_IDENTITY, x, training, 'd%d' % j, lengths=Ls[3])
            tp['dense'].append((x, xe, y1, y1e, y2, stt))
            x, xe = xn, xen
        H = ch // 2
        gi = c.act('d_gi' + tag, B, T, 6 * H)
        gates = c.raw('d_gates' + tag, B * T * 8 * H, c.tdt) if training else None
        # :352-356, and append_emb (:357) rides on the recurrence's stores (third block of cat3 = emb5[c_b] at every t)
        self.gru.fwd(xe, cat3, ch, gi, gates, bcast=(emb[4], cidx, 2 * ch), lengths=Ls[3])
        h5 = c.act('d_h5' + tag, B, T, ch)
        self.dense5.fwd(cat3, out=h5, act=LRELU, slope=ns)                                                # :358-359
        return self._fwd_tail(c, tp, h5, cat3, gates, xe, T, tag, B)

    def _fwd_tail(self, c, tp, h5, cat3, gates, xe, T, tag, B):
        xdec = c.act('d_xdec' + tag, B, T, self.F, dtype=torch.float32)
        self.linear.fwd(h5, out=xdec, act=self.out_act, out_f32=True)                                    # :360-364
        tp.update(cat3=cat3, gates=gates, gru_in=xe, h5=h5, T=T)
        self.tape = tp
        return xdec

    def _in(self, x, out, out2, vec2, cidx, res_mode, res, training, key, lengths=None):
        c = self.ctx
        B, T, C = x.B, x.T, x.ld
        mean = rstd = None
        if training:
            mean = c.f32('d_mean%d_%s_%d_%d' % (self.uid, key, B, T), B * C)
            rstd = c.f32('d_rstd%d_%s_%d_%d' % (self.uid, key, B, T), B * C)
        L.call('zs_instnorm_fwd', 'ZsInstNormFwd', c.stream, dtype=c.dt, x=x.ptr(), ldx=x.ld, out=out.ptr(), ldo=out.ld,
               out2=out2.ptr(), ldo2=out2.ld, vec2=L.ptr(vec2), vec2_ld=vec2.shape[1], vec2_cols=vec2.shape[1], idx=L.ptr(cidx),
               mean=L.ptr(mean), rstd=L.ptr(rstd), B=B, T=T, C=C, eps=EPS_IN, drop_p=0.0, res_mode=res_mode, res=res.ptr(),
               ldres=res.ld, T_res=res.T, res_pad_mode=self.pad_mode, lengths=L.ptr(lengths))
        return (mean, rstd)

    def _in_bwd(self, dout, x, stt, dz):
        c = self.ctx
        L.call('zs_instnorm_bwd', 'ZsInstNormBwd', c.stream, dtype=c.dt, dout=dout.ptr(), ldd=dout.ld, x=x.ptr(), ldx=x.ld,
               mean=L.ptr(stt[0]), rstd=L.ptr(stt[1]), dz=dz.ptr(), ldz=dz.ld, B=x.B, T=x.T, C=x.ld, drop_p=0.0, slope=self.ns)

    def _combine(self, gp, T, pl, pr, out, emb_i=None, res_mode=L.ZS_RES_NONE, res=None, dact=None, unshuffle=0, C=None):
        c = self.ctx
        esum = None
        if emb_i is not None:                      # per-sample sums for embedding emb_i (finished by zs_emb_scatter)
            esum = L.ptr(self._embsum, emb_i * gp.B * self.ch)
        L.call('zs_grad_combine', 'ZsGradCombine', c.stream, dtype=c.dt, gp=gp.ptr(), ldg=gp.ld, pad_left=pl, pad_right=pr,
               pad_mode=self.pad_mode, B=gp.B, T=T, C=(C if C is not None else self.ch),
               emb_sum=esum, emb_ld=self.ch, emb_cols=self.ch, res_mode=res_mode,
               res=(res.ptr() if res is not None else None), ldres=(res.ld if res is not None else 0),
               dact_src=(dact.ptr() if dact is not None else None), dact_ld=(dact.ld if dact is not None else 0),
               slope=self.ns, out=(out.ptr() if out is not None else None), ldo=(out.ld if out is not None else 0),
               unshuffle=unshuffle)

    def backward(self, dlogit, need_dbits=True):
        """dlogit: Act [B,T,F] gradient w.r.t. the pre-sigmoid output.  Fills (overwrites) every decoder parameter
        gradient.  Returns dbits Act [B,T',E]."""
        self.backward_head(dlogit)
        return self.backward_convs(need_dbits)

    def backward_head(self, dlogit):
        """linear, dense5, the GRU and the two dense blocks: the gradients of dense1 .. linear (a contiguous range of the flat
        gradient buffer) are final when this returns and the side streams have been joined -- data parallel training starts their
        all-reduce here, under the conv blocks' backward (trainer.AEStep)."""
        c, ns, ch, tp = self.ctx, self.ns, self.ch, self.tape
        assert tp is not None and tp['training']
        B, T0, T = tp['B'], tp['T0'], tp['T']
        tag = '_%d_%d_%d' % (self.uid, B, T0)
        cat3, h5 = tp['cat3'], tp['h5']
        H = ch // 2
        self._embsum = c.f32('d_embsum' + tag, 6 * B * ch)      # slots 0..4: per-sample sums for emb1..emb5; 5: column sums of dz5
        self._embsum.zero_()
        self.linear.wgrad(dlogit, h5)
        dz5 = c.act('d_dz5' + tag, B, T, ch)
        # the embedding parts of the k = 1 data gradients (per-sample column sums) come out of the GEMM epilogues when whole
        # samples fit a tile (ZsGemmConv.colsum); otherwise a zs_grad_combine pass over the gradient computes them
        fuse = colsum_ok(T) and os.environ.get('ZS_FUSE_COLSUM', '1') == '1'
        slot = lambda k, col0=0: (L.ptr(self._embsum, k * B * ch), ch, col0)
        # append_emb's block of dense5 (model/model.py:357) multiplies a vector that is constant over time, so its input gradient
        # is only needed summed over t: sum_t (dz5[b,t] W5e) = (sum_t dz5[b,t]) W5e -- the column sums of dz5 (epilogue of the
        # GEMM that produces it) times the block, a B-row GEMM, instead of a third of dense5's data gradient
        thin5 = fuse and ch % 128 == 0 and os.environ.get('ZS_THIN_EMB5', '1') == '1'
        self.linear.dgrad(dlogit, T, dz5, dact_src=h5, slope=ns, colsum=(slot(5) if thin5 else None), colsum_post=True)
        self.dense5.wgrad(dz5, cat3)
        dcat3 = c.act('d_dcat3' + tag, B, T, 3 * ch)
        if thin5:
            self.dense5.dgrad(dz5, T, dcat3, n_cols=2 * ch)
            s32 = Act(self._embsum, B, 1, ch, ch, 5 * B * ch)
            sT = c.act('d_s5' + tag, B, 1, ch)
            L.call('zs_cast_rows', 'ZsCastRows', c.stream, dtype=c.dt, src=s32.ptr(), ld_src=ch, src_f32=1, dst=sT.ptr(), ld_dst=sT.ld,
                   dst_f32=0, col_off=0, rows=B, cols=ch, fill_cols=sT.ld, act=L.ZS_ACT_NONE)
            slot4 = Act(self._embsum, B, 1, ch, ch, 4 * B * ch)
            self.dense5.dgrad(sT, 1, slot4, add_src=slot4, out_f32=True, add_f32=True, n_cols=ch, n_off=2 * ch)
        elif fuse:                                                                           # d emb5 via append_emb: the third
            self.dense5.dgrad(dz5, T, dcat3, colsum=slot(4, 2 * ch), out_cols=2 * ch)        # column block is summed, not stored
        else:
            self.dense5.dgrad(dz5, T, dcat3)
            self._combine(dcat3.sub(2 * ch, ch), T, 0, 0, None, emb_i=4)
        dgi = c.act('d_dgi' + tag, B, T, 6 * H)
        dgh = c.act('d_dgh' + tag, B, T, 6 * H)
        gp = c.act('d_gpA' + tag, B, T + 2, ch)
        gpv = Act(gp.t, B, T, ch, gp.ld)
        dx = c.act('d_dxA' + tag, B, T, ch)
        if fuse:                                                                                     # out+emb5 (:353)
            self.gru.bwd(dcat3, ch, cat3, ch, tp['gates'], tp['gru_in'], dgi, dgh, dx, add_src=dcat3.sub(0, ch), colsum=slot(4))
        else:
            self.gru.bwd(dcat3, ch, cat3, ch, tp['gates'], tp['gru_in'], dgi, dgh, gpv)
            self._combine(gpv, T, 0, 0, dx, emb_i=4, res_mode=L.ZS_RES_IDENTITY, res=dcat3.sub(0, ch))
        for j in (1, 0):
            la, lb = self.dense[j]
            xin, xe, y1, y1e, y2, stt = tp['dense'][j]
            dz2 = c.act('d_dz2%d' % j + tag, B, T, ch)           # unique per block: read later by the side-stream wgrad
            self._in_bwd(dx, y2, stt, dz2)
            lb.wgrad(dz2, y1e)
            dz1 = c.act('d_dz1%d' % j + tag, B, T, ch)
            if fuse:
                lb.dgrad(dz2, T, dz1, dact_src=y1, slope=ns, colsum=slot(3))
            else:
                lb.dgrad(dz2, T, gpv)
                self._combine(gpv, T, 0, 0, dz1, emb_i=3, dact=y1)
            la.wgrad(dz1, xe)
            dn = c.act('d_dxD%d' % j + tag, B, T, ch)
            if fuse:
                la.dgrad(dz1, T, dn, add_src=dx, colsum=slot(3))
            else:
                la.dgrad(dz1, T, gpv)
                self._combine(gpv, T, 0, 0, dn, emb_i=3, res_mode=L.ZS_RES_IDENTITY, res=dx)
            dx = dn
        self._bw = (dx, gp, tag)

    def backward_convs(self, need_dbits=True):
        """The three conv blocks, the embeddings and input_emb.  Returns dbits."""
        c, ns, ch, tp = self.ctx, self.ns, self.ch, self.tape
        dx, gp, tag = self._bw
        B, T0 = tp['B'], tp['T0']
        for i in (2, 1, 0):
            la, lb = self.convs[i]
            xin, xe, ya, s, yb, stt, Ti = tp['blocks'][i]
            T2 = 2 * Ti
            dzb = c.act('d_dzb%d' % i + tag, B, T2, ch)
            self._in_bwd(dx, yb, stt, dzb)
            lb.wgrad(dzb, s)
            gp2 = Act(gp.t, B, T2 + 2, ch, gp.ld)
            lb.dgrad(dzb, T2, gp2)
            dza = c.act('d_dza%d' % i + tag, B, Ti, 2 * ch)
            self._combine(gp2, T2, 1, 1, dza, emb_i=i, dact=ya, unshuffle=1)                # pixel_shuffle^T, +emb (:323-324)
            la.wgrad(dza, xe)
            gp1 = Act(gp.t, B, Ti + 2, ch, gp.ld)
            la.dgrad(dza, Ti, gp1)
            dn = c.act('d_dxC%d' % i + tag, B, Ti, ch)
            self._combine(gp1, Ti, 1, 1, dn, emb_i=i, res_mode=L.ZS_RES_UPSAMPLE2, res=dx)   # x+emb (:319) and upsample(x) (:329)
            dx = dn
        # nn.Embedding backward, fixed sample order.  Only the optimizer needs these: off the main chain, side by side
        sts = fork_side(c.device) if c.overlap_wgrad else None
        for k in range(5):
            st_k = sts[k % len(sts)].cuda_stream if sts is not None else c.stream
            L.call('zs_emb_scatter', 'ZsEmbScatter', st_k, emb_sum=L.ptr(self._embsum, k * B * ch), emb_ld=ch,
                   idx=L.ptr(tp['cidx']), B=B, demb=L.ptr(self.gemb[k]), demb_ld=ch, n_rows=self.n_spk, C=ch, accumulate=0)
        bits = tp['bits']
        self.input_emb.wgrad(dx, bits)
        dbits = None
        if need_dbits:
            dbits = c.act('d_dbits' + tag, B, T0, self.E)
            self.input_emb.dgrad(dx, T0, dbits)
        return dbits


class ClassifierEngine(object):
    """SpeakerClassifier.forward (reference model/model.py:262-280) and its backward: four conv blocks on the
    T' = T/8 code sequence (conv k5/k3 + lrelu, InstanceNorm, Dropout, identity residual on blocks 2 and 3) and a
    final un-padded Conv1d whose kernel spans the whole sequence (k = seg_len/8) -> logits [B, n_class]."""

    def __init__(self, ctx, P, G, c_in, c_h, n_class, dp, ns, seg_len):
        self.ctx, self.uid = ctx, _uid()
        self.c_in, self.c_h, self.n_class = c_in, c_h, n_class
        self.ns, self.dp = float(ns), float(dp)
        self.pad_mode = L.ZS_PAD_REFLECT if seg_len >= 64 else L.ZS_PAD_ZERO
        mk = lambda n, **kw: ConvLayer(ctx, P[n + '.weight'], P[n + '.bias'], G[n + '.weight'], G[n + '.bias'],
                                       pad_mode=self.pad_mode, name=n, **kw)
        self.blocks = [(mk('conv1'), mk('conv2'), False), (mk('conv3'), mk('conv4'), True), (mk('conv5'), mk('conv6'), True),
                       (mk('conv7'), mk('conv8'), False)]
        self.conv9 = mk('conv9', padded=False)
        self.layers = [l for b in self.blocks for l in b[:2]] + [self.conv9]
        self.tape = None

    def pack(self):
        with L.pack_batch(self.ctx.stream):
            for l in self.layers:
                l.pack()

    def forward(self, x, training, seed=0, drop_masks=None):
        """x: Act [B, T', c_in] in the compute dtype.  Returns logits Act fp32 [B, 1, n_class]."""
        c, ns = self.ctx, self.ns
        B, T = x.B, x.T
        if T != self.conv9.k:
            raise L.ZsError('SpeakerClassifier expects %d encoded frames (got %d)' % (self.conv9.k, T))
        tag = '_%d_%d_%d' % (self.uid, B, T)
        dp = self.dp if training else 0.0
        masks = drop_masks if drop_masks is not None else [None] * 4
        tp = {'B': B, 'T': T, 'training': training, 'seed': seed, 'masks': masks, 'dp': dp, 'x': x, 'blocks': []}
        a = x
        for i, (la, lb, res) in enumerate(self.blocks):
            ya = c.act('c_ya%d' % i + tag, B, T, la.Cout)
            la.fwd(a, out=ya, act=LRELU, slope=ns)
            yb = c.act('c_yb%d' % i + tag, B, T, lb.Cout)
            lb.fwd(ya, out=yb, act=LRELU, slope=ns)
            out = c.act('c_o%d' % i + tag, B, T, lb.Cout)
            C = yb.ld
            mean = rstd = None
            if training:
                mean, rstd = c.f32('c_mean%d' % i + tag, B * C), c.f32('c_rstd%d' % i + tag, B * C)
            m = masks[i]
            L.call('zs_instnorm_fwd', 'ZsInstNormFwd', c.stream, dtype=c.dt, x=yb.ptr(), ldx=yb.ld, out=out.ptr(), ldo=out.ld,
                   mean=L.ptr(mean), rstd=L.ptr(rstd), B=B, T=T, C=C, eps=EPS_IN, drop_p=dp, seed=seed, stream_id=20 + i,
                   mask=L.ptr(m), mask_ld=(m.shape[-1] if m is not None else 0),
                   res_mode=(L.ZS_RES_IDENTITY if res else L.ZS_RES_NONE), res=(a.ptr() if res else None),
                   ldres=(a.ld if res else 0), T_res=(T if res else 0), res_pad_mode=self.pad_mode)
            tp['blocks'].append((a, ya, yb, (mean, rstd, i), res))
            a = out
        logits = c.act('c_logits' + tag, B, 1, self.n_class, dtype=torch.float32)
        self.conv9.fwd(a, out=logits, out_f32=True)
        tp['last'] = a
        self.tape = tp
        return logits

    def backward(self, dlogits_f32, ld, need_dx=True, param_grads=True):
        """dlogits_f32: fp32 tensor [B, ld] (from zs_softmax_ce).  Returns dx Act [B, T', c_in] (or None)."""
        c, ns, tp = self.ctx, self.ns, self.tape
        B, T = tp['B'], tp['T']
        tag = '_%d_%d_%d' % (self.uid, B, T)
        dl = c.act('c_dl' + tag, B, 1, self.n_class)
        L.call('zs_cast_rows', 'ZsCastRows', c.stream, dtype=c.dt, src=L.ptr(dlogits_f32), ld_src=ld, src_f32=1, dst=dl.ptr(),
               ld_dst=dl.ld, dst_f32=0, col_off=0, rows=B, cols=self.n_class, fill_cols=dl.ld, act=L.ZS_ACT_NONE)
        if param_grads:
            self.conv9.wgrad(dl, tp['last'])
        da = c.act('c_da_last' + tag, B, T, self.conv9.Cin)
        self.conv9.dgrad(dl, T, da)
        for i in (3, 2, 1, 0):
            la, lb, _ = self.blocks[i]
            a, ya, yb, stt, res = tp['blocks'][i]
            mean, rstd, k = stt
            m = tp['masks'][k]
            dzb = c.act('c_dzb%d' % i + tag, B, T, lb.Cout)
            L.call('zs_instnorm_bwd', 'ZsInstNormBwd', c.stream, dtype=c.dt, dout=da.ptr(), ldd=da.ld, x=yb.ptr(), ldx=yb.ld,
                   mean=L.ptr(mean), rstd=L.ptr(rstd), dz=dzb.ptr(), ldz=dzb.ld, B=B, T=T, C=yb.ld, drop_p=tp['dp'], seed=tp['seed'],
                   stream_id=20 + k, mask=L.ptr(m), mask_ld=(m.shape[-1] if m is not None else 0), slope=ns)
            if param_grads:
                lb.wgrad(dzb, ya)
            gp = c.act('c_gp%d' % i + tag, B, T + lb.pad_l + lb.pad_r, lb.Cin)
            lb.dgrad(dzb, T, gp)
            dza = c.act('c_dza%d' % i + tag, B, T, la.Cout)
            self._combine(gp, T, lb.pad_l, lb.pad_r, dza, dact=ya)
            if param_grads:
                la.wgrad(dza, a)
            if i == 0 and not need_dx:
                return None
            gp2 = c.act('c_gq%d' % i + tag, B, T + la.pad_l + la.pad_r, la.Cin)
            la.dgrad(dza, T, gp2)
            dn = c.act('c_da%d' % i + tag, B, T, la.Cin)
            self._combine(gp2, T, la.pad_l, la.pad_r, dn, res_mode=(L.ZS_RES_IDENTITY if res else L.ZS_RES_NONE), res=(da if res else None))
            da = dn
        return da

    def _combine(self, gp, T, pl, pr, out, res_mode=L.ZS_RES_NONE, res=None, dact=None):
        c = self.ctx
        L.call('zs_grad_combine', 'ZsGradCombine', c.stream, dtype=c.dt, gp=gp.ptr(), ldg=gp.ld, pad_left=pl, pad_right=pr,
               pad_mode=self.pad_mode, B=gp.B, T=T, C=gp.ld, res_mode=res_mode, res=(res.ptr() if res is not None else None),
               ldres=(res.ld if res is not None else 0), dact_src=(dact.ptr() if dact is not None else None),
               dact_ld=(dact.ld if dact is not None else 0), slope=self.ns, out=out.ptr(), ldo=out.ld, unshuffle=0)
