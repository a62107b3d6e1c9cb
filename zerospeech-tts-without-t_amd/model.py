"""Drop-in Encoder / Decoder / SpeakerClassifier modules (reference: model/model.py:231-489).

Same constructor arguments, same parameter names and shapes (so reference checkpoints load with
load_state_dict), same forward signatures and return tuples in the reference's [B, C, T] layout.
The torch.nn layers created here are parameter CONTAINERS only (they give identical initialisation and
state_dict keys); their forward is never called -- all arithmetic runs in libzs_amd.so through
zs_amd.engine.  All parameters of a module live in one flat fp32 buffer (`flat`), their gradients in
`gflat` (one RCCL all-reduce / one fused Adam launch per net).
"""
import os

import torch
import torch.nn as nn

from . import _lib as L
from .engine import ClassifierEngine, DecoderEngine, EncoderEngine
from .layers import Act, Ctx, rup


def default_dtype():
    return os.environ.get('ZS_DTYPE', 'fp32')


class ZsModule(nn.Module):
    """Flat-parameter storage + lazy engine construction."""

    def __init__(self, dtype=None):
        super(ZsModule, self).__init__()
        self._zs_dtype = dtype or default_dtype()
        self._zs = {'flat': None, 'gflat': None, 'engine': None, 'ctx': None, 'version': None, 'offsets': None}

    # -- flat storage --------------------------------------------------------------------------
    def _flatten(self):
        params = list(self.named_parameters())
        if not params:
            return
        dev = params[0][1].device
        offs, n = {}, 0
        for name, p in params:
            offs[name] = n
            n += rup(p.numel(), 4)
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        gflat = torch.zeros(n, dtype=torch.float32, device=dev)
        for name, p in params:
            o = offs[name]
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).float())
            p.data = flat[o:o + p.numel()].view(p.shape)
            p.grad = gflat[o:o + p.numel()].view(p.shape)
        self._zs.update(flat=flat, gflat=gflat, offsets=offs, engine=None, ctx=None, version=None)

    def _apply(self, fn, *a, **kw):
        out = super(ZsModule, self)._apply(fn, *a, **kw)
        self._flatten()
        return out

    def flat_params(self):
        if self._zs['flat'] is None:
            self._flatten()
        return self._zs['flat'], self._zs['gflat']

    def grad_view(self, name):
        o = self._zs['offsets'][name]
        p = dict(self.named_parameters())[name]
        return self._zs['gflat'][o:o + p.numel()].view(p.shape)

    def flat_range(self, first, last):
        """(lo, hi) element range of the flat parameter / gradient buffers that holds the parameters `first` .. `last`
        (registration order), e.g. a gradient bucket of the data-parallel all-reduce."""
        if self._zs['flat'] is None:
            self._flatten()
        offs = self._zs['offsets']
        p = dict(self.named_parameters())[last]
        lo, hi = offs[first], offs[last] + rup(p.numel(), 4)
        if lo >= hi:
            raise ValueError('flat_range: %s does not precede %s' % (first, last))
        return lo, hi

    def mark_dirty(self):
        self._zs['version'] = None

    def repack(self):
        """Re-pack the GEMM operands from the fp32 master weights now (stream-ordered) and mark them current."""
        z = self._zs
        if z['engine'] is None:
            self._engine()
        else:
            z['engine'].pack()
            z['version'] = sum(p._version for p in self.parameters())

    def set_compute_dtype(self, dtype):
        if dtype != self._zs_dtype:
            self._zs_dtype = dtype
            self._zs.update(engine=None, ctx=None, version=None)

    # -- engine --------------------------------------------------------------------------------
    def _engine(self):
        z = self._zs
        if z['flat'] is None:
            self._flatten()
        if z['engine'] is None:
            dev = z['flat'].device
            if dev.type != 'cuda':
                raise L.ZsError('zs_amd modules compute on an MI355X only: move the module to a cuda device '
                                '(there is no CPU fallback path)')
            z['ctx'] = Ctx(dev, self._zs_dtype)
            P = {k: p.data for k, p in self.named_parameters()}
            G = {k: self.grad_view(k) for k, _ in self.named_parameters()}
            z['engine'] = self._make_engine(z['ctx'], P, G)
            z['version'] = None
        ver = sum(p._version for p in self.parameters())
        if z['version'] != ver:
            z['engine'].pack()
            z['version'] = ver
        return z['engine']

    def load_state_dict(self, *a, **kw):
        out = super(ZsModule, self).load_state_dict(*a, **kw)
        self.mark_dirty()
        return out


class Encoder(ZsModule):
    """model/model.py:368-489 (enc_mode 'multilabel_binary' is the accelerated path)."""

    def __init__(self, c_in=513, c_h1=128, c_h2=512, c_h3=128, ns=0.2, dp=0.5, enc_size=512, seg_len=64,
                 enc_mode='continues', dtype=None):
        super(Encoder, self).__init__(dtype)
        self.ns, self.dp, self.enc_size, self.seg_len, self.enc_mode = ns, dp, enc_size, seg_len, enc_mode
        self.c_in, self.c_h1, self.c_h2, self.c_h3 = c_in, c_h1, c_h2, c_h3
        if enc_mode != 'multilabel_binary':
            if enc_mode in ('continues', 'one_hot', 'binary', 'gumbel_t'):
                raise NotImplementedError("enc_mode %r is not on the MI355X hot path (only 'multilabel_binary' is)" % enc_mode)
            raise NotImplementedError('Invalid encoding mode!')
        self.conv1s = nn.ModuleList([nn.Conv1d(c_in, c_h1, kernel_size=k) for k in range(1, 8)])
        self.conv2 = nn.Conv1d(len(self.conv1s) * c_h1 + c_in, c_h2, kernel_size=1)
        self.conv3 = nn.Conv1d(c_h2, c_h2, kernel_size=5)
        self.conv4 = nn.Conv1d(c_h2, c_h2, kernel_size=5, stride=2)
        self.conv5 = nn.Conv1d(c_h2, c_h2, kernel_size=5)
        self.conv6 = nn.Conv1d(c_h2, c_h2, kernel_size=5, stride=2)
        self.conv7 = nn.Conv1d(c_h2, c_h2, kernel_size=5)
        self.conv8 = nn.Conv1d(c_h2, c_h2, kernel_size=5, stride=2)
        self.dense1 = nn.Linear(c_h2, c_h2)
        self.dense2 = nn.Linear(c_h2, c_h2)
        self.dense3 = nn.Linear(c_h2, c_h2)
        self.dense4 = nn.Linear(c_h2, c_h2)
        self.RNN = nn.GRU(input_size=c_h2, hidden_size=c_h3, num_layers=1, bidirectional=True)
        self.linear = nn.Linear(c_h2 + 2 * c_h3, enc_size * 2)

    def _make_engine(self, ctx, P, G):
        return EncoderEngine(ctx, P, G, self.c_in, self.c_h1, self.c_h2, self.c_h3, self.enc_size, self.ns, self.dp, self.seg_len)

    def forward(self, x, U=None, G=None, drop_masks=None, seed=None, lengths=None):
        """x [B, c_in, T] -> (enc_act [B, E, T/8], enc [B, 2E, T/8]).  The Gumbel-softmax draws noise in
        eval mode too (reference behaviour, model/model.py:95-98): pass U (uniform) or G (Gumbel) to make
        it reproducible; otherwise a counter-hash stream seeded from torch's RNG is used.
        lengths (eval only): frames per sample of a ragged batch padded to T -- every sample is computed exactly as if it were
        forwarded alone with its own length (the reference forwards such fragments one at a time, convert.py:154-165); the
        first ceil(lengths[b] / 8) output frames of sample b are valid."""
        eng = self._engine()
        xb = x.detach().permute(0, 2, 1).contiguous().float()
        noise, kind = None, 2
        if G is not None:
            noise, kind = G.to(xb.device, torch.float32).contiguous(), 0
        elif U is not None:
            noise, kind = U.to(xb.device, torch.float32).contiguous(), 1
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        masks = None
        if drop_masks is not None:
            masks = [m.to(xb.device, torch.uint8).contiguous() if m is not None else None for m in drop_masks]
        bits, bits_f32, logits = eng.forward(xb, self.training, noise=noise, noise_kind=kind, seed=seed, drop_masks=masks, lengths=lengths)
        self._last_bits = bits
        enc_act = bits_f32.permute(0, 2, 1).clone()
        enc = logits.valid().permute(0, 2, 1).clone()
        return enc_act, enc


class Decoder(ZsModule):
    """model/model.py:283-365."""

    def __init__(self, c_in=512, c_out=513, c_h=512, c_a=8, ns=0.2, seg_len=64, output_mask=False, dtype=None):
        super(Decoder, self).__init__(dtype)
        self.output_mask, self.ns, self.seg_len = output_mask, ns, seg_len
        self.c_in, self.c_out, self.c_h, self.c_a = c_in, c_out, c_h, c_a
        self.conv1 = nn.Conv1d(c_h, 2 * c_h, kernel_size=3)
        self.conv2 = nn.Conv1d(c_h, c_h, kernel_size=3)
        self.conv3 = nn.Conv1d(c_h, 2 * c_h, kernel_size=3)
        self.conv4 = nn.Conv1d(c_h, c_h, kernel_size=3)
        self.conv5 = nn.Conv1d(c_h, 2 * c_h, kernel_size=3)
        self.conv6 = nn.Conv1d(c_h, c_h, kernel_size=3)
        self.dense1 = nn.Linear(c_h, c_h)
        self.dense2 = nn.Linear(c_h, c_h)
        self.dense3 = nn.Linear(c_h, c_h)
        self.dense4 = nn.Linear(c_h, c_h)
        self.RNN = nn.GRU(input_size=c_h, hidden_size=c_h // 2, num_layers=1, bidirectional=True)
        self.dense5 = nn.Linear(2 * c_h + c_h, c_h)
        self.linear = nn.Linear(c_h, c_out)
        self.input_emb = nn.Linear(c_in, c_h)
        self.emb1 = nn.Embedding(c_a, c_h)
        self.emb2 = nn.Embedding(c_a, c_h)
        self.emb3 = nn.Embedding(c_a, c_h)
        self.emb4 = nn.Embedding(c_a, c_h)
        self.emb5 = nn.Embedding(c_a, c_h)

    def _make_engine(self, ctx, P, G):
        return DecoderEngine(ctx, P, G, self.c_in, self.c_out, self.c_h, self.c_a, self.ns, self.seg_len, self.output_mask)

    def emb_grad_views(self):
        return [self.grad_view('emb%d.weight' % i) for i in range(1, 6)]

    def forward(self, x, c, lengths=None):
        """x: enc_act [B, c_in, T'] (fp32);  c: int64 [B]  ->  [B, c_out, 8T'].
        lengths (eval only): encoded frames per sample of a ragged batch; the first 8 * lengths[b] output frames are valid."""
        eng = self._engine()
        ctx = eng.ctx
        xb = x.detach().permute(0, 2, 1).contiguous().float()
        B, T0, E = xb.shape
        bits = ctx.act('d_in_%d_%d_%d' % (eng.uid, B, T0), B, T0, E)
        L.call('zs_cast_rows', 'ZsCastRows', ctx.stream, dtype=ctx.dt, src=L.ptr(xb), ld_src=E, src_f32=1, dst=bits.ptr(),
               ld_dst=bits.ld, dst_f32=0, col_off=0, rows=B * T0, cols=E, fill_cols=bits.ld, act=L.ZS_ACT_NONE)
        cidx = c.detach().to(xb.device, torch.int64).contiguous()
        xdec = eng.forward(bits, cidx, self.training, lengths=lengths)
        return xdec.valid().permute(0, 2, 1).clone()


class SpeakerClassifier(ZsModule):
    """model/model.py:231-280."""

    def __init__(self, c_in=512, c_h=512, n_class=8, dp=0.1, ns=0.01, seg_len=128, dtype=None):
        super(SpeakerClassifier, self).__init__(dtype)
        self.dp, self.ns, self.seg_len = dp, ns, seg_len
        self.c_in, self.c_h, self.n_class = c_in, c_h, n_class
        self.conv1 = nn.Conv1d(c_in, c_h, kernel_size=5)
        self.conv2 = nn.Conv1d(c_h, c_h, kernel_size=5)
        self.conv3 = nn.Conv1d(c_h, c_h, kernel_size=5)
        self.conv4 = nn.Conv1d(c_h, c_h, kernel_size=5)
        self.conv5 = nn.Conv1d(c_h, c_h, kernel_size=5)
        self.conv6 = nn.Conv1d(c_h, c_h, kernel_size=5)
        self.conv7 = nn.Conv1d(c_h, c_h // 2, kernel_size=3)
        self.conv8 = nn.Conv1d(c_h // 2, c_h // 4, kernel_size=3)
        if seg_len == 128:
            self.conv9 = nn.Conv1d(c_h // 4, n_class, kernel_size=16)
        elif seg_len == 64:
            self.conv9 = nn.Conv1d(c_h // 4, n_class, kernel_size=8)
        elif seg_len == 32:
            self.conv9 = nn.Conv1d(c_h // 4, n_class, kernel_size=4)
        else:
            raise NotImplementedError('Segement length {} is not supported!'.format(seg_len))

    def _make_engine(self, ctx, P, G):
        return ClassifierEngine(ctx, P, G, self.c_in, self.c_h, self.n_class, self.dp, self.ns, self.seg_len)

    def input_act(self, x_btc_f32):
        """fp32 [B, T', c_in] (contiguous, on the device) -> Act in the compute dtype."""
        eng = self._engine()
        ctx = eng.ctx
        B, T, C = x_btc_f32.shape
        a = ctx.act('c_in_%d_%d_%d' % (eng.uid, B, T), B, T, C)
        L.call('zs_cast_rows', 'ZsCastRows', ctx.stream, dtype=ctx.dt, src=L.ptr(x_btc_f32), ld_src=x_btc_f32.stride(1), src_f32=1,
               dst=a.ptr(), ld_dst=a.ld, dst_f32=0, col_off=0, rows=B * T, cols=C, fill_cols=a.ld, act=L.ZS_ACT_NONE)
        return a

    def forward(self, x, drop_masks=None, seed=None):
        """x: enc [B, c_in, T'] -> logits [B, n_class]."""
        eng = self._engine()
        xb = x.detach().permute(0, 2, 1).contiguous().float()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        masks = None
        if drop_masks is not None:
            masks = [m.to(xb.device, torch.uint8).contiguous() if m is not None else None for m in drop_masks]
        logits = eng.forward(self.input_act(xb), self.training, seed=seed, drop_masks=masks)
        return logits.valid()[:, 0, :].clone()
