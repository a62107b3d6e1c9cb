"""Stage 2 training steps (reference trainer.py:257-294, 467-560): the patchGAN discriminator / generator updates with the
WGAN-GP penalty and the optional target-guided reconstruction step (--train_p / --train_tgat).

One D step  (trainer.py:470-497):  loss = -beta_dis * mean(D(x_t) - D(x_gen)) + beta_clf * CE(D_clf(x_t), c_t - shift)
                                            + lambda * gp(D; x_t, x_gen);      clip(PatchDiscriminator), Adam(0.5, 0.9)
One G step  (trainer.py:514-533):  loss = beta_clf * CE(D_clf(x_gen), c_t - shift) - beta_gen * mean D(x_gen);
                                            clip(Generator), Adam
Target-guided (trainer.py:535-541): loss_rec = mean|gen_step(Encoder(x_t), c_t) - x_t|;  Adam WITHOUT clipping.
x_gen = gen_step(Encoder(x_s), c_t) = x_dec + x_dec * Generator(enc, c_t - shift)   ('targeted_residual', trainer.py:266-278).

Only the network that the reference's optimizer steps receives gradients here: the reference lets autograd fill the
gradients of Encoder / Decoder (and of the discriminator in the G step) as well but never uses them (reset_grad is called on
the stepped net only and the other optimizers are not stepped in this mode), so they are not computed.
"""
import torch

from . import _lib as L
from . import layers, parallel
from .layers import join_side


class NetOpt(object):
    """clip_grad_norm_(net, max_norm) + Adam(lr, betas) over ONE net's flat parameter buffer (zs_sqnorm + zs_adam_clip)."""

    def __init__(self, net, lr, betas=(0.5, 0.9)):
        self.net, self.lr, self.betas, self.t = net, float(lr), betas, 0
        flat, _ = net.flat_params()
        dev = flat.device
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.part = torch.zeros(1024, dtype=torch.float64, device=dev)
        self.reducer = parallel.GradReducer()

    def step(self, max_norm):
        flat, gflat = self.net.flat_params()
        st = torch.cuda.current_stream(flat.device).cuda_stream
        if parallel.world_size() > 1:
            self.reducer.start(gflat)
            self.reducer.finish()
        L.check(L.lib().zs_sqnorm(L.ptr(gflat), gflat.numel(), L.ptr(self.part), L.ptr(self.sq), st), 'zs_sqnorm')
        self.t += 1
        b1, b2 = self.betas
        L.call('zs_adam_clip', 'ZsAdam', st, p=L.ptr(flat), g=L.ptr(gflat), m=L.ptr(self.m), v=L.ptr(self.v), n=flat.numel(), lr=self.lr,
               beta1=b1, beta2=b2, eps=1e-8, bc1=1.0 - b1 ** self.t, bc2=1.0 - b2 ** self.t, sumsq=L.ptr(self.sq),
               max_norm=float(max_norm) if max_norm else 0.0, write_clipped_grad=0, grad_scale=self.reducer.scale)
        self.net.mark_dirty()


class PatchGANStep(object):
    def __init__(self, encoder, decoder, generator, discriminator, hps, g_mode):
        self.Encoder, self.Decoder, self.Generator, self.D = encoder, decoder, generator, discriminator
        self.hps, self.g_mode = hps, g_mode
        if g_mode not in ('naive', 'targeted', 'targeted_residual'):
            raise NotImplementedError('Invalid generator mode to call gen_step()!')
        self.shift = 0 if g_mode == 'naive' else int(hps.n_speakers - hps.n_target_speakers)
        self.device = encoder.flat_params()[0].device
        self.gen_opt = NetOpt(generator, hps.lr)
        self.patch_opt = NetOpt(discriminator, hps.lr)
        dev = self.device
        self.loss_clf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.correct = torch.zeros(1, dtype=torch.int32, device=dev)
        self.loss_rec = torch.zeros(1, dtype=torch.float32, device=dev)
        self._l1part = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.step_no = 0

    def check_targets(self, c_host):
        """c_host: speaker indices of a target batch, still on the HOST.  Raises like the reference's Generator embedding would
        for a speaker outside the target range (an index outside the table would read past it on the GPU)."""
        lo, hi = self.shift, self.shift + int(self.Generator.c_a)
        if c_host.numel() and (int(c_host.min()) < lo or int(c_host.max()) >= hi):
            raise RuntimeError('This generator can only convert to target speakers!')

    # ---- gen_step (trainer.py:266-278) ---------------------------------------------------------------------------------------
    def gen_forward(self, x_btf, c, train_generator, noise=None, noise_kind=2, drop_masks=None, seed=None):
        """x_btf fp32 [B, T, F] (source batch), c int64 [B] target speakers -> x_gen fp32 [B, T_out, F] contiguous."""
        enc, dec, gen = self.Encoder, self.Decoder, self.Generator
        enc.train(); dec.train(); gen.train()
        ee, de, ge = enc._engine(), dec._engine(), gen._engine()
        if seed is None:
            self.step_no += 1
            seed = (self.step_no * 0x9E3779B97F4A7C15 + 12345) % (1 << 62) + 7 * parallel.rank()
        bits, _, _ = ee.forward(x_btf, True, noise=noise, noise_kind=noise_kind, seed=seed, drop_masks=drop_masks)   # encode_step
        xd = de.forward(bits, c, False)                                       # Decoder(enc, c)
        # c - shift_c (trainer.py:277-279).  The range of c is checked where the batch is still on the host (check_targets(),
        # called once per host batch by the training loop / DevicePrefetcher): no device-to-host sync per step here
        cg = (c - self.shift) if self.shift else c
        m = ge.forward(bits, cg.contiguous(), train_generator)                # Generator(enc, c - shift_c)
        B, T = xd.B, xd.T
        F = de.F
        st = de.ctx.stream
        x_gen = de.ctx.f32('s2_xgen_%d_%d' % (B, T), B * T * F)[:B * T * F].view(B, T, F)
        L.check(L.lib().zs_gen_combine_fwd(xd.ptr(), m.ptr(), xd.ld, L.ptr(x_gen), B * T, F, int(self.g_mode == 'targeted_residual'), st),
                'zs_gen_combine_fwd')
        self._xd, self._m = xd, m
        return x_gen

    def _gen_backward(self, dx_gen):
        """dx_gen fp32 [B, T, F] contiguous -> Generator parameter gradients (x_dec is a constant here)."""
        ge = self.Generator._engine()
        xd, m = self._xd, self._m
        B, T, F = dx_gen.shape
        dpre = ge.ctx.act('s2_dpre_%d_%d' % (B, T), B, T, F)
        L.call('zs_gen_combine_bwd', 'ZsGenCombineBwd', ge.ctx.stream, dtype=ge.ctx.dt, dx_gen=L.ptr(dx_gen), ld_dx=F, xd=xd.ptr(), m=m.ptr(),
               ld_in=m.ld, dpre=dpre.ptr(), ldo=dpre.ld, fill_cols=dpre.ld, rows=B * T, F=F, mode=int(self.g_mode == 'targeted_residual'),
               tanh_out=int(self.Generator.output_mask))
        ge.backward(dpre, need_dbits=False)
        join_side(self.device)

    def _ce(self, logits, c, grad_scale):
        """CE(logits, c - shift) (trainer.py:297-304, shift=True) -> dlogits fp32 [B, n_class] * grad_scale; loss / #correct on device."""
        B, n = logits.shape
        tgt = ((c - self.shift) if self.shift else c).contiguous()
        lg = logits.contiguous()
        dl = torch.zeros(B, n, dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.call('zs_softmax_ce', 'ZsSoftmaxCE', st, logits=L.ptr(lg), ld=n, target=L.ptr(tgt), B=B, n_class=n, loss_out=L.ptr(self.loss_clf),
               dlogits=L.ptr(dl), ldg=n, grad_scale=float(grad_scale), correct_out=L.ptr(self.correct))
        return dl

    # ---- D step -----------------------------------------------------------------------------------------------------------------
    def d_step(self, x_s, x_t, c_t, alpha=None, masks=None, update=True, x_gen=None, **gen_kw):
        """x_s / x_t: fp32 [B, T, F] source / target batches, c_t int64 [B].  masks: optional [real, fake, interpolate] lists of six
        [B, C] Dropout2d keep masks; alpha: optional fp32 [B] interpolation weights (utils.py:59).  Returns device scalars."""
        hps, D = self.hps, self.D
        D.train()
        eng = D._engine()
        B = x_t.shape[0]
        if x_gen is None:
            x_gen = self.gen_forward(x_s, c_t, False, **gen_kw)
        mk = masks if masks is not None else [None, None, None]
        eng.zero_grads()
        val_r, logits_r = eng.forward(x_t.contiguous(), 'real', True, masks=mk[0], classify=True)
        dl = self._ce(logits_r, c_t, hps.beta_clf)                             # loss_clf on the REAL logits (trainer.py:491)
        eng.backward('real', dval=torch.full((B,), -float(hps.beta_dis) / B, device=self.device), dlogits=dl)
        val_f, _ = eng.forward(x_gen, 'fake', True, masks=mk[1], classify=False)
        eng.backward('fake', dval=torch.full((B,), float(hps.beta_dis) / B, device=self.device))
        w_dis = (val_r - val_f).mean()                                         # trainer.py:261 (B scalars: host-side glue)
        if alpha is None:
            alpha = torch.rand(B, device=self.device)
        n = x_t[0].numel()
        xi = eng.ctx.f32('s2_xi_%d' % B, B * n)[:B * n].view(x_t.shape)
        st = eng.ctx.stream
        L.check(L.lib().zs_lerp_rows(L.ptr(x_t.contiguous()), L.ptr(x_gen), L.ptr(alpha.float().contiguous()), L.ptr(xi), B, n, st), 'zs_lerp_rows')
        eng.forward(xi, 'inter', True, masks=mk[2], classify=False)
        gp = eng.gp_backward('inter', float(hps.lambda_))
        eng.flush_grads()
        if update:
            self.patch_opt.step(hps.max_grad_norm)
        return {'w_dis': w_dis, 'gp': gp[:1], 'real_loss_clf': self.loss_clf, 'correct': self.correct, 'real_logits': logits_r}

    # ---- G step -----------------------------------------------------------------------------------------------------------------
    def g_step(self, x_s, x_t, c_t, masks=None, update=True, x_gen=None, **gen_kw):
        hps, D = self.hps, self.D
        D.train()
        eng = D._engine()
        B = x_t.shape[0]
        have_tape = x_gen is None
        if x_gen is None:
            x_gen = self.gen_forward(x_s, c_t, True, **gen_kw)
        val_f, logits_f = eng.forward(x_gen, 'gfake', True, masks=masks, classify=True)
        dl = self._ce(logits_f, c_t, hps.beta_clf)
        loss_adv = -val_f.mean()
        dx = eng.backward('gfake', dval=torch.full((B,), -float(hps.beta_gen) / B, device=self.device), dlogits=dl, need_dx=True,
                          param_grads=False)
        if have_tape:
            self._gen_backward(dx)
            if update:
                self.gen_opt.step(hps.max_grad_norm)
        return {'loss_adv': loss_adv, 'fake_loss_clf': self.loss_clf, 'correct': self.correct, 'dx_gen': dx, 'fake_logits': logits_f}

    # ---- target-guided step (teacher forcing, trainer.py:535-541) -----------------------------------------------------------------
    def tg_step(self, x_t, c_t, update=True, **gen_kw):
        x_gen = self.gen_forward(x_t, c_t, True, **gen_kw)
        n = x_gen.numel()
        d = self.Generator._engine().ctx.f32('s2_dl1_%d' % n, n)[:n].view(x_gen.shape)
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(L.lib().zs_l1_plain(L.ptr(x_gen), L.ptr(x_t.contiguous()), n, 1.0, L.ptr(self._l1part), L.ptr(self.loss_rec), L.ptr(d), st),
                'zs_l1_plain')
        self._gen_backward(d)
        if update:
            self.gen_opt.step(0.0)                                             # no grad_clip in the reference here
        return self.loss_rec
