"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from
/root/reference, build container only) and assert that oracle/zs_oracle.py
reproduces it.  The reference never travels: only inputs/outputs (data) are
written.  Run:  python oracle/make_golden.py

TEST INFRASTRUCTURE ONLY (see oracle/zs_oracle.py header).
"""
import json
import os
import struct
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import zs_oracle as O  # noqa: E402
from model.model import Decoder, Encoder, SpeakerClassifier  # noqa: E402  (the reference)

GOLD = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(4)


def sd_np(prefix, module):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def sd_t(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def ref_encoder_eval(enc, x, seed):
    """Reference eval forward; capture the single torch.rand draw (model/model.py:96)."""
    enc.eval()
    torch.manual_seed(seed)
    with torch.no_grad():
        act, logits = enc(x)
    torch.manual_seed(seed)
    U = torch.rand(x.size(0), logits.size(2), enc.enc_size, 2)
    return act, logits, U


def check(name, a, b, tol=0.0):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    tol = tol * max(1.0, float(b.abs().max())) if a.numel() else tol      # relative to the reference's scale
    assert err <= tol, '%s: oracle != reference, max abs err %g > %g' % (name, err, tol)
    print('  ok %-38s max|d|=%.3g' % (name, err))


def check_param(name, a, b, lr, g, nsteps):
    """Parameters after Adam steps.  The update lr*m/(sqrt(v)+1e-8) is ill-conditioned where |g| is at
    rounding-noise level (e.g. biases in front of an InstanceNorm have an exactly-zero true gradient, and a
    sign flip of such a g moves the parameter by 2*lr per step): there only bound by 2.1*lr*steps; where
    the step-0 gradient is significant (|g| > 1e-6) require 5% of lr."""
    d = (torch.as_tensor(a) - torch.as_tensor(b)).abs()
    sig = torch.as_tensor(g).abs() > 1e-6
    m_sig = d[sig].max().item() if sig.any() else 0.0
    assert d.max().item() <= 2.1 * lr * nsteps and m_sig <= 0.05 * lr * nsteps, (name, d.max().item(), m_sig)
    print('  ok %-38s max|d|=%.3g (significant-g: %.3g, %d/%d)' % (name, d.max().item(), m_sig, int(sig.sum()), d.numel()))


def gen_infer(tag, c_in, c_h1, c_h2, c_h3, E, c_h, n_spk, lengths, B, seed0):
    """Eval-mode Encoder/Decoder vectors at several fragment lengths (SURVEY 8c items 1,2,6)."""
    ns, dp, seg_len = 0.01, 0.5, 128
    torch.manual_seed(seed0)
    enc = Encoder(c_in=c_in, c_h1=c_h1, c_h2=c_h2, c_h3=c_h3, ns=ns, dp=dp, enc_size=E, seg_len=seg_len,
                  enc_mode='multilabel_binary')
    dec = Decoder(c_in=E, c_out=c_in, c_h=c_h, c_a=n_spk, ns=ns, seg_len=seg_len)
    enc.eval(); dec.eval()
    out = dict(sd_np('enc.', enc)); out.update(sd_np('dec.', dec))
    out['meta'] = np.array(json.dumps(dict(c_in=c_in, c_h1=c_h1, c_h2=c_h2, c_h3=c_h3, enc_size=E, c_h=c_h,
                                           n_spk=n_spk, ns=ns, dp=dp, seg_len=seg_len, lengths=lengths, B=B)))
    esd, dsd = sd_t(enc), sd_t(dec)
    for T in lengths:
        x = torch.rand(B, c_in, T) * 0.98 + 1e-3
        c = torch.randint(0, n_spk, (B,))
        act, logits, U = ref_encoder_eval(enc, x, seed0 + T)
        with torch.no_grad():
            xdec = dec(act, c)
            o_act, o_logits = O.encoder_forward(esd, x, ns, dp, E, seg_len, U=U, training=False)
            o_xdec = O.decoder_forward(dsd, o_act, c, ns, seg_len)
        check('%s T=%d enc logits' % (tag, T), o_logits, logits, 1e-4)
        flips = (o_act != act)
        if flips.any():   # only allowed where the Gumbel margin is at rounding level
            G = O.gumbel_from_uniform(U)
            s = (logits.permute(0, 2, 1).reshape(B, -1, E, 2) + G)
            margin = (s[..., 0] - s[..., 1]).abs().permute(0, 2, 1)[flips]
            assert margin.max() < 1e-4, margin.max()
            print('  note: %d bit flips at margin < 1e-4' % int(flips.sum()))
        else:
            print('  ok %-38s bit-exact (%d bits)' % ('%s T=%d enc_act' % (tag, T), act.numel()))
        with torch.no_grad():
            o_xdec_same = O.decoder_forward(dsd, act, c, ns, seg_len)
        check('%s T=%d x_dec' % (tag, T), o_xdec_same, xdec, 2e-5)
        assert xdec.shape[2] == O.out_len(T)
        out['x.%d' % T] = x.numpy(); out['c.%d' % T] = c.numpy(); out['U.%d' % T] = U.numpy()
        out['enc_act.%d' % T] = act.numpy(); out['enc.%d' % T] = logits.numpy(); out['x_dec.%d' % T] = xdec.numpy()
    np.savez_compressed(os.path.join(GOLD, 'infer_%s.npz' % tag), **out)


def gen_train(tag, c_in, c_h1, c_h2, c_h3, E, c_h, n_spk, B, T, steps, seed0):
    """train_ae steps with the reference modules + torch.optim.Adam + clip_grad_norm_ exactly as
    trainer.py:65-66, 326-332 / utils.py:48-55 (dropout p=0 so U is the only RNG draw)."""
    ns, seg_len, lr, max_norm = 0.01, 128, 1e-4, 5.0
    torch.manual_seed(seed0)
    enc = Encoder(c_in=c_in, c_h1=c_h1, c_h2=c_h2, c_h3=c_h3, ns=ns, dp=0.0, enc_size=E, seg_len=seg_len,
                  enc_mode='multilabel_binary')
    dec = Decoder(c_in=E, c_out=c_in, c_h=c_h, c_a=n_spk, ns=ns, seg_len=seg_len)
    enc.train(); dec.train()
    out = dict(sd_np('enc0.', enc)); out.update(sd_np('dec0.', dec))
    hp = dict(ns=ns, enc_dp=0.0, enc_size=E, seg_len=seg_len)
    oracle = O.TrainAE(sd_t(enc), sd_t(dec), hp, lr=lr, max_grad_norm=max_norm)
    opt = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()), lr=lr, betas=(0.5, 0.9))
    x = torch.rand(B, c_in, T) * 0.98 + 1e-3
    c = torch.randint(0, n_spk, (B,))
    out['x'] = x.numpy(); out['c'] = c.numpy()
    out['meta'] = np.array(json.dumps(dict(c_in=c_in, c_h1=c_h1, c_h2=c_h2, c_h3=c_h3, enc_size=E, c_h=c_h,
                                           n_spk=n_spk, ns=ns, seg_len=seg_len, B=B, T=T, steps=steps, lr=lr,
                                           max_grad_norm=max_norm)))
    for s in range(steps):
        torch.manual_seed(seed0 + 100 + s)
        xin = x.clone().requires_grad_(True)                      # to_var(), utils.py:43-45
        enc_act, _ = enc(xin)
        x_dec = dec(enc_act, c)
        loss = torch.mean(torch.abs(x_dec - xin))
        enc.zero_grad(); dec.zero_grad()
        loss.backward()
        if s == 0:
            for k, p in enc.named_parameters():
                out['genc.' + k] = p.grad.numpy().copy()
            for k, p in dec.named_parameters():
                out['gdec.' + k] = (p.grad.numpy().copy() if p.grad is not None else np.zeros(p.shape, np.float32))
        ne = torch.nn.utils.clip_grad_norm_(enc.parameters(), max_norm)
        nd = torch.nn.utils.clip_grad_norm_(dec.parameters(), max_norm)
        opt.step()
        torch.manual_seed(seed0 + 100 + s)
        U = torch.rand(B, T // 8, E, 2)
        o_loss, o_ne, o_nd, o_xdec, o_act = oracle.step(x, c, U=U, training=True)
        print('   step', s, 'loss', o_loss, loss.item(), 'norms', o_ne, ne.item(), o_nd, nd.item(), 'flips', int((o_act != enc_act.detach()).sum()))
        # Step 0 is compared tightly.  Later steps only through the loss: Adam turns rounding-level
        # gradients into +-lr moves and the tau=0.1 straight-through encoder gradient is spiky, so two
        # exact-fp32 implementations already differ by ~10% in the step-1 encoder grad norm (measured
        # here between this oracle and the reference) while their losses agree to 1e-7.
        check('%s step %d loss' % (tag, s), o_loss, loss.item(), 1e-6 if s == 0 else 1e-4)
        if s == 0:
            check('%s step %d norms' % (tag, s), torch.tensor([o_ne, o_nd]), torch.tensor([ne.item(), nd.item()]), 1e-4)
            assert (o_act == enc_act.detach()).all()
        out['U.%d' % s] = U.numpy(); out['loss.%d' % s] = np.float32(loss.item())
        out['norm_enc.%d' % s] = np.float32(ne.item()); out['norm_dec.%d' % s] = np.float32(nd.item())
        out['x_dec.%d' % s] = x_dec.detach().numpy(); out['enc_act.%d' % s] = enc_act.detach().numpy()
        if s == 0:
            for k, v in enc.state_dict().items():
                check_param('%s step %d enc.%s' % (tag, s, k), oracle.enc_sd[k], v, lr, out['genc.' + k], s + 1)
                out['enc%d.%s' % (s + 1, k)] = v.numpy().copy()
            for k, v in dec.state_dict().items():
                check_param('%s step %d dec.%s' % (tag, s, k), oracle.dec_sd[k], v, lr, out['gdec.' + k], s + 1)
                out['dec%d.%s' % (s + 1, k)] = v.numpy().copy()
    # step-0 gradient check of the oracle (fresh copy)
    e0 = {k[5:]: torch.from_numpy(v) for k, v in out.items() if k.startswith('enc0.')}
    d0 = {k[5:]: torch.from_numpy(v) for k, v in out.items() if k.startswith('dec0.')}
    _, (ge, gd), _, _ = O.train_ae_grads(e0, d0, x, c, hp, U=torch.from_numpy(out['U.0']))
    for k in ge:
        check('%s grad enc.%s' % (tag, k), ge[k], out['genc.' + k], 1e-6 + 1e-4 * float(np.abs(out['genc.' + k]).max()))
    for k in gd:
        check('%s grad dec.%s' % (tag, k), gd[k], out['gdec.' + k], 1e-6 + 1e-4 * float(np.abs(out['gdec.' + k]).max()))
    np.savez_compressed(os.path.join(GOLD, 'train_%s.npz' % tag), **out)


def gen_dropout_check():
    """Training-mode forward with dropout ON: oracle (F.dropout, same call order/shapes) must equal the
    reference under the same seed.  Validation only; nothing stored."""
    torch.manual_seed(5)
    enc = Encoder(c_in=80, c_h1=16, c_h2=32, c_h3=16, ns=0.01, dp=0.5, enc_size=8, seg_len=128,
                  enc_mode='multilabel_binary').train()
    x = torch.rand(2, 80, 32)
    torch.manual_seed(9)
    with torch.no_grad():
        act, logits = enc(x)
    torch.manual_seed(9)
    with torch.no_grad():
        o_act, o_logits = O.encoder_forward(sd_t(enc), x, 0.01, 0.5, 8, 128, training=True)
    check('dropout-on train fwd logits', o_logits, logits, 2e-5)
    assert (o_act == act).all()


def gen_classifier(seed0=31):
    """SpeakerClassifier fwd/bwd + CE."""
    ns, seg_len, n_class, c_in, c_h, B = 0.01, 128, 5, 16, 32, 3
    torch.manual_seed(seed0)
    clf = SpeakerClassifier(c_in=c_in, c_h=c_h, n_class=n_class, dp=0.0, ns=ns, seg_len=seg_len).train()
    x = torch.randn(B, c_in, 16)
    y = torch.randint(0, n_class, (B,))
    logits = clf(x)
    loss = torch.nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    out = dict(sd_np('clf.', clf))
    for k, p in clf.named_parameters():
        out['g.' + k] = p.grad.numpy().copy()
    out['x'] = x.numpy(); out['y'] = y.numpy(); out['logits'] = logits.detach().numpy(); out['loss'] = np.float32(loss.item())
    out['meta'] = np.array(json.dumps(dict(c_in=c_in, c_h=c_h, n_class=n_class, ns=ns, seg_len=seg_len, B=B)))
    p = {k: v.clone().requires_grad_(True) for k, v in sd_t(clf).items()}
    o_logits = O.speaker_classifier_forward(p, x, ns, 0.0, seg_len, training=True)
    o_loss = O.cross_entropy(o_logits, y)
    o_loss.backward()
    check('classifier logits', o_logits.detach(), logits.detach(), 2e-5)
    check('classifier loss', o_loss.detach(), loss.detach(), 1e-6)
    for k in p:
        check('classifier grad ' + k, p[k].grad, out['g.' + k], 1e-6 + 1e-4 * float(np.abs(out['g.' + k]).max()))
    np.savez_compressed(os.path.join(GOLD, 'classifier_small.npz'), **out)


def gen_vocoder():
    """Griffin-Lim: librosa is absent (parity unpinned); cross-check the STFT/iSTFT restatement against
    torch.stft/istft and freeze a short oracle run for regression."""
    rng = np.random.RandomState(0)
    y = rng.randn(200 * 30).astype(np.float32) * 0.1
    S = O.stft(y)
    win = torch.from_numpy(O.hann_padded())
    St = torch.stft(torch.from_numpy(y), O.N_FFT, O.HOP, window=win, center=True, pad_mode='reflect',
                    return_complex=True).numpy()
    check('stft vs torch.stft (re)', S.real, St.real, 2e-4)
    check('stft vs torch.stft (im)', S.imag, St.imag, 2e-4)
    yi = O.istft(S)
    yt = torch.istft(torch.from_numpy(St), O.N_FFT, O.HOP, window=win, center=True, length=len(yi)).numpy()
    check('istft vs torch.istft', yi, yt, 2e-5)
    assert len(yi) == O.HOP * (S.shape[1] - 1)
    mag = np.clip(rng.rand(40, 513).astype(np.float32), 1e-8, 1)
    wav = O.spectrogram2wav(mag, n_iter=8, do_trim=False)
    assert len(wav) == O.HOP * (40 - 1)
    np.savez_compressed(os.path.join(GOLD, 'vocoder_small.npz'), mag=mag, wav_iter8=wav,
                        stft_in=y, stft_re=S.real, stft_im=S.imag, istft_out=yi)
    # wav sample-count law from the reference's own sample outputs (docs/exp/**.wav, 16 kHz PCM16)
    counts = {}
    base = os.path.join(REF, 'docs', 'exp')
    for d, _, files in os.walk(base):
        for f in files:
            if f.endswith('.wav'):
                with open(os.path.join(d, f), 'rb') as fh:
                    raw = fh.read()
                i = raw.find(b'data')
                n = struct.unpack('<I', raw[i + 4:i + 8])[0] // 2
                counts[os.path.relpath(os.path.join(d, f), base)] = n
    with open(os.path.join(GOLD, 'docs_exp_wav_samples.json'), 'w') as fh:
        json.dump(counts, fh, indent=1, sort_keys=True)
    print('  wrote %d wav sample counts' % len(counts))


class _MaskDrop(torch.nn.Module):
    """Stands in for one nn.Dropout2d of the reference module so that the keep masks are known: y = x * keep / (1 - p)."""

    def __init__(self, p):
        super(_MaskDrop, self).__init__()
        self.p, self.queue = p, []

    def forward(self, x):
        m = self.queue.pop(0)
        return x * m[:, :, None, None] / (1.0 - self.p)


def gen_stage2():
    """Stage 2 (--train_p / --train_tgat): the reference's PatchDiscriminator, its WGAN-GP penalty
    (utils.calculate_gradients_penalty, double backward) and the patchGAN losses of trainer.py:488-494 / 528-533 on B = 2
    segments.  The 10.9 M discriminator weights are regenerated from a seed (O.synthetic_patch_sd); the fixture holds inputs,
    Dropout2d keep masks, losses, logits, a strided sample + the norm of every parameter gradient, and dLoss_G/dx_gen."""
    import types
    sys.modules.setdefault('tensorboardX', types.SimpleNamespace(SummaryWriter=object))
    from model.model import PatchDiscriminator                      # the reference
    from utils import calculate_gradients_penalty                    # the reference
    ns, seg_len, n_class, dp, B = 0.01, 128, 2, 0.1, 2
    hp = dict(ns=ns, seg_len=seg_len, dp=dp, training=True, beta_dis=1.0, beta_clf=1.0, beta_gen=1.0, lambda_=10.0)
    sd = O.synthetic_patch_sd(n_class, seed=7)
    D = PatchDiscriminator(n_class=n_class, ns=ns, dp=dp, seg_len=seg_len)
    D.load_state_dict(sd)
    D.train()
    drops = [_MaskDrop(dp) for _ in range(6)]
    for i, d in enumerate(drops):
        setattr(D, 'drop%d' % (i + 1), d)
    g = torch.Generator().manual_seed(21)
    x_t = torch.rand(B, 513, seg_len, generator=g)
    x_dec = torch.rand(B, 513, seg_len, generator=g).requires_grad_(True)     # as in training: x_dec comes out of gen_step (the
    c = torch.randint(0, n_class, (B,), generator=g)                           # reference's penalty differentiates w.r.t. it)
    chans = [64, 128, 256, 512, 512, 32]
    masks = [[(torch.rand(B, ch, generator=g) >= dp).float() for ch in chans] for _ in range(4)]      # real, fake, interpolate, G-step

    def feed(m):
        for d, mk in zip(drops, m):
            d.queue.append(mk)

    # ---- D step (trainer.py:480-494) on the reference modules
    feed(masks[0]); D_real, real_logits = D(x_t, classify=True)
    feed(masks[1]); D_fake, fake_logits = D(x_dec, classify=True)
    w_dis = torch.mean(D_real - D_fake)
    feed(masks[2])
    torch.manual_seed(99)
    gp = calculate_gradients_penalty(D, x_t, x_dec)
    torch.manual_seed(99)
    alpha = torch.rand(B)
    loss_clf = torch.nn.CrossEntropyLoss()(real_logits, c)
    loss = -hp['beta_dis'] * w_dis + hp['beta_clf'] * loss_clf + hp['lambda_'] * gp
    D.zero_grad()
    loss.backward()
    gref = {k: p.grad.detach().clone() for k, p in D.named_parameters()}
    # ---- the oracle on the same inputs
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x_dec = x_dec.detach()
    o_loss, o_w, o_clf, o_gp, o_logits = O.patch_d_loss(p, x_t, x_dec, c, alpha, hp, masks=masks[:3])
    o_loss.backward()
    check('stage2 D_real', O.patch_discriminator_forward(sd, x_t, ns, seg_len, dp=dp, drop_masks=masks[0]), D_real.detach(), 2e-5)
    check('stage2 real logits', o_logits.detach(), real_logits.detach(), 2e-5)
    check('stage2 w_dis', o_w.detach(), w_dis.detach(), 2e-5)
    check('stage2 gp', o_gp.detach(), gp.detach(), 2e-5)
    check('stage2 D loss', o_loss.detach(), loss.detach(), 2e-5)
    for k in gref:
        check('stage2 D grad ' + k, p[k].grad, gref[k], 1e-7 + 2e-4 * float(gref[k].abs().max()))
    out = dict(x_t=x_t.numpy(), x_dec=x_dec.numpy(), c=c.numpy(), alpha=alpha.numpy(), D_real=D_real.detach().numpy(),
               D_fake=D_fake.detach().numpy(), real_logits=real_logits.detach().numpy(), w_dis=np.float32(w_dis.item()),
               gp=np.float32(gp.item()), loss_clf=np.float32(loss_clf.item()), loss_d=np.float32(loss.item()))
    for pi, ms in enumerate(masks):
        for li, mk in enumerate(ms):
            out['mask.%d.%d' % (pi, li)] = mk.numpy().astype(np.uint8)
    for k, v in gref.items():
        flat = v.reshape(-1)
        out['gD.norm.' + k] = np.float32(flat.double().norm().item())
        out['gD.sample.' + k] = flat[::max(1, flat.numel() // 512)][:512].numpy().copy()
    # ---- G step (trainer.py:524-533): gradient of the generator loss w.r.t. the generated spectrogram
    x_gen = x_dec.clone().requires_grad_(True)
    feed(masks[3]); D_f, f_logits = D(x_gen, classify=True)
    loss_adv = -torch.mean(D_f)
    loss_g = hp['beta_clf'] * torch.nn.CrossEntropyLoss()(f_logits, c) + hp['beta_gen'] * loss_adv
    dx, = torch.autograd.grad(loss_g, x_gen)
    xo = x_dec.clone().requires_grad_(True)
    o_lg, o_adv, o_c, o_fl = O.patch_g_loss(sd, xo, c, hp, masks=masks[3])
    odx, = torch.autograd.grad(o_lg, xo)
    check('stage2 G loss', o_lg.detach(), loss_g.detach(), 2e-5)
    check('stage2 dLoss_G/dx_gen', odx, dx, 1e-8 + 2e-4 * float(dx.abs().max()))
    out.update(loss_g=np.float32(loss_g.item()), loss_adv=np.float32(loss_adv.item()), fake_logits=f_logits.detach().numpy(),
               dx_gen=dx.numpy())
    out['meta'] = np.array(json.dumps(dict(ns=ns, seg_len=seg_len, n_class=n_class, dp=dp, B=B, seed=7, beta_dis=1.0, beta_clf=1.0,
                                            beta_gen=1.0, lambda_=10.0)))
    np.savez_compressed(os.path.join(GOLD, 'stage2_small.npz'), **out)


if __name__ == '__main__':
    print('[golden] inference vectors')
    gen_infer('f80', c_in=80, c_h1=16, c_h2=32, c_h3=16, E=8, c_h=32, n_spk=4,
              lengths=[9, 10, 16, 24, 127, 129, 201], B=2, seed0=1)
    gen_infer('f513', c_in=513, c_h1=8, c_h2=32, c_h3=8, E=6, c_h=32, n_spk=3, lengths=[24, 33], B=2, seed0=2)
    print('[golden] train_ae vectors')
    gen_train('f80', c_in=80, c_h1=16, c_h2=32, c_h3=16, E=8, c_h=32, n_spk=4, B=3, T=32, steps=3, seed0=11)
    gen_dropout_check()
    print('[golden] classifier')
    gen_classifier()
    print('[golden] vocoder')
    gen_vocoder()
    print('[golden] stage 2 (PatchDiscriminator, WGAN-GP)')
    gen_stage2()
    print('done ->', GOLD)
