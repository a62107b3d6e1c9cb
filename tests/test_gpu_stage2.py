"""Stage 2 on the GPU (BASELINE config 5): PatchDiscriminator forward, the patchGAN discriminator step with
the WGAN-GP double backward, and the generator step, through the product code (zs_amd.patch / zs_amd.stage2) against the golden
vectors captured from the reference's PatchDiscriminator + utils.calculate_gradients_penalty (tests/golden/stage2_small.npz; the
10.9 M weights are regenerated from the seed by oracle.synthetic_patch_sd).

Tolerances (fp32 path): values / logits 1e-3 of scale; parameter gradients of the D loss (including the second-order penalty
term) 5e-3 of each tensor's scale on a strided sample of 512 entries + the tensor norm to 5e-3; dLoss_G/dx_gen 2e-3 of scale."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import zs_amd  # noqa: F401
    return torch.device('cuda:0')


class _Hps(object):
    def __init__(self, m):
        self.beta_dis, self.beta_clf, self.beta_gen, self.lambda_ = m['beta_dis'], m['beta_clf'], m['beta_gen'], m['lambda_']
        self.max_grad_norm, self.lr, self.n_speakers, self.n_target_speakers = 5.0, 1e-4, m['n_class'], m['n_class']


def _setup(dev, dtype='fp32'):
    import zs_oracle as O
    from zs_amd.patch import PatchDiscriminator
    d, m = load_golden('stage2_small.npz')
    D = PatchDiscriminator(n_class=m['n_class'], ns=m['ns'], dp=m['dp'], seg_len=m['seg_len'], dtype=dtype).to(dev)
    D.load_state_dict(O.synthetic_patch_sd(m['n_class'], m['seed']))
    masks = [[torch.from_numpy(d['mask.%d.%d' % (p, l)]).float().to(dev) for l in range(6)] for p in range(4)]
    return d, m, D, masks


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(1e-12, np.abs(b).max())


# Tolerances per compute dtype against the reference's golden vectors.  fp32: the exact-fp32 MFMA path (summation order only).
# bf16 (the default dtype of `main.py --train_p`): bf16 storage of every activation / adjoint through six conv + InstanceNorm2d
# layers and, for the penalty, a double backward; bounds = ~2-3x the errors measured on an MI355X (printed by the tests):
# forward 1.3e-2, w_dis 3.1e-2 of its value, CE 1.6e-2, gp 3.7e-3, worst gradient tensor 0.31 of its scale (conv_classify.bias:
# B = 2 samples, peaky softmax of a random-init critic), dLoss_G/dx_gen 0.19 relative L2.  The same comparison at B = 128
# (test_config5_b128_...) gives 0.4-2.7 % on the bias gradients and 11-18 % (cosine >= 0.984) on the conv weights.
TOL = {'fp32': dict(fwd=1e-3, wdis=1e-3, ce=1e-4, gp=2e-3, grad=5e-3, gnorm=5e-3, dx_l2=5e-3, dx_off=5e-3, ladv=1e-3),
       'bf16': dict(fwd=4e-2, wdis=8e-2, ce=5e-2, gp=2e-2, grad=6e-1, gnorm=6e-1, dx_l2=5e-1, dx_off=1.0, ladv=8e-2)}


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_patch_discriminator_forward_golden(dev, dtype):
    d, m, D, masks = _setup(dev, dtype)
    D.train()
    x_t = torch.from_numpy(d['x_t']).to(dev)                      # [B, 513, T], the reference's layout
    val, logits = D(x_t, classify=True, drop_masks=masks[0])
    ev, el = _rel(val.cpu().numpy(), d['D_real']), _rel(logits.cpu().numpy(), d['real_logits'])
    print('%s forward vs reference: val %.3g logits %.3g' % (dtype, ev, el))
    assert ev < TOL[dtype]['fwd'] and el < TOL[dtype]['fwd']
    assert set(D.state_dict().keys()) == {'%s.%s' % (n, k) for n in ['conv%d' % i for i in range(1, 8)] + ['conv_classify'] for k in ('weight', 'bias')}


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_patchgan_d_step_with_gradient_penalty_golden(dev, dtype):
    """w_dis, CE, gp and every PatchDiscriminator parameter gradient of -beta*w_dis + beta*CE + lambda*gp (trainer.py:488-494)."""
    from zs_amd.stage2 import PatchGANStep
    d, m, D, masks = _setup(dev, dtype)
    tol = TOL[dtype]

    class _Net(object):                                          # gen_step is bypassed: x_gen is given
        def flat_params(self):
            return D.flat_params()

        def mark_dirty(self):
            pass
    step = PatchGANStep.__new__(PatchGANStep)
    step.D, step.hps, step.g_mode, step.shift, step.device = D, _Hps(m), 'naive', 0, dev
    step.loss_clf = torch.zeros(1, device=dev); step.correct = torch.zeros(1, dtype=torch.int32, device=dev)
    x_t = torch.from_numpy(d['x_t']).permute(0, 2, 1).contiguous().to(dev)           # [B, T, F]
    x_gen = torch.from_numpy(d['x_dec']).permute(0, 2, 1).contiguous().to(dev)
    c = torch.from_numpy(d['c']).to(dev)
    r = step.d_step(None, x_t, c, alpha=torch.from_numpy(d['alpha']).to(dev), masks=masks[:3], update=False, x_gen=x_gen)
    torch.cuda.synchronize()
    print('%s D step vs reference: w_dis %.5f (%.5f)  CE %.5f (%.5f)  gp %.4f (%.4f)' %
          (dtype, r['w_dis'].item(), float(d['w_dis']), r['real_loss_clf'].item(), float(d['loss_clf']), r['gp'].item(), float(d['gp'])))
    assert abs(r['w_dis'].item() - float(d['w_dis'])) < tol['wdis'] * max(1.0, abs(float(d['w_dis'])))
    assert abs(r['real_loss_clf'].item() - float(d['loss_clf'])) < tol['ce']
    assert abs(r['gp'].item() - float(d['gp'])) < tol['gp'] * max(1.0, float(d['gp'])), (r['gp'].item(), float(d['gp']))
    worst = ('', 0.0)
    for k, _ in D.named_parameters():
        g = D.grad_view(k).detach().reshape(-1)
        ref = d['gD.sample.' + k]
        got = g[::max(1, g.numel() // 512)][:512].cpu().numpy()
        e = np.abs(got - ref).max() / max(1e-9, np.abs(ref).max())
        en = abs(g.double().norm().item() - float(d['gD.norm.' + k])) / max(1e-9, float(d['gD.norm.' + k]))
        if max(e, en) > worst[1]:
            worst = (k, max(e, en))
        assert e < tol['grad'] and en < tol['gnorm'], (k, e, en)
    print(dtype + ' worst D-gradient error (sample / norm, relative): %s %.3g; gp %.5f (reference %.5f)' % (worst + (r['gp'].item(), float(d['gp']))))


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_patchgan_g_step_input_gradient_golden(dev, dtype):
    """The generator loss and its gradient w.r.t. the generated spectrogram (what flows into the Generator), trainer.py:524-533."""
    from zs_amd.stage2 import PatchGANStep
    d, m, D, masks = _setup(dev, dtype)
    tol = TOL[dtype]
    step = PatchGANStep.__new__(PatchGANStep)
    step.D, step.hps, step.g_mode, step.shift, step.device = D, _Hps(m), 'naive', 0, dev
    step.loss_clf = torch.zeros(1, device=dev); step.correct = torch.zeros(1, dtype=torch.int32, device=dev)
    x_gen = torch.from_numpy(d['x_dec']).permute(0, 2, 1).contiguous().to(dev)
    c = torch.from_numpy(d['c']).to(dev)
    r = step.g_step(None, x_gen, c, masks=masks[3], update=False, x_gen=x_gen)
    torch.cuda.synchronize()
    assert abs(r['loss_adv'].item() - float(d['loss_adv'])) < tol['ladv'] * max(1.0, abs(float(d['loss_adv'])))
    assert _rel(r['fake_logits'].cpu().numpy(), d['fake_logits']) < tol['fwd']
    dx = r['dx_gen'].permute(0, 2, 1).cpu().numpy()                                  # -> [B, 513, T]
    # One of the 135 168 layer-4 pre-activations of this vector is zero to within fp32 rounding (|z| < 4e-6): the GPU's
    # summation order puts it on the other side of the LeakyReLU kink (slope 0.01), its gradient differs by the factor 100 and the
    # transposed convolutions above spread that over a 24 x 128 patch of one sample.  Hence: relative L2 error over everything,
    # and the entries that are off by more than 2e-3 of the scale must be a small patch (< 0.5 %).
    ref = d['dx_gen']
    l2 = np.linalg.norm(dx - ref) / np.linalg.norm(ref)
    off = (np.abs(dx - ref) > 2e-3 * np.abs(ref).max()).mean()
    print('%s dLoss_G/dx_gen: relative L2 error %.3g, entries off by > 2e-3 of scale: %.3g %%' % (dtype, l2, 100 * off))
    assert l2 < tol['dx_l2'] and off < tol['dx_off'], (l2, off)


def test_generator_chain_vs_oracle(dev):
    """gen_step + the generator's parameter gradients through x_gen = x_dec + x_dec * G (targeted_residual) and a linear loss on
    x_gen, against the oracle's autograd; and the target-guided L1 step."""
    import zs_oracle as O
    from zs_amd.model import Decoder, Encoder
    from zs_amd.stage2 import PatchGANStep
    torch.manual_seed(0)
    E, ch, nspk, ntgt = 8, 32, 4, 2
    enc = Encoder(c_in=513, c_h1=8, c_h2=32, c_h3=8, ns=0.01, dp=0.0, enc_size=E, seg_len=128, enc_mode='multilabel_binary', dtype='fp32').to(dev)
    dec = Decoder(c_in=E, c_out=513, c_h=ch, c_a=nspk, ns=0.01, seg_len=128, dtype='fp32').to(dev)
    gen = Decoder(c_in=E, c_out=513, c_h=ch, c_a=ntgt, ns=0.01, seg_len=128, output_mask=True, dtype='fp32').to(dev)

    class H(object):
        beta_dis = beta_clf = beta_gen = 1.0; lambda_ = 10.0; max_grad_norm = 5.0; lr = 1e-4; n_speakers = nspk; n_target_speakers = ntgt
    step = PatchGANStep(enc, dec, gen, gen, H, 'targeted_residual')      # the discriminator slot is unused here
    g = torch.Generator().manual_seed(3)
    B, T = 2, 128
    x = torch.rand(B, T, 513, generator=g)
    c = torch.randint(nspk - ntgt, nspk, (B,), generator=g)
    G = O.gumbel_from_uniform(torch.rand(B, T // 8, E, 2, generator=g))
    wlin = torch.randn(B, T, 513, generator=g) * 1e-3
    x_gen = step.gen_forward(x.to(dev), c.to(dev), True, noise=G.contiguous().to(dev), noise_kind=0)
    step._gen_backward(wlin.to(dev).contiguous())
    torch.cuda.synchronize()
    esd = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    dsd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    gsd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in gen.state_dict().items()}
    with torch.no_grad():
        act, _ = O.encoder_forward(esd, x.permute(0, 2, 1), 0.01, 0.0, E, 128, G=G, training=True)
    og = O.gen_step(dsd, gsd, act, c, nspk - ntgt, 0.01, 128, 'targeted_residual')         # [B, 513, T]
    assert _rel(x_gen.permute(0, 2, 1).cpu().numpy(), og.detach().numpy()) < 1e-3
    (og * wlin.permute(0, 2, 1)).sum().backward()
    for k, _ in gen.named_parameters():
        ref = gsd[k].grad if gsd[k].grad is not None else torch.zeros_like(gsd[k])
        scale = ref.abs().max().item()
        if scale < 1e-7:
            continue
        e = (gen.grad_view(k).cpu() - ref).abs().max().item() / scale
        assert e < 3e-3, (k, e)
    # target-guided step: L1(x_gen, x_t) and its update (no clipping)
    before = gen.flat_params()[0].clone()
    lrec = step.tg_step(x.to(dev), c.to(dev), noise=G.contiguous().to(dev), noise_kind=0)
    torch.cuda.synchronize()
    assert abs(lrec.item() - (og.detach() - x.permute(0, 2, 1)).abs().mean().item()) < 1e-5
    assert not torch.equal(before, gen.flat_params()[0])
    # the target-speaker range is checked on the HOST batch (DevicePrefetcher(check=...)), not by device syncs in every step
    step.check_targets(c)
    with pytest.raises(RuntimeError, match='target speakers'):
        step.check_targets(torch.zeros(B, dtype=torch.int64))


def test_patchgan_loop_runs_and_checkpoints(dev, tmp_path, monkeypatch):
    """Trainer.train(mode='patchGAN', target_guided=True) end to end on a narrow autoencoder (the discriminator has the reference's
    fixed width): console lines, checkpoint dict with the two DataParallel-prefixed stage-2 entries, reload."""
    from zs_amd.dataloader import DataLoader, SyntheticDataset
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    monkeypatch.setenv('ZS_CKPT_EVERY', '2')
    hps = make_hps(enc_size=8, emb_size=32, n_speakers=4, n_target_speakers=2, batch_size=2, patch_iters=2, n_patch_steps=2, max_to_keep=5)
    src = DataLoader(SyntheticDataset(16, seg_len=128, n_speakers=2, seed=1), 2)
    tgt = DataLoader(SyntheticDataset(16, seg_len=128, n_speakers=2, seed=2, speaker_offset=2), 2)
    tr = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=str(tmp_path / 'log'), dtype='bf16', device=dev)
    tr.add_duo_loader(src, tgt)
    buf = io.StringIO()
    with redirect_stdout(buf):
        tr.train(str(tmp_path / 'm.pth'), 'train', mode='patchGAN', target_guided=True)
    txt = buf.getvalue()
    assert 'patch_D-1:[000002/000002], w_dis=' in txt and 'patch_G:[000002/000002], loss_adv=' in txt and 'tg_rec=' in txt
    ck = torch.load(str(tmp_path / 'm.pth-s2-2'), map_location='cpu', weights_only=True)
    assert set(ck.keys()) == {'encoder', 'decoder', 'generator', 'classifier', 'patch_discriminator', 'target_classifier'}
    assert all(k.startswith('module.') for k in ck['patch_discriminator']) and 'module.conv_classify.weight' in ck['target_classifier']
    assert all(torch.isfinite(v).all() for v in ck['patch_discriminator'].values())
    tr2 = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=str(tmp_path / 'log'), dtype='bf16', device=dev)
    with redirect_stdout(io.StringIO()):
        tr2.load_model(str(tmp_path / 'm.pth-s2-2'), 'encoder, decoder, generator, patch_discriminator, target_classifier')
    assert torch.equal(tr2.PatchDiscriminator.flat_params()[0], tr.PatchDiscriminator.flat_params()[0])


def test_main_train_p_synthetic(dev, tmp_path, monkeypatch, capsys):
    """`python main.py --train_p --synthetic` (main.py:155-158): source / target loaders, add_duo_loader, the patchGAN mode."""
    sys.path.insert(0, ROOT)
    import main
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('ZS_CKPT_EVERY', '1')
    dcfg = json.load(open(os.path.join(ROOT, 'hps', 'zerospeech_english.json')))
    dcfg.update(enc_size=8, emb_size=32, n_speakers=4, n_target_speakers=2, batch_size=2, patch_iters=1, n_patch_steps=1)
    hp_path = str(tmp_path / 'hps.json')
    json.dump(dcfg, open(hp_path, 'w'))
    main.main(['--train_p', '--synthetic', '--hps_path', hp_path, '--ckpt_dir', str(tmp_path / 'ck'), '--dtype', 'bf16'])
    out = capsys.readouterr().out
    assert 'patch_G:[000001/000001]' in out and 'pre_AE' not in out
    assert os.path.exists(str(tmp_path / 'ck' / 'model.pth-s2-1'))


# ---- BASELINE config 5 at its own size: B = 128, bf16, english hps with enc_size = emb_size = 1024 ---------------------------

def _keep_masks(B, dp, gen):
    return [(torch.rand(B, C, generator=gen) >= dp).float() for C in (64, 128, 256, 512, 512, 32)]


def _refl(u, n):
    u = np.where(u < 0, -u, u)
    return np.where(u >= n, 2 * (n - 1) - u, u)


def test_config5_b128_bf16_steps_vs_fp32_path_and_fp64_spot_checks(dev, tmp_path):
    """BASELINE config 5 on one GPU as `tools/stage2_bench.py --tgat` / bench.py's secondary record run it (B = 128, bf16,
    enc_size = emb_size = 1024; reference trainer.py:467-560, utils.py:58-77): D step, G step and target-guided step are
    finite and bit-identical when replayed from the same state; w_dis / gp / CE and every discriminator gradient (second-order
    penalty term included) stay within a stated bound of the fp32 HIP path on the same weights, masks, alpha and x_gen; and,
    on the operands the bf16 kernels really read, fp64 dot products taken on the host confirm (1) a conv2d_gather + GEMM forward
    (conv3) and (2) entries of conv3's accumulated weight gradient = real + fake + adjoint-pass + reverse-sweep contributions.

    Bounds (measured values in the test output): |w_dis - fp32| <= 0.02 (|.| + 1), |gp - fp32| <= 0.05 (gp + 1), CE 0.02;
    gradient tensors (dominated by lambda * d gp / d theta at random init: bf16 storage of every activation, gradient and adjoint
    through 6 conv + InstanceNorm2d layers, four sweeps): relative L2 error <= 0.35 and cosine >= 0.94 against the fp32 tensor
    (measured: worst 0.18 / 0.985 on conv2.weight)."""
    from zs_amd import layers
    from zs_amd.hps import make_hps
    from zs_amd.patch import PatchDiscriminator
    from zs_amd.stage2 import PatchGANStep
    from zs_amd.trainer import Trainer
    torch.manual_seed(0)
    B = 128
    hps = make_hps(enc_size=1024, emb_size=1024, batch_size=B)
    tr = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=str(tmp_path / 'log'), dtype='bf16', device=dev)
    s2 = tr.stage2()
    D = tr.PatchDiscriminator
    g = torch.Generator().manual_seed(0)
    x_s, x_t = torch.rand(B, 128, 513, generator=g).to(dev), torch.rand(B, 128, 513, generator=g).to(dev)
    c_t = torch.randint(hps.n_speakers - hps.n_target_speakers, hps.n_speakers, (B,), generator=g).to(dev)
    masks = [[m.to(dev) for m in _keep_masks(B, D.dp, g)] for _ in range(3)]
    alpha = torch.rand(B, generator=g).to(dev)
    x_gen = s2.gen_forward(x_s, c_t, False, seed=77).clone()
    assert torch.isfinite(x_gen).all()

    def d_once(step):
        r = step.d_step(None, x_t, c_t, alpha=alpha, masks=masks, update=False, x_gen=x_gen)
        torch.cuda.synchronize()
        return (r['w_dis'].item(), r['gp'].item(), r['real_loss_clf'].item()), step.D.flat_params()[1].clone()

    v1, g1 = d_once(s2)
    v2, g2 = d_once(s2)
    layers.check_status(dev)
    assert all(np.isfinite(v1)) and torch.isfinite(g1).all()
    assert v1 == v2 and torch.equal(g1, g2), 'the bf16 D step is not deterministic'

    # ---- fp64 spot checks on the tapes of that step (before anything overwrites them) ----------------------------------------
    eng = D._engine()
    rng = np.random.RandomState(1)
    ns = float(D.ns)
    bf = lambda t: t.float().to(torch.bfloat16).double()
    rows = lambda a: a.t[a.off:a.off + a.rows * a.ld].view(a.rows, a.ld)
    Wc = D.conv3.weight.detach()                                   # [256, 128, kf = 5, kt = 5]
    Cout, C, KF, KT = Wc.shape
    W64 = bf(Wc).cpu()
    b64 = D.conv3.bias.detach().double().cpu()
    tp = eng.tapes['real']
    L1, L2 = tp['layers'][1], tp['layers'][2]
    H, Wd, Ho, Wo = L2['H'], L2['W'], L2['Ho'], L2['Wo']
    assert (H, Wd, Ho, Wo, L2['C'], L2['Cout']) == (32, 129, 16, 65, C, Cout)
    a1 = rows(L1['a']).view(B, H, Wd, -1)                          # layer-2 output [B, H, W, C] (rows ordered b, h, w)
    y2 = rows(L2['y']).view(B, Ho, Wo, -1)                         # conv3 + lrelu, before InstanceNorm
    got, ref, aref = [], [], []
    for _ in range(96):
        b, ho, wo, co = rng.randint(B), rng.randint(Ho), rng.randint(Wo), rng.randint(Cout)
        hs, ws = _refl(2 * ho + np.arange(KT) - 2, H), _refl(2 * wo + np.arange(KF) - 2, Wd)
        patch = a1[b][torch.from_numpy(hs).to(dev)][:, torch.from_numpy(ws).to(dev), :C].double().cpu()      # [kt, kf, C]
        prod = patch * W64[co].permute(2, 1, 0)                                                               # W[co, c, kf, kt] -> [kt, kf, c]
        acc = float(prod.sum()) + float(b64[co])
        got.append(float(y2[b, ho, wo, co])); ref.append(acc if acc > 0 else ns * acc); aref.append(float(prod.abs().sum()) + abs(float(b64[co])))
    got, ref, aref = np.array(got), np.array(ref), np.array(aref)
    tol = 2.0 ** -8 * np.abs(ref) + 2e-4 * aref + 1e-12
    assert (np.abs(got - ref) <= tol).all(), 'conv3 forward (gather + GEMM): worst |d|/tol %.3g' % (np.abs(got - ref) / tol).max()

    # conv3.weight gradient = sum over the four passes that feed it of dY^T gather_W(xh)
    name = lambda key, what: eng.ctx.act(eng._name(key, what, B), B, Ho * Wo, Cout)
    passes = []
    for key in ('real', 'fake'):
        passes.append((name(key, 'bgz2'), eng.tapes[key]['layers'][2]['xin']))
    ti = eng.tapes['inter']['layers'][2]
    passes.append((ti['gz'], eng.ctx.act(eng._name('inter', 'gxh2', B), B * Ho, Wd, 5 * C, ld=ti['layer'].cin_pad)))
    passes.append((name('inter', 'rgz2'), ti['xin']))
    gW = D.grad_view('conv3.weight')
    got, ref, aref = [], [], []
    wo_idx = torch.arange(Wo, device=dev)
    for _ in range(24):
        co, c, kf, kt = rng.randint(Cout), rng.randint(C), rng.randint(KF), rng.randint(KT)
        wi = torch.from_numpy(_refl(2 * np.arange(Wo) + kf - 2, Wd)).to(dev)
        tot, tota = 0.0, 0.0
        for dY, xh in passes:
            ycol = rows(dY).view(B * Ho, Wo, -1)[:, :, co].double()                       # [B*Ho, Wo]
            xcol = rows(xh).view(B * Ho, Wd, -1)[:, wi, kt * C + c].double()               # [B*Ho, Wo]
            pr = ycol * xcol
            tot += float(pr.sum()); tota += float(pr.abs().sum())
        got.append(float(gW[co, c, kf, kt])); ref.append(tot); aref.append(tota)
    got, ref, aref = np.array(got), np.array(ref), np.array(aref)
    tol = 2e-4 * aref + 1e-12
    assert np.abs(ref).max() > 0
    assert (np.abs(got - ref) <= tol).all(), 'conv3.weight gradient (4 passes, second-order term included): worst |d|/tol %.3g' % (np.abs(got - ref) / tol).max()

    # ---- the same step on the fp32 HIP path: same weights, masks, alpha, x_gen ------------------------------------------------
    D32 = PatchDiscriminator(n_class=D.n_class, ns=D.ns, dp=D.dp, seg_len=D.seg_len, dtype='fp32').to(dev)
    D32.load_state_dict(D.state_dict())
    st32 = PatchGANStep.__new__(PatchGANStep)
    st32.D, st32.hps, st32.g_mode, st32.shift, st32.device = D32, hps, s2.g_mode, s2.shift, dev
    st32.loss_clf = torch.zeros(1, device=dev); st32.correct = torch.zeros(1, dtype=torch.int32, device=dev)
    v32, g32 = d_once(st32)
    print('B=128 D step: bf16 (w_dis, gp, CE) = %s; fp32 path = %s' % (v1, v32))
    assert abs(v1[0] - v32[0]) <= 0.02 * (abs(v32[0]) + 1) and abs(v1[1] - v32[1]) <= 0.05 * (v32[1] + 1) and abs(v1[2] - v32[2]) <= 0.02
    table = []
    for k, p in D.named_parameters():
        a, r = D.grad_view(k).double().reshape(-1), D32.grad_view(k).double().reshape(-1)
        if k.startswith('conv7') and k.endswith('bias'):
            continue                                               # d/d(conv7.bias): the critic's bias cancels in w_dis and in gp (exactly 0 +- rounding)
        e = ((a - r).norm() / r.norm().clamp_min(1e-30)).item()
        cos = (torch.dot(a, r) / (a.norm() * r.norm()).clamp_min(1e-30)).item()
        table.append((k, e, cos))
    print('B=128 D step, bf16 gradient tensors vs the fp32 path (relative L2 error, cosine): ' +
          '; '.join('%s %.3f %.4f' % t for t in table))
    for k, e, cos in table:
        assert e <= 0.35 and cos >= 0.94, (k, e, cos)
    del D32, st32
    torch.cuda.empty_cache()

    # ---- G step and target-guided step: finite, deterministic ----------------------------------------------------------------
    G = tr.Generator
    res = []
    for _ in range(2):
        r = s2.g_step(x_s, x_t, c_t, masks=masks[0], update=False, seed=78)
        ladv = r['loss_adv'].item()
        gg = G.flat_params()[1].clone()
        lrec = s2.tg_step(x_t, c_t, update=False, seed=79).item()
        res.append((ladv, gg, lrec, G.flat_params()[1].clone()))
    layers.check_status(dev)
    assert np.isfinite(res[0][0]) and np.isfinite(res[0][2]) and torch.isfinite(res[0][1]).all() and torch.isfinite(res[0][3]).all()
    assert res[0][0] == res[1][0] and res[0][2] == res[1][2] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][3], res[1][3])
    assert res[0][1].abs().max().item() > 0 and res[0][3].abs().max().item() > 0
