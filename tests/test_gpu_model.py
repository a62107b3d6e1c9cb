"""GPU parity tests of the whole autoencoder path through the drop-in modules, against (a) the golden
vectors captured from the reference and (b) the oracle on seeded inputs.

Tolerances (fp32 path; north_star: 1e-3 relative, MBV bits bit-exact given logits+noise):
  encoder logits / decoder output: 1e-3 of the reference scale; bits: identical, except where the Gumbel
  margin |(l0+g0)-(l1+g1)| is below the logit error bound (end-to-end fp32 summation-order noise) --
  every mismatch is checked to be such a near-tie.  bf16 path: measured and bounded loosely."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub_sd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import zs_amd  # noqa: F401
    return torch.device('cuda:0')


def _build(m, dtype, dev, d, enc_prefix='enc.', dec_prefix='dec.', dp=None):
    from zs_amd.model import Decoder, Encoder
    enc = Encoder(c_in=m['c_in'], c_h1=m['c_h1'], c_h2=m['c_h2'], c_h3=m['c_h3'], ns=m['ns'], dp=m.get('dp', 0.0) if dp is None else dp,
                  enc_size=m['enc_size'], seg_len=m['seg_len'], enc_mode='multilabel_binary', dtype=dtype).to(dev)
    dec = Decoder(c_in=m['enc_size'], c_out=m['c_in'], c_h=m['c_h'], c_a=m['n_spk'], ns=m['ns'], seg_len=m['seg_len'], dtype=dtype).to(dev)
    enc.load_state_dict(sub_sd(d, enc_prefix))
    dec.load_state_dict(sub_sd(d, dec_prefix))
    return enc, dec


def _rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a - b).abs().max().item() / max(1e-6, b.abs().max().item())


@pytest.mark.parametrize('name', ['infer_f80.npz', 'infer_f513.npz'])
def test_infer_golden_fp32(dev, name):
    import zs_oracle as O
    d, m = load_golden(name)
    enc, dec = _build(m, 'fp32', dev, d)
    enc.eval(); dec.eval()
    total_bits = flips_total = 0
    for T in m['lengths']:
        x = torch.from_numpy(d['x.%d' % T]).to(dev)
        c = torch.from_numpy(d['c.%d' % T]).to(dev)
        U = torch.from_numpy(d['U.%d' % T])
        G = O.gumbel_from_uniform(U)
        act, logits = enc(x, G=G.to(dev))
        ref_logits, ref_act = torch.from_numpy(d['enc.%d' % T]), torch.from_numpy(d['enc_act.%d' % T])
        e = _rel(logits, ref_logits)
        assert e < 1e-3, 'T=%d logits rel err %.3g' % (T, e)
        assert act.shape == ref_act.shape and set(np.unique(act.cpu().numpy())) <= {0.0, 1.0}
        flips = (act.cpu() != ref_act)
        total_bits += act.numel(); flips_total += int(flips.sum())
        if flips.any():
            s = ref_logits.permute(0, 2, 1).reshape(x.size(0), -1, m['enc_size'], 2) + G
            margin = (s[..., 0] - s[..., 1]).abs().permute(0, 2, 1)[flips]
            bound = 4 * (logits.cpu() - ref_logits).abs().max().item() + 1e-6
            assert margin.max().item() <= bound, 'bit flip at margin %.3g > logit error bound %.3g' % (margin.max().item(), bound)
        xdec = dec(torch.from_numpy(d['enc_act.%d' % T]).to(dev), c)
        e = _rel(xdec, d['x_dec.%d' % T])
        assert e < 1e-3, 'T=%d x_dec rel err %.3g' % (T, e)
    print('%s: %d/%d MBV bits differ from the reference (near-ties only)' % (name, flips_total, total_bits))
    assert flips_total <= max(1, total_bits // 200)


def test_mbv_bits_exact_given_reference_logits(dev):
    """The discretiser contract: identical (logits, G) -> identical bits, through the public module path's
    kernel (zs_mbv_fwd) on the golden logits."""
    import zs_oracle as O
    from zs_amd import _lib as L
    d, m = load_golden('infer_f80.npz')
    for T in m['lengths']:
        logits = torch.from_numpy(d['enc.%d' % T]).permute(0, 2, 1).contiguous()          # [B,T',2E]
        B, Tp, E2 = logits.shape
        G = O.gumbel_from_uniform(torch.from_numpy(d['U.%d' % T])).contiguous()
        bits = torch.zeros(B * Tp, E2 // 2, device=dev)
        logits_d, G_d = logits.to(dev), G.to(dev)      # keep alive: the call only takes raw pointers
        L.call('zs_mbv_fwd', 'ZsMbvFwd', torch.cuda.current_stream().cuda_stream, dtype=L.ZS_F32, logits=L.ptr(logits_d),
               ld=E2, logits_f32=1, noise=L.ptr(G_d), noise_kind=0, rows=B * Tp, E=E2 // 2, tau=0.1, bits_f32=L.ptr(bits))
        torch.cuda.synchronize()
        ref = torch.from_numpy(d['enc_act.%d' % T]).permute(0, 2, 1).reshape(B * Tp, E2 // 2)
        assert torch.equal(bits.cpu(), ref)


def _grad_dict(net):
    return {k: net.grad_view(k).detach().cpu().clone() for k, _ in net.named_parameters()}


def test_train_step_golden_fp32(dev):
    """One --train_ae step against the reference's loss / per-parameter grads / clip norms / Adam update."""
    import zs_oracle as O
    from zs_amd.trainer import AEStep
    d, m = load_golden('train_f80.npz')
    enc, dec = _build(m, 'fp32', dev, d, 'enc0.', 'dec0.', dp=0.0)
    ae = AEStep(enc, dec, lr=m['lr'], max_grad_norm=m['max_grad_norm'])
    x = torch.from_numpy(d['x']).permute(0, 2, 1).contiguous().to(dev)
    c = torch.from_numpy(d['c']).to(dev)
    G = O.gumbel_from_uniform(torch.from_numpy(d['U.0'])).contiguous().to(dev)
    loss = ae.step(x, c, noise=G, noise_kind=0, update=False)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(d['loss.0'])) < 1e-5, (loss.item(), float(d['loss.0']))
    _rel(ae.xdec.valid().permute(0, 2, 1), d['x_dec.0'])
    assert _rel(ae.xdec.valid().permute(0, 2, 1), d['x_dec.0']) < 1e-3
    worst = ('', 0.0)
    for net, pre in ((enc, 'genc.'), (dec, 'gdec.')):
        for k, g in _grad_dict(net).items():
            ref = torch.from_numpy(d[pre + k])
            scale = ref.abs().max().item()
            if scale < 1e-6:              # exactly-zero true gradients (biases in front of InstanceNorm): noise only
                assert g.abs().max().item() < 1e-5, (k, g.abs().max().item())
                continue
            e = (g - ref).abs().max().item() / scale
            if e > worst[1]:
                worst = (pre + k, e)
            assert e < 2e-3, 'grad %s%s rel err %.3g' % (pre, k, e)
    print('worst grad rel err: %s %.3g' % worst)
    sq = ae.grad_norms()
    torch.cuda.synchronize()
    ne, nd = float(sq[0].item()) ** 0.5, float(sq[1].item()) ** 0.5
    assert abs(ne - float(d['norm_enc.0'])) < 2e-3 * float(d['norm_enc.0'])
    assert abs(nd - float(d['norm_dec.0'])) < 2e-3 * float(d['norm_dec.0'])
    ae.optimizer_step()
    torch.cuda.synchronize()
    lr = m['lr']
    for net, gpre, ppre in ((enc, 'genc.', 'enc1.'), (dec, 'gdec.', 'dec1.')):
        for k, p in net.state_dict().items():
            if gpre + k not in d:
                continue
            err = (p.detach().cpu() - torch.from_numpy(d[ppre + k])).abs()
            sig = torch.from_numpy(np.abs(d[gpre + k]) > 1e-5)
            assert err.max().item() <= 2.1 * lr, (k, err.max().item())
            if sig.any():
                assert err[sig].max().item() <= 0.1 * lr, (k, err[sig].max().item())
    # two more steps: loss trajectory follows the reference (see oracle/make_golden.py on why only the loss)
    for s in (1, 2):
        G = O.gumbel_from_uniform(torch.from_numpy(d['U.%d' % s])).contiguous().to(dev)
        loss = ae.step(x, c, noise=G, noise_kind=0)
        assert abs(loss.item() - float(d['loss.%d' % s])) < 5e-4, (s, loss.item(), float(d['loss.%d' % s]))


def test_train_with_dropout_masks_vs_oracle(dev):
    """Dropout ON with injected keep-masks: gradients equal the oracle's autograd."""
    import zs_oracle as O
    from zs_amd.trainer import AEStep
    d, m = load_golden('train_f80.npz')
    m = dict(m)
    enc, dec = _build(m, 'fp32', dev, d, 'enc0.', 'dec0.', dp=0.5)
    ae = AEStep(enc, dec, lr=m['lr'], max_grad_norm=m['max_grad_norm'])
    x_bct = torch.from_numpy(d['x'])
    c = torch.from_numpy(d['c'])
    B, _, T = x_bct.shape
    gen = torch.Generator().manual_seed(4)
    c2 = m['c_h2']
    Ts = [T, T // 2, T // 4, T // 8, T // 8, T // 8]
    masks = [(torch.rand(B, c2, t, generator=gen) >= 0.5).float() for t in Ts]           # [B,C,T] for the oracle
    U = torch.from_numpy(d['U.0'])
    G = O.gumbel_from_uniform(U)
    hp = dict(ns=m['ns'], enc_dp=0.5, enc_size=m['enc_size'], seg_len=m['seg_len'])
    o_loss, (ge, gd), _, o_act = O.train_ae_grads(sub_sd(d, 'enc0.'), sub_sd(d, 'dec0.'), x_bct, c, hp, G=G, drop_masks=masks)
    dm = [mk.permute(0, 2, 1).contiguous().to(torch.uint8).to(dev) for mk in masks]      # [B,T,C] for the kernels
    loss = ae.step(x_bct.permute(0, 2, 1).contiguous().to(dev), c.to(dev), noise=G.contiguous().to(dev), noise_kind=0,
                   drop_masks=dm, update=False)
    torch.cuda.synchronize()
    assert abs(loss.item() - o_loss.item()) < 1e-5
    for net, ref in ((enc, ge), (dec, gd)):
        for k, g in _grad_dict(net).items():
            scale = ref[k].abs().max().item()
            if scale < 1e-6:
                continue
            assert (g - ref[k]).abs().max().item() / scale < 2e-3, k


def test_bf16_path_is_close(dev):
    """bf16 storage / MFMA path: measured against the fp32 golden (not bit-level)."""
    import zs_oracle as O
    d, m = load_golden('infer_f80.npz')
    enc, dec = _build(m, 'bf16', dev, d)
    enc.eval(); dec.eval()
    T = 129
    x = torch.from_numpy(d['x.%d' % T]).to(dev)
    G = O.gumbel_from_uniform(torch.from_numpy(d['U.%d' % T])).to(dev)
    act, logits = enc(x, G=G)
    e = _rel(logits, d['enc.%d' % T])
    flips = (act.cpu() != torch.from_numpy(d['enc_act.%d' % T])).float().mean().item()
    xdec = dec(torch.from_numpy(d['enc_act.%d' % T]).to(dev), torch.from_numpy(d['c.%d' % T]).to(dev))
    e2 = _rel(xdec, d['x_dec.%d' % T])
    print('bf16: logits rel err %.3g, bit mismatch rate %.3g, x_dec rel err %.3g' % (e, flips, e2))
    assert e < 0.08 and e2 < 0.05 and flips < 0.1


def test_state_dict_names_match_reference(dev):
    d, m = load_golden('infer_f513.npz')
    enc, dec = _build(m, 'fp32', dev, d)
    assert set(enc.state_dict().keys()) == set(sub_sd(d, 'enc.').keys())
    assert set(dec.state_dict().keys()) == set(sub_sd(d, 'dec.').keys())
    for k, v in enc.state_dict().items():
        assert tuple(v.shape) == tuple(d['enc.' + k].shape)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_full_size_properties(dev, dtype):
    """BASELINE config sizes (english hps, enc_size=1024, emb_size=1024, 102 speakers; B reduced to 32 to keep the
    test short -- bench.py runs B=256): size-independent properties.
      * MBV output is exactly {0,1}; decoder output in (0,1); shapes follow T -> T/8 -> T
      * spot-check: 256 random elements of the first conv-bank GEMM vs fp64 dot products on the host
      * the loss of repeated steps on one batch falls (the reference's 16-item loader re-serves one batch)
      * determinism: the same step replayed from the same state gives bit-identical loss and gradients."""
    from zs_amd.model import Decoder, Encoder
    from zs_amd.trainer import AEStep
    torch.manual_seed(0)
    B, T, Fb, E, ch, nspk = 32, 128, 513, 1024, 1024, 102
    enc = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=128, enc_mode='multilabel_binary', dtype=dtype).to(dev)
    dec = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=128, dtype=dtype).to(dev)
    x = (torch.rand(B, T, Fb) * (1 - 1e-8) + 1e-8).to(dev)
    c = torch.randint(0, nspk, (B,)).to(dev)
    ae = AEStep(enc, dec, lr=1e-4, max_grad_norm=5.0)
    l0 = ae.step(x, c, seed=7, update=False).item()
    g0 = enc.flat_params()[1].clone(), dec.flat_params()[1].clone()
    l0b = ae.step(x, c, seed=7, update=False).item()
    assert l0 == l0b and torch.equal(g0[0], enc.flat_params()[1]) and torch.equal(g0[1], dec.flat_params()[1])
    assert np.isfinite(l0) and all(torch.isfinite(g).all() for g in g0)
    assert g0[0].abs().max() > 0 and g0[1].abs().max() > 0
    # spot check of conv1s.6 (k=7) inside the cat buffer
    ee = enc._engine()
    cat = ee.tape['cat'].valid().float().cpu()
    w = enc.conv1s[6].weight.detach().double().cpu(); bias = enc.conv1s[6].bias.detach().double().cpu()
    xr = x.cpu().double()
    if dtype == 'bf16':
        xr, w = xr.float().bfloat16().double(), w.float().bfloat16().double()
    rng = np.random.RandomState(1)
    worst = 0.0
    for _ in range(256):
        b, t, n = rng.randint(B), rng.randint(T), rng.randint(128)
        acc = bias[n].item()
        for j in range(7):
            s = t + j - 3
            s = -s if s < 0 else (2 * (T - 1) - s if s >= T else s)
            acc += float((xr[b, s] * w[n, :, j]).sum())
        ref = acc if acc > 0 else 0.01 * acc
        worst = max(worst, abs(cat[b, t, 6 * 128 + n].item() - ref) / max(1.0, abs(ref)))
    assert worst < (1e-4 if dtype == 'fp32' else 2e-2), worst
    losses = [ae.step(x, c, seed=100 + i).item() for i in range(12)]
    enc.eval(); dec.eval()
    act, logits = enc(x.permute(0, 2, 1)[:4])
    xd = dec(act, c[:4])
    assert act.shape == (4, E, T // 8) and logits.shape == (4, 2 * E, T // 8) and xd.shape == (4, Fb, T)
    assert set(np.unique(act.cpu().numpy())) <= {0.0, 1.0}
    assert xd.min().item() >= 0 and xd.max().item() <= 1
    print(dtype, 'losses', ['%.4f' % v for v in [l0] + losses])
    assert losses[-1] < l0


def test_speaker_classifier_golden_fp32(dev):
    """SpeakerClassifier forward, CE loss and every parameter gradient against the reference (golden)."""
    from zs_amd import _lib as L
    from zs_amd.model import SpeakerClassifier
    d, m = load_golden('classifier_small.npz')
    clf = SpeakerClassifier(c_in=m['c_in'], c_h=m['c_h'], n_class=m['n_class'], dp=0.0, ns=m['ns'], seg_len=m['seg_len'], dtype='fp32').to(dev)
    clf.load_state_dict(sub_sd(d, 'clf.'))
    clf.train()
    x = torch.from_numpy(d['x']).to(dev)
    logits = clf(x)
    assert _rel(logits, d['logits']) < 1e-3
    eng = clf._engine()
    out = eng.forward(clf.input_act(x.permute(0, 2, 1).contiguous()), True)
    B = x.shape[0]
    y = torch.from_numpy(d['y']).to(dev)
    loss, corr = torch.zeros(1, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    dl = torch.zeros(B, out.ld, device=dev)
    L.call('zs_softmax_ce', 'ZsSoftmaxCE', torch.cuda.current_stream().cuda_stream, logits=out.ptr(), ld=out.ld, target=L.ptr(y), B=B,
           n_class=m['n_class'], loss_out=L.ptr(loss), dlogits=L.ptr(dl), ldg=out.ld, grad_scale=1.0, correct_out=L.ptr(corr))
    eng.backward(dl, out.ld, need_dx=True)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(d['loss'])) < 1e-5
    assert corr.item() == int((torch.from_numpy(d['logits']).argmax(1) == torch.from_numpy(d['y'])).sum())
    for k, g in _grad_dict(clf).items():
        ref = torch.from_numpy(d['g.' + k])
        scale = ref.abs().max().item()
        if scale < 1e-6:
            continue
        assert (g - ref).abs().max().item() / scale < 2e-3, k
    assert set(clf.state_dict().keys()) == set(sub_sd(d, 'clf.').keys())


def test_adversarial_steps_vs_oracle(dev):
    """D step (classifier CE) and G step (loss_rec - alpha*loss_clf, trainer.py:444) gradients against the oracle's autograd."""
    import zs_oracle as O
    from zs_amd.model import SpeakerClassifier
    from zs_amd.trainer import AEStep, ClfStep
    d, m = load_golden('train_f80.npz')
    enc, dec = _build(m, 'fp32', dev, d, 'enc0.', 'dec0.', dp=0.0)
    torch.manual_seed(3)
    clf = SpeakerClassifier(c_in=2 * m['enc_size'], c_h=32, n_class=m['n_spk'], dp=0.0, ns=m['ns'], seg_len=128, dtype='fp32').to(dev)
    ae = AEStep(enc, dec, lr=1e-4, max_grad_norm=5.0)
    cs = ClfStep(ae, clf, lr=1e-4, max_grad_norm=5.0)
    gen = torch.Generator().manual_seed(8)
    B, T = 3, 128
    x_bct = torch.rand(B, m['c_in'], T, generator=gen) * 0.98 + 1e-3
    c = torch.randint(0, m['n_spk'], (B,), generator=gen)
    G = O.gumbel_from_uniform(torch.rand(B, T // 8, m['enc_size'], 2, generator=gen))
    alpha = 0.37
    # oracle
    esd = {k: v.clone().requires_grad_(True) for k, v in sub_sd(d, 'enc0.').items()}
    dsd = {k: v.clone().requires_grad_(True) for k, v in sub_sd(d, 'dec0.').items()}
    csd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in clf.state_dict().items()}
    act, logits = O.encoder_forward(esd, x_bct, m['ns'], 0.0, m['enc_size'], 128, G=G, training=True)
    xdec = O.decoder_forward(dsd, act, c, m['ns'], 128)
    l_rec = O.l1_loss(xdec, x_bct)
    l_clf = O.cross_entropy(O.speaker_classifier_forward(csd, logits, m['ns'], 0.0, 128, training=True), c)
    (l_rec - alpha * l_clf).backward()
    xd = x_bct.permute(0, 2, 1).contiguous().to(dev)
    lr_, lc_, corr = cs.g_step(xd, c.to(dev), alpha, noise=G.contiguous().to(dev), noise_kind=0, update=False)
    torch.cuda.synchronize()
    assert abs(lr_.item() - l_rec.item()) < 1e-5 and abs(lc_.item() - l_clf.item()) < 1e-4
    for net, ref in ((enc, esd), (dec, dsd)):
        for k, g in _grad_dict(net).items():
            r = ref[k].grad if ref[k].grad is not None else torch.zeros_like(ref[k])
            scale = r.abs().max().item()
            if scale < 1e-6:
                continue
            assert (g - r).abs().max().item() / scale < 3e-3, k
    # D step: classifier gradients only
    lc2, _ = cs.d_step(xd, c.to(dev), alpha_dis=1.0, noise=G.contiguous().to(dev), noise_kind=0, update=False)
    torch.cuda.synchronize()
    csd2 = {k: v.detach().clone().requires_grad_(True) for k, v in csd.items()}
    with torch.no_grad():
        _, lg = O.encoder_forward({k: v.detach() for k, v in esd.items()}, x_bct, m['ns'], 0.0, m['enc_size'], 128, G=G, training=True)
    l2 = O.cross_entropy(O.speaker_classifier_forward(csd2, lg, m['ns'], 0.0, 128, training=True), c)
    l2.backward()
    assert abs(lc2.item() - l2.item()) < 1e-4
    for k, g in _grad_dict(clf).items():
        r = csd2[k].grad
        scale = r.abs().max().item()
        if scale < 1e-6:
            continue
        assert (g - r).abs().max().item() / scale < 3e-3, k
    cs.optimizer_step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in clf.parameters())


def test_graph_replay_equals_eager(dev):
    """hipGraph capture/replay of the training step: bit-identical loss trajectory and parameters to the same
    launches issued eagerly (device-side seed and Adam step count), and the noise really changes between steps."""
    from zs_amd.model import Decoder, Encoder
    from zs_amd.trainer import AEStep
    res = []
    for graph_warmup in (2, 10 ** 9):
        torch.manual_seed(0)
        enc = Encoder(c_in=80, c_h1=16, c_h2=64, c_h3=32, ns=0.01, dp=0.5, enc_size=32, seg_len=128, enc_mode='multilabel_binary',
                      dtype='bf16').to(dev)
        dec = Decoder(c_in=32, c_out=80, c_h=64, c_a=4, ns=0.01, seg_len=128, dtype='bf16').to(dev)
        ae = AEStep(enc, dec, lr=1e-3, max_grad_norm=5.0, use_graph=True)
        ae.graph_warmup = graph_warmup
        g = torch.Generator().manual_seed(1)
        losses = []
        for i in range(7):
            x = torch.rand(4, 128, 80, generator=g).to(dev)
            c = torch.randint(0, 4, (4,), generator=g).to(dev)
            losses.append(ae.step(x, c).item())
        res.append((losses, enc.flat_params()[0].clone(), dec.flat_params()[0].clone(), len(ae._graphs)))
    (la, ea, da, na), (lb, eb, db, nb) = res
    assert na == 1 and nb == 0
    assert la == lb, (la, lb)
    assert torch.equal(ea, eb) and torch.equal(da, db)
    assert len(set(la)) == len(la)
