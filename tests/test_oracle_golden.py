"""CPU: the oracle (oracle/zs_oracle.py) against the golden vectors captured from the reference
(oracle/make_golden.py).  Tolerances: 1e-4 relative-to-scale for encoder logits (InstanceNorm over a
few frames amplifies fp32 summation-order noise), 2e-5 for decoder outputs, bit-exact MBV bits."""
import json
import os

import numpy as np
import pytest
import torch

import zs_oracle as O
from conftest import GOLD, load_golden, sub_sd


def _close(a, b, tol):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


@pytest.mark.parametrize('name', ['infer_f80.npz', 'infer_f513.npz'])
def test_infer_golden(name):
    d, m = load_golden(name)
    esd, dsd = sub_sd(d, 'enc.'), sub_sd(d, 'dec.')
    for T in m['lengths']:
        x, c, U = (torch.from_numpy(d['%s.%d' % (k, T)]) for k in ('x', 'c', 'U'))
        with torch.no_grad():
            act, logits = O.encoder_forward(esd, x, m['ns'], m['dp'], m['enc_size'], m['seg_len'], U=U)
            xdec = O.decoder_forward(dsd, torch.from_numpy(d['enc_act.%d' % T]), c, m['ns'], m['seg_len'])
        _close(logits, d['enc.%d' % T], 1e-4)
        assert torch.equal(act, torch.from_numpy(d['enc_act.%d' % T]))          # bits: exact
        assert set(np.unique(act.numpy())) <= {0.0, 1.0}
        _close(xdec, d['x_dec.%d' % T], 2e-5)
        assert xdec.shape[2] == O.out_len(T) and act.shape[2] == O.out_len(T) // 8


def test_train_step0_golden():
    d, m = load_golden('train_f80.npz')
    hp = dict(ns=m['ns'], enc_dp=0.0, enc_size=m['enc_size'], seg_len=m['seg_len'])
    x, c = torch.from_numpy(d['x']), torch.from_numpy(d['c'])
    tr = O.TrainAE(sub_sd(d, 'enc0.'), sub_sd(d, 'dec0.'), hp, lr=m['lr'], max_grad_norm=m['max_grad_norm'])
    _, (ge, gd), _, _ = O.train_ae_grads(tr.enc_sd, tr.dec_sd, x, c, hp, U=torch.from_numpy(d['U.0']))
    for k, g in ge.items():
        _close(g, d['genc.' + k], 1e-4) if np.abs(d['genc.' + k]).max() > 1e-5 else None
    for k, g in gd.items():
        _close(g, d['gdec.' + k], 1e-4) if np.abs(d['gdec.' + k]).max() > 1e-5 else None
    losses = []
    for s in range(m['steps']):
        loss, ne, nd, x_dec, act = tr.step(x, c, U=torch.from_numpy(d['U.%d' % s]))
        losses.append(loss)
        if s == 0:
            assert abs(loss - float(d['loss.0'])) < 1e-6
            assert abs(ne - float(d['norm_enc.0'])) < 1e-3 * float(d['norm_enc.0'])
            assert abs(nd - float(d['norm_dec.0'])) < 1e-3 * float(d['norm_dec.0'])
            assert torch.equal(act, torch.from_numpy(d['enc_act.0']))
            # parameters after one Adam step, where the gradient is significant
            for k, v in tr.enc_sd.items():
                sig = np.abs(d['genc.' + k]) > 1e-6
                err = np.abs(v.numpy() - d['enc1.' + k])
                assert err.max() <= 2.1 * m['lr'] and (not sig.any() or err[sig].max() <= 0.05 * m['lr'])
        else:
            assert abs(loss - float(d['loss.%d' % s])) < 1e-4


def test_classifier_golden():
    d, m = load_golden('classifier_small.npz')
    p = {k: v.clone().requires_grad_(True) for k, v in sub_sd(d, 'clf.').items()}
    logits = O.speaker_classifier_forward(p, torch.from_numpy(d['x']), m['ns'], 0.0, m['seg_len'], training=True)
    loss = O.cross_entropy(logits, torch.from_numpy(d['y']))
    loss.backward()
    _close(logits.detach(), d['logits'], 2e-5)
    assert abs(loss.item() - float(d['loss'])) < 1e-6
    for k in p:
        _close(p[k].grad, d['g.' + k], 1e-4)


def test_mbv_contract():
    torch.manual_seed(0)
    logits = torch.randn(2, 16, 5)
    U = torch.rand(2, 5, 8, 2)
    act = O.mbv(logits, 8, U=U)
    G = O.gumbel_from_uniform(U)
    s = logits.permute(0, 2, 1).reshape(2, 5, 8, 2) + G
    ref = (s[..., 0] / 0.1 >= s[..., 1] / 0.1).float().permute(0, 2, 1)      # ties -> index 0
    assert torch.equal(act, ref)
    assert act.dtype == torch.float32 and set(act.unique().tolist()) <= {0.0, 1.0}


def test_known_answers():
    # pixel-shuffle index law out[b,c,2w+r] = in[b,2c+r,w]  (model/model.py:43-51)
    x = torch.arange(2 * 6 * 5, dtype=torch.float32).view(2, 6, 5)
    y = O.pixel_shuffle_1d(x)
    for c in range(3):
        for w in range(5):
            for r in range(2):
                assert y[1, c, 2 * w + r] == x[1, 2 * c + r, w]
    assert O.pad_amounts(4) == (2, 1) and O.pad_amounts(5) == (2, 2) and O.pad_amounts(1) == (0, 0)
    assert [O.out_len(t) for t in (9, 10, 127, 128, 129, 201, 254)] == [16, 16, 128, 128, 136, 208, 256]


def test_fragment_plan():
    # convert.py:139-168 restated; tail fragment drops the last frame, short tails are skipped
    assert O.fragment_plan(5, 128) == (9, [(0, 9)], 1)
    assert O.fragment_plan(9, 128) == (9, [(0, 9)], None)
    assert O.fragment_plan(128, 128) == (128, [(0, 128)], None)
    assert O.fragment_plan(129, 128) == (129, [(0, 128)], None)           # idx=0: 0+256>129 -> spec[0:-1]
    assert O.fragment_plan(255, 128) == (255, [(0, 254)], None)
    assert O.fragment_plan(256, 128) == (256, [(0, 128)], None)               # tail spec[128:-1] is 127 < seg_len: skipped
    assert O.fragment_plan(300, 128) == (300, [(0, 128), (128, 299)], None)
    assert O.fragment_plan(700, 128) == (700, [(0, 128), (128, 256), (256, 384), (384, 512), (512, 699)], None)
    assert O.encodings_text(np.array([[1., 0., 1.], [0., 0., 1.]])) == '1 0 1\n0 0 1\n'


def test_wav_sample_count_law():
    """Every sample wav the reference ships (docs/exp/**.wav) has 200*(n-1) samples with n a length the
    decoder can emit for some utterance under fragment_plan."""
    with open(os.path.join(GOLD, 'docs_exp_wav_samples.json')) as fh:
        counts = json.load(fh)
    assert len(counts) >= 20
    emit = set()
    for L in range(9, 900):
        _, frags, _ = O.fragment_plan(L, 128)
        emit.add(sum(O.out_len(b - a) for a, b in frags))
    synth = [n for k, n in counts.items() if not k.startswith('original') and 'orig' not in k]
    ok = [n for n in synth if n % 200 == 0 and (n // 200 + 1) in emit]
    assert len(ok) >= 0.5 * len(synth)        # trimmed / natural recordings need not obey the law


def test_vocoder_regression():
    d, _ = load_golden('vocoder_small.npz')
    S = O.stft(d['stft_in'])
    assert np.abs(S.real - d['stft_re']).max() < 1e-4 and np.abs(S.imag - d['stft_im']).max() < 1e-4
    y = O.istft(S)
    assert len(y) == 200 * (S.shape[1] - 1) and np.abs(y - d['istft_out']).max() < 1e-5
    assert np.abs(y[512:-512] - d['stft_in'][512:len(y) - 512]).max() < 1e-4   # STFT->iSTFT identity
    wav = O.spectrogram2wav(d['mag'], n_iter=8, do_trim=False)
    assert np.abs(wav - d['wav_iter8']).max() < 1e-3 * np.abs(d['wav_iter8']).max()
    win = torch.from_numpy(O.hann_padded())
    St = torch.stft(torch.from_numpy(d['stft_in']), 1024, 200, window=win, center=True, pad_mode='reflect',
                    return_complex=True).numpy()
    assert np.abs(S - St).max() < 5e-4


def test_vocoder_pieces_pinned_to_scipy():
    """What CAN be pinned in the vocoder oracle: scipy IS the reference's dependency for the de-preemphasis filter
    (`signal.lfilter([1], [1, -hp.preemphasis], wav)`, convert.py:60) and, through librosa, for the window
    (`scipy.signal.get_window('hann', win_length, fftbins=True)` centre-padded to n_fft); the STFT framing is checked
    against scipy.signal.stft on the reflect-padded signal.  librosa-specific conventions (istft normalisation, trim) stay
    'parity unpinned'."""
    import scipy.signal
    rng = np.random.RandomState(0)
    x = (rng.randn(140000) * 0.1).astype(np.float32)
    ref = scipy.signal.lfilter([1], [1, -0.97], x)
    got = O.de_preemphasis(x)
    assert got.dtype == ref.dtype == np.float64 and got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    w = scipy.signal.get_window('hann', 800, fftbins=True)
    pad = np.zeros(1024)
    pad[112:912] = w                                                 # librosa.util.pad_center(w, 1024)
    assert np.array_equal(O.hann_padded(), pad.astype(np.float32))
    # framing + transform: scipy.signal.stft without its own padding on the reflect-padded signal; 'spectrum' scaling
    # divides by sum(window)
    y = x[:200 * 50]
    yp = np.pad(y, 512, mode='reflect')
    _, _, Z = scipy.signal.stft(yp, window=pad, nperseg=1024, noverlap=1024 - 200, nfft=1024, boundary=None, padded=False,
                                return_onesided=True, scaling='spectrum')
    S = O.stft(y)
    assert S.shape == Z.shape == (513, 1 + len(y) // 200)
    assert np.abs(S - Z * pad.sum()).max() < 1e-4 * np.abs(S).max()
    # preprocess.py:240 pre-emphasis np.append(y[0], y[1:] - 0.97*y[:-1]) is lfilter([1, -0.97], [1], y)
    pre = np.append(y[0], y[1:] - O.PREEMPH * y[:-1])
    assert np.abs(pre - scipy.signal.lfilter([1, -0.97], [1], y.astype(np.float64))).max() < 1e-6
    # ... and de_preemphasis inverts it
    assert np.abs(O.de_preemphasis(pre) - y).max() < 1e-4


def test_stage2_golden():
    """Stage 2 oracle (PatchDiscriminator forward, WGAN-GP double backward, patchGAN D / G losses) against the vectors
    captured from the reference's own PatchDiscriminator + utils.calculate_gradients_penalty (oracle/make_golden.py)."""
    d, m = load_golden('stage2_small.npz')
    sd = O.synthetic_patch_sd(m['n_class'], m['seed'])
    hp = dict(ns=m['ns'], seg_len=m['seg_len'], dp=m['dp'], training=True, beta_dis=m['beta_dis'], beta_clf=m['beta_clf'],
              beta_gen=m['beta_gen'], lambda_=m['lambda_'])
    masks = [[torch.from_numpy(d['mask.%d.%d' % (p, l)]).float() for l in range(6)] for p in range(4)]
    x_t, x_dec, c, alpha = (torch.from_numpy(d[k]) for k in ('x_t', 'x_dec', 'c', 'alpha'))
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, w_dis, l_clf, gp, logits = O.patch_d_loss(p, x_t, x_dec, c, alpha, hp, masks=masks[:3])
    loss.backward()
    assert abs(w_dis.item() - float(d['w_dis'])) < 1e-5 and abs(gp.item() - float(d['gp'])) < 1e-5 * max(1.0, float(d['gp']))
    assert abs(loss.item() - float(d['loss_d'])) < 1e-5 * max(1.0, abs(float(d['loss_d'])))
    assert np.abs(logits.detach().numpy() - d['real_logits']).max() < 1e-5
    for k in p:
        g = p[k].grad.reshape(-1)
        ref = d['gD.sample.' + k]
        got = g[::max(1, g.numel() // 512)][:512].numpy()
        assert np.abs(got - ref).max() <= 1e-7 + 3e-4 * np.abs(ref).max(), k
        assert abs(g.double().norm().item() - float(d['gD.norm.' + k])) <= 3e-4 * float(d['gD.norm.' + k]) + 1e-9, k
    xo = x_dec.clone().requires_grad_(True)
    lg, _, _, fl = O.patch_g_loss(sd, xo, c, hp, masks=masks[3])
    dx, = torch.autograd.grad(lg, xo)
    assert abs(lg.item() - float(d['loss_g'])) < 1e-5 and np.abs(dx.numpy() - d['dx_gen']).max() <= 3e-4 * np.abs(d['dx_gen']).max()
