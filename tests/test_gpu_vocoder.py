"""GPU: Griffin-Lim vocoder kernels and the segmenting inference drivers against the oracle.
The STFT/iSTFT restatement is "parity unpinned" against librosa itself (not installed; version unpinned by
the reference); it is pinned to torch.stft/istft in oracle/make_golden.py.  Tolerances: single STFT/iSTFT
1e-4 of scale; Griffin-Lim after 8 iterations 1e-3 of the waveform scale (north_star audio tolerance); at the
reference's 300 iterations the fixed point is compared through its spectral convergence."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cv():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import zs_amd  # noqa: F401
    from zs_amd import convert
    return convert


def test_stft_istft_kernels(cv):
    import zs_oracle as O
    from zs_amd import _lib as L
    d, _ = load_golden('vocoder_small.npz')
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    S = (d['stft_re'] + 1j * d['stft_im']).astype(np.complex64)          # [513, T]
    T = S.shape[1]
    spec = torch.zeros(1, T, 513, 2, device=dev)
    spec[0, :, :, 0] = torch.from_numpy(np.ascontiguousarray(S.real.T)).to(dev)
    spec[0, :, :, 1] = torch.from_numpy(np.ascontiguousarray(S.imag.T)).to(dev)
    lengths = torch.tensor([T], dtype=torch.int32, device=dev)
    wav = torch.zeros(1, 200 * (T - 1), device=dev)
    frames = torch.empty(1, T, 1024, device=dev)
    mag = torch.ones(1, T, 513, device=dev)
    L.call('zs_gl_istft', 'ZsGlIstft', st, spec=L.ptr(spec), mag=L.ptr(mag), lengths=L.ptr(lengths), n_utt=1, T_max=T, wav=L.ptr(wav),
           wav_ld=wav.shape[1], frames_ws=L.ptr(frames))
    torch.cuda.synchronize()
    ref = d['istft_out']
    assert np.abs(wav[0].cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()
    # forward: spec = mag * E/|E| with mag = |E_ref| reproduces E
    E = O.stft(ref)
    magE = torch.from_numpy(np.ascontiguousarray(np.abs(E).T)).to(dev).unsqueeze(0).contiguous()
    L.call('zs_gl_stft_project', 'ZsGlStft', st, wav=L.ptr(wav), wav_ld=wav.shape[1], mag=L.ptr(magE), lengths=L.ptr(lengths), n_utt=1,
           T_max=T, spec=L.ptr(spec))
    torch.cuda.synchronize()
    got = (spec[0, :, :, 0] + 1j * spec[0, :, :, 1]).cpu().numpy().T
    assert np.abs(got - E).max() < 2e-4 * np.abs(E).max()


@pytest.mark.parametrize('T,tile', [(4, 0), (5, 4), (17, 4), (40, 10), (53, 26), (131, 0), (131, 42)])
def test_fused_griffin_lim_iteration_vs_oracle(cv, T, tile):
    """zs_gl_iter (one fused kernel: istft -> overlap-add -> stft -> projection, 512-point complex FFT per wave) against the
    oracle's istft / stft on a random complex spectrogram, for every tile size class: single tile, many small tiles (halo
    frames from both neighbours), utterance ends inside a tile.  With mag = |E_ref| the projection reproduces E itself.
    Tolerance 2e-4 of the scale (fp32 transforms, different butterfly order than numpy's)."""
    import zs_oracle as O
    from zs_amd import _lib as L
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.RandomState(T)
    X = (rng.randn(513, T) + 1j * rng.randn(513, T)).astype(np.complex64)
    x_ref = O.istft(X)
    E = O.stft(x_ref)
    Tm = T + 3                                                      # T_max > T: rows past the length must stay untouched
    spec_in = torch.zeros(1, Tm, 513, 2, device=dev)
    spec_in[0, :T, :, 0] = torch.from_numpy(np.ascontiguousarray(X.real.T)).to(dev)
    spec_in[0, :T, :, 1] = torch.from_numpy(np.ascontiguousarray(X.imag.T)).to(dev)
    spec_out = torch.full((1, Tm, 513, 2), 7.0, device=dev)
    mag = torch.zeros(1, Tm, 513, device=dev)
    mag[0, :T] = torch.from_numpy(np.ascontiguousarray(np.abs(E).T)).to(dev)
    lengths = torch.tensor([T], dtype=torch.int32, device=dev)
    wav = torch.full((1, 200 * (Tm - 1)), 3.0, device=dev)
    L.call('zs_gl_iter', 'ZsGlIter', st, spec_in=L.ptr(spec_in), spec_out=L.ptr(spec_out), mag=L.ptr(mag), lengths=L.ptr(lengths), n_utt=1,
           T_max=Tm, wav=L.ptr(wav), wav_ld=wav.shape[1], tile_frames=tile)
    L.call('zs_gl_iter', 'ZsGlIter', st, spec_in=L.ptr(spec_in), spec_out=None, mag=L.ptr(mag), lengths=L.ptr(lengths), n_utt=1,
           T_max=Tm, wav=L.ptr(wav), wav_ld=wav.shape[1], tile_frames=tile)
    torch.cuda.synchronize()
    got = (spec_out[0, :T, :, 0] + 1j * spec_out[0, :T, :, 1]).cpu().numpy().T
    assert np.abs(got - E).max() < 2e-4 * np.abs(E).max(), np.abs(got - E).max() / np.abs(E).max()
    assert (spec_out[0, T:] == 7.0).all()
    w = wav[0].cpu().numpy()
    assert np.abs(w[:200 * (T - 1)] - x_ref).max() < 2e-4 * np.abs(x_ref).max()
    assert (w[200 * (T - 1):] == 3.0).all()


def test_fused_griffin_lim_equals_split_kernels_and_is_tile_invariant(cv):
    """Ragged batch, 12 iterations: the fused loop (zs_griffin_lim) agrees with the per-transform kernels it replaces to 1e-3 of
    the scale, and its result does not depend on the tile size bit for bit (the overlap-add order is fixed by frame index mod 4)."""
    rng = np.random.RandomState(9)
    mags = [np.abs(rng.randn(513, T)).astype(np.float32) * np.linspace(1, 0.01, 513, dtype=np.float32)[:, None] for T in (16, 61, 300, 97)]
    ref, _, lens = cv.griffin_lim_batch(mags, n_iter=12, impl='split')
    outs = [cv.griffin_lim_batch(mags, n_iter=12, impl='fused', tile_frames=f)[0] for f in (0, 10, 42)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    for i, T in enumerate(lens):
        a, b = outs[0][i, :200 * (T - 1)].cpu().numpy(), ref[i, :200 * (T - 1)].cpu().numpy()
        assert np.abs(a - b).max() < 1e-3 * np.abs(b).max(), (T, np.abs(a - b).max() / np.abs(b).max())
        assert (outs[0][i, 200 * (T - 1):] == 0).all()
    with pytest.raises(ValueError):
        cv.griffin_lim_batch([mags[0][:, :3]], n_iter=1)


def test_spectrogram2wav_vs_oracle(cv):
    import zs_oracle as O
    d, _ = load_golden('vocoder_small.npz')
    wav = cv.spectrogram2wav_batch([d['mag']], n_iter=8, do_trim=False)[0]
    ref = d['wav_iter8']
    assert wav.shape == ref.shape == (200 * (40 - 1),)
    assert np.abs(wav - ref).max() < 1e-3 * np.abs(ref).max(), np.abs(wav - ref).max() / np.abs(ref).max()
    # ragged batch: every utterance equals its own single-utterance run (zero-padded frames never leak)
    rng = np.random.RandomState(3)
    mags = [np.clip(rng.rand(T, 513).astype(np.float32), 1e-8, 1) for T in (17, 40, 29)]
    batch = cv.spectrogram2wav_batch(mags, n_iter=5, do_trim=False)
    for m, w in zip(mags, batch):
        single = cv.spectrogram2wav_batch([m], n_iter=5, do_trim=False)[0]
        assert w.shape == (200 * (m.shape[0] - 1),) and np.array_equal(w, single)
    o = O.spectrogram2wav(mags[0], n_iter=5, do_trim=False)
    assert np.abs(batch[0] - o).max() < 1e-3 * np.abs(o).max()


def _speechlike(n_frames, seed):
    """A harmonic + noise signal with a moving pitch and formant-like envelope: 200*(n_frames-1) samples at 16 kHz."""
    rng = np.random.RandomState(seed)
    n = 200 * (n_frames - 1)
    t = np.arange(n) / 16000.0
    f0 = 120 + 40 * np.sin(2 * np.pi * 0.7 * t)
    ph = 2 * np.pi * np.cumsum(f0) / 16000.0
    y = sum((0.5 / k) * np.sin(k * ph + rng.rand() * 6.28) * (1 + 0.5 * np.sin(2 * np.pi * (0.3 + 0.1 * k) * t)) for k in range(1, 24))
    y = y * (0.2 + 0.8 * (np.sin(2 * np.pi * 1.3 * t) > -0.3)) + 0.02 * rng.randn(n)
    return (0.2 * y / np.abs(y).max()).astype(np.float32)


def test_deemphasis_kernel_vs_scipy_lfilter(cv):
    """zs_gl_deemphasis on a 140 000-sample row (an 8.75 s utterance) against scipy.signal.lfilter([1], [1, -0.97], x), the
    reference's own call (convert.py:60).  The recursion y[n] = x[n] + 0.97 y[n-1] is a scan on the GPU (fp32)."""
    import scipy.signal
    from zs_amd import _lib as L
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(1)
    T = 701
    n = 200 * (T - 1)
    x = (rng.randn(2, n) * 0.1).astype(np.float32)
    x[1, 90000:] = 0                                               # second row shorter: only its own length is filtered
    lens = torch.tensor([T, 451], dtype=torch.int32, device=dev)
    w = torch.from_numpy(x).to(dev)
    L.check(L.lib().zs_gl_deemphasis(L.ptr(w), n, L.ptr(lens), 2, 0.97, torch.cuda.current_stream().cuda_stream), 'zs_gl_deemphasis')
    got = w.cpu().numpy()
    for i, m in enumerate((n, 200 * 450)):
        ref = scipy.signal.lfilter([1], [1, -0.97], x[i, :m].astype(np.float64))
        err = np.abs(got[i, :m] - ref).max() / np.abs(ref).max()
        assert err < 1e-3, (i, err)
    print('deemphasis rel err', err)


@pytest.mark.parametrize('n_iter', [8, 60])
def test_griffin_lim_audio_700_frames_vs_oracle(cv, n_iter):
    """A 700-frame utterance (the longest of BASELINE config 4): audio of the GPU Griffin-Lim against the oracle's at the
    north_star tolerance, 1e-3 of the waveform scale, after 8 and after 60 iterations."""
    import zs_oracle as O
    y = _speechlike(700, 2)
    S = np.abs(O.stft(y)).astype(np.float32)                       # [513, 700]
    w_gpu = cv.griffin_lim(S, n_iter=n_iter)
    w_ref = O.griffin_lim(S, n_iter=n_iter)
    assert w_gpu.shape == w_ref.shape == (200 * 699,)
    err = np.abs(w_gpu - w_ref).max() / np.abs(w_ref).max()
    rms = np.sqrt(np.mean((w_gpu - w_ref) ** 2)) / np.sqrt(np.mean(w_ref ** 2))
    print('GL %d iterations, 700 frames: max err %.3g of scale, rms err %.3g' % (n_iter, err, rms))
    assert err < 1e-3, err


def test_griffin_lim_audio_at_hp_n_iter_300_vs_oracle(cv):
    """hp.n_iter = 300 (hps/hps.py:31), the product setting: the AUDIO of the GPU Griffin-Lim against the oracle's after all 300
    iterations, 350 speech-like frames, at the north_star tolerance (1e-3 of the waveform scale)."""
    import zs_oracle as O
    y = _speechlike(350, 7)
    S = np.abs(O.stft(y)).astype(np.float32)
    w_gpu = cv.griffin_lim(S, n_iter=300)
    w_ref = O.griffin_lim(S, n_iter=300)
    assert w_gpu.shape == w_ref.shape == (200 * 349,)
    err = np.abs(w_gpu - w_ref).max() / np.abs(w_ref).max()
    rms = np.sqrt(np.mean((w_gpu - w_ref) ** 2)) / np.sqrt(np.mean(w_ref ** 2))
    print('GL 300 iterations, 350 frames: max err %.3g of scale, rms err %.3g' % (err, rms))
    assert err < 1e-3, err


def test_trim_statistics_on_the_device_equal_the_host_trim(cv):
    """zs_gl_frame_mse + trim_bounds (librosa.effects.trim restated, convert.py:61) against the host-side trim() on signals with
    silent ends, lengths that are and are not multiples of the hop, and an all-quiet signal; through spectrogram2wav_batch the
    two paths return identical samples."""
    from zs_amd import _lib as L
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(4)
    Ts = [700, 351, 40, 8, 123]
    n_max = 200 * (max(Ts) - 1)
    x = np.zeros((len(Ts), n_max), dtype=np.float32)
    for i, T in enumerate(Ts):
        n = 200 * (T - 1)
        y = (rng.randn(n) * 0.1).astype(np.float32)
        lo, hi = int(0.13 * n), int(0.81 * n)
        y[:lo] *= 1e-5; y[hi:] *= 1e-5                          # "silence" 100 dB down at both ends
        x[i, :n] = y
    x[3, :] *= 1e-7                                             # all quiet: everything is within 60 dB of its own maximum
    w = torch.from_numpy(x).to(dev)
    lens = torch.tensor(Ts, dtype=torch.int32, device=dev)
    nf = 1 + n_max // cv.TRIM_HOP
    mse = torch.zeros(len(Ts), nf, dtype=torch.float64, device=dev)
    L.check(L.lib().zs_gl_frame_mse(L.ptr(w), n_max, L.ptr(lens), len(Ts), cv.TRIM_FRAME, cv.TRIM_HOP, L.ptr(mse), nf,
                                    torch.cuda.current_stream().cuda_stream), 'zs_gl_frame_mse')
    m = mse.cpu().numpy()
    for i, T in enumerate(Ts):
        n = 200 * (T - 1)
        _, (a, b) = cv.trim(x[i, :n])
        assert cv.trim_bounds(m[i, :1 + n // cv.TRIM_HOP], n) == (a, b), (T, cv.trim_bounds(m[i, :1 + n // cv.TRIM_HOP], n), (a, b))
        assert 0 <= a < b <= n


def test_griffin_lim_300_iterations_converges_like_oracle(cv):
    """n_iter = 300 (hps/hps.py:31): compare spectral convergence |STFT(x)| vs target of GPU and oracle results."""
    import zs_oracle as O
    rng = np.random.RandomState(5)
    t = np.arange(200 * 30) / 16000.0
    y = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 1320 * t) + 0.01 * rng.randn(len(t))).astype(np.float32)
    S = np.abs(O.stft(y))                                 # consistent magnitude spectrogram [513, 31]
    w_gpu = cv.griffin_lim(S, n_iter=300)
    w_ref = O.griffin_lim(S, n_iter=300)
    def sc(w):
        return np.linalg.norm(np.abs(O.stft(w)) - S) / np.linalg.norm(S)
    a, b = sc(w_gpu), sc(w_ref)
    print('spectral convergence gpu %.4g oracle %.4g' % (a, b))
    assert w_gpu.shape == w_ref.shape and a < 1.5 * b + 1e-3


def test_convert_and_encode_pipeline_vs_oracle(cv, tmp_path):
    """Trainer.test_step / encoder_test_step through convert()/encode() on a 300-frame utterance: fragment rule,
    encodings text, output length law 200*(T_out-1)."""
    import zs_oracle as O
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    torch.manual_seed(0)
    hps = make_hps(enc_size=8, emb_size=32, n_speakers=4, n_target_speakers=2, g_mode='targeted_residual')
    tr = Trainer(hps, None, 'targeted_residual', 'multilabel_binary', log_dir=str(tmp_path / 'log'), dtype='fp32')
    rng = np.random.RandomState(1)
    spec = np.clip(rng.rand(300, 513).astype(np.float32), 1e-8, 1)
    esd = {k: v.detach().cpu() for k, v in tr.Encoder.state_dict().items()}
    dsd = {k: v.detach().cpu() for k, v in tr.Decoder.state_dict().items()}
    # encode(): deterministic noise through U is not part of the reference signature, so compare logits-level pieces:
    _, frags, _ = O.fragment_plan(300, 128)
    assert cv.fragments(300, 128) == frags == [(0, 128), (128, 299)]
    outs = []
    for a, b in frags:
        x = torch.from_numpy(spec[a:b]).unsqueeze(0)
        U = torch.rand(1, O.out_len(b - a) // 8, 8, 2)
        G = O.gumbel_from_uniform(U)
        xd, enc = tr.test_step(x, torch.tensor([1]), enc_only=True, verbose=False, G=G)
        with torch.no_grad():
            o_act, _ = O.encoder_forward(esd, x.permute(0, 2, 1), hps.ns, hps.enc_dp, 8, 128, G=G)
            o_xd = O.decoder_forward(dsd, torch.from_numpy(enc), torch.tensor([1]), hps.ns, 128)
        assert xd.shape == (1, 513, O.out_len(b - a))
        assert (enc != o_act.numpy()).mean() < 0.02
        assert np.abs(xd - o_xd.numpy()).max() < 1e-3
        outs.append(xd[0].T)
    wav_data, encodings = cv.convert(tr, 128, spec, 'S1', 'V1', 'u1', {'V1': 1}, str(tmp_path), enc_only=True, save=[])
    n_out = sum(o.shape[0] for o in outs)
    assert encodings.shape == (n_out // 8, 8) and set(np.unique(encodings)) <= {0.0, 1.0}
    assert len(wav_data) <= 200 * (n_out - 1) and wav_data.dtype == np.float32
    enc_only = cv.encode(spec, tr, 128, save=False)
    assert enc_only.shape == encodings.shape
    cv.write_encodings(str(tmp_path / 'e.txt'), encodings)
    assert open(str(tmp_path / 'e.txt')).read() == O.encodings_text(encodings)
    short = cv.encode(spec[:5], tr, 128, save=False)                      # < MIN_LEN: zero-padded to 9, encoding truncated to 1 row
    assert short.shape == (1, 8)
    tr.save_model(str(tmp_path / 'm.pth'), 'ae', 1000)
    tr2 = Trainer(hps, None, 'targeted_residual', 'multilabel_binary', log_dir=str(tmp_path / 'log'), dtype='fp32')
    tr2.load_model(str(tmp_path / 'm.pth-ae-1000'), hps.load_model_list, verbose=False)
    for (k, a), (_, b) in zip(tr.Decoder.state_dict().items(), tr2.Decoder.state_dict().items()):
        assert torch.equal(a, b)


def test_batched_encode_resynthesis_64_utterances(cv, tmp_path):
    """BASELINE config 4 shape: 64 utterances of 200..700 frames, batched fragments + batched Griffin-Lim.
    * batched encodings / spectrograms equal the per-fragment (reference-style) path given the same Gumbel noise
    * a sample of utterances is checked against the oracle (bits: identical up to near-ties; spectrogram 1e-3)
    * output lengths follow the fragment rule and the 200*(T-1) sample law."""
    import zs_oracle as O
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    torch.manual_seed(1)
    hps = make_hps(enc_size=16, emb_size=32, n_speakers=4, n_target_speakers=2)
    tr = Trainer(hps, None, 'targeted_residual', 'multilabel_binary', log_dir=str(tmp_path / 'log'), dtype='fp32')
    rng = np.random.RandomState(0)
    lens = rng.randint(200, 701, size=64)
    specs = [np.clip(rng.rand(int(n), 513).astype(np.float32), 1e-8, 1) for n in lens]
    spk = [int(rng.randint(0, 4)) for _ in specs]
    store = {}

    def noise_fn(n, Tp, E):                                   # deterministic noise keyed by (Tp, call count)
        g = torch.Generator().manual_seed(1000 * Tp + len([k for k in store if k[0] == Tp]))
        G = O.gumbel_from_uniform(torch.rand(n, Tp, E, 2, generator=g))
        store[(Tp, len([k for k in store if k[0] == Tp]))] = G
        return G

    encs, decs = cv.encode_batch(specs, tr, 128, decode_speakers=spk, noise_fn=noise_fn)
    assert len(encs) == 64 and len(decs) == 64
    for n, e, d in zip(lens, encs, decs):
        frags = O.fragment_plan(int(n), 128)[1]
        t_out = sum(O.out_len(b - a) for a, b in frags)
        assert d.shape == (t_out, 513) and e.shape == (t_out // 8, 16) and set(np.unique(e)) <= {0.0, 1.0}
    # oracle on 3 sampled 128-frame fragments with the very noise rows the batch used
    esd = {k: v.detach().cpu() for k, v in tr.Encoder.state_dict().items()}
    dsd = {k: v.detach().cpu() for k, v in tr.Decoder.state_dict().items()}
    G128 = store[(16, 0)]
    order = [u for u in range(64)]                            # fragments of length 128 in batch order: (u, k) sorted by utterance
    frag128 = [(u, k, a, b) for u in order for k, (a, b) in enumerate(O.fragment_plan(int(lens[u]), 128)[1]) if b - a == 128]
    flips = total = 0
    for i in (0, 7, 100):
        u, k, a, b = frag128[i]
        x = torch.from_numpy(specs[u][a:b]).unsqueeze(0).permute(0, 2, 1)
        with torch.no_grad():
            o_act, _ = O.encoder_forward(esd, x, hps.ns, hps.enc_dp, 16, 128, G=G128[i:i + 1])
            o_dec = O.decoder_forward(dsd, torch.from_numpy(encs[u][16 * k:16 * k + 16].T.copy()).unsqueeze(0), torch.tensor([spk[u]]), hps.ns, 128)
        flips += int((o_act[0].numpy().T != encs[u][16 * k:16 * k + 16]).sum()); total += 16 * 16
        assert np.abs(o_dec[0].numpy().T - decs[u][128 * k:128 * k + 128]).max() < 1e-3
    assert flips <= total // 50
    wavs = cv.spectrogram2wav_batch(decs[:16], n_iter=20, do_trim=False)
    for d, w in zip(decs[:16], wavs):
        assert w.shape == (200 * (d.shape[0] - 1),) and np.isfinite(w).all()
    # device-resident hand-over (encode_batch(to_host=False) -> spectrogram2wav_batch): same spectrograms, same waveforms
    store.clear()
    encs_d, decs_d = cv.encode_batch(specs, tr, 128, decode_speakers=spk, noise_fn=noise_fn, to_host=False)
    assert all(torch.is_tensor(d) and d.is_cuda for d in decs_d)
    for e, e2, d, d2 in zip(encs, encs_d, decs, decs_d):
        assert (e == e2).all() and np.array_equal(d, d2.cpu().numpy())
    wavs_d = cv.spectrogram2wav_batch(decs_d[:16], n_iter=20, do_trim=False)
    for w, w2 in zip(wavs, wavs_d):
        assert np.array_equal(w, w2)
    # the whole pipeline as ONE enqueue (resynth_batch), from host arrays and from utterances already resident in HBM:
    # the same encodings and waveforms as encode_batch + spectrogram2wav_batch on the same utterances and noise
    for do_trim in (False, True):
        store.clear()
        e16, d16 = cv.encode_batch(specs[:16], tr, 128, decode_speakers=spk[:16], noise_fn=noise_fn)
        w16 = cv.spectrogram2wav_batch(d16, n_iter=20, do_trim=do_trim)
        for src in (specs[:16], [torch.from_numpy(s).cuda() for s in specs[:16]]):
            store.clear()
            e_r, w_r = cv.resynth_batch(src, tr, 128, spk[:16], n_iter=20, do_trim=do_trim, noise_fn=noise_fn)
            assert len(e_r) == 16 and len(w_r) == 16
            for a, b, w, w2 in zip(e16, e_r, w16, w_r):
                assert np.array_equal(a, b) and w.dtype == np.float32 and np.array_equal(w, w2)
    # a short utterance (< MIN_LEN frames, zero-padded by the fragment rule) resident on the device takes the same path
    short = [np.clip(rng.rand(5, 513).astype(np.float32), 1e-8, 1), specs[0]]
    store.clear()
    e_h, _ = cv.encode_batch(short, tr, 128, decode_speakers=[0, 1], noise_fn=noise_fn)
    store.clear()
    e_d, _ = cv.encode_batch([torch.from_numpy(s).cuda() for s in short], tr, 128, decode_speakers=[0, 1], noise_fn=noise_fn)
    assert all(np.array_equal(a, b) for a, b in zip(e_h, e_d)) and e_h[0].shape[0] == 1


def test_encode_batch_draws_fresh_noise_and_decodes_its_own_bits(cv, tmp_path):
    """encode_batch without injected noise (the product path of --test / --test_encode): for every call the decoded
    spectrograms equal the Decoder applied to the encodings that call returned (the decoder is deterministic), the
    encodings are bits, and the Gumbel noise is fresh at every call (reference: noise is drawn in eval mode too)."""
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    torch.manual_seed(2)
    hps = make_hps(enc_size=16, emb_size=32, n_speakers=4, n_target_speakers=2)
    tr = Trainer(hps, None, 'targeted_residual', 'multilabel_binary', log_dir=str(tmp_path / 'log'), dtype='fp32')
    rng = np.random.RandomState(1)
    lens = [300, 300, 131, 520, 129, 9, 200]
    specs = [np.clip(rng.rand(n, 513).astype(np.float32), 1e-8, 1) for n in lens]
    spk = [1, 0, 3, 2, 1, 0, 2]
    dev = tr.device
    prev = None
    for call in range(4):
        encs, decs = cv.encode_batch(specs, tr, 128, decode_speakers=spk)
        for u, (e, d) in enumerate(zip(encs, decs)):
            assert set(np.unique(e)) <= {0.0, 1.0} and d.shape[1] == 513 and e.shape[0] * 8 == d.shape[0]
        # utterance 2 (131 frames -> one 130-frame fragment): decode its returned bits eagerly
        e = torch.from_numpy(encs[2].T[None]).to(dev)                       # [1, E, T']
        xd = tr.Decoder(e, torch.tensor([spk[2]], device=dev))[0].T.cpu().numpy()
        assert np.abs(xd - decs[2]).max() < 1e-6
        if prev is not None:
            assert any((a != b).any() for a, b in zip(prev, encs)), 'the Gumbel noise did not change between calls'
        prev = encs


def test_resynth_batch_small_and_short_inputs(cv, tmp_path):
    """resynth_batch on the corner shapes of the fragment rule: a single utterance shorter than seg_len (one ragged chunk, no full
    fragment), one shorter than MIN_LEN (zero-padded, encodings truncated), one that is an exact multiple of seg_len + 1 (the rule
    drops the last frame: only full fragments) -- each equal to encode_batch + spectrogram2wav_batch."""
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    torch.manual_seed(2)
    hps = make_hps(enc_size=16, emb_size=32, n_speakers=4, n_target_speakers=2)
    tr = Trainer(hps, None, 'targeted_residual', 'multilabel_binary', log_dir=str(tmp_path / 'log'), dtype='fp32')
    rng = np.random.RandomState(3)

    def noise_fn(n, Tp, E):
        g = torch.Generator().manual_seed(77 + Tp)
        return -torch.log(-torch.log(torch.rand(n, Tp, E, 2, generator=g) + 1e-20) + 1e-20)

    for lens in ([100], [5, 257], [385, 129, 64]):
        specs = [np.clip(rng.rand(n, 513).astype(np.float32), 1e-8, 1) for n in lens]
        spk = [i % 4 for i in range(len(lens))]
        e0, d0 = cv.encode_batch(specs, tr, 128, decode_speakers=spk, noise_fn=noise_fn)
        w0 = cv.spectrogram2wav_batch(d0, n_iter=6, do_trim=False)
        e1, w1 = cv.resynth_batch(specs, tr, 128, spk, n_iter=6, do_trim=False, noise_fn=noise_fn)
        for a, b, w, w2 in zip(e0, e1, w0, w1):
            assert np.array_equal(a, b) and np.array_equal(w, w2) and np.isfinite(w2).all()
    # two batches in flight (defer=True: batch 2 is enqueued before batch 1 is waited for) give what they give one at a time
    sa = [np.clip(rng.rand(n, 513).astype(np.float32), 1e-8, 1) for n in (300, 140)]
    sb = [np.clip(rng.rand(n, 513).astype(np.float32), 1e-8, 1) for n in (129, 260, 90)]
    ra = cv.resynth_batch(sa, tr, 128, [0, 1], n_iter=6, noise_fn=noise_fn)
    rb = cv.resynth_batch(sb, tr, 128, [2, 3, 0], n_iter=6, noise_fn=noise_fn)
    fa = cv.resynth_batch(sa, tr, 128, [0, 1], n_iter=6, noise_fn=noise_fn, defer=True)
    fb = cv.resynth_batch(sb, tr, 128, [2, 3, 0], n_iter=6, noise_fn=noise_fn, defer=True)
    for (e_ref, w_ref), (e_got, w_got) in ((ra, fa()), (rb, fb())):
        assert all(np.array_equal(a, b) for a, b in zip(e_ref, e_got)) and all(np.array_equal(a, b) for a, b in zip(w_ref, w_got))
