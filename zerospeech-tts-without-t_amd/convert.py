"""Inference / resynthesis surface of the reference's convert.py (lines 36-265, 303-360).

convert() / encode() reproduce the reference's fragment rule exactly (MIN_LEN zero padding, `spec[idx:-1]`
tail that drops the last frame, tails shorter than seg_len skipped, RuntimeError for a too-short input),
call Trainer.test_step / encoder_test_step per fragment and concatenate.  spectrogram2wav() /
griffin_lim() run the 300-iteration Griffin-Lim on the MI355X (zs_gl_* kernels: 1024-point FFTs in LDS),
batched over utterances by griffin_lim_batch(); de-preemphasis on the device, librosa.effects.trim restated
on the host (librosa itself is not required).  Utterances are independent: multi-GPU = replicas over a
sharded utterance list (zs_amd.parallel.shard_range), no collective.
"""
import ctypes
import glob
import json
import os

import numpy as np
import torch

from . import _lib as L
from .hps import Hps, hp

MIN_LEN = 9


# ---------------------------------------------------------------------------------------------------
# vocoder (convert.py:39-62)
# ---------------------------------------------------------------------------------------------------

def _device():
    if not torch.cuda.is_available():
        raise L.ZsError('zs_amd.convert needs an MI355X; there is no CPU fallback')
    return torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))


def griffin_lim_batch(mags, n_iter=None, device=None, impl=None, tile_frames=0, mag_padded=None):
    """mags: list of real [513, T_i] amplitude spectrograms (or mag_padded).  Returns (wav [n, 200*(T_max-1)] on the device, lengths, lens).
    X = S (zero phase); n_iter x { x = istft(X); E = stft(x); X = S * E / max(1e-8, |E|) }; x = istft(X)  (convert.py:39-52).
    impl 'fused' (default): zs_griffin_lim -- one fused kernel per iteration, the whole loop issued by one C call;
    impl 'split' (ZS_GL_IMPL=split): the older per-transform kernels (zs_gl_istft + zs_gl_stft_project per iteration), kept as
    the variant the tests compare the fused kernel with."""
    n_iter = hp.n_iter if n_iter is None else n_iter
    impl = impl or os.environ.get('ZS_GL_IMPL', 'fused')
    dev = device or _device()
    st = torch.cuda.current_stream(dev).cuda_stream
    if mag_padded is not None:                    # (amplitudes [n, T_max, 513] fp32 on the device, frame counts)
        mag, lens = mag_padded
        n, Tm = mag.shape[0], mag.shape[1]
    else:
        n = len(mags)
        lens = [int(m.shape[1]) for m in mags]
        Tm = max(lens)
    if min(lens) < 4:
        raise ValueError('griffin_lim needs at least 4 frames (reflect padding of n_fft//2 = 512 samples on 200*(T-1) samples)')
    if mag_padded is None:
        mag = torch.zeros(n, Tm, 513, dtype=torch.float32, device=dev)
        for i, m in enumerate(mags):
            t = m if torch.is_tensor(m) else torch.as_tensor(np.asarray(m, dtype=np.float32))
            mag[i, :lens[i]] = t.to(dev, torch.float32).t()
    lengths = torch.tensor(lens, dtype=torch.int32, device=dev)
    wav_ld = 200 * (Tm - 1)
    wav = torch.zeros(n, wav_ld, dtype=torch.float32, device=dev)
    if impl == 'fused':
        spec = torch.empty(n, Tm, 513, 2, dtype=torch.float32, device=dev)     # scratch: the first iteration reads X0 = S from `mag`
        spec_b = torch.empty_like(spec)
        host_lens = np.asarray(lens, dtype=np.int32)                # the chains of zs_griffin_lim are cut at equal frame counts
        S = L.STRUCTS['ZsGlIter'](mag=L.ptr(mag), lengths=L.ptr(lengths), n_utt=n, T_max=Tm, wav=L.ptr(wav), wav_ld=wav_ld,
                                  tile_frames=int(tile_frames), host_lengths=host_lens.ctypes.data)
        L.check(L.lib().zs_griffin_lim(ctypes.addressof(S), L.ptr(spec), L.ptr(spec_b), int(n_iter), st), 'zs_griffin_lim')
        return wav, lengths, lens
    if impl != 'split':
        raise ValueError("griffin_lim impl must be 'fused' or 'split'")
    spec = torch.zeros(n, Tm, 513, 2, dtype=torch.float32, device=dev)
    spec[..., 0] = mag                            # X0 = S, zero phase; rows past an utterance's length are never read
    frames = torch.empty(n, Tm, 1024, dtype=torch.float32, device=dev)
    ist = dict(spec=L.ptr(spec), mag=L.ptr(mag), lengths=L.ptr(lengths), n_utt=n, T_max=Tm, wav=L.ptr(wav), wav_ld=wav_ld,
               frames_ws=L.ptr(frames))
    stf = dict(wav=L.ptr(wav), wav_ld=wav_ld, mag=L.ptr(mag), lengths=L.ptr(lengths), n_utt=n, T_max=Tm, spec=L.ptr(spec))
    for _ in range(n_iter):
        L.call('zs_gl_istft', 'ZsGlIstft', st, **ist)
        L.call('zs_gl_stft_project', 'ZsGlStft', st, **stf)
    L.call('zs_gl_istft', 'ZsGlIstft', st, **ist)
    return wav, lengths, lens


def griffin_lim(spectrogram, n_iter=None, impl=None):
    """convert.py:39-52 for one [513, T] amplitude spectrogram."""
    wav, _, lens = griffin_lim_batch([spectrogram], n_iter=n_iter, impl=impl)
    return wav[0, :200 * (lens[0] - 1)].cpu().numpy()


def trim(wav, top_db=60, frame_length=2048, hop_length=512):
    """librosa.effects.trim defaults restated (RMS of centred frames, dB relative to the max).  The frames overlap four
    times (2048 / 512): the mean squares come from sums over 512-sample blocks, one pass over the signal."""
    y = np.asarray(wav, dtype=np.float64)
    if len(y) == 0:
        return y, (0, 0)
    yp = np.pad(y, frame_length // 2, mode='reflect')
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    if frame_length % hop_length == 0:
        r = frame_length // hop_length
        nb = n_frames + r - 1
        blocks = np.einsum('ij,ij->i', yp[:nb * hop_length].reshape(nb, hop_length), yp[:nb * hop_length].reshape(nb, hop_length))
        cs = np.concatenate(([0.0], np.cumsum(blocks)))
        mse = (cs[r:r + n_frames] - cs[:n_frames]) / frame_length
        mse = np.maximum(mse, 0.0)
    else:
        idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n_frames)[:, None]
        mse = np.mean(yp[idx] ** 2, axis=1)
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(np.maximum(1e-10, np.max(mse)))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return y[0:0], (0, 0)
    start, end = int(nz[0]) * hop_length, min(len(y), (int(nz[-1]) + 1) * hop_length)
    return y[start:end], (start, end)


TRIM_FRAME, TRIM_HOP = 2048, 512        # librosa.effects.trim defaults


def trim_bounds(mse, n, top_db=60, hop_length=TRIM_HOP):
    """(start, stop) of librosa.effects.trim given the frame mean squares (see trim())."""
    mse = np.asarray(mse, dtype=np.float64)
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(np.maximum(1e-10, np.max(mse)))
    nz = np.flatnonzero(db > -top_db)
    if nz.size == 0:
        return 0, 0
    return int(nz[0]) * hop_length, min(n, (int(nz[-1]) + 1) * hop_length)


def _enqueue_vocoder(t, lens, n_iter, do_trim, dev):
    """t: normalised magnitudes [n, T_max, 513] fp32 on the device.  Enqueues convert.py:55-62 and returns the function that
    waits for it and cuts the waveforms (librosa.effects.trim restated on device-computed frame statistics)."""
    st = torch.cuda.current_stream(dev).cuda_stream
    amp = torch.empty_like(t)
    L.check(L.lib().zs_gl_denormalize(L.ptr(t), L.ptr(amp), t.numel(), st), 'zs_gl_denormalize')     # convert.py:56-58
    wav, lengths, lens = griffin_lim_batch(None, n_iter=n_iter, device=dev, mag_padded=(amp, lens))
    L.check(L.lib().zs_gl_deemphasis(L.ptr(wav), wav.shape[1], L.ptr(lengths), len(lens), float(hp.preemphasis), st),
            'zs_gl_deemphasis')                               # signal.lfilter([1], [1, -0.97], wav)
    mse_h = None
    if do_trim and min(lens) > 6:                            # frame statistics of librosa.effects.trim on the device (convert.py:61)
        nf = 1 + wav.shape[1] // TRIM_HOP
        mse = torch.zeros(len(lens), nf, dtype=torch.float64, device=dev)
        L.check(L.lib().zs_gl_frame_mse(L.ptr(wav), wav.shape[1], L.ptr(lengths), len(lens), TRIM_FRAME, TRIM_HOP, L.ptr(mse), nf, st),
                'zs_gl_frame_mse')
        mse_h = _to_host(mse)
    w_h = _to_host(wav)
    from . import layers
    st_h = _to_host(layers.device_status(dev))                # the sticky GRU status word as of this batch (checked by the caller)
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream(dev))

    def finish():
        done.synchronize()                                    # this batch's copies only: later batches may already be enqueued
        finish.status = int(st_h[0])
        w = w_h.numpy()
        out = []
        for i, T in enumerate(lens):
            y = w[i, :200 * (T - 1)]
            if do_trim:
                if mse_h is not None:
                    a, b = trim_bounds(mse_h[i, :1 + len(y) // TRIM_HOP].numpy(), len(y))
                    y = y[a:b]
                else:
                    y, _ = trim(y)
            out.append(np.asarray(y, dtype=np.float32))
        return out
    return finish


def spectrogram2wav_batch(mags_tf, n_iter=None, do_trim=True):
    """Batched spectrogram2wav (convert.py:55-62): list of [T_i, 513] normalised magnitudes -> list of float32 wavs.
    One padded buffer, one H2D copy (unless the magnitudes are on the device already), one de-normalisation launch, one
    zs_griffin_lim call, one de-emphasis launch, one D2H copy; librosa.effects.trim (restated) from frame statistics computed on
    the device."""
    dev = _device()
    lens = [int(m.shape[0]) for m in mags_tf]
    n, Tm = len(lens), max(lens)
    if all(torch.is_tensor(m) and m.is_cuda for m in mags_tf):          # decoder outputs still on the device (encode_batch(to_host=False))
        t = torch.zeros(n, Tm, 513, dtype=torch.float32, device=dev)
        for i, m in enumerate(mags_tf):
            t[i, :lens[i]] = m
    else:
        host = np.zeros((n, Tm, 513), dtype=np.float32)
        for i, m in enumerate(mags_tf):
            host[i, :lens[i]] = np.asarray(m.cpu() if torch.is_tensor(m) else m, dtype=np.float32)
        t = torch.from_numpy(host).to(dev)
    return _enqueue_vocoder(t, lens, n_iter, do_trim, dev)()


def spectrogram2wav(mag, n_iter=None):
    return spectrogram2wav_batch([mag], n_iter=n_iter)[0]


# ---------------------------------------------------------------------------------------------------
# fragment drivers (convert.py:70-221)
# ---------------------------------------------------------------------------------------------------

def convert_x(x, c, trainer, enc_only, verbose=False):
    c_var = torch.from_numpy(np.array([c]))
    tensor = torch.from_numpy(np.expand_dims(x, axis=0)).type(torch.FloatTensor)
    converted, enc = trainer.test_step(tensor, c_var, enc_only=enc_only, verbose=verbose)
    return converted.squeeze(axis=0).transpose((1, 0)), enc.squeeze(axis=0).transpose((1, 0))


def encode_x(x, trainer):
    tensor = torch.from_numpy(np.expand_dims(x, axis=0)).type(torch.FloatTensor)
    return trainer.encoder_test_step(tensor).squeeze(axis=0).transpose((1, 0))


def get_trainer(hps_path, model_path, g_mode, enc_mode, clf_path=None):
    from .trainer import Trainer
    hps = Hps(hps_path).get_tuple()
    global MIN_LEN
    MIN_LEN = MIN_LEN if hps.enc_mode != 'gumbel_t' else hps.seg_len
    trainer = Trainer(hps, None, g_mode, enc_mode)
    trainer.load_model(model_path, load_model_list=hps.load_model_list, clf_path=clf_path)
    return trainer


def fragments(n_frames, seg_len):
    """(start, stop) slices the reference sends through the network for an utterance of n_frames >= MIN_LEN
    (convert.py:151-165): full seg_len pieces, then a tail `spec[idx:-1]` once idx + 2*seg_len > len; pieces
    shorter than seg_len are skipped."""
    out = []
    for idx in range(0, n_frames, seg_len):
        if idx + (seg_len * 2) > n_frames:
            start, stop = idx, n_frames - 1
        else:
            start, stop = idx, idx + seg_len
        if stop - start >= seg_len:
            out.append((start, stop))
        elif idx == 0:
            raise RuntimeError('Please check if input is too short!')
    return out


def _pad_min(spec):
    if len(spec) < MIN_LEN:
        padding = np.zeros((MIN_LEN - spec.shape[0], spec.shape[1]))
        return np.concatenate((spec, padding), axis=0), True
    return spec, False


def parse_encodings(encodings):
    return [' '.join([str(int(e)) for e in enc]) for enc in encodings]


def write_encodings(path, encodings):
    with open(path, 'w') as file:
        for enc in encodings:
            for i, e in enumerate(enc):
                file.write(str(int(e)) + (' ' if i < len(enc) - 1 else ''))
            file.write('\n')


def convert(trainer, seg_len, src_speaker_spec, src_speaker, tar_speaker, utt_id, speaker2id, result_dir, enc_only=True,
            save=['wav', 'enc']):
    src_speaker_spec, PADDED = _pad_min(src_speaker_spec)
    if len(src_speaker_spec) <= seg_len:
        converted_results, encodings = convert_x(src_speaker_spec, speaker2id[tar_speaker], trainer, enc_only=enc_only)
        if PADDED:
            encodings = encodings[:MIN_LEN // 8]
    else:
        converted_results, encodings = [], []
        for a, b in fragments(len(src_speaker_spec), seg_len):
            converted_x, enc = convert_x(src_speaker_spec[a:b], speaker2id[tar_speaker], trainer, enc_only=enc_only)
            converted_results.append(converted_x)
            encodings.append(enc)
        converted_results = np.concatenate(converted_results, axis=0)
        encodings = np.concatenate(encodings, axis=0)
    wav_data = spectrogram2wav(converted_results)
    if len(save) != 0:
        wav_path = None
        if 'wav' in save:
            wav_path = os.path.join(result_dir, f'{tar_speaker}_{utt_id}.wav')
            write_wav(wav_path, wav_data, hp.sr)
        if 'enc' in save:
            write_encodings(os.path.join(result_dir, f'{src_speaker}_{utt_id}.txt'), encodings)
        return wav_path, len(converted_results)
    return wav_data, encodings


def encode(src_speaker_spec, trainer, seg_len, s_speaker=None, utt_id=None, result_dir=None, save=True):
    if save:
        assert result_dir is not None and s_speaker is not None and utt_id is not None
    src_speaker_spec, PADDED = _pad_min(src_speaker_spec)
    if len(src_speaker_spec) <= seg_len:
        encodings = encode_x(src_speaker_spec, trainer)
        if PADDED:
            encodings = encodings[:MIN_LEN // 8]
    else:
        encodings = [encode_x(src_speaker_spec[a:b], trainer) for a, b in fragments(len(src_speaker_spec), seg_len)]
        encodings = np.concatenate(encodings, axis=0)
    if save:
        write_encodings(os.path.join(result_dir, f'{s_speaker}_{utt_id}.txt'), encodings)
    else:
        return encodings


_POOL = None


_STAGE_THREADS = 4


def _pool():
    """A few host threads for the memcpy-bound staging of batches into pinned memory."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=_STAGE_THREADS)
    return _POOL


def _stage_rows(arrays, dev):
    """list of [T_i, C] arrays -> (device fp32 tensor [sum T_i + 1, C] whose LAST row is zeros, row offsets).
    Host arrays: one pass over the bytes into a pinned buffer (torch's caching host allocator recycles it between calls),
    several threads, then ONE asynchronous H2D copy at the link rate -- a pageable .to(device) of the same bytes is 3-4 x
    slower.  Utterances already resident in HBM (torch tensors on `dev`): one concatenation on the device, no host pass."""
    C = int(arrays[0].shape[1])
    offs = np.concatenate(([0], np.cumsum([int(a.shape[0]) for a in arrays]))).astype(np.int64)
    total = int(offs[-1])
    if all(torch.is_tensor(a) and a.is_cuda for a in arrays):
        return torch.cat([a.to(dev, torch.float32) for a in arrays] + [torch.zeros(1, C, dtype=torch.float32, device=dev)], dim=0), offs
    host = torch.empty(total + 1, C, dtype=torch.float32, pin_memory=True)
    host[total].zero_()
    hn = host.numpy()
    src = [a.detach().cpu().float().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float32) for a in arrays]
    k = max(1, min(_STAGE_THREADS, len(src)))
    cuts = [len(src) * i // k for i in range(k + 1)]

    def put(g):                                   # numpy's copy loop (GIL released); torch's copy_ of many small slices is 3 x slower
        a, b = cuts[g], cuts[g + 1]
        if b > a:
            np.concatenate(src[a:b], axis=0, out=hn[offs[a]:offs[b]])
    list(_pool().map(put, range(k)))
    return host.to(dev, non_blocking=True), offs


def _to_host(t):
    """Device tensor -> numpy array backed by pinned memory (asynchronous D2H; the caller synchronises once for all of them)."""
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    return h


def _t8(n):
    return ((((n + 1) // 2 + 1) // 2) + 1) // 2


class _Enqueued(object):
    """What _enqueue_encode leaves behind: nothing here has been waited for."""
    __slots__ = ('n_utt', 'chunks', 'enc_host', 'dec_dev', 'dec_host', 'enc_done', 'dev')


def _enqueue_encode(specs, trainer, seg_len, decode_speakers, noise_fn, max_batch, dec_to_host):
    """Fragment rule + every Encoder / Decoder launch of a batch of utterances, WITHOUT a host synchronisation: the caller decides
    what else to put on the stream (the vocoder) before it waits."""
    trainer.set_eval()
    enc, dec = trainer.Encoder, trainer.Decoder
    dev = trainer.device
    items = []                                            # (utt, order, start, stop, truncate)
    padded = []
    for u, spec in enumerate(specs):
        if torch.is_tensor(spec) and spec.is_cuda:
            was_padded = spec.shape[0] < MIN_LEN
            if was_padded:
                spec = torch.nn.functional.pad(spec, (0, 0, 0, MIN_LEN - spec.shape[0]))
        else:
            spec, was_padded = _pad_min(np.asarray(spec, dtype=np.float32))
        padded.append(spec)
        if len(spec) <= seg_len:
            items.append((u, 0, 0, len(spec), (MIN_LEN // 8) if was_padded else None))
        else:
            for k, (a, b) in enumerate(fragments(len(spec), seg_len)):
                items.append((u, k, a, b, None))
    # Two kinds of batches instead of one per distinct length (the fragment rule leaves ~60 tail lengths per 64 utterances):
    # the full seg_len fragments as plain batches, everything else (tails of seg_len .. 2 seg_len - 2 frames, short utterances)
    # as RAGGED batches padded to their longest member, with per-sample lengths: the kernels pad / normalise / run the reverse
    # GRU at each sample's own end, so a fragment's result does not depend on what it is batched with.
    full = [it for it in items if it[3] - it[2] == seg_len]
    rest = sorted((it for it in items if it[3] - it[2] != seg_len), key=lambda it: it[2] - it[3])      # longest first
    chunks = [(full[lo:lo + max_batch], None) for lo in range(0, len(full), max_batch)]
    chunks += [(rest[lo:lo + max_batch], True) for lo in range(0, len(rest), max_batch)]
    # every utterance goes to the device once, whole; the fragments are gathered there (row index tables, a few hundred KB)
    rows, offs = _stage_rows(padded, dev)
    zero_row = int(offs[-1])
    q = _Enqueued()
    q.n_utt, q.dev, q.chunks, q.enc_host, q.dec_dev, q.dec_host = len(specs), dev, [], [], [], []
    for chunk, ragged in chunks:
        lens = [b - a for (_, _, a, b, _) in chunk]
        Tm = max(lens)
        idx = np.full((len(chunk), Tm), zero_row, dtype=np.int64)
        for i, (u, _, a, b, _) in enumerate(chunk):
            idx[i, :b - a] = np.arange(offs[u] + a, offs[u] + b)
        x = rows[torch.from_numpy(idx).to(dev, non_blocking=True)]                                     # [n, Tm, 513]
        G = noise_fn(len(chunk), _t8(Tm), enc.enc_size) if noise_fn is not None else None
        act, _ = enc(x.permute(0, 2, 1), G=G, lengths=(lens if ragged else None))
        xd = None
        if decode_speakers is not None:
            ch = torch.tensor([decode_speakers[u] for (u, _, _, _, _) in chunk], dtype=torch.int64)
            lens_p = [_t8(n) for n in lens]
            xd = dec(act, ch.to(dev, non_blocking=True), lengths=(lens_p if ragged else None)).permute(0, 2, 1).contiguous()
        q.chunks.append((chunk, lens))
        q.enc_host.append(_to_host(act.permute(0, 2, 1).contiguous()))
        q.dec_dev.append(xd)
        q.dec_host.append(_to_host(xd) if (xd is not None and dec_to_host) else None)
    q.enc_done = torch.cuda.Event()
    q.enc_done.record(torch.cuda.current_stream(dev))
    return q


def _assemble(q, want_dec):
    """Per-utterance encodings (and decoded spectrograms: 'host', 'device' or None) from the chunk outputs, in fragment order.
    The caller has waited for the copies."""
    enc_out, dec_out = {}, {}
    for ci, (chunk, lens) in enumerate(q.chunks):
        e = q.enc_host[ci].numpy()
        xd = None
        if want_dec == 'host':
            xd = q.dec_host[ci].numpy()
        elif want_dec == 'device':
            xd = q.dec_dev[ci]
        for i, (u, k, _, _, trunc) in enumerate(chunk):
            tp = _t8(lens[i])
            enc_out[(u, k)] = e[i][:trunc] if trunc is not None else e[i][:tp]
            if xd is not None:
                dec_out[(u, k)] = xd[i][:8 * tp]
    by_utt = {}
    for (u, k) in enc_out:
        by_utt.setdefault(u, []).append(k)
    encs, decs = [], ([] if want_dec else None)
    for u in range(q.n_utt):
        ks = sorted(by_utt[u])
        encs.append(np.concatenate([enc_out[(u, k)] for k in ks], axis=0))
        if decs is not None:
            parts = [dec_out[(u, k)] for k in ks]
            decs.append(np.concatenate(parts, axis=0) if want_dec == 'host' else (parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)))
    return encs, decs


def encode_batch(specs, trainer, seg_len, decode_speakers=None, noise_fn=None, max_batch=256, to_host=True):
    """Batched encode()/convert() for many utterances: the same fragments the reference would send through the network one by
    one (convert.py:151-165), as a few large batches on the GPU.
    specs: list of [T_i, 513] arrays (host arrays, or torch tensors already on the device).  decode_speakers: optional list of
    target speaker ids -> also returns the decoded spectrograms (enc_only path of convert()).  noise_fn(n_frag, T', E) -> Gumbel
    noise [n, T', E, 2] or None makes the stochastic discretiser reproducible (called once per batch; T' = encoded frames of the
    longest member).  Returns (encodings list, decoded list or None); to_host=False leaves the decoded spectrograms on the
    device (torch tensors [T_out, 513]) for spectrogram2wav_batch."""
    q = _enqueue_encode(specs, trainer, seg_len, decode_speakers, noise_fn, max_batch, dec_to_host=to_host)
    torch.cuda.current_stream(q.dev).synchronize()
    from . import layers
    layers.check_status(q.dev)                                # synchronised: a timed-out GRU pass raises here
    return _assemble(q, None if decode_speakers is None else ('host' if to_host else 'device'))


def resynth_batch(specs, trainer, seg_len, speakers, n_iter=None, do_trim=True, noise_fn=None, max_batch=256, defer=False):
    """test_encode + convert(enc_only) + spectrogram2wav for a batch of utterances as ONE enqueue (convert.py:151-165, 55-62):
    the fragments' Encoder / Decoder launches, the gather of the decoded fragments into the vocoder's padded [n, T_max, 513]
    input (one index kernel on the device), de-normalisation, the Griffin-Lim loop, de-emphasis, the trim statistics and the
    D2H copies go onto the stream back to back; the host assembles the encodings while the vocoder runs and waits ONCE.
    Returns (encodings list, wav list) -- the same values as encode_batch(...) followed by spectrogram2wav_batch(...).
    defer=True returns a function instead that waits for THIS batch (an event behind its last copy) and returns that pair: a
    serving loop enqueues batch i + 1 before it calls the function of batch i, so the host-side assembly and the D2H copies of one
    batch run under the GPU work of the next."""
    q = _enqueue_encode(specs, trainer, seg_len, speakers, noise_fn, max_batch, dec_to_host=False)
    dev = q.dev
    # row table of all decoded fragments + a zero row; utterance u's frames are its fragments' rows in order
    bases, total = [], 0
    for ci, (chunk, lens) in enumerate(q.chunks):
        bases.append(total)
        total += q.dec_dev[ci].shape[0] * q.dec_dev[ci].shape[1]
    frag_rows = {}
    for ci, (chunk, lens) in enumerate(q.chunks):
        ld = q.dec_dev[ci].shape[1]
        for i, (u, k, _, _, _) in enumerate(chunk):
            frag_rows.setdefault(u, []).append((k, bases[ci] + i * ld, 8 * _t8(lens[i])))
    ulens = []
    for u in range(q.n_utt):
        frag_rows[u].sort()
        ulens.append(sum(n for (_, _, n) in frag_rows[u]))
    Tm = max(ulens)
    idx = np.full((q.n_utt, Tm), total, dtype=np.int64)
    for u in range(q.n_utt):
        at = 0
        for (_, r0, n) in frag_rows[u]:
            idx[u, at:at + n] = np.arange(r0, r0 + n)
            at += n
    table = torch.cat([x.reshape(-1, x.shape[2]) for x in q.dec_dev] + [torch.zeros(1, q.dec_dev[0].shape[2], dtype=q.dec_dev[0].dtype, device=dev)], dim=0)
    t = table[torch.from_numpy(idx).to(dev, non_blocking=True)].float()                               # [n, Tm, 513]
    fin = _enqueue_vocoder(t, ulens, n_iter, do_trim, dev)

    def finish():
        q.enc_done.synchronize()                              # the encodings are on the host; the vocoder is still running
        encs, _ = _assemble(q, None)
        wavs = fin()
        if fin.status:                                        # a persistent GRU pass timed out before this batch's end
            from . import layers
            layers.check_status(dev)                          # synchronises, clears the word and raises
        return encs, wavs
    return finish if defer else finish()


def write_wav(path, wav, sr):
    """16-bit PCM mono wav (the reference uses soundfile.write(..., 'PCM_16'), convert.py:174)."""
    import wave
    pcm = np.clip(np.round(np.asarray(wav, dtype=np.float64) * 32767.0), -32768, 32767).astype('<i2')
    with wave.open(path, 'wb') as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(pcm.tobytes())


def _open_h5(path):
    """The preprocessed dataset: HDF5 when h5py is importable, else the .npz container zs_amd.preprocess writes."""
    from .dataloader import open_store
    return open_store(path, 'r')


def test_from_list(trainer, seg_len, synthesis_list, data_path, speaker2id_path, result_dir, enc_only, flag='test', run_asr=False):
    """convert.py:224-265 (the ASR scoring branch needs the network and is not reproduced)."""
    with open(speaker2id_path, 'r') as f_json:
        speaker2id = json.load(f_json)
    feeds = []
    with open(synthesis_list, 'r') as f:
        for line in f.readlines():
            line = line.split('\n')[0].split(' ')
            feeds.append({'s_id': line[0].split('/')[1].split('_')[0], 'utt_id': line[0].split('/')[1].split('_')[1], 't_id': line[1]})
    print('[Tester] - Number of files to be resynthesize: ', len(feeds))
    dir_path = os.path.join(result_dir, f'{flag}/')
    os.makedirs(dir_path, exist_ok=True)
    with _open_h5(data_path) as f_h5:
        for feed in feeds:
            convert(trainer, seg_len, src_speaker_spec=f_h5[f"test/{feed['s_id']}/{feed['utt_id']}/lin"][()], src_speaker=feed['s_id'],
                    tar_speaker=feed['t_id'], utt_id=feed['utt_id'], speaker2id=speaker2id, result_dir=dir_path, enc_only=enc_only,
                    save=['wav'])


SINGLE_FILES = {            # convert.py:308-317: the utterances --test_single knows per source speaker
    'S015': './data/english/train/unit/S015_0361841101.wav',
    'S119': './data/english/train/unit/S119_1561145062.wav',
    'S130': './data/english/test/S130_3516588097.wav',
    'S089': './data/english/test/S089_1810826781.wav',
    'S378': './data/surprise/test/S378_117437.wav',
}


def test_single(trainer, seg_len, speaker2id_path, result_dir, enc_only, s_speaker, t_speaker, filename=None):
    """convert.py:303-340: one wav -> spectrogram (zs_amd.preprocess, STFT on the GPU) -> convert() -> result.wav (PCM16) +
    result.txt.  The reference then scores both files with the Google Web Speech API (convert.py:96-113); that needs the
    network and is not reproduced.  `filename` overrides the per-speaker table (not in the reference signature)."""
    from .preprocess import get_spectrograms
    with open(speaker2id_path, 'r') as f_json:
        speaker2id = json.load(f_json)
    if filename is None:
        if s_speaker not in SINGLE_FILES:
            raise NotImplementedError('Please modify path manually!')
        filename = SINGLE_FILES[s_speaker]
    _, spec = get_spectrograms(filename)
    wav_data, encodings = convert(trainer, seg_len, src_speaker_spec=spec, src_speaker=s_speaker, tar_speaker=t_speaker, utt_id='',
                                  speaker2id=speaker2id, result_dir=result_dir, enc_only=enc_only, save=[])
    write_wav(os.path.join(result_dir, 'result.wav'), wav_data, hp.sr)
    write_encodings(os.path.join(result_dir, 'result.txt'), encodings)
    print('Testing on source speaker {} and target speaker {}, output shape: {}'.format(s_speaker, t_speaker, wav_data.shape))
    print('Comparing ASR result - skipped (the reference calls the Google Web Speech API here; no network)')
    return wav_data, encodings


def test_encode(trainer, seg_len, test_path, data_path, result_dir, flag='test'):
    """convert.py:342-360."""
    files = sorted(glob.glob(os.path.join(test_path, '*.wav')))
    feeds = [{'s_id': f.split('/')[-1].split('_')[0], 'utt_id': f.split('/')[-1].split('_')[1].split('.')[0]} for f in files]
    print('[Tester] - Number of files to encoded: ', len(feeds))
    dir_path = os.path.join(result_dir, f'{flag}/')
    os.makedirs(dir_path, exist_ok=True)
    with _open_h5(data_path) as f_h5:
        for feed in feeds:
            encode(f_h5[f"test/{feed['s_id']}/{feed['utt_id']}/lin"][()], trainer, seg_len, s_speaker=feed['s_id'],
                   utt_id=feed['utt_id'], result_dir=dir_path)
