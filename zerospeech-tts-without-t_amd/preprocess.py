"""Feature extraction and dataset files feeding the --train_ae path (reference preprocess.py).

* `get_spectrograms(sound_file)` / `get_spectrograms_batch(wavs)` -- preprocess.py:227-258: trim -> pre-emphasis -> STFT ->
  |.| -> mel(80) -> dB -> normalise, with the STFT / dB / mel product on the GPU (`zs_pre_spectrogram`, `zs_pre_mel`); the trim
  (librosa.effects.trim defaults) is host NumPy.  Sound files are read with scipy (16-bit / float wav at hp.sr; the reference's
  librosa.load would resample, which is not reproduced: other rates raise).
* `make_dataset`, `Sampler`, `make_samples`, `preprocess` -- preprocess.py:26-225: the `{train,test}/<speaker>/<utt>/{lin,mel}`
  float32 layout and the `[{speaker, i, t}]` index JSON.  The container is HDF5 when h5py is importable (the reference's
  format); otherwise the same keys go into an .npz archive (`NpzStore`), which `dataloader.Dataset` also reads.
"""
import glob
import json
import os
import random
from collections import defaultdict, namedtuple

import numpy as np
import torch

from . import _lib as L
from .convert import trim
from .dataloader import NpzStore, open_store
from .hps import hp


def mel_basis(sr=None, n_fft=None, n_mels=None):
    """librosa.filters.mel(sr, n_fft, n_mels) defaults (Slaney scale, area-normalised triangles), float32 [n_mels, 1+n_fft//2]."""
    sr, n_fft, n_mels = sr or hp.sr, n_fft or hp.n_fft, n_mels or hp.n_mels
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-12) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


_BASIS = {}


def get_spectrograms_batch(wavs, do_trim=True, device=None):
    """wavs: list of 1-D float arrays at hp.sr.  Returns a list of (mel [T, 80], mag [T, 513]) float32 NumPy pairs."""
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    if dev.type != 'cuda':
        raise L.ZsError('zs_amd.preprocess runs on an MI355X only (there is no CPU path)')
    ys = []
    for w in wavs:
        y = np.asarray(w, dtype=np.float32)
        if do_trim:
            y = trim(y)[0].astype(np.float32)                        # librosa.effects.trim(y)   preprocess.py:238
        if len(y) < 2:
            raise ValueError('utterance too short after trimming')
        ys.append(y)
    n = len(ys)
    lens = np.array([len(y) for y in ys], dtype=np.int32)
    ld = int(lens.max())
    T_max = 1 + ld // hp.hop_length
    host = np.zeros((n, ld), dtype=np.float32)
    for i, y in enumerate(ys):
        host[i, :len(y)] = y
    wav = torch.from_numpy(host).to(dev)
    ns = torch.from_numpy(lens).to(dev)
    mag = torch.zeros(n, T_max, 513, dtype=torch.float32, device=dev)
    amp = torch.zeros(n, T_max, 513, dtype=torch.float32, device=dev)
    mel = torch.zeros(n, T_max, hp.n_mels, dtype=torch.float32, device=dev)
    key = (dev.index, hp.sr, hp.n_fft, hp.n_mels)
    if key not in _BASIS:
        _BASIS[key] = torch.from_numpy(mel_basis()).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    L.call('zs_pre_spectrogram', 'ZsPreSpec', st, wav=L.ptr(wav), wav_ld=ld, n_samples=L.ptr(ns), n_utt=n, T_max=T_max,
           preemph=hp.preemphasis, ref_db=float(hp.ref_db), max_db=float(hp.max_db), mag=L.ptr(mag), mag_ld=513, amp=L.ptr(amp))
    fn = L.lib().zs_pre_mel
    L.check(fn(L.ptr(amp), L.ptr(_BASIS[key]), L.ptr(mel), n * T_max, hp.n_mels, float(hp.ref_db), float(hp.max_db), st), 'zs_pre_mel')
    mag_h, mel_h = mag.cpu().numpy(), mel.cpu().numpy()
    out = []
    for i in range(n):
        T = 1 + int(lens[i]) // hp.hop_length
        out.append((mel_h[i, :T].copy(), mag_h[i, :T].copy()))
    return out


def load_wav(sound_file):
    """16 kHz mono waveform in [-1, 1] (librosa.load(sound_file, sr=hp.sr) for files that already are at hp.sr)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(sound_file)
    if sr != hp.sr:
        raise ValueError('%s is sampled at %d Hz; resampling to %d Hz (librosa.load) is not reproduced' % (sound_file, sr, hp.sr))
    if data.ndim > 1:
        data = data.mean(axis=1)
    if data.dtype == np.int16:
        data = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float32) / 2147483648.0
    return data.astype(np.float32)


def get_spectrograms(sound_file):
    """preprocess.py:227-258: (mel [T, n_mels], mag [T, 1+n_fft/2]) normalised log spectrograms of a sound file."""
    return get_spectrograms_batch([load_wav(sound_file)])[0]


def make_dataset(grps, seg_len, root_dir, make_test=False, pad=True, batch=32):
    """preprocess.py:77-107: every `<speaker>_<segment>.wav` under root_dir -> `<speaker>/<segment>/{mel,lin}`."""
    filenames = sorted(glob.glob(os.path.join(root_dir, '*_*.wav')))
    grp = grps[1] if make_test else grps[0]
    print('Number of speakers: ', len({os.path.basename(f).split('_')[0] for f in filenames}))
    for lo in range(0, len(filenames), batch):
        chunk = filenames[lo:lo + batch]
        feats = get_spectrograms_batch([load_wav(f) for f in chunk])
        for filename, (mel_spec, lin_spec) in zip(chunk, feats):
            speaker_id, segment_id = os.path.basename(filename)[:-len('.wav')].split('_')
            if pad and len(lin_spec) <= seg_len:                                       # preprocess.py:95-100
                mel_spec = np.concatenate((mel_spec, np.zeros((seg_len - mel_spec.shape[0] + 1, mel_spec.shape[1]))), axis=0)
                lin_spec = np.concatenate((lin_spec, np.zeros((seg_len - lin_spec.shape[0] + 1, lin_spec.shape[1]))), axis=0)
            grp.create_dataset('{}/{}/mel'.format(speaker_id, segment_id), data=mel_spec, dtype=np.float32)
            grp.create_dataset('{}/{}/lin'.format(speaker_id, segment_id), data=lin_spec, dtype=np.float32)


class Sampler(object):
    """preprocess.py:124-214: draws `index(speaker, i, t)` training segments, speakers weighted by their utterance counts."""

    def __init__(self, h5_path, dset='train', seg_len=64, speaker2id_path='', make_object='all'):
        self.dset, self.seg_len, self.speaker2id_path = dset, seg_len, speaker2id_path
        self.f = open_store(h5_path, 'r')
        if 'english' in h5_path:
            self.target_speakers = ['V001', 'V002']
        elif 'surprise' in h5_path:
            self.target_speakers = ['V001']
        else:
            raise NotImplementedError('Invalid dataset.hdf5 name!')
        speakers = self._keys(dset)
        if make_object == 'all':
            self.speaker_used = speakers
            self.save_speaker2id()
        elif make_object == 'source':
            self.get_speaker2id()
            self.speaker_used = [s for s in speakers if s not in self.target_speakers]
        elif make_object == 'target':
            self.get_speaker2id()
            self.speaker_used = self.target_speakers
        else:
            raise NotImplementedError('Invalid make object!')
        self.speaker2utts = {s: self._keys('%s/%s' % (dset, s)) for s in self.speaker_used}
        self.rm_too_short_utt()
        self.speaker_weight = [len(self.speaker2utts[s]) / self.total_utt for s in self.speaker_used]
        self.indexer = namedtuple('index', ['speaker', 'i', 't'])

    def _keys(self, prefix):
        return self.f.keys(prefix) if isinstance(self.f, NpzStore) else sorted(list(self.f[prefix].keys()))

    def _len(self, speaker, utt):
        return self.f['%s/%s/%s/lin' % (self.dset, speaker, utt)].shape[0]

    def get_num_utts(self):
        return sum(len(self.speaker2utts[s]) for s in self.speaker_used)

    def rm_too_short_utt(self, limit=None):
        self.total_utt = self.get_num_utts()
        limit = self.seg_len if limit is None else limit
        for s in self.speaker_used:
            self.speaker2utts[s] = [u for u in self.speaker2utts[s] if self._len(s, u) > limit]
        print('[Sampler] - %i too short utterences out of a total of %i are removed.' % (self.total_utt - self.get_num_utts(), self.total_utt))

    def sample(self):
        speaker = np.random.choice(self.speaker_used, p=self.speaker_weight)
        utt_id = random.sample(self.speaker2utts[speaker], 1)[0]
        t = random.randint(0, self._len(speaker, utt_id) - self.seg_len)
        return self.indexer(speaker=self.speaker2id[speaker], i='%s/%s' % (speaker, utt_id), t=t)

    def save_speaker2id(self):
        self.speaker2id = {s: i for i, s in enumerate(self.speaker_used)}
        with open(self.speaker2id_path, 'w') as f:
            f.write(json.dumps(self.speaker2id))

    def get_speaker2id(self):
        with open(self.speaker2id_path, 'r') as f:
            self.speaker2id = json.load(f)


def make_samples(h5py_path, json_path, speaker2id_path, make_object, seg_len=64, n_samples=200000, dset='train'):
    """preprocess.py:110-121."""
    sampler = Sampler(h5py_path, dset, seg_len, speaker2id_path, make_object)
    samples = [sampler.sample()._asdict() for _ in range(n_samples)]
    with open(json_path, 'w') as f_json:
        json.dump(samples, f_json, indent=4, separators=(',', ': '))


def preprocess(source_path, target_path, test_path, dataset_path, index_path, index_source_path, index_target_path, speaker2id_path,
               seg_len=128, n_samples=200000, dset='train', remake=True):
    """preprocess.py:26-74."""
    if remake or not (os.path.isfile(dataset_path) or os.path.isfile(dataset_path + '.npz')):
        with open_store(dataset_path, 'w') as f:
            grps = [f.create_group('train'), f.create_group('test')]
            make_dataset(grps, seg_len, root_dir=source_path)
            make_dataset(grps, seg_len, root_dir=target_path)
            make_dataset(grps, seg_len, root_dir=test_path, make_test=True, pad=False)
    for path, obj in ((index_path, 'all'), (index_source_path, 'source'), (index_target_path, 'target')):
        make_samples(dataset_path, path, speaker2id_path, make_object=obj, seg_len=seg_len, n_samples=n_samples, dset=dset)
