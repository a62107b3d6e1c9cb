"""Secondary metric (BASELINE config 4): test_encode + Griffin-Lim resynthesis of 64 utterances of 200..700 frames on one
MI355X -- full-size english model (enc_size 1024, random weights), n_iter = 300 -- in utterances/s and frames/s, with the CPU
oracle timed on a bounded sample beside it.   python tools/resynth_bench.py [--dtype bf16|fp32] [--utts 64] [--cpu-utts 1]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zs_amd  # noqa: E402,F401
from zs_amd import convert as cv  # noqa: E402
from zs_amd.hps import make_hps  # noqa: E402
from zs_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16'); ap.add_argument('--utts', type=int, default=64)
ap.add_argument('--cpu-utts', type=int, default=1); ap.add_argument('--n-iter', type=int, default=300)
a = ap.parse_args()
torch.manual_seed(1)
hps = make_hps(enc_size=1024, emb_size=1024, n_speakers=102)
tr = Trainer(hps, None, 'targeted_residual', 'multilabel_binary', log_dir=tempfile.mkdtemp(), dtype=a.dtype)
rng = np.random.RandomState(0)
lens = rng.randint(200, 701, size=a.utts)
specs = [np.clip(rng.rand(int(n), 513).astype(np.float32), 1e-8, 1) for n in lens]
spk = [int(rng.randint(0, 102)) for _ in specs]


def run():
    encs, decs = cv.encode_batch(specs, tr, 128, decode_speakers=spk)
    wavs = cv.spectrogram2wav_batch(decs, n_iter=a.n_iter, do_trim=True)
    return encs, decs, wavs


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
encs, decs, wavs = run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t1 = time.perf_counter()
cv.encode_batch(specs, tr, 128, decode_speakers=spk)
torch.cuda.synchronize()
dt_enc = time.perf_counter() - t1
frames = int(sum(lens))
print('GPU (%s): %d utterances / %d frames: encode+decode+Griffin-Lim(%d) %.3f s = %.1f utterances/s, %.0f frames/s (encode+decode alone %.3f s; host-side '
      'fragmenting, trim and copies included)' % (a.dtype, a.utts, frames, a.n_iter, dt, a.utts / dt, frames / dt, dt_enc), flush=True)
if a.cpu_utts > 0:
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import zs_oracle as O      # CPU baseline beside the measurement only
    torch.set_num_threads(int(os.environ.get('ZS_CPU_THREADS', '16')))
    t0 = time.perf_counter()
    fr = 0
    for d in decs[:a.cpu_utts]:
        O.spectrogram2wav(np.asarray(d, dtype=np.float32), n_iter=a.n_iter)
        fr += d.shape[0]
    dc = time.perf_counter() - t0
    print('CPU oracle Griffin-Lim(%d) on %d utterance(s), %d frames: %.2f s = %.2f utterances/s, %.0f frames/s' % (a.n_iter, a.cpu_utts, fr, dc, a.cpu_utts / dc, fr / dc))
