"""Feature extraction rate: get_spectrograms on 64 synthetic utterances of 2-7 s (english test utterance
lengths) in one GPU batch, against the oracle's NumPy restatement of the reference (librosa-style) path on a bounded sample.
  python tools/preprocess_bench.py [--utts 64] [--cpu-utts 8]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zs_amd  # noqa: E402,F401
from zs_amd import preprocess as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--utts', type=int, default=64); ap.add_argument('--cpu-utts', type=int, default=8)
a = ap.parse_args()
rng = np.random.RandomState(0)
wavs = [(0.1 * rng.randn(int(n))).astype(np.float32) for n in rng.randint(2 * 16000, 7 * 16000, size=a.utts)]
P.get_spectrograms_batch(wavs)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = P.get_spectrograms_batch(wavs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
frames = sum(m.shape[0] for _, m in out)
t1 = time.perf_counter()
P.get_spectrograms_batch(wavs, do_trim=False)
torch.cuda.synchronize()
dt_nt = time.perf_counter() - t1
print('GPU: %d utterances, %d frames: %.3f s = %.0f frames/s (host trim + H2D/D2H included; without the host trim %.3f s)' % (a.utts, frames, dt, frames / dt, dt_nt))
if a.cpu_utts:
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import zs_oracle as O      # CPU baseline beside the measurement
    t0 = time.perf_counter()
    fr = sum(O.get_spectrograms(w)[1].shape[0] for w in wavs[:a.cpu_utts])
    dc = time.perf_counter() - t0
    print('CPU oracle (NumPy, 1 thread FFT): %d utterances, %d frames: %.3f s = %.0f frames/s' % (a.cpu_utts, fr, dc, fr / dc))
