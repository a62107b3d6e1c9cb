"""Griffin-Lim alone on BASELINE config 4's shape: 64 utterances of U{200..700} frames, n_iter = 300 (hps/hps.py:31).
usage: python tools/gl_bench.py [n_iter]   -- fused kernel at several tile sizes against the per-transform kernels."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zs_amd  # noqa: E402,F401
from zs_amd import convert as cv  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(0)
lens = rng.randint(200, 701, size=64)
mags = [torch.from_numpy(np.abs(rng.randn(513, int(T))).astype(np.float32)).cuda() for T in lens]
frames = int(lens.sum())


def run(**kw):
    cv.griffin_lim_batch(mags, n_iter=2, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cv.griffin_lim_batch(mags, n_iter=n_iter, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


tiles = [int(t) for t in os.environ.get('GL_TILES', '6,8,10,12,14,18,26').split(',')]
from zs_amd import _lib as L  # noqa: E402
L.set_option('gl_prefetch', int(os.environ.get('GL_PREFETCH', '1')))
chains = [int(c) for c in os.environ.get('GL_CHAINS', '3').split(',')]
for kw in [dict(impl='split')] + [dict(impl='fused', tile_frames=t, chains=c) for c in chains for t in tiles]:
    label = dict(kw)
    if 'chains' in kw:
        L.set_option('gl_chains', kw.pop('chains'))
    dt = min(run(**kw) for _ in range(3))
    # 2 real 1024-point transforms per frame and iteration, 5 N log2 N / 2 flops each (N = 1024, real input)
    fl = frames * (2 * n_iter + 1) * 2.5 * 1024 * 10
    print('%-52s %8.2f ms  %7.1f utt/s  %9.0f frames/s  %6.2f TFLOP/s' % (label, dt * 1e3, 64 / dt, frames / dt, fl / dt / 1e12), flush=True)
