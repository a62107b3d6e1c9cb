import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zs_amd
from zs_amd import convert as cv, _lib as L
from zs_amd.hps import make_hps
from zs_amd.trainer import Trainer
rng = np.random.RandomState(0)
lens = rng.randint(200, 701, size=64)
mags = [torch.from_numpy(np.abs(rng.randn(513, int(T))).astype(np.float32)).cuda() for T in lens]
def gl(tag):
    if os.environ.get('PROBE_SKIP_STANDALONE') == '1':
        return
    for c in (1, 3):
        L.set_option('gl_chains', c)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            cv.griffin_lim_batch(mags, n_iter=300)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print('%-40s chains=%d  %.2f ms' % (tag, c, best * 1e3), flush=True)
gl('fresh process')
hps = make_hps(enc_size=1024, emb_size=1024, n_speakers=102)
tr = Trainer(hps, None, 'targeted_residual', 'multilabel_binary', log_dir=tempfile.mkdtemp(), dtype='bf16', device=torch.device('cuda', 0))
gl('after Trainer()')
specs = [np.clip(rng.rand(int(n), 513).astype(np.float32), 1e-8, 1) for n in lens]
spk = [int(rng.randint(0, 102)) for _ in specs]
encs, decs = cv.encode_batch(specs, tr, 128, decode_speakers=spk, to_host=False)
gl('after encode_batch (host specs)')
mags2 = [d.t().float().contiguous() for d in decs]
mags, keep = mags2, mags
gl('decoder-shaped input')
mags = keep
e, w = cv.resynth_batch(specs, tr, 128, spk, n_iter=5)
gl('after resynth_batch')
# inside the pipeline (as bench.py --mode resynth times it)
specs_dev = [torch.from_numpy(s).cuda() for s in specs]
for c in ((3, 1, 2, 3) if os.environ.get('PROBE_SKIP_STANDALONE') == '1' else (1, 2, 3)):
    L.set_option('gl_chains', c)
    for src, name in ((specs_dev, 'device specs'), (specs, 'host specs')):
        best = (1e9, 0)
        for _ in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            orig = cv.griffin_lim_batch
            def timed_gl(*a, **kw):
                s.record(); r = orig(*a, **kw); e.record(); return r
            cv.griffin_lim_batch = timed_gl
            torch.cuda.synchronize(); t0 = time.perf_counter()
            try:
                cv.resynth_batch(src, tr, 128, spk, n_iter=300)
            finally:
                cv.griffin_lim_batch = orig
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = min(best, (dt * 1e3, s.elapsed_time(e)))
        print('pipeline %-14s chains=%d  batch %.2f ms, GL by events %.2f ms' % (name, c, best[0], best[1]), flush=True)
