"""gen_forward of stage 2 alone (B = 128, bf16): wall time per call against the host time to enqueue it.  python tools/genfwd_probe.py"""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zs_amd  # noqa: E402,F401
from zs_amd.hps import make_hps  # noqa: E402
from zs_amd.trainer import Trainer  # noqa: E402

dev = torch.device('cuda:0')
B = 128
hps = make_hps(enc_size=1024, emb_size=1024, batch_size=B)
tr = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=tempfile.mkdtemp(), dtype='bf16', device=dev)
s2 = tr.stage2()
g = torch.Generator().manual_seed(0)
x_s = torch.rand(B, 128, 513, generator=g).to(dev)
c_t = torch.randint(hps.n_speakers - hps.n_target_speakers, hps.n_speakers, (B,), generator=g).to(dev)
for _ in range(3):
    s2.gen_forward(x_s, c_t, False)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    s2.gen_forward(x_s, c_t, False)
t_host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print('gen_forward: %.2f ms per call back to back, host enqueue %.2f ms' % (t_all * 1e3, t_host * 1e3))
