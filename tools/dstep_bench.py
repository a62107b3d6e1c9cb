"""The stage-2 discriminator step alone (BASELINE config 5 shapes: B = 128, bf16, english hps with enc_size = emb_size = 1024):
n D steps on a fixed x_gen, for kernel-level profiling.   python tools/dstep_bench.py [n]
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 $R/tools/dstep_bench.py 5"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda:0')
B = 128
hps = make_hps(enc_size=1024, emb_size=1024, batch_size=B)
tr = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=tempfile.mkdtemp(), dtype='bf16', device=dev)
s2 = tr.stage2()
g = torch.Generator().manual_seed(0)
x_s, x_t = torch.rand(B, 128, 513, generator=g).to(dev), torch.rand(B, 128, 513, generator=g).to(dev)
c_t = torch.randint(hps.n_speakers - hps.n_target_speakers, hps.n_speakers, (B,), generator=g).to(dev)
x_gen = s2.gen_forward(x_s, c_t, False).clone()
for _ in range(2):
    s2.d_step(None, x_t, c_t, x_gen=x_gen)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    r = s2.d_step(None, x_t, c_t, x_gen=x_gen)
torch.cuda.synchronize()
print('D step (x_gen given): %.2f ms; gp %.3f' % ((time.perf_counter() - t0) / n * 1e3, r['gp'].item()), flush=True)
