"""BASELINE config 5 on one MI355X: train_p / train_tgat (patchGAN stage from an ae checkpoint), batch 128, bf16, english hps with
enc_size = emb_size = 1024, 102 speakers / 2 targets.  One "iteration" = n_patch_steps (5) discriminator steps (each with the
WGAN-GP double backward) + one generator step (+ the target-guided step with --tgat), as trainer.py:467-560.
  python tools/stage2_bench.py [--iters 3] [--batch 128] [--tgat]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zs_amd  # noqa: E402,F401
from zs_amd import layers  # noqa: E402
from zs_amd.hps import make_hps  # noqa: E402
from zs_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=3); ap.add_argument('--batch', type=int, default=128)
ap.add_argument('--tgat', action='store_true'); ap.add_argument('--dtype', default='bf16')
a = ap.parse_args()
dev = torch.device('cuda:0')
hps = make_hps(enc_size=1024, emb_size=1024, batch_size=a.batch)
tr = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir='/tmp/zs_s2_log', dtype=a.dtype, device=dev)
s2 = tr.stage2()
g = torch.Generator().manual_seed(0)
B = a.batch
x_s = torch.rand(B, 128, 513, generator=g).to(dev); x_t = torch.rand(B, 128, 513, generator=g).to(dev)
c_t = torch.randint(hps.n_speakers - hps.n_target_speakers, hps.n_speakers, (B,), generator=g).to(dev)


def iteration():
    for _ in range(hps.n_patch_steps):
        r = s2.d_step(x_s, x_t, c_t)
    r2 = s2.g_step(x_s, x_t, c_t)
    lrec = s2.tg_step(x_t, c_t) if a.tgat else None
    return r, r2, lrec


iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    r, r2, lrec = iteration()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
layers.check_status(dev)
t1 = time.perf_counter(); s2.d_step(x_s, x_t, c_t); torch.cuda.synchronize(); td = time.perf_counter() - t1
t1 = time.perf_counter(); s2.g_step(x_s, x_t, c_t); torch.cuda.synchronize(); tg = time.perf_counter() - t1
print('stage 2 (%s, B=%d%s): %.1f ms per iteration (%d D steps + 1 G step%s) = %.0f frames/s;  one D step %.1f ms, one G step %.1f ms;  '
      'w_dis %.3f gp %.3f loss_clf %.3f loss_adv %.3f%s;  peak memory %.1f GB' %
      (a.dtype, B, ', target-guided' if a.tgat else '', dt * 1e3, hps.n_patch_steps, ' + tg step' if a.tgat else '', B * 128 / dt, td * 1e3, tg * 1e3,
       r['w_dis'].item(), r['gp'].item(), r['real_loss_clf'].item(), r2['loss_adv'].item(), (' tg_rec %.4f' % lrec.item()) if lrec is not None else '',
       torch.cuda.max_memory_allocated() / 1e9), flush=True)
