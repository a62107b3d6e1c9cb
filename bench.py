"""bench.py -- throughput of the --train_ae step (BASELINE.json metric) on N MI355X GPUs.

  python bench.py --gpus N --steps K --warmup W [--dtype bf16|fp32] [--batch 256] [--no-cpu-baseline]
  N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / WORLD_SIZE in
  the environment), or plainly as `python bench.py --gpus N`: the parent then starts N rank processes itself (before it makes
  any GPU call) and returns rank 0's JSON line.  A world size that differs from --gpus is an error, never a silent 1-rank run.

A "step" is one pass of the hot path over one batch of synthetic segments resident in HBM: Encoder fwd,
Decoder fwd, L1, full backward, per-net clip, Adam (dropout and Gumbel noise ON, nothing skipped).
Workload = BASELINE.json configs[1]: english hps, seg_len=128, enc_size=1024, emb_size=1024, 102 speakers,
513-bin frames, batch 256 per GPU.  One JSON line is printed by rank 0.

roofline: the dominant kernel is gemm_conv_kernel (implicit-GEMM conv / linear forward + data gradient).
  achieved = algorithmic FLOPs of those launches (2*M*Cout*Cin*k per call, unpadded sizes) / their summed
  duration, measured with HIP events on the launch stream during the timed steps; peak = dense MFMA peak
  of the dtype (bf16 2.5 PFLOP/s, exact-fp32 157.3 TFLOP/s; MI355X_MICROARCH.md).
cpu_baseline: the oracle (CPU port of the reference, oracle/zs_oracle.py) timed on the host cores on a bounded
  sample of the same workload (full-size model, reference batch 16), rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--dtype', default=os.environ.get('ZS_BENCH_DTYPE', 'bf16'))
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='issue every launch from the host instead of replaying hipGraphs')
    ap.add_argument('--mode', choices=['train_ae', 'resynth'], default='train_ae',
                    help='resynth: BASELINE config 4 (test_encode + Griffin-Lim resynthesis of 64 utterances) instead of the headline train_ae step')
    ap.add_argument('--utts', type=int, default=64)
    ap.add_argument('--host-input', action='store_true', help='copy the batch from pinned host memory every step (PCIe-inclusive rate; never the headline value)')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the secondary records of the train_ae line (bf16-vs-fp32 parity of the benchmarked model, BASELINE configs 4 and 5)')
    return ap.parse_args()


T0 = time.time()


def log(msg):
    sys.stderr.write('[bench %7.1fs] %s\n' % (time.time() - T0, msg))
    sys.stderr.flush()


def cpu_share():
    """CPU threads this process may really use: min(affinity, cgroup quota), capped by ZS_CPU_THREADS."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get('ZS_CPU_THREADS', '16'))))


def cpu_baseline(seg_len, F, E, ch, nspk, steps=3, batch=16):
    """Oracle train_ae step on the host cores (bounded sample).  Returns dict for the JSON line."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import zs_oracle as O      # checker / baseline only
    torch.manual_seed(0)
    cores = cpu_share()
    torch.set_num_threads(cores)
    log('cpu_baseline: %d threads' % cores)
    # random-init weights of the reference architecture, through torch.nn containers (names = reference state_dict)
    from zs_amd.model import Decoder, Encoder
    enc = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=seg_len, enc_mode='multilabel_binary')
    dec = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=seg_len)
    esd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    dsd = {k: v.detach().clone() for k, v in dec.state_dict().items()}
    hp = dict(ns=0.01, enc_dp=0.5, enc_size=E, seg_len=seg_len)
    tr = O.TrainAE(esd, dsd, hp, lr=1e-4, max_grad_norm=5.0)
    x = torch.rand(batch, F, seg_len) * (1 - 1e-8) + 1e-8
    c = torch.randint(0, nspk, (batch,))
    t0 = time.perf_counter()
    tr.step(x, c)                                 # warm-up
    log('cpu_baseline: warm-up step %.2f s' % (time.perf_counter() - t0))
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        tr.step(x, c)
        ts.append(time.perf_counter() - t0)
        log('cpu_baseline: step %.2f s' % ts[-1])
    ts.sort()
    med = ts[len(ts) // 2]
    return {'value': batch * seg_len / med, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': '%d train_ae steps of %d x %d-frame segments (full-size model, fp32), median %.3f s/step' % (steps, batch, seg_len, med)}


def _p8_dispatch(M, N, n_pad):
    """zs_gemm_conv's rule for its 256x256 ping-pong kernel (csrc/zs_gemm.hip, default options)."""
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    return n_pad % 256 == 0 and tiles >= 200


def isolated_kernel_rate(dev, dtype, launches=200):
    """The dominant kernel alone on the GPU: the decoder's largest conv (k3 1024 -> 2048 at T=64, M = 16384 rows), `launches`
    back-to-back launches (~40 ms: long enough for the power-capped clocks to settle), HIP events on the launch stream."""
    from zs_amd import _lib as L, layers
    ctx = layers.Ctx(dev, dtype)
    w = torch.randn(2048, 1024, 3, device=dev) * 0.02
    b = torch.zeros(2048, device=dev)
    l = layers.ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b))
    l.pack()
    X = ctx.act('iso_x', 256, 64, 1024)
    X.t.normal_()
    Y = ctx.act('iso_y', 256, 64, 2048)
    for _ in range(20):
        l.fwd(X, out=Y, act=L.ZS_ACT_LRELU, slope=0.01)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(launches):
        l.fwd(X, out=Y, act=L.ZS_ACT_LRELU, slope=0.01)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / launches
    fl = 2.0 * 256 * 64 * 2048 * 1024 * 3
    return {'achieved': fl / ms / 1e9, 'unit': 'TFLOP/s', 'avg_launch_ms': ms, 'launches': launches,
            'shape': 'conv k3 1024->2048, B=256 T=64 (M=16384, N=2048, K=3072), random normal operands'}


class KernelEvents(object):
    """HIP-event timing, on the launch stream, of every zs_gemm_conv launch made through ConvLayer.fwd / ConvLayer.dgrad that
    dispatches to the dominant kernel (gemm_conv_p8m16_kernel: the 256x256 ping-pong tile); the smaller layers go to the
    ring / 128x128 kernels and are not counted."""

    def __init__(self, all_launches=False):
        self.pairs = []      # (start, end, flops)
        self.enabled = False
        self.all = all_launches          # True: every zs_gemm_conv launch through ConvLayer, whatever kernel it dispatches to
        self._orig = None

    def install(self):
        from zs_amd import layers
        ke = self
        ofwd, odgrad = layers.ConvLayer.fwd, layers.ConvLayer.dgrad
        self._orig = (ofwd, odgrad)

        def fwd(self, A, *a, **kw):
            if not ke.enabled or not (ke.all or _p8_dispatch(A.B * self.t_out(A.T), self.Cout, self.n_pad)):
                return ofwd(self, A, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = ofwd(self, A, *a, **kw)
            e.record()
            ke.pairs.append((s, e, 2.0 * A.B * self.t_out(A.T) * self.Cout * self.Cin * self.k))
            return r

        def dgrad(self, dY, T_x, out, *a, **kw):
            n = kw.get('n_cols') or self.Cin                  # columns this launch really computes (ConvLayer.dgrad n_cols)
            n_pad = self.n_pad_d if n == self.Cin else min(self.n_pad_d, (n + 255) // 256 * 256)
            if not ke.enabled or not (ke.all or _p8_dispatch(dY.B * (T_x + self.pad_l + self.pad_r), n, n_pad)):
                return odgrad(self, dY, T_x, out, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = odgrad(self, dY, T_x, out, *a, **kw)
            e.record()
            ke.pairs.append((s, e, 2.0 * dY.B * dY.T * self.Cout * n * self.k))
            return r

        layers.ConvLayer.fwd, layers.ConvLayer.dgrad = fwd, dgrad

    def uninstall(self):
        from zs_amd import layers
        if self._orig is not None:
            layers.ConvLayer.fwd, layers.ConvLayer.dgrad = self._orig
            self._orig = None

    def summary(self):
        tot_ms = sum(s.elapsed_time(e) for s, e, _ in self.pairs)
        tot_fl = sum(f for _, _, f in self.pairs)
        return tot_fl, tot_ms, len(self.pairs)


def spawn_ranks(args):
    """`python bench.py --gpus N` without torchrun: start N rank processes (one per GPU) with the torchrun environment
    contract.  The parent has made no GPU call (importing torch does not initialise HIP) and only waits; rank 0's stdout is
    the parent's, so the single JSON line comes out unchanged.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', str(max(1, cpu_share() // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL)))
    rc = 0
    for r, pr in enumerate(procs):
        code = pr.wait()
        if code != 0:
            log('rank %d exited with code %d' % (r, code))
            rc = rc or code
    return rc


def resynth_record(args, rank, world, dev, steps, warmup, cpu=True):
    """BASELINE config 4: test_encode + convert.py Griffin-Lim resynthesis of 64 utterances of U{200..700} frames on one MI355X
    (full-size english model, random weights, bf16 network, fp32 vocoder, n_iter = 300).  A "step" = the whole batch once:
    fragmenting, Encoder + Decoder over every fragment, de-normalisation, 300 Griffin-Lim iterations, de-emphasis, trim.
    `value`: the utterances' spectrograms resident in HBM when the timed region starts; the host-side fragment logic, every launch
    and the D2H copies of the encodings and waveforms are inside it.  `host_input_utt_per_s`: the same batch handed over as host
    numpy arrays as the reference's loader does (pinned staging + H2D inside the time).  Multi-GPU: replicas over a sharded
    utterance list, no collective.
    roofline: the dominant kernel gl_iter_kernel (one fused Griffin-Lim iteration per launch) is bound by the memory system: every
    iteration streams the complex spectrogram of all utterances in and out (more than the 256 MiB Infinity Cache holds at 64
    utterances).  achieved = ALGORITHMIC bytes per launch (per frame: 513 complex64 read + 513 complex64 written + 513 fp32
    magnitudes read = 10.3 KB) / the HIP-event time of a launch on the launch stream, against 8 TB/s; the FLOP rate of the
    transforms (2 real 1024-point FFTs per frame and iteration, 2.5 N log2 N each) is reported beside it.
    Returns the record (rank 0) or None."""
    import tempfile
    import numpy as np
    from zs_amd import convert as cv, layers, parallel
    from zs_amd.hps import hp, make_hps
    from zs_amd.trainer import Trainer
    torch.manual_seed(1)
    hps = make_hps(enc_size=1024, emb_size=1024, n_speakers=102)
    tr = Trainer(hps, None, 'targeted_residual', 'multilabel_binary', log_dir=tempfile.mkdtemp(), dtype=args.dtype, device=dev)
    rng = np.random.RandomState(0)
    lens_all = rng.randint(200, 701, size=args.utts * world)
    lo, hi = parallel.shard_range(len(lens_all), rank, world)
    lens = lens_all[lo:hi]
    specs = [np.clip(rng.rand(int(n), 513).astype(np.float32), 1e-8, 1) for n in lens]
    spk = [int(rng.randint(0, 102)) for _ in specs]
    n_iter = hp.n_iter
    gl_ms = []

    specs_dev = [torch.from_numpy(s).to(dev) for s in specs]     # the utterances resident in HBM when the timed region starts

    def run(timed, src):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        orig = cv.griffin_lim_batch

        def timed_gl(*a, **kw):
            s.record()
            r = orig(*a, **kw)
            e.record()
            return r
        cv.griffin_lim_batch = timed_gl
        try:
            encs, wavs = cv.resynth_batch(src, tr, 128, spk, n_iter=n_iter, do_trim=True)
        finally:
            cv.griffin_lim_batch = orig
        if timed:
            torch.cuda.synchronize()
            gl_ms.append(s.elapsed_time(e))
        return encs, wavs

    for _ in range(max(1, warmup)):
        run(False, specs_dev)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        encs, wavs = run(True, specs_dev)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    layers.check_status(dev)
    # the same batches as a serving loop issues them: batch i + 1 enqueued before the results of batch i are waited for
    # (resynth_batch(defer=True)); reported beside `value`, which keeps one host wait per batch inside every step
    pend = None
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        f = cv.resynth_batch(specs_dev, tr, 128, spk, n_iter=n_iter, do_trim=True, defer=True)
        if pend is not None:
            pend()
        pend = f
    pend()
    torch.cuda.synchronize()
    dt_pipe = (time.perf_counter() - t1) / steps
    # beside it: the same batch handed over as host arrays (the reference's loader), pinned staging + H2D inside the time
    run(False, specs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        run(False, specs)
    torch.cuda.synchronize()
    dt_host = (time.perf_counter() - t1) / steps
    t1 = time.perf_counter()
    _, decs = cv.encode_batch(specs_dev, tr, 128, decode_speakers=spk, to_host=False)
    torch.cuda.synchronize()
    dt_enc = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    out_frames = int(sum(d.shape[0] for d in decs))
    n_utt = len(lens_all)
    value = n_utt * steps / dt
    log('resynth: Griffin-Lim per step (ms): ' + ' '.join('%.2f' % g for g in gl_ms))
    gl = sum(gl_ms) / len(gl_ms) * 1e-3
    gl_traffic = None
    from zs_amd import _lib as ZL
    chains = int(ZL.lib().zs_gl_chains_used())
    tpath = os.path.join(ROOT, 'profiles', GL_TRAFFIC_FILE)
    if os.path.exists(tpath) and args.utts == 64:
        kk = json.load(open(tpath)).get('kernels', {})
        # counters are per kernel launch; an iteration of all utterances is (launches of gl_iter_kernel) / (batches x (n_iter + 1)) of
        # them -- one under the counter passes (the profiler serialises kernels, so the stream probe finds no concurrency and
        # zs_griffin_lim stays on one chain), `chains` otherwise.  Batches of the profiled run = launches of the de-emphasis scan.
        gks = [v for k, v in kk.items() if k.startswith('gl_iter_kernel')]          # (the first-iteration variant is a kernel of its own)
        nb = next((v.get('launches') for k, v in kk.items() if k.startswith('gl_deemph_scan_kernel')), None)
        if gks and nb:
            gl_traffic = sum(v['traffic_bytes_per_launch'] * v['launches'] for v in gks) / float(nb * (n_iter + 1))
    fl = out_frames * (2 * n_iter + 1) * 2.5 * 1024 * 10          # per rank-0 shard
    by = out_frames * 513.0 * (8 + 8 + 4)                         # algorithmic bytes per launch: spectrum in + out, magnitudes in
    launch_s = gl / (n_iter + 1)
    out = {'metric': 'utterances/sec (test_encode + Griffin-Lim resynthesis, 64 utterances of 200..700 frames, n_iter=300)',
           'value': value, 'unit': 'utterances/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
           'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f32 (vocoder) / %s (network)' % args.dtype, 'data': 'synthetic',
           'config': {'workload': 'BASELINE config 4: encode + decode (english hps, enc_size=1024, emb_size=1024) + spectrogram2wav '
                                  '(Griffin-Lim n_iter=%d, de-emphasis, trim) of %d utterances of U{200..700} frames per GPU, spectrograms '
                                  'resident in HBM, host fragmenting and D2H of the results included' % (n_iter, len(lens)), 'parallelism': 'replicas%d' % world},
           'frames_per_s': float(sum(lens_all)) * steps / dt, 'host_input_utt_per_s': len(lens_all) / dt_host,
           'host_input_ms_per_step': 1e3 * dt_host, 'pipelined_utt_per_s': len(lens_all) / dt_pipe, 'pipelined_ms_per_step': 1e3 * dt_pipe,
           'encode_decode_ms': 1e3 * dt_enc, 'griffin_lim_ms': 1e3 * gl,
           'roofline': {'bound': 'hbm', 'achieved': by / launch_s / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': by / launch_s / 8e12,
                        'traffic': gl_traffic, 'traffic_unit': 'bytes/launch (committed rocprofv3 --pmc passes of this command)',
                        'kernel': 'gl_iter_kernel (one fused Griffin-Lim iteration of ALL utterances = one "launch" here; issued as up to 3 '
                                  'concurrent launches over utterance ranges on independent streams, %d iterations per batch)' % (n_iter + 1),
                        'avg_launch_ms': 1e3 * launch_s, 'algorithmic_bytes_per_launch': by, 'concurrent_kernel_launches_per_launch': chains,
                        'fft_tflops': fl / gl / 1e12, 'fft_frac_of_fp32_vector_peak': fl / gl / 1e12 / 157.3}}
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import zs_oracle as O      # CPU baseline beside the measurement only
        cores = cpu_share()
        torch.set_num_threads(cores)
        d = np.asarray(decs[int(np.argmin([abs(x.shape[0] - 376) for x in decs]))].cpu(), dtype=np.float32)
        t0 = time.perf_counter()
        O.spectrogram2wav(d, n_iter=n_iter)
        dc = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': 1.0 / dc, 'unit': 'utterances/s', 'cores': cores, 'kind': 'port',
                               'sample': 'oracle spectrogram2wav (Griffin-Lim n_iter=%d, numpy FFT) of ONE %d-frame utterance: %.2f s; the '
                                         'network forward is not included (the vocoder is >95 %% of the CPU path)' % (n_iter, d.shape[0], dc)}
    del tr
    torch.cuda.empty_cache()
    return out


def resynth_main(args):
    import zs_amd  # noqa: F401
    from zs_amd import parallel
    rank, world, local = parallel.init_from_env(os.environ.get('ZS_DIST_BACKEND', 'nccl'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE is %d\n' % (args.gpus, world))
        sys.exit(2)
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    os.environ['LOCAL_RANK'] = str(local)
    out = resynth_record(args, rank, world, dev, args.steps, args.warmup, cpu=not args.no_cpu_baseline)
    if out is not None:
        emit_json(out)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def stage2_record(args, dev, iters=2, batch=128, cpu=True):
    """BASELINE config 5 on one GPU: train_p / train_tgat (patchGAN stage), batch 128, bf16, english hps with enc_size = emb_size =
    1024.  One "iteration" = n_patch_steps (5) discriminator steps (each with the WGAN-GP double backward) + one generator step +
    the target-guided step (trainer.py:467-560).  roofline: the conv GEMM launches of one discriminator step (zs_gemm_conv through
    ConvLayer.fwd / dgrad: the critic's forward, data-gradient and adjoint convolutions), HIP events on the launch stream."""
    import tempfile
    from zs_amd import layers
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    hps = make_hps(enc_size=1024, emb_size=1024, batch_size=batch)
    tr = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=tempfile.mkdtemp(), dtype=args.dtype, device=dev)
    s2 = tr.stage2()
    g = torch.Generator().manual_seed(0)
    B = batch
    x_s = torch.rand(B, 128, 513, generator=g).to(dev)
    x_t = torch.rand(B, 128, 513, generator=g).to(dev)
    c_t = torch.randint(hps.n_speakers - hps.n_target_speakers, hps.n_speakers, (B,), generator=g).to(dev)

    def iteration():
        for _ in range(hps.n_patch_steps):
            r = s2.d_step(x_s, x_t, c_t)
        r2 = s2.g_step(x_s, x_t, c_t)
        lrec = s2.tg_step(x_t, c_t)
        return r, r2, lrec

    iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        r, r2, lrec = iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    layers.check_status(dev)
    t1 = time.perf_counter(); s2.d_step(x_s, x_t, c_t); torch.cuda.synchronize(); td = time.perf_counter() - t1
    t1 = time.perf_counter(); s2.g_step(x_s, x_t, c_t); torch.cuda.synchronize(); tg = time.perf_counter() - t1
    ke = KernelEvents(all_launches=True)
    ke.install()
    ke.enabled = True
    s2.d_step(x_s, x_t, c_t)
    torch.cuda.synchronize()
    ke.enabled = False
    ke.uninstall()
    fl, ms, n = ke.summary()
    peak = 2500.0 if args.dtype == 'bf16' else 157.3
    out = {'metric': 'frames/sec (train_tgat iteration: 5 D steps with WGAN-GP + 1 G step + target-guided step)', 'value': B * 128 / dt,
           'unit': 'frames/s', 'ms_per_iteration': 1e3 * dt, 'd_step_ms': 1e3 * td, 'g_step_ms': 1e3 * tg, 'dtype': args.dtype, 'data': 'synthetic',
           'config': {'workload': 'BASELINE config 5 on one GPU: patchGAN stage, english hps enc_size=1024 emb_size=1024, batch=%d' % B},
           'w_dis': float(r['w_dis'].item()), 'gp': float(r['gp'].item()), 'loss_adv': float(r2['loss_adv'].item()), 'tg_rec': float(lrec.item()),
           'peak_memory_gb': torch.cuda.max_memory_allocated() / 1e9,
           'roofline': {'bound': 'mfma', 'achieved': fl / (ms * 1e-3) / 1e12 if ms else None, 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': (fl / (ms * 1e-3) / 1e12 / peak) if ms else None, 'traffic': None,
                        'kernel': 'zs_gemm_conv launches of one discriminator step (forward, data-gradient and adjoint convolutions)',
                        'launches_timed': n, 'gemm_ms_per_d_step': ms, 'gemm_tflop_per_d_step': fl / 1e12}}
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import zs_oracle as O      # CPU baseline beside the measurement only
        cores = cpu_share()
        torch.set_num_threads(cores)
        Bc = 2
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in tr.PatchDiscriminator.state_dict().items()}
        hp_ = dict(ns=hps.ns, seg_len=hps.seg_len, beta_dis=hps.beta_dis, beta_clf=hps.beta_clf, lambda_=hps.lambda_)
        xt_c, xg_c = x_t[:Bc].permute(0, 2, 1).cpu().contiguous(), x_s[:Bc].permute(0, 2, 1).cpu().contiguous()
        cc = (c_t[:Bc] - (hps.n_speakers - hps.n_target_speakers)).cpu()
        dc = 1e30
        for _ in range(2):                                      # (first pass: allocator / thread-pool warm-up)
            for v in sd.values():
                v.grad = None
            t0 = time.perf_counter()
            loss = O.patch_d_loss(sd, xt_c, xg_c, cc, torch.rand(Bc), hp_)[0]
            loss.backward()
            dc = min(dc, time.perf_counter() - t0)
        out['cpu_baseline'] = {'value': Bc * 128 / (dc * hps.n_patch_steps), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
                               'sample': 'oracle discriminator step (3 critic forwards, WGAN-GP double backward, all gradients) on %d segments: '
                                         '%.2f s; x %d D steps per iteration, the generator steps are not included' % (Bc, dc, hps.n_patch_steps)}
    del tr, s2
    torch.cuda.empty_cache()
    return out


def parity_record(enc, dec, dev, B=256):
    """The benchmarked bf16 model against the fp32 HIP path on the SAME weights, batch and Gumbel noise (eval mode: dropout off),
    SURVEY 8(d)'s secondary metrics: MBV bit-mismatch rate, relative error of the encoder logits and of x_dec (decoder error
    alone: both decoders fed the fp32 path's bits; and end to end, bit flips included).  Relative = max|d| / max|ref|."""
    from zs_amd.model import Decoder, Encoder
    seg_len, F, E, ch, nspk = 128, 513, enc.enc_size, dec.c_h, dec.c_a
    g = torch.Generator().manual_seed(4242)
    x = (torch.rand(B, F, seg_len, generator=g) * (1 - 1e-8) + 1e-8).to(dev)
    c = torch.randint(0, nspk, (B,), generator=g).to(dev)
    U = torch.rand(B, seg_len // 8, E, 2, generator=g).to(dev)
    G = -torch.log(-torch.log(U + 1e-20) + 1e-20)
    enc32 = Encoder(ns=enc.ns, dp=0.5, enc_size=E, seg_len=seg_len, enc_mode='multilabel_binary', dtype='fp32').to(dev)
    dec32 = Decoder(ns=dec.ns, c_in=E, c_h=ch, c_a=nspk, seg_len=seg_len, dtype='fp32').to(dev)
    enc32.load_state_dict(enc.state_dict()); dec32.load_state_dict(dec.state_dict())
    was = enc.training, dec.training
    for m in (enc, dec, enc32, dec32):
        m.eval()
    with torch.no_grad():
        act_b, log_b = enc(x, G=G)
        act_b, log_b = act_b.clone(), log_b.clone()
        xd_b = dec(act_b, c).clone()
        act_f, log_f = enc32(x, G=G)
        xd_f = dec32(act_f, c)
        xd_b_same = dec(act_f.clone(), c)
    rel = lambda a, r: float((a.float() - r.float()).abs().max() / r.float().abs().max())
    rms = lambda a, r: float((a.float() - r.float()).norm() / r.float().norm())
    out = {'mbv_bit_mismatch_rate': float((act_b != act_f).float().mean()), 'mbv_bits_compared': int(act_f.numel()),
           'enc_logits_rel_err': rel(log_b, log_f), 'x_dec_rel_err': rel(xd_b_same, xd_f), 'x_dec_rel_err_end_to_end': rel(xd_b, xd_f),
           'x_dec_mean_abs_err': float((xd_b_same - xd_f).abs().mean()), 'enc_logits_rel_rms_err': rms(log_b, log_f),
           'x_dec_rel_rms_err': rms(xd_b_same, xd_f), 'x_dec_rel_rms_err_end_to_end': rms(xd_b, xd_f),
           'reference': 'fp32 HIP path (exact-fp32 MFMA; parity-tested against the oracle at 1e-3), same weights / batch of %d / Gumbel noise, eval mode' % B}
    enc.train(was[0]); dec.train(was[1])
    del enc32, dec32
    torch.cuda.empty_cache()
    return out


GL_TRAFFIC_FILE = 'r03_resynth_pmc_traffic.json'

_JSON_FD = None


def claim_stdout():
    """The bench's stdout carries ONE JSON line.  Libraries write there too (RCCL prints a version banner on its first collective):
    keep a private duplicate of the real stdout for the JSON line and point fd 1 at stderr for everything else."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit_json(out):
    line = (json.dumps(out) + '\n').encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))                    # (the parent keeps its stdout as it is: rank 0 inherits it)
    claim_stdout()
    if args.mode == 'resynth':
        return resynth_main(args)
    import zs_amd  # noqa: F401
    from zs_amd import parallel
    from zs_amd.model import Decoder, Encoder
    from zs_amd.trainer import AEStep
    import torch.distributed as dist

    torch.set_num_threads(cpu_share())
    # ZS_DIST_BACKEND / ZS_FORCE_DEVICE: rehearsal of the multi-rank path on a one-GPU box (gloo, every rank on cuda:0)
    force_dev = os.environ.get('ZS_FORCE_DEVICE')
    if force_dev is not None:
        torch.cuda.set_device(int(force_dev))
    rank, world, local = parallel.init_from_env(os.environ.get('ZS_DIST_BACKEND', 'nccl'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE is %d: refusing to report a %d-GPU number from %d rank(s)\n' %
                         (args.gpus, world, args.gpus, world))
        sys.exit(2)
    dev = torch.device('cuda', int(force_dev) if force_dev is not None else local)
    torch.cuda.set_device(dev)
    seg_len, F, E, ch, nspk, B = 128, 513, 1024, 1024, 102, args.batch
    torch.manual_seed(1234 + rank)
    enc = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=seg_len, enc_mode='multilabel_binary', dtype=args.dtype).to(dev)
    dec = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=seg_len, dtype=args.dtype).to(dev)
    if world > 1 or parallel.multi_rank():             # identical initial weights on every rank
        for net in (enc, dec):
            dist.broadcast(net.flat_params()[0], src=0)
            net.mark_dirty()
    ae = AEStep(enc, dec, lr=1e-4, max_grad_norm=5.0, use_graph=not args.no_graph)
    g = torch.Generator().manual_seed(99 + rank)       # distinct per-rank data
    x = (torch.rand(B, seg_len, F, generator=g) * (1 - 1e-8) + 1e-8).to(dev)
    c = torch.randint(0, nspk, (B,), generator=g).to(dev)
    if ae.use_graph and not args.host_input:
        # resident batch: it lives in the buffers the captured step reads (as a device-side loader would deliver it), so no
        # device-to-device copy of the batch precedes a replay
        xs, cs = ae.static_inputs(B, seg_len, F)
        xs.copy_(x); cs.copy_(c)
        x, c = xs, cs

    feeder = None
    if args.host_input:
        # every step's batch comes from HOST memory through trainer.HostFedStep: staged into pinned memory by the host, copied
        # H2D by a node of the captured step (the copy of batch i+1 runs beside the kernels of step i)
        class _HostBatches(object):
            def __init__(self, c_host, x_host):
                self.c, self.x = c_host, x_host

            def __next__(self):
                return self.c, self.x

        feeder = ae.host_feeder(_HostBatches(c.cpu(), x.cpu()))

    def one_step():
        if feeder is not None:
            return next(feeder)                       # 67.2 MB H2D per step at B=256
        return ae.step(x, c)

    ke = KernelEvents()
    if not args.no_kernel_events:
        ke.install()
    log('model built (%s, B=%d, graph=%s)' % (args.dtype, B, ae.use_graph))
    if ae.use_graph:                                   # set-up, not warm-up: 2 eager steps + the capture step
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        log('hipGraph captured (%d graph segment(s))' % (sum(len(v['graphs']) for v in ae._graphs.values()) + (sum(len(e['graphs']) for e in feeder.ents) if feeder is not None and feeder.ents else 0)))
    for i in range(args.warmup):
        one_step()
        torch.cuda.synchronize()
        log('warm-up step %d done, loss %.4f' % (i, ae._loss.item()))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ke.enabled = (not args.no_kernel_events) and not ae.use_graph
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    t_host = time.perf_counter() - t0                 # host time to ENQUEUE the steps (no synchronisation inside)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ke.enabled = False
    log('timed %d steps: %.3f s (host enqueue %.3f s)' % (args.steps, dt, t_host))
    if ae.use_graph and not args.no_kernel_events:
        # HIP events cannot bracket kernels inside a replayed graph: the per-kernel durations for the roofline come from
        # instrumented EAGER steps of the same workload, run right after the timed region in this process
        ae.use_graph = False
        ke.enabled = True
        for _ in range(min(10, max(3, args.steps))):
            ae.step(x, c)                              # (resident batch: only the kernel durations are read from these steps)
        torch.cuda.synchronize()
        ke.enabled = False
        ae.use_graph = True
        log('instrumented eager steps for the roofline: %d gemm_conv launches timed' % len(ke.pairs))
    isolated = None
    if rank == 0 and not args.no_kernel_events:
        isolated = isolated_kernel_rate(dev, args.dtype)
    loss = float(ae._loss.item())
    from zs_amd import layers
    layers.check_status(dev)                           # sticky status word: raises if ANY persistent GRU pass of this run timed out
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        dist.barrier()                                # rank 0 is still measuring / printing: leave together
        dist.destroy_process_group()
        return
    frames = float(B) * seg_len * args.steps * world
    value = frames / dt
    out = {
        'metric': 'mel-frames/sec (train_ae, seg_len=128, enc_size=1024)', 'value': value, 'unit': 'frames/s',
        'per_gpu': value / world, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'train_ae english hps seg_len=128 enc_size=1024 emb_size=1024 n_speakers=102 F=513, batch=%d/GPU, '
                               'dropout+Gumbel on, fwd+bwd+clip+Adam' % B, 'global_batch': B * world, 'parallelism': 'dp%d' % world},
        'final_loss': loss, 'host_input': bool(args.host_input), 'host_enqueue_ms_per_step': 1e3 * t_host / args.steps, 'hipgraph': bool(ae.use_graph and (world == 1 or os.environ.get('ZS_GRAPH_MULTI', '1') == '1')),
        'graph_segments': sum(len(v['graphs']) for v in ae._graphs.values()) if ae._graphs else None,
    }
    peak = 2500.0 if args.dtype == 'bf16' else 157.3
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'r03_pmc_traffic.json')
    if args.dtype == 'bf16' and B == 256 and os.path.exists(tpath):
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); counters cannot be read from inside the process
        traffic = json.load(open(tpath)).get('traffic_bytes_per_launch')
    if ke.pairs:
        fl, ms, n = ke.summary()
        ach = fl / (ms * 1e-3) / 1e12
        out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': traffic, 'traffic_unit': 'bytes/launch',
                           'kernel': 'gemm_conv_p8m16_kernel<%s> (zs_gemm_conv launches with >= 200 tiles of 256x256)' % ('bf16' if args.dtype == 'bf16' else 'float'),
                           'launches_timed': n, 'avg_launch_ms': ms / n, 'avg_launch_gflop': fl / n / 1e9,
                           'measured_on': ('instrumented eager steps after the timed region' if ae.use_graph else 'the timed steps')}
    else:
        out['roofline'] = {'bound': 'mfma', 'achieved': None, 'peak': peak, 'unit': 'TFLOP/s', 'frac': None, 'traffic': None}
    if isolated is not None:
        # secondary figure: the same kernel with nothing else on the GPU (in the step it shares the chip, and its power budget,
        # with the side-stream weight gradients)
        isolated['frac'] = isolated['achieved'] / peak
        out['roofline']['isolated'] = isolated
    out['step_tflops'] = 180.7e6 * value / 1e12          # 180.7 MFLOP per frame for the whole step
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(seg_len, F, E, ch, nspk)                    # the reference's own batch size
        if B != 16:
            big = cpu_baseline(seg_len, F, E, ch, nspk, steps=1, batch=B)              # and the GPU configuration's
            out['cpu_baseline']['at_gpu_batch'] = {k: big[k] for k in ('value', 'unit', 'sample')}
    if world == 1 and not parallel.multi_rank() and not args.no_secondary and args.dtype == 'bf16':
        # secondary records, all outside the timed region: parity of the benchmarked bf16 model, BASELINE configs 4 and 5
        ke.uninstall()
        sec = {}
        if not args.host_input and ae.use_graph:
            # the PCIe-inclusive rate (the reference copies every batch from host memory): the same step with the
            # next batch fetched from pinned host memory by a branch of the captured step (trainer.HostFedStep); never the headline
            class _HostBatches(object):
                def __init__(self, c_host, x_host):
                    self.c, self.x = c_host, x_host

                def __next__(self):
                    return self.c, self.x
            hf = ae.host_feeder(_HostBatches(c.cpu(), x.cpu()))
            for _ in range(4):
                next(hf)
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(10):
                next(hf)
            torch.cuda.synchronize()
            out['host_input_ms_per_step'] = 1e3 * (time.perf_counter() - th) / 10
            log('secondary: batch from host memory every step: %.2f ms/step' % out['host_input_ms_per_step'])
        sec['parity_bf16_vs_fp32'] = parity_record(enc, dec, dev, B=B)
        log('secondary: parity %s' % json.dumps(sec['parity_bf16_vs_fp32']))
        out['mbv_bit_mismatch_rate'] = sec['parity_bf16_vs_fp32']['mbv_bit_mismatch_rate']
        out['x_dec_rel_err'] = sec['parity_bf16_vs_fp32']['x_dec_rel_err']
        del ae, enc, dec
        torch.cuda.empty_cache()
        sec['resynth'] = resynth_record(args, 0, 1, dev, steps=3, warmup=1, cpu=not args.no_cpu_baseline)
        log('secondary: config 4 %.0f utterances/s' % sec['resynth']['value'])
        sec['stage2'] = stage2_record(args, dev, iters=2, cpu=not args.no_cpu_baseline)
        log('secondary: config 5 %.1f ms per iteration' % sec['stage2']['ms_per_iteration'])
        out['secondary'] = sec
    emit_json(out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
