"""Trainer surface of the reference (trainer.py:34-347) for the stage-1 autoencoder path.

`Trainer(hps, data_loader, g_mode, enc_mode, log_dir)` builds Encoder / Decoder / Generator with the
reference's constructor wiring (trainer.py:48-98), `train(model_path, flag, mode='pretrain_AE')` runs the
`--train_ae` loop (trainer.py:320-347), `save_model` / `load_model` use the reference's checkpoint dict,
`test_step` / `encoder_test_step` are the inference entry points convert.py calls.

One training step (`ae_step`) is: Encoder fwd -> Decoder fwd -> L1 -> Decoder bwd -> [async RCCL
all-reduce of decoder grads] -> Encoder bwd -> [all-reduce encoder grads] -> per-net grad norm -> fused
clip + Adam(lr, betas=(0.5, 0.9)) on the flat parameter buffers.  Everything is enqueued on one HIP
stream without host synchronisation; the loss is read back only when it is logged.
"""
import os

import numpy as np
import torch

from . import _lib as L
from . import parallel
from . import layers
from .layers import Act, join_side
from .model import Decoder, Encoder, SpeakerClassifier
from .patch import PatchDiscriminator, TargetClassifier
from .stage2 import PatchGANStep
from .utils import Logger


class AEStep(object):
    """The fused --train_ae iteration (trainer.py:322-332) for one Encoder/Decoder pair:
    fwd, L1, bwd, gradient all-reduce, per-net clip (utils.py:53-55) and Adam(betas=(0.5,0.9)) over both nets
    (trainer.py:65-66).  All state lives on the device; nothing here synchronises with the host.

    Graph mode (ZS_GRAPH=1, default): after two eager warm-up steps the ~600 kernel launches of a step are captured
    into hipGraphs and replayed, so the host cost per step is a handful of calls instead of ~15 ms of Python/ctypes
    launches.  Everything that changes from step to step lives in device memory (RNG seed, Adam step count:
    zs_step_counters).  With more than one rank the step is one graph per segment of _multi_actions (forward + decoder tail
    backward | decoder conv blocks | encoder backward | conv bank weight gradients | decoder optimizer | encoder optimizer) with
    the RCCL all-reduces of the gradient buckets launched eagerly between them, each as soon as its bucket is final."""

    def __init__(self, encoder, decoder, lr=1e-4, betas=(0.5, 0.9), max_grad_norm=5.0, use_graph=None):
        self.Encoder, self.Decoder = encoder, decoder
        self.lr, self.betas, self.max_grad_norm = float(lr), betas, float(max_grad_norm)
        self.adam_step = 0
        self._opt = {}
        dev = encoder.flat_params()[0].device
        self.device = dev
        for name, net in (('enc', encoder), ('dec', decoder)):
            flat, gflat = net.flat_params()
            self._opt[name] = dict(m=torch.zeros_like(flat), v=torch.zeros_like(flat),
                                   sq=torch.zeros(1, dtype=torch.float32, device=dev),
                                   part=torch.zeros(1024, dtype=torch.float64, device=dev))
        self._loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self._lpart = torch.zeros(1024, dtype=torch.float32, device=dev)
        self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)      # uint64 bits, advanced by zs_step_counters
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)      # Adam step count on the device
        self.reducer = parallel.GradReducer()
        self.xdec = None
        self.use_graph = (os.environ.get('ZS_GRAPH', '1') == '1') if use_graph is None else bool(use_graph)
        self.early_dec_update = os.environ.get('ZS_EARLY_DEC_UPDATE', '1') == '1'
        self._dec_updated = False
        self.fetch_by_kernel = os.environ.get('ZS_FETCH_KERNEL', '1') == '1'     # in-graph H2D by zs_host_fetch instead of a memcpy node
        self.fetch_wgs = int(os.environ.get('ZS_FETCH_WGS', '32'))
        # data parallel: gradient buckets in backward order (decoder tail layers first, the encoder's conv bank last), each
        # all-reduce started as soon as its range of the flat gradient buffer is final (ZS_DP_BUCKETS=0: one per net)
        self.buckets = os.environ.get('ZS_DP_BUCKETS', '1') == '1'
        self._graphs = {}            # (B, T, F) -> dict(graphs=[...], x=static x, c=static c)
        self._statics = {}           # (B, T, F) -> (x, c) handed out by static_inputs()
        self._eager_calls = 0
        self.graph_warmup = 2        # eager steps (same launches) before the capture

    # ---- the stream-ordered segments of a step --------------------------------------------------------------------
    def _seg_forward(self, x_btf, c, noise, noise_kind, drop_masks, seed, seed_ptr):
        """Encoder + Decoder forward and the L1 loss; returns dlogit (the gradient w.r.t. the pre-sigmoid output)."""
        enc, dec = self.Encoder, self.Decoder
        ee, de = enc._engine(), dec._engine()
        ctx = ee.ctx
        B, T, F = x_btf.shape
        bits, _, _ = ee.forward(x_btf, True, noise=noise, noise_kind=noise_kind, seed=seed, drop_masks=drop_masks, seed_ptr=seed_ptr)
        xdec = de.forward(bits, c, True)
        dlogit = de.ctx.act('t_dlogit_%d_%d' % (B, T), B, xdec.T, F)
        L.call('zs_l1_loss', 'ZsL1Loss', ctx.stream, dtype=ctx.dt, x_dec=xdec.ptr(), ld_dec=xdec.ld, x=L.ptr(x_btf), ldx=F,
               rows=B * xdec.T, F=F, dlogits=dlogit.ptr(), ldg=dlogit.ld, fill_cols=dlogit.ld, partial=L.ptr(self._lpart),
               loss_out=L.ptr(self._loss), grad_scale=1.0)                                   # trainer.py:328
        self.xdec = xdec
        return dlogit

    def _seg_forward_decbwd(self, x_btf, c, noise, noise_kind, drop_masks, seed, seed_ptr):
        dlogit = self._seg_forward(x_btf, c, noise, noise_kind, drop_masks, seed, seed_ptr)
        self._dbits = self.Decoder._engine().backward(dlogit)                                # loss.backward(), trainer.py:330

    def _multi_actions(self, x_btf, c, noise, noise_kind, drop_masks, seed, seed_ptr, counted):
        """The data-parallel step as an ordered list of actions: ('run', fn) = launches on the current stream (one hipGraph each
        in graph mode), ('reduce', net name, lo, hi) = start the all-reduce of that range of the net's flat gradient buffer (a
        bucket), ('wait', net name) = the current stream waits for that net's all-reduces, ('update', net name) = the per-net clip +
        Adam.  Buckets in backward order: decoder dense1 .. linear (50 MB) start under the decoder's conv blocks, the conv blocks +
        embeddings (120 MB) under the encoder's backward, the encoder without its conv bank (47 MB) under the bank's seven weight
        gradients, the bank (7 MB) last; every 'run' ends with the side streams joined, so its gradients are final."""
        enc, dec = self.Encoder, self.Decoder
        ee, de = enc._engine(), dec._engine()
        st = {}

        def fwd_and_dec_head():
            if counted:
                L.check(L.lib().zs_step_counters(L.ptr(self._seed_dev), L.ptr(self._step_dev), torch.cuda.current_stream(self.device).cuda_stream),
                        'zs_step_counters')
            st['dlogit'] = self._seg_forward(x_btf, c, noise, noise_kind, drop_masks, seed, seed_ptr)
            if self.buckets:
                de.backward_head(st['dlogit'])
            else:
                self._dbits = de.backward(st['dlogit'])
            join_side(self.device)

        def dec_convs():
            self._dbits = de.backward_convs()
            join_side(self.device)

        def enc_main():
            ee.backward_main(self._dbits)
            join_side(self.device)

        def enc_bank():
            ee.backward_bank()
            join_side(self.device)

        acts = [('run', fwd_and_dec_head)]
        if self.buckets:
            acts += [('reduce', 'dec') + dec.flat_range('dense1.weight', 'linear.bias'), ('run', dec_convs),
                     ('reduce', 'dec') + dec.flat_range('conv1.weight', 'conv6.bias'),
                     ('reduce', 'dec') + dec.flat_range('input_emb.weight', 'emb5.weight'),
                     ('run', enc_main), ('reduce', 'enc') + enc.flat_range('conv2.weight', 'linear.bias'),
                     ('run', enc_bank), ('reduce', 'enc') + enc.flat_range('conv1s.0.weight', 'conv1s.6.bias')]
        else:
            acts += [('reduce', 'dec', 0, dec.flat_params()[1].numel()), ('run', self._seg_encbwd),
                     ('reduce', 'enc', 0, enc.flat_params()[1].numel())]
        # the decoder's clip + Adam + re-pack run while the encoder's all-reduces are still in flight
        return acts + [('wait', 'dec'), ('update', 'dec'), ('wait', 'enc'), ('update', 'enc')]

    def _do_action(self, a):
        """Execute a 'reduce' / 'wait' action (the 'run' / 'update' actions are launches: eager or captured)."""
        net = self.Decoder if a[1] == 'dec' else self.Encoder
        if a[0] == 'reduce':
            self.reducer.start(net.flat_params()[1][a[2]:a[3]], tag=a[1])
        elif a[0] == 'wait':
            self.reducer.finish(a[1])

    def _seg_encbwd(self):
        self.Encoder._engine().backward(self._dbits)
        join_side(self.device)

    def step(self, x_btf, c, noise=None, noise_kind=2, drop_masks=None, seed=None, update=True):
        """x_btf: fp32 [B, T, F] on the device (the loader's native layout), c: int64 [B].
        Returns the device scalar holding loss_rec."""
        enc, dec = self.Encoder, self.Decoder
        enc.train(); dec.train()
        multi = parallel.multi_rank()
        plain = noise is None and drop_masks is None and seed is None and update
        # multi-rank: three graphs with the all-reduces launched between them (ZS_GRAPH_MULTI=0: every launch from the host)
        if plain and self.use_graph and (not multi or os.environ.get('ZS_GRAPH_MULTI', '1') == '1'):
            return self._graph_step(x_btf, c, multi)
        if seed is None:
            seed = (self.adam_step + 1) * 0x9E3779B97F4A7C15 % (1 << 63) + parallel.rank()
        if multi:
            for a in self._multi_actions(x_btf, c, noise, noise_kind, drop_masks, seed, None, counted=False):
                if a[0] == 'run':
                    a[1]()
                elif a[0] != 'update':             # (the update below: host-side bias correction, optional)
                    self._do_action(a)
        else:
            self._seg_forward_decbwd(x_btf, c, noise, noise_kind, drop_masks, seed, None)
            self._seg_encbwd()
        if update:
            self.optimizer_step()
        return self._loss

    # ---- graph mode ------------------------------------------------------------------------------------------------
    def _counted_step_eager(self, x_btf, c, multi):
        """Same launches as the captured step (device-side seed / step count), executed eagerly."""
        if multi:
            for a in self._multi_actions(x_btf, c, None, 2, None, parallel.rank(), L.ptr(self._seed_dev), counted=True):
                if a[0] == 'run':
                    a[1]()
                elif a[0] == 'update':
                    self._net_device_update(a[1], self.Decoder if a[1] == 'dec' else self.Encoder)
                else:
                    self._do_action(a)
            self.adam_step += 1
            return
        st = torch.cuda.current_stream(self.device).cuda_stream
        L.check(L.lib().zs_step_counters(L.ptr(self._seed_dev), L.ptr(self._step_dev), st), 'zs_step_counters')
        self._seg_forward_decbwd(x_btf, c, None, 2, None, parallel.rank(), L.ptr(self._seed_dev))
        if self.early_dec_update:
            self._early_decoder_update()
        self._seg_encbwd()
        self._optimizer_device_step()

    def _net_device_update(self, name, net):
        """Per-net clip (utils.py:53-55) + Adam with the step count read from device memory + re-pack, on the current stream."""
        o = self._opt[name]
        st = torch.cuda.current_stream(self.device).cuda_stream
        flat, gflat = net.flat_params()
        L.check(L.lib().zs_sqnorm(L.ptr(gflat), gflat.numel(), L.ptr(o['part']), L.ptr(o['sq']), st), 'zs_sqnorm')
        b1, b2 = self.betas
        L.call('zs_adam_clip', 'ZsAdam', st, p=L.ptr(flat), g=L.ptr(gflat), m=L.ptr(o['m']), v=L.ptr(o['v']),
               n=flat.numel(), lr=self.lr, beta1=b1, beta2=b2, eps=1e-8, bc1=1.0, bc2=1.0, sumsq=L.ptr(o['sq']),
               max_norm=self.max_grad_norm, write_clipped_grad=0, step_ptr=L.ptr(self._step_dev), grad_scale=self.reducer.scale)
        net.repack()

    def _early_decoder_update(self):
        """Single rank: the decoder's gradients are final once its backward (main stream) and its weight gradients (side
        stream) are done, so its clip + Adam + re-pack run on a third stream under the encoder's backward."""
        main = torch.cuda.current_stream(self.device)
        os_ = layers.opt_stream(self.device)
        ev_main = torch.cuda.Event()
        ev_main.record(main)
        os_.wait_event(ev_main)
        for ev in layers.side_events(self.device):
            os_.wait_event(ev)
        with torch.cuda.stream(os_):
            self._net_device_update('dec', self.Decoder)
        self._dec_updated = True

    def _optimizer_device_step(self):
        """clip + Adam with the step count read from device memory, then re-pack the GEMM operands."""
        if getattr(self, '_dec_updated', False):
            self._net_device_update('enc', self.Encoder)
            ev = torch.cuda.Event()
            ev.record(layers.opt_stream(self.device))
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._dec_updated = False
        else:
            self._net_device_update('enc', self.Encoder)
            self._net_device_update('dec', self.Decoder)
        self.adam_step += 1

    def _graph_step(self, x_btf, c, multi):
        key = tuple(x_btf.shape)
        ent = self._graphs.get(key)
        if ent is None:
            if self._eager_calls < self.graph_warmup:          # warm-up: allocate every buffer, grow the workspaces
                self._eager_calls += 1
                self._counted_step_eager(x_btf, c, multi)
                return self._loss
            ent = self._capture(x_btf, c, multi, statics=self._statics.get(key))
            self._graphs[key] = ent
        if x_btf.data_ptr() != ent['x'].data_ptr():        # (a loader that fills static_inputs() directly skips these copies)
            ent['x'].copy_(x_btf, non_blocking=True)
        if c.data_ptr() != ent['c'].data_ptr():
            ent['c'].copy_(c, non_blocking=True)
        self._replay(ent, multi)
        return self._loss

    def static_inputs(self, B, T, F):
        """(x fp32 [B, T, F], c int64 [B]): the device buffers the captured step of this shape reads its batch from.  A loader that
        writes the next batch straight into them and passes them to step() saves the device-to-device copy of the batch (67 MB
        at B = 256) that step() otherwise makes in front of every replay.  Call before the first step of that shape."""
        key = (B, T, F)
        if key not in self._statics:
            if key in self._graphs:
                ent = self._graphs[key]
                self._statics[key] = (ent['x'], ent['c'])
            else:
                self._statics[key] = (torch.zeros(B, T, F, dtype=torch.float32, device=self.device),
                                      torch.zeros(B, dtype=torch.int64, device=self.device))
        return self._statics[key]

    def _capture(self, x_btf, c, multi, statics=None, prefetch=None):
        """Capture the step on static inputs.  prefetch = (dst_x, src_x_pinned, dst_c, src_c_pinned): an H2D copy of the NEXT
        batch captured as a parallel branch of the (first) graph -- it runs on its own stream beside the kernels of this step."""
        xs, cs = statics if statics is not None else (x_btf.clone(), c.clone())
        torch.cuda.synchronize(self.device)
        st_ptr = L.ptr(self._seed_dev)
        graphs = []
        step0 = self.adam_step

        state = {'ev': None}

        def join_prefetch():
            if state['ev'] is not None:                            # join inside the same graph
                torch.cuda.current_stream(self.device).wait_event(state['ev'])
                state['ev'] = None

        def fork_prefetch():
            main = torch.cuda.current_stream(self.device)
            if prefetch is not None:
                # The fetch of the NEXT batch forks at the very start of the graph and joins at its very end: measured with kernel
                # traces, wherever the branch forks the executor of this ROCm stack does not run it truly
                # beside the chain (the 1.2 ms transfer costs 1.1-1.4 ms of step time); this placement is the cheapest.
                cs_ = layers.copy_stream(self.device)
                ev0 = torch.cuda.Event()
                ev0.record(main)
                cs_.wait_event(ev0)
                with torch.cuda.stream(cs_):
                    if self.fetch_by_kernel:                       # kernel nodes that read the pinned buffers over PCIe (zs_amd.h)
                        for src, dst in ((prefetch[3], prefetch[2]), (prefetch[1], prefetch[0])):
                            L.check(L.lib().zs_host_fetch(L.ptr(src), L.ptr(dst), dst.numel() * dst.element_size(), self.fetch_wgs,
                                                          cs_.cuda_stream), 'zs_host_fetch')
                    else:
                        prefetch[0].copy_(prefetch[1], non_blocking=True)
                        prefetch[2].copy_(prefetch[3], non_blocking=True)
                    state['ev'] = torch.cuda.Event()
                    state['ev'].record(cs_)

        def single():
            fork_prefetch()
            L.check(L.lib().zs_step_counters(L.ptr(self._seed_dev), L.ptr(self._step_dev), torch.cuda.current_stream(self.device).cuda_stream),
                    'zs_step_counters')
            self._seg_forward_decbwd(xs, cs, None, 2, None, parallel.rank(), st_ptr)
            if self.early_dec_update:
                self._early_decoder_update()
            self._seg_encbwd()
            self._optimizer_device_step()
            self.adam_step = step0                             # capture does not execute: the caller counts the step
            # the fetch branch joins at the very END of the graph (wherever the executor schedules it, it then runs beside the
            # encoder's backward instead of holding it up)
            join_prefetch()

        pool = [None]

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool[0], capture_error_mode='thread_local'):   # other threads (RCCL watchdog) may call HIP
                fn()
            pool[0] = g.pool()
            graphs.append(g)
            return g

        if not multi:
            capture(single)
            return {'graphs': graphs, 'x': xs, 'c': cs, 'plan': None}
        # data parallel: one graph per 'run' / 'update' action, the all-reduces are issued between the replays
        plan = []
        first = [True]
        for a in self._multi_actions(xs, cs, None, 2, None, parallel.rank(), st_ptr, counted=True):
            if a[0] == 'run':
                def seg(fn=a[1], with_fetch=first[0]):
                    if with_fetch:
                        fork_prefetch()
                    fn()
                    if with_fetch:
                        join_prefetch()                            # several graphs: the branch must end inside the first
                first[0] = False
                plan.append(('graph', capture(seg)))
            elif a[0] == 'update':
                plan.append(('graph', capture(lambda n=a[1]: self._net_device_update(n, self.Decoder if n == 'dec' else self.Encoder))))
            else:
                plan.append(a)
        return {'graphs': graphs, 'x': xs, 'c': cs, 'plan': plan}

    def _replay(self, ent, multi):
        if not multi:
            ent['graphs'][0].replay()
        else:
            for a in ent['plan']:
                if a[0] == 'graph':
                    a[1].replay()
                else:
                    self._do_action(a)
        self.adam_step += 1

    def host_feeder(self, loader):
        """Iterator that runs one training step per `next()` on the batches of a HOST loader (`next(loader)` -> (c int64 [B],
        x fp32 [B, T, F]) CPU tensors, the reference's DataLoader contract): see HostFedStep."""
        return HostFedStep(self, loader)

    def grad_norms(self):
        """Squared per-net gradient norms as device scalars (Encoder, Decoder are clipped separately).  With more than one
        rank the gradient buffers hold the SUM over ranks after the reduce (the 1/world is applied inside zs_adam_clip):
        these are the squared norms of that sum, i.e. world^2 times those of the averaged gradient."""
        out = []
        st = torch.cuda.current_stream(self.device).cuda_stream
        for name, net in (('enc', self.Encoder), ('dec', self.Decoder)):
            o = self._opt[name]
            _, gflat = net.flat_params()
            L.check(L.lib().zs_sqnorm(L.ptr(gflat), gflat.numel(), L.ptr(o['part']), L.ptr(o['sq']), st), 'zs_sqnorm')
            out.append(o['sq'])
        return out

    def optimizer_step(self):
        """grad_clip([Encoder, Decoder], max_grad_norm) + ae_opt.step()  (trainer.py:331-332)."""
        self.grad_norms()
        self.adam_step += 1
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.adam_step, 1.0 - b2 ** self.adam_step
        st = torch.cuda.current_stream(self.device).cuda_stream
        for name, net in (('enc', self.Encoder), ('dec', self.Decoder)):
            o = self._opt[name]
            flat, gflat = net.flat_params()
            L.call('zs_adam_clip', 'ZsAdam', st, p=L.ptr(flat), g=L.ptr(gflat), m=L.ptr(o['m']), v=L.ptr(o['v']),
                   n=flat.numel(), lr=self.lr, beta1=b1, beta2=b2, eps=1e-8, bc1=bc1, bc2=bc2, sumsq=L.ptr(o['sq']),
                   max_norm=self.max_grad_norm, write_clipped_grad=0, grad_scale=self.reducer.scale)
            net.mark_dirty()
        self._step_dev.fill_(self.adam_step)


class HostFedStep(object):
    """The --train_ae step fed from host memory with the H2D copy INSIDE the captured step (the reference
    copies the 67 MB batch synchronously in front of every iteration, trainer.py:238-244).

    Two static device inputs X[0], X[1] and two pinned staging buffers P[0], P[1]; graph k computes the step on X[k] and, as a
    parallel branch on the copy stream, copies P[1-k] -> X[1-k] (the NEXT batch).  Step i replays graph i % 2, so the copy of batch
    i+1 runs on the DMA engine beside the kernels of step i and costs no time on the critical path.  The host stages batch i+2 into
    P[i % 2] after it has seen the end of step i-1 (the last reader of that buffer): it runs at most two steps ahead of the GPU.
    Until the graphs exist (the AEStep's eager warm-up steps) batches go through a plain synchronous copy."""

    def __init__(self, ae, loader):
        self.ae, self.loader, self.dev = ae, loader, ae.device
        self.i = 0
        self.P = self.PC = self.X = self.C = None
        self.ents = None
        self.done = [torch.cuda.Event(), torch.cuda.Event()]
        self._next = self._fetch()

    def _fetch(self):
        c, x = next(self.loader)[:2]
        c = c.long().contiguous()
        n_spk = self.ae.Decoder.c_a
        if c.numel() and (int(c.min()) < 0 or int(c.max()) >= n_spk):         # an index outside the embedding tables would fault the GPU
            raise ValueError('speaker index outside [0, %d) in the batch' % n_spk)
        return c, x.float().contiguous()

    def _alloc(self, c, x):
        def pair(shape, dtype):
            # storage rounded up to 16 bytes (zs_host_fetch granularity); zero-initialised: never a stale speaker index
            n = 1
            for d in shape:
                n *= d
            es = torch.empty(0, dtype=dtype).element_size()
            n_pad = (n * es + 15) // 16 * 16 // es
            host = [torch.zeros(n_pad, dtype=dtype).pin_memory() for _ in range(2)]
            devt = [torch.zeros(n_pad, dtype=dtype, device=self.dev) for _ in range(2)]
            return host, devt, [h[:n].view(shape) for h in host], [d[:n].view(shape) for d in devt]
        self._Pfull, self._Xfull, self.P, self.X = pair(tuple(x.shape), torch.float32)
        self._PCfull, self._Cfull, self.PC, self.C = pair(tuple(c.shape), torch.int64)

    def __iter__(self):
        return self

    def __next__(self):
        ae = self.ae
        multi = parallel.multi_rank()
        c, x = self._next
        if self.P is None:
            self._alloc(c, x)
        if self.ents is None:
            if ae._eager_calls < ae.graph_warmup or not ae.use_graph:       # warm-up (or graphs off): synchronous copy, eager launches
                ae._eager_calls += 1
                self.X[0].copy_(x); self.C[0].copy_(c)
                if ae.use_graph:
                    ae._counted_step_eager(self.X[0], self.C[0], multi)
                else:
                    ae.step(self.X[0], self.C[0])
                self._next = self._fetch()
                return ae._loss
            # capture graph k on X[k] with the prefetch of the other slot as a branch; batch i -> X[0], batch i+1 -> P[1]
            self.ents = [ae._capture(None, None, multi, statics=(self.X[k], self.C[k]),
                                     prefetch=(self._Xfull[1 - k], self._Pfull[1 - k], self._Cfull[1 - k], self._PCfull[1 - k]))
                         for k in range(2)]
            self.X[0].copy_(x); self.C[0].copy_(c)
            self.i = 0
            self._next = self._fetch()
            self.P[1].copy_(self._next[1]); self.PC[1].copy_(self._next[0])
        k = self.i % 2
        ae._replay(self.ents[k], multi)                                       # step i on X[k]; P[1-k] -> X[1-k] beside it
        self.done[k].record(torch.cuda.current_stream(self.dev))
        # stage batch i+2 into P[k]: its last reader was graph 1-k at step i-1
        self._next = self._fetch()                                            # = batch i+2 (batch i+1 is already in P[1-k])
        if self.i >= 1:
            self.done[1 - k].synchronize()
        self.PC[k].copy_(self._next[0])
        self.P[k].copy_(self._next[1])
        self.i += 1
        return ae._loss


class ClfStep(object):
    """Speaker-classifier side of stage 1 (trainer.py:349-465): the D step trains SpeakerClassifier on the encoder's
    pre-activation `enc` (CE, per-net clip, Adam betas (0.5, 0.9)); the G step trains Encoder+Decoder with
    loss_rec - alpha * loss_clf (the classifier only passes the gradient through)."""

    def __init__(self, ae, classifier, lr=1e-4, betas=(0.5, 0.9), max_grad_norm=5.0):
        self.ae, self.clf = ae, classifier
        self.lr, self.betas, self.max_grad_norm = float(lr), betas, float(max_grad_norm)
        self.adam_step = 0
        dev = ae.device
        self.device = dev
        flat, _ = classifier.flat_params()
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.part = torch.zeros(1024, dtype=torch.float64, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.correct = torch.zeros(1, dtype=torch.int32, device=dev)
        self._dl = None

    def _classify(self, logits_act, c, grad_scale, seed, drop_masks=None):
        """enc logits Act fp32 [B,T',2E] -> CE loss (device scalar), fills self._dl = grad_scale * dCE/dlogits."""
        clf = self.clf
        clf.train()
        ce = clf._engine()
        B, T = logits_act.B, logits_act.T
        x = clf.input_act(logits_act.valid())
        out = ce.forward(x, True, seed=seed, drop_masks=drop_masks)
        st = torch.cuda.current_stream(self.device).cuda_stream
        if self._dl is None or self._dl.shape != (B, out.ld):
            self._dl = torch.zeros(B, out.ld, dtype=torch.float32, device=self.device)
        L.call('zs_softmax_ce', 'ZsSoftmaxCE', st, logits=out.ptr(), ld=out.ld, target=L.ptr(c), B=B, n_class=clf.n_class,
               loss_out=L.ptr(self.loss), dlogits=L.ptr(self._dl), ldg=out.ld, grad_scale=float(grad_scale),
               correct_out=L.ptr(self.correct))                                              # cal_loss / cal_acc, trainer.py:297-313
        return ce, out

    def d_step(self, x_btf, c, alpha_dis=1.0, seed=None, update=True, noise=None, noise_kind=2, drop_masks=None, clf_masks=None):
        """One classifier update (trainer.py:352-366 / 396-411).  Returns (loss_clf, n_correct) device scalars."""
        enc = self.ae.Encoder
        enc.train()
        ee = enc._engine()
        if seed is None:
            seed = (self.adam_step + 1) * 0x9E3779B97F4A7C15 % (1 << 62) + 17 * parallel.rank()
        _, _, logits = ee.forward(x_btf, True, noise=noise, noise_kind=noise_kind, seed=seed, drop_masks=drop_masks)
        ce, out = self._classify(logits, c, alpha_dis, seed + 1, clf_masks)
        ce.backward(self._dl, out.ld, need_dx=False)
        join_side(self.device)
        if parallel.world_size() > 1:
            self.ae.reducer.start(self.clf.flat_params()[1])
            self.ae.reducer.finish()
        if update:
            self.optimizer_step()
        return self.loss, self.correct

    def optimizer_step(self):
        st = torch.cuda.current_stream(self.device).cuda_stream
        flat, gflat = self.clf.flat_params()
        L.check(L.lib().zs_sqnorm(L.ptr(gflat), gflat.numel(), L.ptr(self.part), L.ptr(self.sq), st), 'zs_sqnorm')
        self.adam_step += 1
        b1, b2 = self.betas
        L.call('zs_adam_clip', 'ZsAdam', st, p=L.ptr(flat), g=L.ptr(gflat), m=L.ptr(self.m), v=L.ptr(self.v), n=flat.numel(),
               lr=self.lr, beta1=b1, beta2=b2, eps=1e-8, bc1=1.0 - b1 ** self.adam_step, bc2=1.0 - b2 ** self.adam_step,
               sumsq=L.ptr(self.sq), max_norm=self.max_grad_norm, write_clipped_grad=0, grad_scale=self.ae.reducer.scale)
        self.clf.mark_dirty()

    def g_step(self, x_btf, c, alpha, seed=None, update=True, noise=None, noise_kind=2, drop_masks=None, clf_masks=None):
        """One autoencoder update with the adversarial term (trainer.py:427-447): loss = loss_rec - alpha * loss_clf.
        Returns (loss_rec, loss_clf, n_correct) device scalars."""
        ae = self.ae
        enc, dec = ae.Encoder, ae.Decoder
        enc.train(); dec.train()
        ee, de = enc._engine(), dec._engine()
        ctx = ee.ctx
        st = ctx.stream
        B, T, F = x_btf.shape
        if seed is None:
            seed = (ae.adam_step + 1) * 0x9E3779B97F4A7C15 % (1 << 62) + 31 * parallel.rank()
        bits, _, logits = ee.forward(x_btf, True, noise=noise, noise_kind=noise_kind, seed=seed, drop_masks=drop_masks)
        xdec = de.forward(bits, c, True)
        dlogit = de.ctx.act('t_dlogit_%d_%d' % (B, T), B, xdec.T, F)
        L.call('zs_l1_loss', 'ZsL1Loss', st, dtype=ctx.dt, x_dec=xdec.ptr(), ld_dec=xdec.ld, x=L.ptr(x_btf), ldx=F,
               rows=B * xdec.T, F=F, dlogits=dlogit.ptr(), ldg=dlogit.ld, fill_cols=dlogit.ld, partial=L.ptr(ae._lpart),
               loss_out=L.ptr(ae._loss), grad_scale=1.0)
        ce, out = self._classify(logits, c, -float(alpha), seed + 1, clf_masks)          # maximise the classification loss
        dbits = de.backward(dlogit)
        dx = ce.backward(self._dl, out.ld, need_dx=True, param_grads=False)
        multi = parallel.multi_rank()
        if multi:
            join_side(self.device)
            ae.reducer.start(dec.flat_params()[1])
        ee.backward(dbits, dlogits_extra=dx)
        join_side(self.device)
        if multi:
            ae.reducer.start(enc.flat_params()[1])
            ae.reducer.finish()
        ae.xdec = xdec
        if update:
            ae.optimizer_step()
        return ae._loss, self.loss, self.correct


class Trainer(object):
    def __init__(self, hps, data_loader, g_mode, enc_mode, log_dir='./log/', dtype=None, device=None):
        self.hps = hps
        self.data_loader = data_loader
        self._prefetch = None
        self.model_kept = []
        self.max_keep = hps.max_to_keep
        self.logger = Logger(log_dir)
        self.g_mode = g_mode
        self.enc_mode = enc_mode
        self.dtype = dtype or os.environ.get('ZS_DTYPE', 'fp32')
        if device is None:
            if not torch.cuda.is_available():
                raise L.ZsError('zs_amd.Trainer needs an MI355X (torch.cuda.is_available() is False); no CPU fallback')
            device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
        self.device = torch.device(device)
        self.log_every = int(os.environ.get('ZS_LOG_EVERY', '1'))
        self.h2d_in_graph = os.environ.get('ZS_H2D_IN_GRAPH', '1') == '1'
        self.ckpt_every = int(os.environ.get('ZS_CKPT_EVERY', '1000'))     # the reference saves every 1000 iterations (trainer.py:345)
        self.build_model()

    # ---- model (trainer.py:48-98) -------------------------------------------------------------
    def build_model(self):
        hps, ns = self.hps, self.hps.ns
        dev, dt = self.device, self.dtype
        self.Encoder = Encoder(ns=ns, dp=hps.enc_dp, enc_size=hps.enc_size, seg_len=hps.seg_len, enc_mode=self.enc_mode,
                               dtype=dt).to(dev)
        self.Decoder = Decoder(ns=ns, c_in=hps.enc_size, c_h=hps.emb_size, c_a=hps.n_speakers, seg_len=hps.seg_len,
                               dtype=dt).to(dev)
        if self.g_mode == 'naive':
            self.Generator = Decoder(ns=ns, c_in=hps.enc_size, c_h=hps.emb_size, c_a=hps.n_speakers, seg_len=hps.seg_len,
                                     dtype=dt).to(dev)
        elif self.g_mode in ('targeted', 'targeted_residual'):
            self.Generator = Decoder(ns=ns, c_in=hps.enc_size, c_h=hps.emb_size, c_a=hps.n_target_speakers, seg_len=hps.seg_len,
                                     output_mask=(self.g_mode == 'targeted_residual'), dtype=dt).to(dev)
        elif self.g_mode in ('enhanced', 'spectrogram', 'tacotron'):
            raise NotImplementedError('g_mode %r (stage-2 generator) is outside the MI355X hot path' % self.g_mode)
        else:
            raise NotImplementedError('Invalid Generator mode!')
        enc_size = hps.enc_size
        self.SpeakerClassifier = SpeakerClassifier(ns=ns, c_in=(enc_size * enc_size if self.enc_mode == 'binary' else
                                                                (2 * enc_size if self.enc_mode == 'multilabel_binary' else enc_size)),
                                                   c_h=hps.emb_size, n_class=hps.n_speakers, dp=hps.dis_dp, seg_len=hps.seg_len,
                                                   dtype=dt).to(dev)
        # ---stage two--- (trainer.py:85-97; nn.DataParallel there is a single-GPU identity: one process per GPU here)
        self.PatchDiscriminator = PatchDiscriminator(ns=ns, n_class=(hps.n_speakers if self.g_mode == 'naive' else hps.n_target_speakers),
                                                     seg_len=hps.seg_len, dtype=dt).to(dev)
        self.TargetClassifier = TargetClassifier(ns=ns, n_class=3, seg_len=hps.seg_len, dtype=dt).to(dev)
        self.ae = AEStep(self.Encoder, self.Decoder, lr=hps.lr, betas=(0.5, 0.9), max_grad_norm=hps.max_grad_norm)
        self.clf = ClfStep(self.ae, self.SpeakerClassifier, lr=hps.lr, betas=(0.5, 0.9), max_grad_norm=hps.max_grad_norm)
        self.s2 = None                      # PatchGANStep, built on first use (its optimizer state is 2 x 53 M floats)
        self.source_loader = self.target_loader = None
        self.testing_shift_c = None
        self.sync_params()

    def nets(self):
        return [self.Encoder, self.Decoder, self.Generator, self.SpeakerClassifier, self.PatchDiscriminator, self.TargetClassifier]

    def add_duo_loader(self, source_loader, target_loader):
        """trainer.py:171-173."""
        self.source_loader, self.target_loader = source_loader, target_loader
        self._duo = None

    def stage2(self):
        if self.s2 is None:
            self.s2 = PatchGANStep(self.Encoder, self.Decoder, self.Generator, self.PatchDiscriminator, self.hps, self.g_mode)
        return self.s2

    def sync_params(self):
        """Data parallel: every rank starts from rank 0's weights (each process seeds torch's RNG differently, so the freshly
        built replicas differ); the Adam moments start at zero everywhere.  Called after build_model and load_model."""
        parallel.broadcast_params(self.nets())

    def reset_keep(self):
        self.model_kept = []

    # ---- checkpoints (trainer.py:103-168) -----------------------------------------------------
    def save_model(self, model_path, name, iteration, model_all=True):
        all_model = {
            'encoder': self.Encoder.state_dict(),
            'decoder': self.Decoder.state_dict(),
            'generator': self.Generator.state_dict(),
            'classifier': self.SpeakerClassifier.state_dict(),
        }
        if model_all:                                  # the reference wraps these two in nn.DataParallel: keys carry 'module.'
            all_model['patch_discriminator'] = {'module.' + k: v for k, v in self.PatchDiscriminator.state_dict().items()}
            all_model['target_classifier'] = {'module.' + k: v for k, v in self.TargetClassifier.state_dict().items()}
        else:
            del all_model['classifier']
        all_model = {k: {n: t.detach().cpu().clone() for n, t in sd.items()} for k, sd in all_model.items()}
        new_model_path = '{}-{}-{}'.format(model_path, name, iteration)
        torch.save(all_model, new_model_path)
        self.model_kept.append(new_model_path)
        if len(self.model_kept) >= self.max_keep:
            os.remove(self.model_kept[0])
            self.model_kept.pop(0)

    def load_model(self, model_path, load_model_list, verbose=True, clf_path=None):
        if verbose:
            print('[Trainer] - load model from {}'.format(model_path))
        load_model_list = load_model_list.split(', ')
        all_model = torch.load(model_path, map_location='cpu', weights_only=True)
        if verbose:
            print('[Trainer] - ', end='')
        strip = lambda sd: {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
        for key, net, tag in (('encoder', self.Encoder, 'encoder'), ('decoder', self.Decoder, 'decoder'),
                              ('generator', self.Generator, 'generator'), ('classifier', self.SpeakerClassifier, 'classifier'),
                              ('patch_discriminator', self.PatchDiscriminator, 'patch_discriminator'),
                              ('target_classifier', self.TargetClassifier, 'target_classifier')):
            if key in load_model_list:
                try:
                    if key == 'target_classifier' and clf_path is not None:        # trainer.py:157-161: taken from ANOTHER checkpoint
                        clf_model = torch.load(clf_path, map_location='cpu', weights_only=True)
                        net.load_state_dict(strip(clf_model['target_classifier']))
                        tag = 'target_classifier_another'
                    else:
                        net.load_state_dict(strip(all_model[key]))
                    if verbose:
                        print('[%s], ' % tag, end='')
                except Exception as e:                                   # reference: bare except, prints [x - X]
                    print('[%s - X] (%s), ' % (tag, type(e).__name__), end='')
        if verbose:
            print('Loaded!')
        self.sync_params()

    # ---- inference (trainer.py:180-228) -------------------------------------------------------
    def set_eval(self):
        self.testing_shift_c = torch.tensor([int(self.hps.n_speakers - self.hps.n_target_speakers)], device=self.device)
        self.Encoder.eval()
        self.Decoder.eval()
        self.SpeakerClassifier.eval()
        self.Generator.eval()

    def test_step(self, x, c, enc_only=False, verbose=True, U=None, G=None):
        self.set_eval()
        x = x.to(self.device).permute(0, 2, 1)
        c = c.to(self.device)
        enc, _ = self.Encoder(x, U=U, G=G)
        x_dec = self.Decoder(enc, c)
        if not enc_only:
            if verbose:
                print('Testing with Autoencoder + Generator, encoding: ', enc.cpu().numpy())
            if self.g_mode != 'naive' and int((c - self.testing_shift_c)[0]) not in range(self.hps.n_target_speakers):
                raise RuntimeError('This generator can only convert to target speakers!')
            if self.g_mode == 'naive':
                x_dec = x_dec + self.Generator(enc, c)
            elif self.g_mode == 'targeted':
                x_dec = x_dec + self.Generator(enc, c - self.testing_shift_c)
            elif self.g_mode == 'targeted_residual':
                x_dec = (x_dec * 1.0) + (1.0 * x_dec * self.Generator(enc, c - self.testing_shift_c))
            else:
                raise NotImplementedError('Invalid Generator mode!')
        elif verbose:
            print('Testing with Autoencoder only, encoding: ', enc.cpu().numpy())
        out = x_dec.cpu().numpy(), enc.cpu().numpy()
        layers.check_status(self.device)                  # the .cpu() above synchronised: a timed-out GRU pass raises here
        return out

    def encoder_test_step(self, x, U=None, G=None):
        self.set_eval()
        x = x.to(self.device).permute(0, 2, 1)
        enc, _ = self.Encoder(x, U=U, G=G)
        out = enc.cpu().numpy()
        layers.check_status(self.device)
        return out

    # ---- training pieces (trainer.py:238-254) --------------------------------------------------
    def permute_data(self, data, load_mel=False):
        C = data[0].to(self.device, non_blocking=True)
        X = data[1].to(self.device, non_blocking=True).permute(0, 2, 1)
        return C, X

    def encode_step(self, x):
        return self.Encoder(x)

    def decode_step(self, enc, c):
        return self.Decoder(enc, c)

    def clf_step(self, enc):
        return self.SpeakerClassifier(enc)

    def ae_step(self, x_btf, c, **kw):
        """One fused --train_ae iteration; see AEStep.step."""
        return self.ae.step(x_btf, c, **kw)

    # ---- the loop (trainer.py:316-347) -----------------------------------------------------------
    def _batch(self):
        if self._prefetch is None:
            from .dataloader import DevicePrefetcher
            self._prefetch = DevicePrefetcher(self.data_loader, self.device, n_speakers=self.hps.n_speakers)   # next batch copied under the current step
        return next(self._prefetch)

    def _duo_batch(self):
        """next(source_loader), next(target_loader) through the prefetcher (trainer.py:472-475)."""
        if getattr(self, '_duo', None) is None:
            from .dataloader import DevicePrefetcher
            if self.source_loader is None or self.target_loader is None:
                raise RuntimeError('patchGAN needs add_duo_loader(source_loader, target_loader)')
            self._duo = (DevicePrefetcher(self.source_loader, self.device, n_speakers=self.hps.n_speakers),
                         DevicePrefetcher(self.target_loader, self.device, n_speakers=self.hps.n_speakers, check=self.stage2().check_targets))
        return next(self._duo[0]), next(self._duo[1])

    def train(self, model_path, flag='train', mode='train', target_guided=False):
        hps = self.hps
        is_main = parallel.rank() == 0
        if mode == 'pretrain_AE':                                                         # trainer.py:320-347
            feeder = self.ae.host_feeder(self.data_loader) if (self.h2d_in_graph and self.ae.use_graph) else None
            for iteration in range(hps.enc_pretrain_iters):
                if feeder is not None:
                    loss_t = next(feeder)                                                 # H2D of the next batch inside the captured step
                else:
                    c, x = self._batch()
                    loss_t = self.ae_step(x, c)
                if (iteration % self.log_every == 0) or (iteration + 1 == hps.enc_pretrain_iters):
                    loss_rec = loss_t.item()                                              # the only host sync
                    layers.check_status(self.device)                                      # ... so the GRU status word is read here
                    info = {f'{flag}/pre_loss_rec': loss_rec}
                    slot_value = (iteration + 1, hps.enc_pretrain_iters) + tuple(info.values())
                    if is_main:
                        print('pre_AE:[%06d/%06d], loss_rec=%.3f' % slot_value, end='\r')
                        if iteration % 100 == 0:
                            for tag, value in info.items():
                                self.logger.scalar_summary(tag, value, iteration + 1)
                if (iteration + 1) % self.ckpt_every == 0 and is_main:
                    self.save_model(model_path, 'ae', iteration + 1)
            layers.check_status(self.device)                                              # every rank, after the last step
            if is_main:
                print()
        elif mode == 'pretrain_C':                                                        # trainer.py:349-382
            for iteration in range(hps.dis_pretrain_iters):
                c, x = self._batch()
                loss_t, corr = self.clf.d_step(x, c)
                if (iteration % self.log_every == 0) or (iteration + 1 == hps.dis_pretrain_iters):
                    info = {f'{flag}/pre_loss_clf': loss_t.item(), f'{flag}/pre_acc': corr.item() / float(c.shape[0])}
                    layers.check_status(self.device)
                    slot_value = (iteration + 1, hps.dis_pretrain_iters) + tuple(info.values())
                    if is_main:
                        print('pre_C:[%06d/%06d], loss_clf=%.2f, acc=%.2f' % slot_value, end='\r')
                        if iteration % 100 == 0:
                            for tag, value in info.items():
                                self.logger.scalar_summary(tag, value, iteration + 1)
                if (iteration + 1) % self.ckpt_every == 0 and is_main:
                    self.save_model(model_path, 'c', iteration + 1)
            layers.check_status(self.device)                                              # every rank, after the last step
            if is_main:
                print()
        elif mode == 'train':                                                             # trainer.py:384-465
            for iteration in range(hps.iters):
                if iteration < hps.lat_sched_iters:
                    current_alpha = hps.alpha_enc * (iteration / hps.lat_sched_iters)
                else:
                    current_alpha = hps.alpha_enc
                log_now = (iteration % self.log_every == 0) or (iteration + 1 == hps.iters)
                for step in range(hps.n_latent_steps):                                    # train D
                    c, x = self._batch()
                    loss_t, corr = self.clf.d_step(x, c, alpha_dis=hps.alpha_dis)
                    if log_now and is_main:
                        info = {f'{flag}/D_loss_clf': loss_t.item(),
                                f'{flag}/D_acc': corr.item() / float(c.shape[0])}
                        print('D-%d:[%06d/%06d], loss_clf=%.2f, acc=%.2f' % ((step, iteration + 1, hps.iters) + tuple(info.values())), end='\r')
                        if iteration % 100 == 0:
                            for tag, value in info.items():
                                self.logger.scalar_summary(tag, value, iteration + 1)
                c, x = self._batch()                                                      # train G
                l_rec, l_clf, corr = self.clf.g_step(x, c, current_alpha)
                if log_now and is_main:
                    info = {f'{flag}/loss_rec': l_rec.item(), f'{flag}/G_loss_clf': l_clf.item(), f'{flag}/alpha': current_alpha,
                            f'{flag}/G_acc': corr.item() / float(c.shape[0])}
                    layers.check_status(self.device)
                    print('G:[%06d/%06d], loss_rec=%.3f, loss_clf=%.2f, alpha=%.2e, acc=%.2f' % ((iteration + 1, hps.iters) + tuple(info.values())),
                          end='\r')
                    if iteration % 100 == 0:
                        for tag, value in info.items():
                            self.logger.scalar_summary(tag, value, iteration + 1)
                if (iteration + 1) % self.ckpt_every == 0 and is_main:
                    self.save_model(model_path, 's1', iteration + 1)
            layers.check_status(self.device)                                              # every rank, after the last step
            if is_main:
                print()
        elif mode == 'patchGAN':                                                          # trainer.py:467-560
            s2 = self.stage2()
            B = float(hps.batch_size)
            for iteration in range(hps.patch_iters):
                log_now = (iteration % self.log_every == 0) or (iteration + 1 == hps.patch_iters)
                for step in range(hps.n_patch_steps):                                     # train D
                    (_, x_s), (c_t, x_t) = self._duo_batch()
                    r = s2.d_step(x_s, x_t, c_t)
                    if log_now and is_main:
                        info = {f'{flag}/w_dis': r['w_dis'].item(), f'{flag}/gp': r['gp'].item(), f'{flag}/real_loss_clf': r['real_loss_clf'].item(),
                                f'{flag}/real_acc': r['correct'].item() / B}
                        print('patch_D-%d:[%06d/%06d], w_dis=%.2f, gp=%.2f, loss_clf=%.2f, acc=%.2f' %
                              ((step, iteration + 1, hps.patch_iters) + tuple(info.values())), end='\r')
                        if iteration % 100 == 0:
                            for tag, value in info.items():
                                self.logger.scalar_summary(tag, value, iteration + 1)
                (_, x_s), (c_t, x_t) = self._duo_batch()                                  # train G
                r = s2.g_step(x_s, x_t, c_t)
                loss_rec = s2.tg_step(x_t, c_t) if target_guided else None                # teacher forcing (trainer.py:535-541)
                if log_now:
                    layers.check_status(self.device)
                if log_now and is_main:
                    info = {f'{flag}/loss_adv': r['loss_adv'].item(), f'{flag}/fake_loss_clf': r['fake_loss_clf'].item(),
                            f'{flag}/fake_acc': r['correct'].item() / B, f'{flag}/tg_rec': loss_rec.item() if target_guided else 0.000}
                    print('patch_G:[%06d/%06d], loss_adv=%.2f, loss_clf=%.2f, acc=%.2f, tg_rec=%.3f' %
                          ((iteration + 1, hps.patch_iters) + tuple(info.values())), end='\r')
                    if iteration % 100 == 0:
                        for tag, value in info.items():
                            self.logger.scalar_summary(tag, value, iteration + 1)
                if (iteration + 1) % self.ckpt_every == 0 and is_main:
                    self.save_model(model_path, 's2', iteration + 1)
            layers.check_status(self.device)
            if is_main:
                print()
        else:
            raise NotImplementedError("mode %r (autolocker / t_classify / Tacotron) is outside the path this build covers" % mode)
