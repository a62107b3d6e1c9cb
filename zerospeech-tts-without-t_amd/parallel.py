"""Data parallelism for the --train_ae step: one process per GPU, torch.distributed (backend 'nccl' is
RCCL over xGMI on ROCm; 'gloo' for the CPU tests).  The segments of a batch are independent
(InstanceNorm is per sample), so the ONLY exchange is the gradient average: one all-reduce per flat
per-net gradient buffer (decoder 170 MB, encoder 54 MB fp32), the decoder's launched as soon as its
backward is done so it overlaps the encoder's backward.  The per-net clip norm is computed after the
reduce on identical averaged gradients (no extra collective).  The reference itself has no multi-process
path."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or os.environ.get('ZS_FORCE_MULTI') == '1') and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl' and os.environ.get('ZS_FORCE_DEVICE') is None:
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def multi_rank():
    """True when the data-parallel code path is to be taken: more than one rank, or ZS_FORCE_MULTI=1 (a one-rank rehearsal of the
    multi-rank step -- one hipGraph per segment with the bucketed RCCL all-reduces between them -- on a single GPU)."""
    return world_size() > 1 or (os.environ.get('ZS_FORCE_MULTI') == '1' and dist.is_available() and dist.is_initialized())


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class GradReducer(object):
    """Sum flat gradient buffers across ranks.  start(buf) launches an asynchronous all-reduce (SUM) ordered after
    everything already enqueued on the current stream; finish() makes the current stream wait for all of them.
    The buffers then hold the SUM over ranks: the 1/world of the average is `scale`, which the consumer folds into
    its own pass over the gradients (zs_adam_clip's grad_scale) -- no separate pass over 224 MB to apply it.

    One collective per net by default (decoder 170 MB, encoder 54 MB: large messages run RCCL's rings at their
    per-link rate); ZS_BUCKET_MB splits them.  ZS_REDUCE_BF16=1 sends bf16 (half the xGMI bytes, rounding once on
    the way in; the sum is accumulated by RCCL in bf16) -- off by default, the fp32 reduce keeps the averaged
    gradients exact to fp32."""

    def __init__(self, bucket_bytes=None, bf16=None):
        self.pending = []
        if bucket_bytes is None:
            mb = int(os.environ.get('ZS_BUCKET_MB', '0'))
            bucket_bytes = (mb << 20) if mb > 0 else (1 << 62)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.bf16 = (os.environ.get('ZS_REDUCE_BF16', '0') == '1') if bf16 is None else bool(bf16)
        self._half = {}

    @property
    def scale(self):
        return 1.0 / world_size()

    def start(self, flat, tag=None):
        """tag: any label (e.g. the net's name); finish(tag) waits for the collectives started under it."""
        if not multi_rank():
            return
        n = flat.numel()
        if self.bf16:
            h = self._half.get(flat.data_ptr())
            if h is None or h.numel() != n:
                h = self._half[flat.data_ptr()] = torch.empty(n, dtype=torch.bfloat16, device=flat.device)
            h.copy_(flat)
            self.pending.append((tag, flat, h, dist.all_reduce(h, op=dist.ReduceOp.SUM, async_op=True)))
            return
        for lo in range(0, n, self.bucket_elems):
            chunk = flat[lo:min(n, lo + self.bucket_elems)]
            self.pending.append((tag, chunk, None, dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True)))

    def finish(self, tag=None):
        """Make the current stream wait for the pending collectives started under `tag` (all of them when tag is None)."""
        keep = []
        for ent in self.pending:
            t, dst, half, work = ent
            if tag is not None and t != tag:
                keep.append(ent)
                continue
            work.wait()
            if half is not None:
                dst.copy_(half)
        self.pending = keep


def broadcast_params(nets, src=0):
    """Make every rank start from rank `src`'s weights: the modules draw their initial parameters from torch's
    default RNG, whose seed differs per process.  Broadcasts each net's flat fp32 parameter buffer."""
    if world_size() == 1:
        return
    for net in nets:
        flat, _ = net.flat_params()
        dist.broadcast(flat, src=src)
        net.mark_dirty()


def shard_range(n_items, rank_, world):
    """Contiguous shard of a work list (inference / vocoder replicas: no collective)."""
    per = (n_items + world - 1) // world
    return min(n_items, rank_ * per), min(n_items, (rank_ + 1) * per)
