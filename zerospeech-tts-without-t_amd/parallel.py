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
    if world > 1 and not dist.is_initialized():
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


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class GradReducer(object):
    """Average flat gradient buffers across ranks.  start(buf) launches an asynchronous all-reduce (SUM)
    ordered after everything already enqueued on the current stream; finish() waits for all of them and
    applies the 1/world scale."""

    def __init__(self, bucket_bytes=64 << 20):
        self.pending = []
        self.bucket_elems = max(1, bucket_bytes // 4)

    def start(self, flat):
        w = world_size()
        if w == 1:
            return
        n = flat.numel()
        for lo in range(0, n, self.bucket_elems):
            chunk = flat[lo:min(n, lo + self.bucket_elems)]
            self.pending.append((chunk, dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True)))

    def finish(self):
        w = world_size()
        for chunk, work in self.pending:
            work.wait()
            chunk.mul_(1.0 / w)
        self.pending = []


def shard_range(n_items, rank_, world):
    """Contiguous shard of a work list (inference / vocoder replicas: no collective)."""
    per = (n_items + world - 1) // world
    return min(n_items, rank_ * per), min(n_items, (rank_ + 1) * per)
