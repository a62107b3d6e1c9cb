"""utils.py surface of the reference (utils.py:20-24, 43-55, 80-85) that the AE path uses."""
import json
import os

import torch


def cc(net):
    return net.cuda() if torch.cuda.is_available() else net


def to_var(x, requires_grad=True):
    """utils.py:43-45.  requires_grad on the input batch is not needed by the HIP path (the reference
    computes an unused dL/dx); kept for signature compatibility."""
    return x.cuda() if torch.cuda.is_available() else x


def reset_grad(net_list):
    for net in net_list:
        net.zero_grad()


def grad_clip(net_list, max_grad_norm):
    """utils.py:53-55: one clip_grad_norm_ PER net.  The training steps here never call it -- the per-net norm and the
    clip coefficient are fused into zs_sqnorm + zs_adam_clip -- it is kept for callers of the reference's utils API and
    works on the modules' flat gradient views (p.grad are views of one buffer per net)."""
    for net in net_list:
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_grad_norm)


class Logger(object):
    """tensorboardX.SummaryWriter when available (utils.py:80-85); otherwise the same (tag, value, step)
    scalars go to <log_dir>/scalars.jsonl."""

    def __init__(self, log_dir='./log'):
        self.writer, self.fh = None, None
        try:
            from tensorboardX import SummaryWriter
            self.writer = SummaryWriter(log_dir)
        except ImportError:
            os.makedirs(log_dir, exist_ok=True)
            self.fh = open(os.path.join(log_dir, 'scalars.jsonl'), 'a')

    def scalar_summary(self, tag, value, step):
        if self.writer is not None:
            self.writer.add_scalar(tag, value, step)
        else:
            self.fh.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step)}) + '\n')
            self.fh.flush()
