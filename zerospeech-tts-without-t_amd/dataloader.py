"""Data surface of the reference (dataloader.py:22-78): `DataLoader(dataset, batch_size)` whose
`next()` returns `(LongTensor[B], FloatTensor[B, seg_len, 513])` with the reference's sliding-index /
wrap rule, `Dataset` over the preprocessed HDF5 + JSON index, and a `SyntheticDataset` of the same item
shape (no dataset ships with the reference; h5py is optional)."""
import json
from collections import namedtuple

import numpy as np
import torch


class DataLoader(object):
    """dataloader.py:22-52.  `rank` / `world` (data parallel, not in the reference): `index` is the start of the GLOBAL
    batch of world*batch_size consecutive items, rank r serves items [index + r*B, index + (r+1)*B) of it and every rank
    advances by world*B with the reference's wrap rule applied to the global batch -- so the ranks never serve the same item
    in one pass.  world == 1 is exactly the reference."""

    def __init__(self, dataset, batch_size=16, rank=0, world=1):
        self.dataset = dataset
        self.n_elements = len(self.dataset[0])
        self.batch_size = batch_size
        self.rank, self.world = int(rank), max(1, int(world))
        self.index = 0

    def _fetch(self, size):
        n = len(self.dataset)
        base = self.index + self.rank * size
        samples = [self.dataset[(base + i) % n if self.world > 1 else base + i] for i in range(size)]
        batch = [[s for s in sample] for sample in zip(*samples)]
        batch_tensor = [torch.from_numpy(np.array(data)) for data in batch]
        step = self.world * self.batch_size
        if self.index + 2 * step >= n:                                 # dataloader.py:48-51 wrap rule
            self.index = 0
        else:
            self.index += step
        return tuple(batch_tensor)

    def all(self, size=1000):
        return self._fetch(size)

    def __iter__(self):
        return self

    def __next__(self):
        return self._fetch(self.batch_size)


class DevicePrefetcher(object):
    """Keeps the GPU fed from a host DataLoader (rows a1/a2 of SURVEY section 8: the reference builds the batch on the host
    and copies 67 MB per step synchronously, trainer.py:238-244): the NEXT batch is staged in pinned memory and copied on a
    separate HIP stream while the current step runs; `next()` returns device tensors `(c int64[B], x fp32[B, seg_len, F])`
    that stay valid until the call after next (double buffering)."""

    def __init__(self, loader, device, n_speakers=None, check=None):
        """check: optional callable on the HOST index tensor of every batch (e.g. stage2.PatchGANStep.check_targets): range checks
        happen here, where the indices are still host memory, not as device-to-host syncs inside the training step."""
        self.loader, self.device, self.n_speakers, self.check = loader, torch.device(device), n_speakers, check
        self.stream = torch.cuda.Stream(self.device)
        self.slots = [None, None]            # (pinned c, pinned x, dev c, dev x, event)
        self.turn = 0
        self._issue(0)

    def _issue(self, i):
        c, x = next(self.loader)[:2]
        x = x.float().contiguous()
        c = c.long().contiguous()
        if self.n_speakers is not None and c.numel() and (int(c.min()) < 0 or int(c.max()) >= self.n_speakers):
            raise ValueError('speaker index outside [0, %d) in the batch' % self.n_speakers)   # would fault the GPU in the embedding lookups
        if self.check is not None:
            self.check(c)
        sl = self.slots[i]
        if sl is None or sl[1].shape != x.shape or sl[0].shape != c.shape:
            sl = [torch.empty(c.shape, dtype=torch.int64).pin_memory(), torch.empty(x.shape, dtype=torch.float32).pin_memory(),
                  torch.empty(c.shape, dtype=torch.int64, device=self.device), torch.empty(x.shape, dtype=torch.float32, device=self.device),
                  torch.cuda.Event()]
            self.slots[i] = sl
        src_c, src_x = (c, x) if (c.is_pinned() and x.is_pinned()) else (sl[0], sl[1])
        if src_x is sl[1]:
            # the slot's previous H2D copy reads these pinned buffers asynchronously: the host may be several steps ahead of
            # the GPU (graph replay, no per-step .item()), so wait for that copy before overwriting its source
            sl[4].synchronize()
            sl[0].copy_(c); sl[1].copy_(x)          # pageable batch: stage it (a 67 MB host memcpy per step at B=256)
        # the device buffers of this slot were last read by the step issued two calls ago on the consumer's stream
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            sl[2].copy_(src_c, non_blocking=True)
            sl[3].copy_(src_x, non_blocking=True)
            sl[4].record(self.stream)

    def __iter__(self):
        return self

    def __next__(self):
        i = self.turn
        sl = self.slots[i]
        torch.cuda.current_stream(self.device).wait_event(sl[4])
        self.turn = 1 - i
        self._issue(self.turn)               # start the copy of the following batch now: it runs under this step
        return sl[2], sl[3]


class NpzStore(object):
    """h5py-shaped container on an .npz file: `create_group(name)`, `group.create_dataset(key, data=, dtype=)`, `store[key]`."""

    class _Group(object):
        def __init__(self, store, prefix):
            self.store, self.prefix = store, prefix

        def create_dataset(self, key, data, dtype=np.float32):
            self.store.arrays[self.prefix + '/' + key] = np.asarray(data, dtype=dtype)

    def __init__(self, path, mode='r'):
        self.path, self.mode, self.arrays = path, mode, {}
        if mode == 'r':
            with np.load(path, allow_pickle=False) as z:
                self.arrays = {k: z[k] for k in z.files}

    def create_group(self, name):
        return NpzStore._Group(self, name)

    def __getitem__(self, key):
        return self.arrays[key]

    def keys(self, prefix):
        """Names directly under `prefix` (h5py: list(f[prefix].keys()))."""
        p = prefix.rstrip('/') + '/'
        return sorted({k[len(p):].split('/')[0] for k in self.arrays if k.startswith(p)})

    def close(self):
        if self.mode == 'w':
            np.savez(self.path, **self.arrays)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def open_store(path, mode='r'):
    """HDF5 (h5py) when available and the path is not .npz; the NpzStore otherwise."""
    if not path.endswith('.npz'):
        try:
            import h5py
            return h5py.File(path, mode)
        except ImportError:
            path = path + '.npz'
    return NpzStore(path, mode)


class Dataset(torch.utils.data.Dataset):
    """HDF5- (or .npz-) backed segments, dataloader.py:56-78 (layout `{dset}/{speaker}/{utt}/lin|mel`)."""

    def __init__(self, h5_path, index_path, dset='train', seg_len=64, load_mel=False):
        self.dataset = open_store(h5_path, 'r')         # HDF5 (h5py) or the .npz container written by zs_amd.preprocess
        with open(index_path) as f_index:
            self.indexes = json.load(f_index)
        self.indexer = namedtuple('index', ['speaker', 'i', 't'])
        self.seg_len, self.dset, self.load_mel = seg_len, dset, load_mel

    def __getitem__(self, i):
        index = self.indexer(**self.indexes[i])
        i, t = index.i, index.t
        data = [index.speaker, self.dataset['%s/%s/lin' % (self.dset, i)][t:t + self.seg_len]]
        if self.load_mel:
            data.append(self.dataset['%s/%s/mel' % (self.dset, i)][t:t + self.seg_len])
        return tuple(data)

    def __len__(self):
        return len(self.indexes)


class SyntheticDataset(object):
    """n segments of U[1e-8,1) 'lin' spectrogram frames (the value range preprocess.py:247-252 produces)
    with random speaker ids; same item contract as Dataset.__getitem__."""

    def __init__(self, n_items, seg_len=128, n_bins=513, n_speakers=102, seed=0, rank=0, speaker_offset=0):
        rng = np.random.RandomState(seed * 1000003 + rank)
        self.spk = (rng.randint(0, n_speakers, size=n_items) + speaker_offset).astype(np.int64)
        self.lin = np.clip(rng.rand(n_items, seg_len, n_bins).astype(np.float32), 1e-8, 1.0)

    def __getitem__(self, i):
        return (self.spk[i], self.lin[i])

    def __len__(self):
        return len(self.spk)
