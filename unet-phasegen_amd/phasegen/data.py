"""Spectrogram batching of the reference's ``data.py`` (data.py:7-47), device resident.

``get_spec_and_angle`` runs the polar transform on the GPU (pg_polar).  ``get_fft_npy_loader`` keeps the reference's
signature and yield format -- ``[x, label]`` with x (b, 2, bins, frames) float32, label (b, 1) float32, shuffled
every epoch, short last batch kept -- but the whole dataset lives in HBM (288 GB per MI355X) after ONE upload, so a
training step does no host->device copy at all (the reference uploads the batch four times per step,
train.py:42,49,50,57; ``.cuda()`` on what this loader yields is a no-op, so the reference loop runs unchanged).
For data-parallel training rank r of W takes clips r::W of each epoch's permutation (same seed on every rank), after the
permutation has been cut to a whole number of GLOBAL batches (W x batch_size clips): every rank then yields the same number
of full batches per epoch, so the ranks issue the same number of gradient all-reduces (a rank that ran one step more than
its peers would pair its collectives with the next epoch's and hang or silently diverge).
"""
import os

import numpy as np
import torch

from . import ops


def _device(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("phasegen.data needs an MI355X (the polar transform and batching run on the device)")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def get_spec_and_angle(data, use_exp=True, device=None, as_numpy=True, chunk=256):
    """data.py:39-47.  ``data``: (N, 2, bins, frames) [re; im] (numpy, memmap or tensor) -> [log1p|z|; angle], float32."""
    dev = _device(device)
    if torch.is_tensor(data) and data.is_cuda:
        out = ops.polar(data.contiguous().float(), use_exp=use_exp)
        return out.cpu().numpy() if as_numpy else out
    n = data.shape[0]
    out = torch.empty(tuple(data.shape), device=dev, dtype=torch.float32)
    for s0 in range(0, n, chunk):                    # stream the (possibly memory-mapped) array through the device
        blk = torch.from_numpy(np.array(data[s0:s0 + chunk], dtype=np.float32)).to(dev)
        ops.polar(blk, out[s0:s0 + chunk], use_exp=use_exp)
    return out.cpu().numpy() if as_numpy else out


class SpectrogramLoader:
    """Iterable with torch DataLoader's surface as the reference uses it (``for i, d in enumerate(loader)``,
    ``loader.__iter__().__next__()``, ``len(loader)``, ``.dataset``, ``.batch_size``)."""

    def __init__(self, data, labels, batch_size, shuffle=True, rank=0, world=1, seed=None):
        self.data, self.labels = data, labels
        self.batch_size, self.shuffle = batch_size, shuffle
        self.rank, self.world = rank, world
        self.dataset = self
        self._gen = torch.Generator(device="cpu")
        self._gen.manual_seed(torch.initial_seed() if seed is None else seed)

    def _usable(self):
        """Clips of an epoch that are dealt out: all of them on one GPU (the short last batch is yielded and dropped by the
        caller, train.py:38-39); with W ranks a whole number of global batches, so that every rank steps equally often."""
        n = self.data.shape[0]
        if self.world == 1:
            return n
        g = self.world * self.batch_size
        if n < g:
            raise ValueError(f"data-parallel loader: {n} clips do not fill one global batch of {self.world} x {self.batch_size}")
        return n // g * g

    def __len__(self):
        n = len(range(self.rank, self._usable(), self.world))
        return (n + self.batch_size - 1) // self.batch_size

    def num_clips(self):
        return self.data.shape[0]

    def __iter__(self):
        n = self.data.shape[0]
        perm = torch.randperm(n, generator=self._gen) if self.shuffle else torch.arange(n)
        perm = perm[:self._usable()][self.rank::self.world].to(self.data.device)
        for s0 in range(0, perm.numel(), self.batch_size):
            idx = perm[s0:s0 + self.batch_size]
            yield [self.data.index_select(0, idx), self.labels.index_select(0, idx)]


def get_fft_npy_loader(paths, labels=None, batch_size=1, norm=True, precon=False, device=None, rank=0, world=1, seed=None):
    """data.py:7-28.  ``norm`` is accepted and ignored exactly as in the reference (it is never read there)."""
    dev = _device(device)
    if not isinstance(paths, list):
        paths = [paths]
    if labels is None:
        labels = [0]
    datas, targets = [], []
    for p, l in zip(paths, labels):                  # zip truncation is reference behaviour (train.py:18-20)
        if os.path.exists(p):                        # missing paths are silently skipped (data.py:17)
            print("{} exists, start loading ...".format(p))
            d = np.load(p, mmap_mode="r")
            if precon:
                d = get_spec_and_angle(d, device=dev, as_numpy=False)
            else:
                d = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
            datas.append(d)
            targets.append(torch.ones(d.shape[0], 1, device=dev) * l)
    assert len(datas) > 0, "datasets should not be an empty iterable"     # what ConcatDataset([]) raises (data.py:26)
    return SpectrogramLoader(torch.cat(datas) if len(datas) > 1 else datas[0],
                             torch.cat(targets) if len(targets) > 1 else targets[0], batch_size, True, rank, world, seed)
