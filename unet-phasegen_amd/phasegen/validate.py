"""Validation metrics of train.py:69-124 without the TensorBoard rendering.

For every validation clip: orig = audio of the true spectrum, hyb = true magnitude + PREDICTED phase (the model's
product), nop = magnitude only (zero phase), lim = 250-iteration Griffin-Lim from the magnitude.  The reference's
"MSE" is ``np.sqrt((a - b)**2)`` averaged, i.e. the mean absolute error (train.py:103-108,122) -- reproduced as is.
Everything heavy (forward, ISTFT, Griffin-Lim) runs on the device.
"""
import numpy as np
import torch

from . import audio


@torch.no_grad()
def validation_metrics(model, val_batch, hop_length=512, n_fft=2048, gl_iters=250, gl_seed=0, group=None, shard=False):
    """val_batch: (n, 2, bins, frames) = [logmag; angle] on the device.  Returns {"MSE", "NOPMSE", "LMSE"} floats.
    Nothing leaves the device before the three means: the forwards are batch-of-one (train-mode BatchNorm, train.py:76: the
    statistics of a clip must not see the others), everything else -- the three ISTFTs and the Griffin-Lim comparator -- is
    batched over the clips.
    ``shard=True`` (data-parallel training; EVERY rank of ``group`` must call it with the same batch): rank r evaluates clips
    r::W and the sums are all-reduced, so no rank sits in the next gradient all-reduce while another one validates, and the
    250 Griffin-Lim iterations run on W GPUs.  Clip c draws its Griffin-Lim start with seed gl_seed + c on whichever rank it
    lands: the result does not depend on W (up to the order of the final fp32 sums).
    BatchNorm buffers: the forwards are train-mode and update running_mean / running_var / num_batches_tracked (the reference's
    validation does, train.py:76 runs the model as it is), so sharded ranks see different clips and their buffers would drift apart;
    after the sharded forwards rank 0's buffers are broadcast, which keeps the replicas identical (what Trainer's construction-time
    check asserts once).  The values then depend on W -- rank 0 saw clips 0, W, 2W, ... -- and on nothing else; the model never reads
    them (no .eval() anywhere), they only travel in checkpoints.
    A non-finite Griffin-Lim or ISTFT result raises, as librosa.util.valid_audio does in the reference (utils.py:41,130)."""
    import torch.distributed as dist
    val_batch = val_batch.contiguous()
    n, _, bins, _ = val_batch.shape
    world = dist.get_world_size(group) if (shard and dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    mine = list(range(rank, n, world))
    sums = torch.zeros(5, device=val_batch.device, dtype=torch.float64)   # sum |orig - hyb|, |orig - nop|, |orig - lim|, elements, non-finite
    if mine:
        vb = val_batch[mine]
        logmag, ang = vb[:, 0].contiguous(), vb[:, 1].contiguous()
        phase = torch.empty_like(logmag)
        for i in range(len(mine)):
            phase[i] = model.forward(vb[i:i + 1, 0])[0, :bins]                   # batch of one, train-mode BN (train.py:76)
        orig = audio.synthesize(logmag, ang, hop_length)
        hyb = audio.synthesize(logmag, phase, hop_length)
        nop = audio.synthesize(logmag, torch.zeros_like(logmag), hop_length)
        lim, _, _ = audio.griffin_lim_batch(torch.exp(logmag) - 1.0, n_fft, hop_length, gl_iters, seed=[gl_seed + c for c in mine])
        sums[0] = torch.abs(orig - hyb).double().sum()
        sums[1] = torch.abs(orig - nop).double().sum()
        sums[2] = torch.abs(orig - lim).double().sum()
        sums[3] = orig.numel()
        sums[4] = sum((~torch.isfinite(t)).sum() for t in (orig, hyb, nop, lim)).double()
    if world > 1:
        dist.all_reduce(sums, group=group)
        arena = model.engine.arena
        src = dist.get_global_rank(group, 0) if group is not None else 0
        for k in sorted(arena.buffers):
            dist.broadcast(arena.buffers[k], src=src, group=group)
    # (clips have equal length: the mean over everything == the reference's mean of per-clip means, train.py:122)
    host = sums.cpu()
    if host[4] != 0:
        raise ValueError("Audio buffer is not finite everywhere")      # librosa.util.valid_audio's message (utils.py:41,130)
    res = host[:3] / host[3]
    return {"MSE": float(res[0]), "NOPMSE": float(res[1]), "LMSE": float(res[2])}
