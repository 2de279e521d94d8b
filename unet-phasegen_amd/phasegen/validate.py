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
def validation_metrics(model, val_batch, hop_length=512, n_fft=2048, gl_iters=250, gl_seed=0):
    """val_batch: (n, 2, bins, frames) = [logmag; angle] on the device.  Returns {"MSE", "NOPMSE", "LMSE"} floats.
    Nothing leaves the device before the three means: the forwards are batch-of-one (train-mode BatchNorm, train.py:76: the
    statistics of a clip must not see the others), everything else -- the three ISTFTs and the Griffin-Lim comparator -- is
    batched over the clips."""
    val_batch = val_batch.contiguous()
    n, _, bins, _ = val_batch.shape
    logmag, ang = val_batch[:, 0].contiguous(), val_batch[:, 1].contiguous()
    phase = torch.empty_like(logmag)
    for c in range(n):
        phase[c] = model.forward(val_batch[c:c + 1, 0])[0, :bins]       # batch of one, train-mode BN (train.py:76)
    orig = audio.synthesize(logmag, ang, hop_length)
    hyb = audio.synthesize(logmag, phase, hop_length)
    nop = audio.synthesize(logmag, torch.zeros_like(logmag), hop_length)
    lim, _, _ = audio.griffin_lim_batch(torch.exp(logmag) - 1.0, n_fft, hop_length, gl_iters, seed=gl_seed)
    # (clips have equal length: the mean over everything == the reference's mean of per-clip means)
    res = torch.stack([torch.abs(orig - hyb).mean(), torch.abs(orig - nop).mean(), torch.abs(orig - lim).mean()]).cpu()
    return {"MSE": float(res[0]), "NOPMSE": float(res[1]), "LMSE": float(res[2])}
