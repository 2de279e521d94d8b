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
    """val_batch: (n, 2, bins, frames) = [logmag; angle] on the device.  Returns {"MSE", "NOPMSE", "LMSE"} floats."""
    val_batch = val_batch.contiguous()
    n, _, bins, _ = val_batch.shape
    mses, nops, lims = [], [], []
    for c in range(n):
        vd = val_batch[c:c + 1]
        pred = model.forward(vd[:, 0])                                   # batch of one, train-mode BN (train.py:76)
        logmag, ang = vd[:, 0], vd[:, 1]
        orig = audio.synthesize(logmag, ang.contiguous(), hop_length)[0]
        hyb = audio.synthesize(logmag, pred[:, :bins].contiguous(), hop_length)[0]
        nop = audio.synthesize(logmag, torch.zeros_like(logmag), hop_length)[0]
        mag = (torch.exp(logmag[0]) - 1.0).cpu().numpy()
        lim, _, _ = audio.griffin_lim(mag, n_fft, hop_length, gl_iters, seed=gl_seed + c)
        lim = torch.from_numpy(lim).to(orig.device)
        mses.append(torch.abs(orig - hyb).mean())
        nops.append(torch.abs(orig - nop).mean())
        lims.append(torch.abs(orig - lim).mean())
    f = lambda v: float(torch.stack(v).mean())      # noqa: E731  (clips have equal length: mean of means == global mean)
    return {"MSE": f(mses), "NOPMSE": f(nops), "LMSE": f(lims)}
