"""GPU: values (not just finiteness) for the SURVEY.md §8f "next" rows and for M0's initialisation.

  N4  validation metrics MSE / NOPMSE / LMSE (train.py:69-124; the reference's "MSE" is np.sqrt((a - b)**2) averaged, i.e. the
      mean absolute error) against the same quantities computed by oracle/signal_ref.py + oracle/unet_ref.py on the same clips
  M0  UNetModel.__init__'s default initialisation (model.py:23-36 builds stock torch modules; weights_init is never called):
      conv weights U(+-1/sqrt(fan_in)), gamma 1, beta 0, running stats (0, 1)
"""
import math

import numpy as np
import pytest
import torch

from oracle import signal_ref, unet_ref
from phasegen import detgen

pytestmark = pytest.mark.gpu


def test_validation_metrics_vs_oracle():
    from phasegen.model import UNetModel
    from phasegen.validate import validation_metrics
    C, L, n_fft, hop, n_clips, iters = 16, 24, 32, 8, 3, 4
    n = hop * (L - 1)
    clips = [detgen.make_clip(n, seed=120 + i) for i in range(n_clips)]
    P = np.ascontiguousarray(signal_ref.get_spec_and_angle(np.stack([signal_ref.chunk_and_stft(c, n_fft, hop) for c in clips])), dtype=np.float32)
    pn = detgen.make_params(C, seed=0)
    model = UNetModel(C, 2 * C, precision="fp32").load_numpy(pn)
    got = validation_metrics(model, torch.from_numpy(P).cuda(), hop, n_fft, gl_iters=iters, gl_seed=0)
    po = unet_ref.to_torch(pn)
    mses, nops, lims = [], [], []
    for c in range(n_clips):
        with torch.no_grad():
            pred = unet_ref.unet_forward(po, torch.from_numpy(P[c:c + 1, 0].copy())).numpy()[0, :C]      # batch of one, train-mode BN
        mag = np.exp(P[c, 0]) - 1
        orig = signal_ref.generate_audio(mag * np.exp(P[c, 1] * 1.j), hop, is_stft=True)                # train.py:83,99
        hyb = signal_ref.generate_audio(mag * np.exp(pred * 1.j), hop, is_stft=True)                    # train.py:84,100
        nop = signal_ref.generate_audio(mag.astype(np.complex64), hop, is_stft=True)                    # train.py:85,101
        g = torch.Generator(device="cpu")
        g.manual_seed(c)                                                                                 # validate.py: seed = gl_seed + c
        init = torch.randn(n, generator=g, dtype=torch.float64).numpy()
        lim, _, _ = signal_ref.griffin_lim(mag, n_fft, hop, iters, init)                                 # train.py:102
        mses.extend(np.sqrt((orig - hyb) ** 2)); nops.extend(np.sqrt((orig - nop) ** 2)); lims.extend(np.sqrt((orig - lim) ** 2))
    want = {"MSE": float(np.mean(mses)), "NOPMSE": float(np.mean(nops)), "LMSE": float(np.mean(lims))}    # train.py:122
    print("\nvalidation metrics", got, want)
    assert abs(got["MSE"] - want["MSE"]) < 1e-5 * want["MSE"]              # measured 7e-8
    assert abs(got["NOPMSE"] - want["NOPMSE"]) < 1e-5 * want["NOPMSE"]
    assert abs(got["LMSE"] - want["LMSE"]) < 1e-4 * want["LMSE"]           # 4 Griffin-Lim iterations; measured 7e-8
    assert want["NOPMSE"] > 0 and want["MSE"] > 0


def test_default_initialisation_matches_torch_defaults():
    from phasegen.model import UNetModel
    torch.manual_seed(123)
    C = 32
    m = UNetModel(C, 2 * C)
    a = m.engine.arena
    shapes = detgen.conv_shapes(C)
    for k, shp in shapes.items():
        w = a.p(k).double()
        b = 1.0 / math.sqrt(shp[1] * shp[2])              # torch: kaiming_uniform(a=sqrt(5)) -> U(+-1/sqrt(fan_in)), fan_in = shape[1]*k
        assert float(w.abs().max()) <= b and float(w.abs().max()) > 0.98 * b, k
        assert abs(float(w.mean())) < 0.02 * b and abs(float(w.std()) - b / math.sqrt(3)) < 0.02 * b, k
    for k in detgen.BN_KEYS:
        assert torch.all(a.p(k + ".weight") == 1) and torch.all(a.p(k + ".bias") == 0)
        assert torch.all(a.buffers[k + ".running_mean"] == 0) and torch.all(a.buffers[k + ".running_var"] == 1)
        assert int(a.buffers[k + ".num_batches_tracked"]) == 0
    m2 = UNetModel(C, 2 * C)                              # a second model continues the generator stream: different weights
    assert not torch.equal(m2.engine.arena.flat, a.flat)
    torch.manual_seed(123)
    m3 = UNetModel(C, 2 * C)                              # same seed: same weights
    assert torch.equal(m3.engine.arena.flat, a.flat)


def test_autograd_surface_refuses_stale_forward_and_accumulation():
    """the engine keeps ONE set of activations and overwrites ONE gradient arena; backward of an older forward or
    a second backward without zero_grad must raise instead of returning wrong gradients."""
    from phasegen.model import UNetModel
    from phasegen.optim import Adam
    C, L = 8, 24
    m = UNetModel(C, 2 * C).load_numpy(detgen.make_params(C, seed=0))
    opt = Adam(m.parameters())
    x = torch.from_numpy(detgen.make_batch(2, C, L, seed=4)[:, 0].copy()).cuda()
    out = m.forward(x)
    with torch.no_grad():
        m.forward(x)                                      # e.g. a validation forward in between
    with pytest.raises(RuntimeError, match="no longer the engine's latest"):
        out.sum().backward()
    opt.zero_grad()
    m.forward(x).sum().backward()
    with pytest.raises(RuntimeError, match="zero_grad"):
        m.forward(x).sum().backward()                     # p.grad still holds the previous gradients
    opt.zero_grad()
    m.forward(x).sum().backward()                         # fine again
