"""ORACLE (test infrastructure only) -- CPU restatement of the reference U-Net hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  The product path (``unet-phasegen_amd/``) never does: it runs the HIP kernels or fails.

Pinning: checked against the *imported reference* in the build container by
``oracle/gen_golden.py`` (which also writes ``tests/golden/*.npz``), and against those committed
fixtures everywhere by ``tests/test_oracle_golden.py``.

Everything is stock fp32 ``torch.nn.functional`` on CPU, restating (file:line in /root/reference):

  * network wiring            model.py:27-34 (UNetModel.__init__), model.py:85-105 (UNetBlock recipes)
  * in-place LeakyReLU skip   model.py:80,109-113  -> the skip branch of cat([x, f]) carries LeakyReLU(x)
  * BatchNorm (train mode)    model.py:81,83; per-channel over (B, L), eps 1e-5, momentum 0.1,
                              biased var to normalise, unbiased var into running_var
  * loss                      train.py:45-60
  * Adam                      train.py:26-27,62 (torch.optim.Adam defaults)
"""
import math

import torch
import torch.nn.functional as F

# ENVIRONMENT BUG GUARD (found while pinning this oracle, see DESIGN.md "oneDNN"): torch 2.10.0's oneDNN
# (v3.7.1) ConvTranspose1d kernel returns WRONG values (0.56 relative error against a float64 einsum) when
# run multi-threaded at the innermost up-conv's full-size shape (4096 -> 2048 channels, k5, s2, L=14);
# single-threaded or with oneDNN disabled the same call is correct to 3e-7.  The oracle (and the imported
# reference, when oracle/gen_golden.py runs it) therefore always uses ATen's native conv path.
torch.backends.mkldnn.enabled = False

K_D0 = "model.0.weight"
K_D1 = "model.1.model.1.weight"
K_D2 = "model.1.model.3.model.1.weight"
K_D3 = "model.1.model.3.model.3.model.1.weight"
K_U3 = "model.1.model.3.model.3.model.3.weight"
K_U2 = "model.1.model.3.model.5.weight"
K_U1 = "model.1.model.5.weight"
K_U0 = "model.3.weight"
BN_D1 = "model.1.model.2"
BN_D2 = "model.1.model.3.model.2"
BN_U3 = "model.1.model.3.model.3.model.4"
BN_U2 = "model.1.model.3.model.6"
BN_U1 = "model.1.model.6"
BN_U0 = "model.4"

EPS = 1e-5
MOMENTUM = 0.1
SLOPE = 0.2


def batch_norm_train(x, gamma, beta, stats=None, prefix=None):
    """Train-mode batch norm over (B, L) per channel (model.py:81,83 with a 3-D input).

    ``stats`` (optional dict) receives updated running_mean / running_var / num_batches_tracked
    under ``prefix`` exactly as nn.BatchNorm1d would (unbiased variance, momentum 0.1).
    """
    n = x.shape[0] * x.shape[2]
    mean = x.mean(dim=(0, 2))
    var = ((x - mean[None, :, None]) ** 2).mean(dim=(0, 2))
    y = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + EPS)
    y = y * gamma[None, :, None] + beta[None, :, None]
    if stats is not None:
        with torch.no_grad():
            unbiased = var * (n / max(n - 1, 1))
            stats[prefix + ".running_mean"] = (1 - MOMENTUM) * stats[prefix + ".running_mean"] + MOMENTUM * mean
            stats[prefix + ".running_var"] = (1 - MOMENTUM) * stats[prefix + ".running_var"] + MOMENTUM * unbiased
            stats[prefix + ".num_batches_tracked"] = stats[prefix + ".num_batches_tracked"] + 1
    return y


def unet_forward(p, x, stats=None, capture=None, masks=None):
    """(B, C, L) -> (B, 2C, L).  ``p``: dict of torch tensors keyed like the reference state-dict.

    ``capture`` (optional dict) receives every intermediate named as in DESIGN.md:
      a0, c1, h1, c2, h2, d3, r3, u3, r2, u2, r1, u1, r0, out

    ``masks`` (optional dict of bool tensors keyed a0, h1, h2, d3, u3, u2, u1; test aid for full-width GRADIENT parity):
    every (Leaky)ReLU is evaluated with the given sign pattern instead of the tensor's own -- LeakyReLU(t) = t * where(m, 1,
    0.2), ReLU(t) = t * m.  With the masks of the run under test the function is the same network wherever the two runs agree
    on signs (everywhere except pre-activations within rounding of zero, where t itself is ~0), but its gradient no longer
    jumps when a pre-activation that is zero to rounding lands on the other side: a flipped mask otherwise changes whole rows
    of the weight gradients by O(1/sqrt(B L')) and makes end-to-end gradient comparison at 1e-4 meaningless.
    """
    cap = capture if capture is not None else {}

    def bn(t, key):
        return batch_norm_train(t, p[key + ".weight"], p[key + ".bias"], stats, key)

    if masks is not None:
        class F:                                                  # noqa: N801  (shadows torch.nn.functional below on purpose)
            conv1d = staticmethod(torch.nn.functional.conv1d)
            conv_transpose1d = staticmethod(torch.nn.functional.conv_transpose1d)
            _m = None

            @staticmethod
            def leaky_relu(t, slope):
                m = F._m
                return t * torch.where(m, torch.ones((), dtype=t.dtype), torch.full((), slope, dtype=t.dtype))

            @staticmethod
            def relu(t):
                return t * F._m.to(t.dtype)

        def site(*names):                                         # the sign pattern for the next activation call
            F._m = torch.cat([masks[n] for n in names], 1) if len(names) > 1 else masks[names[0]]
    else:
        F = torch.nn.functional

        def site(*names):
            pass

    # outermost down: Conv1d(C -> 2C, k32, s2, p16), no activation in front     model.py:33,90
    a0 = F.conv1d(x, p[K_D0], stride=2, padding=16)
    # block b3: LeakyReLU (in place => skip carries it), Conv1d k8 s1 p2, BN        model.py:31,103
    site("a0")
    a0l = F.leaky_relu(a0, SLOPE)
    c1 = F.conv1d(a0l, p[K_D1], stride=1, padding=2)
    h1 = bn(c1, BN_D1)
    # block b2: LeakyReLU, Conv1d k8 s2 p1, BN                                       model.py:29,103
    site("h1")
    h1l = F.leaky_relu(h1, SLOPE)
    c2 = F.conv1d(h1l, p[K_D2], stride=2, padding=1)
    h2 = bn(c2, BN_D2)
    # block b1 (innermost): LeakyReLU, Conv1d k4 s2 p1, ReLU, ConvT k5 s2 p1, BN     model.py:27,96-97
    site("h2")
    h2l = F.leaky_relu(h2, SLOPE)
    d3 = F.conv1d(h2l, p[K_D3], stride=2, padding=1)
    site("d3")
    r3 = F.conv_transpose1d(F.relu(d3), p[K_U3], stride=2, padding=1)
    u3 = bn(r3, BN_U3)
    cat2 = torch.cat([h2l, u3], 1)                                                   # model.py:113
    # b2 up: ReLU, ConvT(4C -> 2C, k8, s2, p1), BN                                   model.py:101-104
    site("h2", "u3")
    r2 = F.conv_transpose1d(F.relu(cat2), p[K_U2], stride=2, padding=1)
    u2 = bn(r2, BN_U2)
    cat1 = torch.cat([h1l, u2], 1)
    # b3 up: ReLU, ConvT(4C -> 2C, k8, s1, p2), BN
    site("h1", "u2")
    r1 = F.conv_transpose1d(F.relu(cat1), p[K_U1], stride=1, padding=2)
    u1 = bn(r1, BN_U1)
    cat0 = torch.cat([a0l, u1], 1)
    # outermost up: ReLU, ConvT(4C -> 2C, k32, s2, p16), BN; no output non-linearity  model.py:88-92
    site("a0", "u1")
    r0 = F.conv_transpose1d(F.relu(cat0), p[K_U0], stride=2, padding=16)
    out = bn(r0, BN_U0)
    cap.update(a0=a0, c1=c1, h1=h1, c2=c2, h2=h2, d3=d3, r3=r3, u3=u3, r2=r2, u2=u2, r1=r1, u1=u1,
               r0=r0, out=out)
    return out


def phase_loss(pred, batch):
    """train.py:45-60.  ``batch`` = (B, 2, C, L) = [logmag, angle].  Returns (loss, ang, mag)."""
    C = batch.shape[2]
    pred_p, pred_m = pred[:, :C], pred[:, C:]
    theta = batch[:, 1]
    cos_loss = F.mse_loss(torch.cos(pred_p), torch.cos(theta))
    sin_loss = F.mse_loss(torch.sin(pred_p), torch.sin(theta))
    ang = cos_loss + sin_loss
    mag = F.mse_loss(pred_m, batch[:, 0])
    return ang + mag * 0.2, ang, mag


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam update (defaults; no weight decay / amsgrad) on plain tensors, in place.

    ``step`` is the 1-based step number.  Mirrors torch/optim/adam.py's single-tensor path.
    """
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def train_step(p, batch, opt_state, stats=None, lr=1e-3, masks=None):
    """One full step of train.py:41-62 on CPU.  ``p`` holds leaf tensors (requires_grad for the 20
    parameters).  ``opt_state`` = {"step": int, "m": {k: t}, "v": {k: t}}.  Returns (loss, ang, mag, grads).
    ``masks``: see unet_forward (full-width gradient parity tests only)."""
    names = [k for k in p if p[k].dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))]
    for k in names:
        p[k].requires_grad_(True)
        p[k].grad = None
    out = unet_forward(p, batch[:, 0], stats, masks=masks)
    loss, ang, mag = phase_loss(out, batch)
    loss.backward()
    grads = {k: p[k].grad.detach().clone() for k in names}
    opt_state["step"] += 1
    with torch.no_grad():
        for k in names:
            adam_step(p[k], grads[k], opt_state["m"][k], opt_state["v"][k], opt_state["step"], lr=lr)
    return loss.detach(), ang.detach(), mag.detach(), grads


def new_opt_state(p):
    names = [k for k in p if p[k].dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))]
    return {"step": 0, "m": {k: torch.zeros_like(p[k]) for k in names}, "v": {k: torch.zeros_like(p[k]) for k in names}}


def to_torch(params_np):
    return {k: torch.from_numpy(v.copy()) if hasattr(v, "shape") and getattr(v, "ndim", 0) > 0
            else torch.tensor(int(v)) for k, v in params_np.items()}
