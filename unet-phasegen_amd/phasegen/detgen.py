"""Build-owned deterministic data generator (counter-based, pure integer arithmetic).

Golden fixtures, parity tests, ``bench.py`` and ``smoke()`` all need the *same* tensors
in this container (where the reference can be imported) and on the GPU box (where it
cannot), so nothing here depends on torch's or numpy's RNG streams: value ``i`` of
stream ``seed`` is a SplitMix64-style hash of ``(seed, i)`` mapped to a float.  The
hash is plain uint64 wrap-around arithmetic, hence bit-identical everywhere.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _hash(seed, n, offset=0):
    with np.errstate(over="ignore"):
        i = np.arange(offset, offset + n, dtype=np.uint64)
        z = (i + np.uint64(seed) * _GOLD + _GOLD)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed, shape, lo=-1.0, hi=1.0, chunk=1 << 24):
    """float32 array of ``shape`` with values in [lo, hi); 24 random mantissa bits."""
    n = int(np.prod(shape)) if len(shape) else 1
    out = np.empty(n, dtype=np.float32)
    for off in range(0, n, chunk):
        m = min(chunk, n - off)
        u = (_hash(seed, m, off) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
        out[off:off + m] = np.float32(lo) + u * np.float32(hi - lo)
    return out.reshape(shape)


def normal(seed, shape, std=1.0):
    """float32 approx-normal values (sum of 4 uniforms, variance-matched); exact same everywhere."""
    acc = np.zeros(shape, dtype=np.float32)
    for k in range(4):
        acc += uniform(seed * 4 + k + 1000003, shape, -1.0, 1.0)
    return (acc * np.float32(std * (3.0 / 4.0) ** 0.5)).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# U-Net parameter set with the reference's state-dict names and shapes (model.py:27-34, 77-105).
# ---------------------------------------------------------------------------------------------
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

CONV_KEYS = (K_D0, K_D1, K_D2, K_D3, K_U3, K_U2, K_U1, K_U0)
BN_KEYS = (BN_D1, BN_D2, BN_U3, BN_U2, BN_U1, BN_U0)


def conv_shapes(C):
    """Weight shapes. Conv1d: (Cout, Cin, k). ConvTranspose1d: (Cin, Cout, k)."""
    return {
        K_D0: (2 * C, C, 32), K_D1: (2 * C, 2 * C, 8), K_D2: (2 * C, 2 * C, 8), K_D3: (4 * C, 2 * C, 4),
        K_U3: (4 * C, 2 * C, 5), K_U2: (4 * C, 2 * C, 8), K_U1: (4 * C, 2 * C, 8), K_U0: (4 * C, 2 * C, 32),
    }


def param_order():
    """Order of ``UNetModel.parameters()`` in the reference (registration order, buffers excluded)."""
    return [K_D0, K_D1, BN_D1 + ".weight", BN_D1 + ".bias", K_D2, BN_D2 + ".weight", BN_D2 + ".bias",
            K_D3, K_U3, BN_U3 + ".weight", BN_U3 + ".bias", K_U2, BN_U2 + ".weight", BN_U2 + ".bias",
            K_U1, BN_U1 + ".weight", BN_U1 + ".bias", K_U0, BN_U0 + ".weight", BN_U0 + ".bias"]


def state_dict_order():
    """Order of the reference's ``model.state_dict()`` keys on modern torch (38 entries)."""
    out = []
    for k in param_order():
        out.append(k)
        if k.endswith(".bias"):
            base = k[:-5]
            out += [base + ".running_mean", base + ".running_var", base + ".num_batches_tracked"]
    return out


def fan_in(key, shape):
    # torch default init: kaiming_uniform(a=sqrt(5)) => U(+-1/sqrt(fan_in)); fan_in = shape[1]*k for
    # both Conv1d and ConvTranspose1d (for the transposed conv that is Cout*k) -- SURVEY.md §8a M0.
    return shape[1] * shape[2]


def make_params(C, seed=0, affine_jitter=True):
    """Deterministic parameter dict (numpy float32) keyed like the reference state-dict.

    Conv weights ~ U(+-1/sqrt(fan_in)) like torch's default init.  BN gamma/beta are jittered away
    from (1, 0) when ``affine_jitter`` so parity tests exercise the affine path.
    """
    p = {}
    for n, (k, shp) in enumerate(conv_shapes(C).items()):
        b = 1.0 / np.sqrt(fan_in(k, shp))
        p[k] = uniform(seed * 101 + n + 1, shp, -b, b)
    for n, k in enumerate(BN_KEYS):
        if affine_jitter:
            p[k + ".weight"] = uniform(seed * 101 + 50 + n, (2 * C,), 0.5, 1.5)
            p[k + ".bias"] = uniform(seed * 101 + 70 + n, (2 * C,), -0.5, 0.5)
        else:
            p[k + ".weight"] = np.ones((2 * C,), np.float32)
            p[k + ".bias"] = np.zeros((2 * C,), np.float32)
        p[k + ".running_mean"] = np.zeros((2 * C,), np.float32)
        p[k + ".running_var"] = np.ones((2 * C,), np.float32)
        p[k + ".num_batches_tracked"] = np.zeros((), np.int64)
    return p


def make_batch(B, C, L, seed=1):
    """Synthetic training batch shaped like ``data.get_fft_npy_loader`` output: (B, 2, C, L) float32.

    ch0 = log1p(|N(0,1) + j N(0,1)|) (log-magnitude), ch1 = U(-pi, pi) (angle).  SURVEY.md §8d.
    """
    re = normal(seed * 7 + 1, (B, C, L))
    im = normal(seed * 7 + 2, (B, C, L))
    logmag = np.log1p(np.sqrt(re * re + im * im)).astype(np.float32)
    ang = uniform(seed * 7 + 3, (B, C, L), -np.pi, np.pi)
    return np.stack([logmag, ang], axis=1).astype(np.float32)


def make_clip(n_samples, seed=2):
    """Synthetic mono clip: N(0, 0.1^2) noise plus two sinusoids."""
    t = np.arange(n_samples, dtype=np.float64)
    y = normal(seed, (n_samples,), 0.1).astype(np.float64)
    y += 0.3 * np.sin(2 * np.pi * 440.0 / 16000.0 * t) + 0.2 * np.sin(2 * np.pi * 1730.0 / 16000.0 * t + 0.5)
    return y.astype(np.float32)
