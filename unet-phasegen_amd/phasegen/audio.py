"""STFT / ISTFT framing of the reference (preproc_mdb.py:84-97, utils.py:11-44, demo.py:36-40) on device."""
import numpy as np
import torch

from . import ops
from ._lib import SCHED_TILE_PER_WG as _lib_sched_tile_per_wg


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("phasegen.audio needs an MI355X (STFT/ISTFT run on the device; no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def chunk_and_stft(audio, start, t_slice, n_fft, hop_length, polar=False):
    """preproc_mdb.py:84-97: slice [start, start+t_slice) of every channel (zero-padded tail), librosa-convention STFT,
    DC bin dropped, [re; im] stacked -> (n_channels, 2, n_fft/2, 1 + t_slice // hop) float32 device tensor.
    ``polar=True`` fuses data.py:39-47 ([log1p|z|; angle]) into the same kernel."""
    a = torch.as_tensor(np.asarray(audio, dtype=np.float32))
    if a.dim() == 1:
        a = a[None]
    chunk = a[:, start:start + t_slice]
    if chunk.shape[1] < t_slice:
        chunk = torch.nn.functional.pad(chunk, (0, t_slice - chunk.shape[1]))
    return ops.stft(chunk.contiguous().to(_dev()), n_fft, hop_length, polar=polar)


def generate_audio(spec, sr, hop_length, is_stft=False):
    """utils.py:11-44.  ``spec``: complex (bins, frames) if ``is_stft`` else real (2, bins, frames) = [re; im].
    Zero DC row prepended, inverse STFT, finite check (librosa.util.valid_audio), peak-normalised.  -> float32 numpy."""
    dev = _dev()
    if is_stft:
        z = np.asarray(spec)
        re, im = np.real(z), np.imag(z)
    else:
        re, im = np.asarray(spec[0]), np.asarray(spec[1])
    re = torch.from_numpy(np.ascontiguousarray(re, dtype=np.float32)).to(dev)[None]
    im = torch.from_numpy(np.ascontiguousarray(im, dtype=np.float32)).to(dev)[None]
    y = ops.istft(re, im, hop_length, mode=1, normalize=True)[0].cpu().numpy()
    if not np.all(np.isfinite(y)):
        raise ValueError("Audio buffer is not finite everywhere")       # librosa.util.valid_audio's ParameterError
    return y


def synthesize(logmag, phase, hop_length, normalize=True):
    """demo.py:39-40 fused on device: istft((exp(logmag) - 1) * exp(1j * phase)) for a batch of clips.
    logmag, phase: (n, bins, frames) device tensors (phase may be the first half of the network output) -> (n, samples)."""
    return ops.istft(logmag, phase, hop_length, mode=0, normalize=normalize)


_synth = {}


def _synthesis_matrix(bins, device):
    """Inverse real DFT of the reference's Griffin-Lim as a (n, 2*bins-2) matrix: n_fft' = 2*(bins-1) (utils.py:114,127
    invert the DC-DROPPED matrix, so 1024 rows mean a 2046-point transform), periodic Hann and 1/n_fft' folded in.
    Columns: Re of bins 0..bins-1, then Im of bins 1..bins-2."""
    key = (bins, str(device))
    if key not in _synth:
        N = 2 * (bins - 1)
        n = np.arange(N, dtype=np.float64)[:, None]
        k = np.arange(bins, dtype=np.float64)[None, :]
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N)
        cw = np.where((k == 0) | (k == bins - 1), 1.0, 2.0)
        Wre = cw * np.cos(2.0 * np.pi * k * n / N)
        Wim = -2.0 * np.sin(2.0 * np.pi * k[:, 1:bins - 1] * n / N)
        W = np.concatenate([Wre, Wim], axis=1) * win / N
        _synth[key] = torch.from_numpy(W.astype(np.float32)).to(device)[:, :, None].contiguous()
    return _synth[key]


def griffin_lim_batch(mag, n_fft, hop_length, n_iter, init=None, seed=None):
    """utils.py:85-134 for a BATCH of clips, device tensors in and out: ``mag`` (n, n_fft/2, frames) magnitudes (DC dropped).
    Every iteration is four launches whatever n is -- STFT (pg_stft) -> keep phase, impose magnitude (pg_gl_project) -> the
    2046-point inverse transform as ONE 1x1 convolution on the fp32-MFMA conv kernel (batch axis = clips) -> overlap-add
    (pg_ola_nt) -- and clip c's result is bit-identical to running it alone (no kernel reduces across clips).
    ``init``: (n, hop * (frames - 1)) start signals; else clip c starts from N(0, 1) drawn with seed ``seed + c`` (or ``seed[c]``
    when a list is given; replaces ``np.random.randn``, utils.py:116).  Returns device tensors (audio (n, length) peak-normalised per clip, new_spec
    (n, 2, bins, frames) = [re; im] of the last projection, loss (n,))."""
    dev = _dev()
    if not (torch.is_tensor(mag) and mag.is_cuda and mag.dtype == torch.float32 and mag.dim() == 3):
        raise ValueError("griffin_lim_batch: mag must be a float32 device tensor (n, bins, frames)")
    mag = mag.contiguous()
    n, bins, frames = mag.shape
    if bins != n_fft // 2:
        raise ValueError(f"griffin_lim: spec has {bins} rows, expected n_fft/2 = {n_fft // 2}")
    length = hop_length * (frames - 1)
    if init is None:
        if isinstance(seed, (list, tuple)):                     # one seed per clip
            seeds = list(seed)
            if len(seeds) != n:
                raise ValueError("griffin_lim_batch: one seed per clip expected")
        else:
            base = torch.initial_seed() if seed is None else seed
            seeds = [base + c for c in range(n)]
        rows = []
        for sd in seeds:
            g = torch.Generator(device="cpu")
            g.manual_seed(sd)
            rows.append(torch.randn(length, generator=g, dtype=torch.float64))
        init = torch.stack(rows).to(torch.float32)
    audio = torch.empty(n, length, device=dev)
    new_spec = torch.empty(n, 2, bins, frames, device=dev)
    loss = torch.full((n,), float("nan"), device=dev)
    W = _synthesis_matrix(bins, dev)
    N = 2 * (bins - 1)
    for c0 in range(0, n, 64):                                   # pg_ola_nt keeps one peak word per clip in 256 bytes
        c1 = min(n, c0 + 64)
        m = c1 - c0
        recon = torch.as_tensor(init[c0:c1]).to(dev, torch.float32).contiguous()
        x = torch.zeros(m, N, frames, device=dev)
        fr = torch.empty(m, N, frames, device=dev)
        S = torch.empty(m, 2, bins, frames, device=dev)
        prev = torch.empty_like(recon)
        for _ in range(n_iter):
            ops.stft(recon, n_fft, hop_length, out=S)
            ops.gl_project(S, mag[c0:c1], x, new_spec[c0:c1])
            # the inverse DFT is always exact-fp32 MFMA, one whole tile per workgroup: every output sums its 2046 terms in the
            # same order whatever the number of clips (a stream-K split would depend on the tile count, i.e. on n)
            ops.conv_fwd(x, W, fr, 1, 0, precision="fp32", schedule=_lib_sched_tile_per_wg)
            prev.copy_(recon)
            ops.ola_nt(fr, hop_length, recon)
        if n_iter > 0:
            for c in range(m):          # one 1-D reduction per clip: the summation order does not depend on how many clips run together
                loss[c0 + c] = torch.sqrt(torch.sum((recon[c] - prev[c]) ** 2) / length)
        peak = recon.abs().amax(dim=1, keepdim=True)
        audio[c0:c1] = torch.where(peak > torch.finfo(torch.float32).tiny, recon / peak, recon)
    return audio, new_spec, loss


def griffin_lim(spec, n_fft, hop_length, n_iter, init=None, seed=None):
    """utils.py:85-134 with the reference's signature: ``spec`` magnitudes (n_fft/2, frames) (numpy, DC already dropped).
    One clip of ``griffin_lim_batch``.  Returns (audio float32 numpy peak-normalised, new_spec complex numpy, loss float)."""
    dev = _dev()
    mag = torch.from_numpy(np.ascontiguousarray(np.abs(spec) if np.iscomplexobj(spec) else spec, dtype=np.float32)).to(dev)
    if init is not None:
        init = torch.from_numpy(np.ascontiguousarray(init, dtype=np.float32))[None]
    a, ns, loss = griffin_lim_batch(mag[None], n_fft, hop_length, n_iter, init=init, seed=seed)
    audio = a[0].cpu().numpy()
    if not np.all(np.isfinite(audio)):
        raise ValueError("Audio buffer is not finite everywhere")
    ns = ns[0].cpu().numpy()
    return audio, (ns[0] + 1j * ns[1]).astype(np.complex64), float(loss[0])
