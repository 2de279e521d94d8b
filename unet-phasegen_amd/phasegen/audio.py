"""STFT / ISTFT framing of the reference (preproc_mdb.py:84-97, utils.py:11-44, demo.py:36-40) on device."""
import numpy as np
import torch

from . import ops


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
