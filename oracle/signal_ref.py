"""ORACLE (test infrastructure only) -- CPU restatement of the STFT / polar / ISTFT framing.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

PARITY UNPINNED for stft/istft/griffin_lim: the arithmetic lives in third-party ``librosa``, which is
not vendored under /root/reference, has no pinned version (API usage brackets it to ~0.5-0.7:
``librosa.output.write_wav`` demo.py:6, ``librosa.display.waveplot`` utils.py:138), and is not
installed in this pipeline; the reference holds no golden vectors for it.  This file restates
librosa's *published* definition (``librosa.core.spectrum.stft/istft``) and is cross-checked against
two independent in-container implementations (``torch.stft/istft`` and a direct DFT) by
``tests/test_oracle_signal.py``.  ``get_spec_and_angle`` IS pinned: fixture G4 is produced by the
imported reference ``data.py`` (oracle/gen_golden.py).

Reference call sites restated (file:line in /root/reference):
  * stft + DC drop + (re, im) stacking      preproc_mdb.py:84-97
  * log1p|z| / angle                        data.py:39-47
  * DC re-insert + istft + peak normalise   utils.py:34-42
  * hybrid spectrum (exp(m)-1) e^{j phi}    demo.py:39
  * Griffin-Lim                             utils.py:112-134
"""
import numpy as np


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True) == 0.5 - 0.5 cos(2 pi i / n)."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n))


def n_frames_for(n_samples, hop):
    """librosa center=True framing: 1 + len(y) // hop frames."""
    return 1 + n_samples // hop


def reflect_index(i, n):
    """Source index in y (len n) of padded position i - pad, numpy 'reflect' (edge not repeated)."""
    i = np.asarray(i)
    period = 2 * (n - 1)
    m = np.mod(i, period)
    return np.where(m < n, m, period - m)


def frame_indices(n_samples, n_fft, hop):
    """INTEGER frame index map (bit-exact contract): idx[t, k] = index into y of tap k of frame t.

    y_pad = np.pad(y, n_fft//2, 'reflect'); frame t = y_pad[t*hop : t*hop + n_fft].
    """
    nf = n_frames_for(n_samples, hop)
    pos = (np.arange(nf)[:, None] * hop + np.arange(n_fft)[None, :]) - n_fft // 2
    return reflect_index(pos, n_samples).astype(np.int64)


def stft(y, n_fft, hop):
    """librosa.stft(y, n_fft, hop_length) with defaults: complex64 (1 + n_fft/2, n_frames)."""
    y = np.asarray(y, dtype=np.float32)
    idx = frame_indices(len(y), n_fft, hop)
    win = hann_periodic(n_fft).astype(np.float32)
    frames = y[idx] * win[None, :]
    return np.fft.rfft(frames.astype(np.float64), axis=1).T.astype(np.complex64)


def chunk_and_stft(chunk, n_fft, hop):
    """preproc_mdb.py:90-96 for one mono chunk: drop bin 0, stack [re; im] -> (2, n_fft/2, frames) f32."""
    s = np.delete(stft(chunk, n_fft, hop), 0, axis=0)
    return np.stack([np.real(s), np.imag(s)], axis=0).astype(np.float32)


def get_spec_and_angle(data):
    """data.py:39-47 (use_exp=True): (N, 2, bins, frames) [re, im] -> [log1p|z|, angle z]."""
    z = data[:, 0] + data[:, 1] * 1j
    return np.concatenate([np.log1p(np.abs(z))[:, None], np.angle(z)[:, None]], axis=1)


def window_sumsquare(n_frames, n_fft, hop):
    wss = np.zeros(n_fft + hop * (n_frames - 1), dtype=np.float64)
    w2 = hann_periodic(n_fft) ** 2
    for t in range(n_frames):
        wss[t * hop: t * hop + n_fft] += w2
    return wss


def istft(S, hop):
    """librosa.istft(S, hop_length): Hermitian irfft per frame x hann, overlap-add, / window-sum-square
    where > tiny, trim n_fft//2 both ends.  Returns float32 of length hop * (n_frames - 1)."""
    n_fft = 2 * (S.shape[0] - 1)
    nf = S.shape[1]
    win = hann_periodic(n_fft)
    frames = np.fft.irfft(S.T.astype(np.complex128), n=n_fft, axis=1) * win[None, :]
    y = np.zeros(n_fft + hop * (nf - 1), dtype=np.float64)
    for t in range(nf):
        y[t * hop: t * hop + n_fft] += frames[t]
    wss = window_sumsquare(nf, n_fft, hop)
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2: len(y) - n_fft // 2].astype(np.float32)


def generate_audio(spec, hop, is_stft=False):
    """utils.py:34-42: DC row of zeros prepended, istft, divide by max|y| (librosa.util.normalize inf-norm;
    a zero signal is returned unchanged)."""
    S = spec if is_stft else spec[0] + 1j * spec[1]
    S = np.concatenate([np.zeros((1, S.shape[1]), np.complex64), S.astype(np.complex64)], axis=0)
    y = istft(S, hop)
    if not np.all(np.isfinite(y)):
        raise ValueError("Audio buffer is not finite everywhere")
    peak = np.max(np.abs(y))
    return y / peak if peak > np.finfo(np.float32).tiny else y


def hybrid_spectrum(logmag, phase):
    """demo.py:39: (exp(m) - 1) * exp(j phi)."""
    return (np.exp(logmag) - 1.0) * np.exp(1j * phase)


def griffin_lim(mag, n_fft, hop, n_iter, init):
    """utils.py:112-134 with the random start vector passed in (``init`` replaces np.random.randn).

    Reference quirk kept: ``librosa.istft(new_spec)`` is called on the DC-dropped (n_fft/2, frames)
    matrix WITHOUT re-inserting the DC row (utils.py:114,127), so the inverse transform infers
    n_fft' = 2*(n_fft/2 - 1) = n_fft - 2 (2046 for 2048); the output length is still hop*(frames-1)."""
    recon = np.asarray(init, dtype=np.float64)
    new_spec, loss = None, None
    for _ in range(n_iter):
        rs = np.delete(stft(recon, n_fft, hop), 0, axis=0)
        new_spec = mag * np.exp(1j * np.angle(rs))
        prev = recon
        recon = istft(new_spec, hop).astype(np.float64)
        loss = np.sqrt(np.sum((recon - prev) ** 2 / recon.size))
    peak = np.max(np.abs(recon))
    return (recon / peak if peak > 0 else recon).astype(np.float32), new_spec, loss
