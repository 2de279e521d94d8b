"""CPU: the librosa-convention STFT/ISTFT oracle cross-checked against two independent implementations present
in this image (torch.stft/istft and a direct DFT).  PARITY UNPINNED w.r.t. librosa itself (absent, unpinned,
no reference fixtures) -- see oracle/signal_ref.py."""
import numpy as np
import pytest
import torch

from oracle import signal_ref
from phasegen import detgen


@pytest.mark.parametrize("n_fft,hop,n", [(2048, 512, 65024), (1024, 256, 65280), (32, 8, 184), (64, 16, 1000)])
def test_frame_count_and_index_map(n_fft, hop, n):
    idx = signal_ref.frame_indices(n, n_fft, hop)
    assert idx.shape == (1 + n // hop, n_fft) and idx.dtype == np.int64
    ypad = np.pad(np.arange(n), n_fft // 2, mode="reflect")      # numpy's own reflect padding of a ramp
    for t in (0, 1, idx.shape[0] // 2, idx.shape[0] - 1):
        assert np.array_equal(idx[t], ypad[t * hop: t * hop + n_fft])


def test_reference_default_shapes():
    # preproc_mdb.py:202-209: 4.064 s @ 16 kHz, 2048/512 -> 1024 bins x 128 frames after the DC drop
    assert signal_ref.n_frames_for(65024, 512) == 128
    assert signal_ref.n_frames_for(65280, 256) == 256
    assert signal_ref.n_frames_for(64000, 256) == 251           # exact 4 s is NOT a valid U-Net length


@pytest.mark.parametrize("n_fft,hop,n", [(64, 16, 1008), (1024, 256, 8192)])
def test_stft_vs_torch_and_dft(n_fft, hop, n):
    y = detgen.make_clip(n, seed=3)
    S = signal_ref.stft(y, n_fft, hop)
    T = torch.stft(torch.from_numpy(y), n_fft, hop, window=torch.hann_window(n_fft, periodic=True), center=True,
                   pad_mode="reflect", return_complex=True).numpy()
    assert S.shape == T.shape == (n_fft // 2 + 1, 1 + n // hop)
    assert np.max(np.abs(S - T)) < 2e-5 * np.max(np.abs(T))
    # direct DFT of frame 3
    idx = signal_ref.frame_indices(n, n_fft, hop)[3]
    fr = y[idx].astype(np.float64) * signal_ref.hann_periodic(n_fft)
    k = np.arange(n_fft // 2 + 1)[:, None] * np.arange(n_fft)[None, :]
    D = (fr[None, :] * np.exp(-2j * np.pi * k / n_fft)).sum(1)
    assert np.max(np.abs(S[:, 3] - D)) < 2e-5 * np.max(np.abs(D))


@pytest.mark.parametrize("n_fft,hop,n", [(64, 16, 1008), (1024, 256, 8192)])
def test_istft_vs_torch_and_roundtrip(n_fft, hop, n):
    y = detgen.make_clip(n, seed=4)
    S = signal_ref.stft(y, n_fft, hop)
    r = signal_ref.istft(S, hop)
    assert r.shape == (hop * (S.shape[1] - 1),)
    t = torch.istft(torch.from_numpy(S), n_fft, hop, window=torch.hann_window(n_fft, periodic=True), center=True).numpy()
    m = min(len(r), len(t))
    # interior (torch.istft refuses / differs only where the window envelope vanishes at the very edges)
    assert np.max(np.abs(r[n_fft:m - n_fft] - t[n_fft:m - n_fft])) < 1e-5
    assert np.max(np.abs(r[n_fft:m - n_fft] - y[n_fft:m - n_fft])) < 1e-4     # COLA round trip


def test_generate_audio_and_chunk_layout(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "demo_g5.npz"))
    spec = signal_ref.chunk_and_stft(g["clip"], 32, 8)
    assert spec.shape == (2, 16, 24) and spec.dtype == np.float32          # DC dropped, [re; im] stacked
    assert np.array_equal(spec, g["spec"])
    a = signal_ref.generate_audio(g["spec"], 8)                             # is_stft=False path: spec[0] + j spec[1]
    assert a.shape == (8 * 23,) and abs(np.max(np.abs(a)) - 1.0) < 1e-6     # peak-normalised
    z = signal_ref.generate_audio(np.zeros((2, 16, 24), np.float32), 8)     # silent clip is returned unchanged
    assert np.all(z == 0)


def test_griffin_lim_shapes_and_normalisation():
    """utils.py:112-134 incl. its quirk: istft on the DC-dropped matrix infers n_fft - 2, so consistency is not
    guaranteed to improve monotonically; pin what the reference does guarantee (shape, finiteness, peak = 1)."""
    n_fft, hop = 64, 16
    y = detgen.make_clip(16 * 31, seed=5)
    mag = np.abs(np.delete(signal_ref.stft(y, n_fft, hop), 0, axis=0))
    init = detgen.normal(8, (16 * 31,)).astype(np.float64)
    a, spec, loss = signal_ref.griffin_lim(mag, n_fft, hop, 5, init)
    assert a.shape == (16 * 31,) and a.dtype == np.float32 and np.all(np.isfinite(a))
    assert abs(np.max(np.abs(a)) - 1.0) < 1e-6 and spec.shape == mag.shape and np.isfinite(loss)
    assert np.allclose(np.abs(spec), mag, rtol=1e-5, atol=1e-7)          # magnitudes are imposed every iteration
