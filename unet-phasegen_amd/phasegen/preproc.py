"""Feature extraction of preproc_mdb.py on device: chunking + STFT + global normalisation +
shuffled split, producing the on-disk format data.py consumes: (N, 2, n_fft/2, frames) float32 ``{genre}_audio_{train,
val}.npy``.  MedleyDB walking, stem mixing and resampling (preproc_mdb.py:15-64,105-116) stay out of scope: the input here
is already-loaded mono audio at the target rate.

  chunk starts   preproc_mdb.py:66-82  one aligned chunk every t_slice samples plus n_random uniformly random crops in
                 [0, a_len - t_slice // 1.3) after each -- the random starts come from a numpy Generator you pass, so a
                 run is reproducible (the reference uses the global np.random state)
  chunk + STFT   preproc_mdb.py:84-97  zero-padded tail, librosa-convention STFT, DC dropped, [re; im]: ONE pg_stft launch
                 for all chunks of a track, reading the chunks in place (pg_stft_args.chunk_start / chunk_row: no gathered
                 or zero-padded copy of the audio) and writing straight into the dataset array
  normalise      preproc_mdb.py:182    (x - mean) / std over the WHOLE array (re and im together, population std):
                 pg_moments (double accumulators, two-pass) + pg_standardize in place
  split          preproc_mdb.py:174-184 shuffled indices, first n_val clips -> val, rest -> train
"""
import os

import numpy as np
import torch

from . import ops


def chunk_starts(a_len, t_slice, n_random, rng):
    """Start offsets in the reference's order: aligned start, then its n_random random crops (preproc_mdb.py:73-80)."""
    bnd = a_len - t_slice // 1.3
    starts = []
    for i in range(0, a_len, t_slice):
        starts.append(i)
        for _ in range(n_random):
            starts.append(int(rng.integers(0, bnd)))
    return starts


def n_chunks(a_len, t_slice, n_random):
    """Chunks chunk_starts yields for a track of a_len samples (preproc_mdb.py:73-80)."""
    return len(range(0, a_len, t_slice)) * (1 + n_random)


def chunk_audio(audio, t_slice, n_fft, hop_length, n_random, rng, device=None, out=None):
    """audio: mono float array (or (channels, samples)).  -> (n_chunks, channels, 2, n_fft/2, frames) device tensor (``out``,
    when given, is that tensor: a slice of the dataset array)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    a = np.asarray(audio, dtype=np.float32)
    if a.ndim == 1:
        a = a[None]
    n_ch, a_len = a.shape
    starts = chunk_starts(a_len, t_slice, n_random, rng)
    ad = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # signal order (chunk, channel): the layout of the dataset array, so the STFT writes it directly
    st = torch.tensor(np.repeat(np.asarray(starts, np.int64), n_ch), device=dev)
    rows = torch.tensor(np.tile(np.arange(n_ch, dtype=np.int32), len(starts)), device=dev)
    bins, frames = n_fft // 2, 1 + t_slice // hop_length
    if out is None:
        out = torch.empty(len(starts), n_ch, 2, bins, frames, device=dev)
    ops.stft(ad, n_fft, hop_length, out=out.view(len(starts) * n_ch, 2, bins, frames), chunk_start=st, chunk_row=rows,
             chunk_len=t_slice)
    return out


def build_dataset(tracks, chunk_seconds=4.064, rsr=16000, n_fft=2048, hop_length=512, n_random=0, n_val=40, seed=0,
                  out_dir=None, genre="Pop", device=None):
    """tracks: list of mono float arrays at ``rsr``.  Returns (train, val) float32 numpy arrays; also writes
    ``{out_dir}/{genre}_audio_{train,val}.npy`` when ``out_dir`` is given (preproc_mdb.py:195-196)."""
    rng = np.random.default_rng(seed)
    t_slice = int(chunk_seconds * rsr)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    shaped = [np.asarray(t, np.float32).reshape(-1, np.shape(t)[-1]).shape for t in tracks]     # (channels, samples) per track
    n_ch = shaped[0][0]
    counts = [n_chunks(a_len, t_slice, n_random) for _, a_len in shaped]
    x = torch.empty(sum(counts), n_ch, 2, n_fft // 2, 1 + t_slice // hop_length, device=dev)    # (N, channels, 2, bins, frames)
    o = 0
    for t, c in zip(tracks, counts):
        chunk_audio(t, t_slice, n_fft, hop_length, n_random, rng, dev, out=x[o:o + c])
        o += c
    if x.shape[1] == 1:
        x = x[:, 0]                                                      # np.squeeze(axis=1), preproc_mdb.py:179-180
    ops.standardize_(x)                                                  # numpy .std() is the population std
    x = x.cpu().numpy()
    idx = np.linspace(0, len(x) - 1, len(x), dtype=int)
    rng.shuffle(idx)
    val, train = x[idx][:n_val], x[idx][n_val:]
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        np.save(os.path.join(out_dir, f"{genre}_audio_val.npy"), val)
        np.save(os.path.join(out_dir, f"{genre}_audio_train.npy"), train)
    return train, val
