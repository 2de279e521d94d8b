"""Feature extraction of preproc_mdb.py on device: chunking + STFT + global normalisation +
shuffled split, producing the on-disk format data.py consumes: (N, 2, n_fft/2, frames) float32 ``{genre}_audio_{train,
val}.npy``.  MedleyDB walking, stem mixing and resampling (preproc_mdb.py:15-64,105-116) stay out of scope: the input here
is already-loaded mono audio at the target rate.

  chunk starts   preproc_mdb.py:66-82  one aligned chunk every t_slice samples plus n_random uniformly random crops in
                 [0, a_len - t_slice // 1.3) after each -- the random starts come from a numpy Generator you pass, so a
                 run is reproducible (the reference uses the global np.random state)
  chunk + STFT   preproc_mdb.py:84-97  zero-padded tail, librosa-convention STFT, DC dropped, [re; im]: ONE pg_stft launch
                 for all chunks of a track (chunks are gathered on the device)
  normalise      preproc_mdb.py:182    (x - mean) / std over the WHOLE array (re and im together, population std)
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


def chunk_audio(audio, t_slice, n_fft, hop_length, n_random, rng, device=None):
    """audio: mono float array (or (channels, samples)).  -> (n_chunks, channels, 2, n_fft/2, frames) device tensor."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    a = np.asarray(audio, dtype=np.float32)
    if a.ndim == 1:
        a = a[None]
    a_len = a.shape[1]
    starts = chunk_starts(a_len, t_slice, n_random, rng)
    ad = torch.zeros(a.shape[0], a_len + t_slice, device=dev)            # zero tail = the reference's np.pad "constant"
    ad[:, :a_len] = torch.from_numpy(a).to(dev)
    idx = torch.tensor(starts, device=dev)[:, None] + torch.arange(t_slice, device=dev)[None, :]
    chunks = ad[:, idx]                                                  # (channels, n_chunks, t_slice)
    n_ch, n_chunks = chunks.shape[0], chunks.shape[1]
    S = ops.stft(chunks.reshape(n_ch * n_chunks, t_slice).contiguous(), n_fft, hop_length)
    return S.reshape(n_ch, n_chunks, *S.shape[1:]).transpose(0, 1).contiguous()


def build_dataset(tracks, chunk_seconds=4.064, rsr=16000, n_fft=2048, hop_length=512, n_random=0, n_val=40, seed=0,
                  out_dir=None, genre="Pop", device=None):
    """tracks: list of mono float arrays at ``rsr``.  Returns (train, val) float32 numpy arrays; also writes
    ``{out_dir}/{genre}_audio_{train,val}.npy`` when ``out_dir`` is given (preproc_mdb.py:195-196)."""
    rng = np.random.default_rng(seed)
    t_slice = int(chunk_seconds * rsr)
    parts = [chunk_audio(t, t_slice, n_fft, hop_length, n_random, rng, device) for t in tracks]
    x = torch.cat(parts)                                                 # (N, 1, 2, bins, frames)
    if x.shape[1] == 1:
        x = x[:, 0]                                                      # np.squeeze(axis=1), preproc_mdb.py:179-180
    mean = x.double().mean()
    std = x.double().std(unbiased=False)                                 # numpy .std() is the population std
    x = ((x - mean.float()) / std.float()).cpu().numpy().astype(np.float32)
    idx = np.linspace(0, len(x) - 1, len(x), dtype=int)
    rng.shuffle(idx)
    val, train = x[idx][:n_val], x[idx][n_val:]
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        np.save(os.path.join(out_dir, f"{genre}_audio_val.npy"), val)
        np.save(os.path.join(out_dir, f"{genre}_audio_train.npy"), train)
    return train, val
