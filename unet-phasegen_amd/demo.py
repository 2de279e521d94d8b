#!/usr/bin/env python3
"""Counterpart of the reference's ``demo.py``: same flags and defaults (demo.py:9-16), same input
(``dataset/{genre}_audio_val.npy``), same outputs (``demo/unet_{genre}_{c}.wav``), same timed region per clip
(forward + device->host + ISTFT, demo.py:35-41) and the same printed line ``UNet - avg {} sec per clip.``.

The Griffin-Lim comparator (demo.py:47-60, utils.griffin_lim) runs batched on the device
(``phasegen.audio.griffin_lim_batch``) and prints the reference's second line, ``GL - avg {} sec per clip``.
WAV files are written with scipy (float32 PCM, what librosa.output.write_wav produced; librosa is not installed).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    parser = argparse.ArgumentParser(description="Arguments for generating demo clips.")
    parser.add_argument("--genre", required=True)
    parser.add_argument("--n_songs", default=5, type=int)
    parser.add_argument("--n_fft", default=2048, type=int)
    parser.add_argument("--sr", default=16000, type=int)
    parser.add_argument("--hop", default=512, type=int)
    parser.add_argument("--gpu", default=0, type=int, help="reference default: 3 (the author's box); here 0")
    parser.add_argument("--weight", required=True)
    parser.add_argument("--channels", default=1024, type=int, help="bins = model width (reference hard-codes 1024)")
    parser.add_argument("--dataset_dir", default="dataset")
    parser.add_argument("--out_dir", default="demo")
    parser.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"], default="fp32",
                    help="MFMA operand mode of the convolutions (pg_conv_args.precision): fp32 = the reference's arithmetic")
    args = parser.parse_args()

    import numpy as np
    import torch
    from scipy.io import wavfile
    from data import get_fft_npy_loader
    from utils import generate_audio
    from cycleGAN import UNetModel
    from phasegen import audio as pg_audio

    torch.cuda.set_device(args.gpu)
    loader = get_fft_npy_loader([os.path.join(args.dataset_dir, args.genre + "_audio_val.npy")], [0, 1],
                                batch_size=args.n_songs, precon=True)
    model = UNetModel(args.channels, args.channels * 2, gpu_ids=[args.gpu], precision=args.precision).cuda(args.gpu)
    model.load(args.weight)
    data = loader.__iter__().__next__()[0]
    os.makedirs(args.out_dir, exist_ok=True)

    runtimes = []
    with torch.no_grad():
        for c, d in enumerate(data):
            d = d.unsqueeze(0)
            start = time.time()
            pred = model.forward(d[:, 0].cuda(args.gpu))
            mag = d.cpu().numpy()[0]
            pred = pred.data.cpu().numpy()[0, :args.channels, ...]
            stft = (np.exp(mag[0]) - 1) * np.exp(pred * 1.j)
            audio = generate_audio(stft, sr=args.sr, hop_length=args.hop, is_stft=True)
            end = time.time() - start
            runtimes.append(end)
            wavfile.write(os.path.join(args.out_dir, "unet_{}_{}.wav".format(args.genre, c)), args.sr, audio.astype(np.float32))
    print("UNet - avg {} sec per clip.".format(np.mean(runtimes)))

    # Griffin-Lim comparator (demo.py:50-58 -> utils.py:112-134): all clips in ONE batched run of 250 iterations -- four
    # launches per iteration whatever the number of clips -- timed as a whole; the printed average is that time per clip
    start = time.time()
    mags = torch.exp(data[:, 0].cuda(args.gpu)) - 1
    lims, _, _ = pg_audio.griffin_lim_batch(mags, n_fft=args.n_fft, hop_length=args.hop, n_iter=250)
    lims = lims.cpu().numpy()
    per_clip = (time.time() - start) / len(lims)
    for c, lim in enumerate(lims):
        wavfile.write(os.path.join(args.out_dir, "gl_{}_{}.wav".format(args.genre, c)), args.sr, lim.astype(np.float32))
    print("GL - avg {} sec per clip".format(per_clip))


if __name__ == "__main__":
    main()
