#!/usr/bin/env python3
"""dev tool: STFT (+ polar) and ISTFT at the e2e shape (64 signals x 256 frames of 2048 / 512) and the demo shape (1 x 128), event-timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
from phasegen import ops

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us

for nsig, frames, n_fft, hop in ((64, 256, 2048, 512), (1, 128, 2048, 512), (64, 256, 1024, 256)):
    n = hop * (frames - 1)
    y = torch.randn(nsig, n, device="cuda") * 0.1
    out = torch.empty(nsig, 2, n_fft // 2, frames, device="cuda")
    a_in, a_out = nsig * n * 4, out.numel() * 4
    t = timeit(lambda: ops.stft(y, n_fft, hop, out=out))
    print(f"[{nsig} x {frames} @ {n_fft}/{hop}] stft        {t:8.1f} us  {(a_in + a_out) / t / 1e6:6.2f} TB/s algorithmic")
    t = timeit(lambda: ops.stft(y, n_fft, hop, polar=True, out=out))
    print(f"[{nsig} x {frames} @ {n_fft}/{hop}] stft+polar  {t:8.1f} us  {(a_in + a_out) / t / 1e6:6.2f} TB/s")
    p2 = torch.empty_like(out)
    t = timeit(lambda: ops.polar(out, p2))
    print(f"[{nsig} x {frames} @ {n_fft}/{hop}] polar       {t:8.1f} us  {2 * a_out / t / 1e6:6.2f} TB/s")
    lm, ph = p2[:, 0].contiguous(), p2[:, 1].contiguous()
    for mode, norm in ((0, True), (0, False), (1, True)):
        t = timeit(lambda: ops.istft(lm, ph, hop, mode=mode, normalize=norm))
        alg = a_out + a_in * (3 if norm else 1)
        print(f"[{nsig} x {frames} @ {n_fft}/{hop}] istft mode {mode} normalize {int(norm)}  {t:8.1f} us  {alg / t / 1e6:6.2f} TB/s algorithmic")
    t = timeit(lambda: ops.istft(lm, ph, hop, mode=0, normalize=True, single_frame=True))
    print(f"[{nsig} x {frames} @ {n_fft}/{hop}] istft three-kernel path (one frame per workgroup)  {t:8.1f} us")
