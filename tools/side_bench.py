#!/usr/bin/env python3
"""Side-path measurements (dev tool): STFT / polar / ISTFT bandwidth, the preprocessing kernels (chunked STFT source, whole-array
standardisation), forward-only throughput (BASELINE config 2),
per-clip inference latency (demo.py's timed region), Griffin-Lim time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from phasegen import ops, audio
from phasegen.model import UNetModel

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for n_fft, hop, n, nsig in ((2048, 512, 65024, 512), (1024, 256, 65280, 512)):
    y = torch.randn(nsig, n, device="cuda")
    nf = 1 + n // hop
    out = torch.empty(nsig, 2, n_fft // 2, nf, device="cuda")
    ms = timeit(lambda: ops.stft(y, n_fft, hop, out=out))
    fr = nsig * nf
    byt = nsig * n * 4 + out.numel() * 4
    print(f"stft {n_fft}/{hop}: {ms:.3f} ms for {fr} frames = {fr/ms*1e3/1e6:.2f} M frames/s, {byt/ms/1e6:.0f} GB/s algorithmic")
    ms = timeit(lambda: ops.stft(y, n_fft, hop, polar=True, out=out))
    print(f"stft+polar fused: {ms:.3f} ms, {byt/ms/1e6:.0f} GB/s")
    p2 = torch.empty_like(out)
    ms = timeit(lambda: ops.polar(out, p2))
    print(f"polar: {ms:.3f} ms, {2*out.numel()*4/ms/1e6:.0f} GB/s")
    # row N2: the same frames read as chunks of 8 long tracks in place (odd starts), and the whole-array standardisation
    src = torch.randn(8, 64 * n + 999, device="cuda")
    st = (torch.arange(nsig, device="cuda", dtype=torch.int64) // 8) * n + 1
    rows = (torch.arange(nsig, device="cuda") % 8).int()
    ms = timeit(lambda: ops.stft(src, n_fft, hop, out=out, chunk_start=st, chunk_row=rows, chunk_len=n))
    print(f"stft chunked in place: {ms:.3f} ms, {byt/ms/1e6:.0f} GB/s algorithmic")
    ms = timeit(lambda: ops.standardize_(out))
    print(f"standardize ({out.numel()/1e6:.0f} M values): {ms:.3f} ms, {out.numel()*16/ms/1e6:.0f} GB/s (2 reads for the moments + read/write)")
    del src
    re, im = out[:64, 0].contiguous(), out[:64, 1].contiguous()
    ms = timeit(lambda: ops.istft(re, im, hop, mode=1, normalize=True))
    print(f"istft (64 clips): {ms:.3f} ms = {64*nf/ms*1e3/1e6:.2f} M frames/s, {(2*re.numel()*4 + 64*hop*(nf-1)*4)/ms/1e6:.0f} GB/s algorithmic")
    del y, out, p2

C = 1024
m = UNetModel(C, 2 * C)
for B, L in ((32, 256), (64, 256)):
    x = torch.randn(B, C, L, device="cuda")
    ms = timeit(lambda: m.engine.forward(x, update_stats=False), reps=5)
    print(f"forward only B={B} L={L}: {ms:.2f} ms = {B*L/ms*1e3:.0f} frames/s ({128.748e9*B/ms/1e9:.1f} TFLOP/s)")
x1 = torch.randn(1, 2, C, 128, device="cuda").abs()
def demo_clip():
    with torch.no_grad():
        pred = m.forward(x1[:, 0])
        a = audio.synthesize(x1[:, 0], pred[:, :C].contiguous(), 512)
    return a.cpu()
demo_clip(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): demo_clip()
print(f"demo timed region (forward + ISTFT + D2H, 1 clip 1024x128): {(time.perf_counter()-t0)/5*1e3:.2f} ms per clip")
mag = np.expm1(x1[0, 0].cpu().numpy())
t0 = time.perf_counter(); audio.griffin_lim(mag, 2048, 512, 250, seed=0); print(f"griffin_lim 250 iterations: {(time.perf_counter()-t0)*1e3:.0f} ms per clip")
