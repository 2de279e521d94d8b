"""dev: STFT 2048/512 (+ polar) timing with phases compiled out (tools/abl/libphasegen_stft<n>.so built with -DPG_STFT_ABL=n)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
from phasegen import _lib
if len(sys.argv) > 1 and sys.argv[1] != "0":
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl", "libphasegen_stft" + sys.argv[1] + ".so")
import torch
from phasegen import ops
n_fft, hop, n, nsig = 2048, 512, 255 * 512, 64
y = torch.randn(nsig, n, device="cuda") * 0.1
nf = 1 + n // hop
out = torch.empty(nsig, 2, n_fft // 2, nf, device="cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
by = nsig * nf * (n_fft * 4 // 4 * 1 + 0)  # placeholder
bytes_ = nsig * (n * 4 + 2 * (n_fft // 2) * nf * 4)
for polar in (False, True):
    us = t(lambda: ops.stft(y, n_fft, hop, polar=polar, out=out))
    print(f"abl {sys.argv[1] if len(sys.argv) > 1 else 0} polar={polar}: {us:7.1f} us  {bytes_ / us / 1e6:6.2f} TB/s", flush=True)
re, im = out[:, 0].contiguous(), out[:, 1].contiguous()
us = t(lambda: ops.istft(re, im, hop, mode=0, normalize=True))
print(f"istft mode 0 (64 x {nf} frames): {us:7.1f} us  {bytes_ / us / 1e6:6.2f} TB/s")
