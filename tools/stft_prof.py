#!/usr/bin/env python3
"""Dev tool: STFT / ISTFT launches only, for rocprofv3 --kernel-trace --stats (side path of SURVEY.md row 8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
from phasegen import ops
n_fft, hop, n, nsig = 2048, 512, 65024, 512
y = torch.randn(nsig, n, device="cuda")
nf = 1 + n // hop
out = torch.empty(nsig, 2, n_fft // 2, nf, device="cuda")
for _ in range(10):
    ops.stft(y, n_fft, hop, out=out)
re, im = out[:64, 0].contiguous(), out[:64, 1].contiguous()
for _ in range(10):
    ops.istft(re, im, hop, mode=1, normalize=True)
torch.cuda.synchronize()
