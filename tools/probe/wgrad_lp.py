#!/usr/bin/env python3
"""Dev probe: does the sample-crossing slab path of the raw wgrad kernel cost time?  U0-shaped wgrad at Lin = 128 (LP % 16 == 0:
no slab crosses a sample) vs 129 (every 8th slab does)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd"))
import torch
from phasegen import ops
B, Cin, Cout, k, s, p = 64, 4096, 2048, 32, 2, 16
for Lin in (128, 129, 144, 130):
    Lout = ops.convt_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda"); dy = torch.randn(B, Cout, Lout, device="cuda"); dw = torch.empty(Cin, Cout, k, device="cuda")
    for _ in range(2): ops.conv_wgrad(x, dy, dw, s, p, transposed=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv_wgrad(x, dy, dw, s, p, transposed=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * Lin * Cin * Cout * k
    print(f"Lin {Lin}: {ms:.3f} ms {fl / ms / 1e9:.1f} TFLOP/s ({fl / ms / 1e9 / 157.3 * 100:.1f} %)", flush=True)
