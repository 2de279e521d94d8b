#!/usr/bin/env python3
"""dev tool: the k = 32 wgrad (D0 geometry) at frame counts whose samples do / do not end inside a 16-frame slab."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
from phasegen import ops
B, Cin, Cout, k, s, p = 64, 1024, 2048, 32, 2, 16
for Lin in [int(a) for a in sys.argv[1:]] or [256, 254, 222, 286]:
    Lout = ops.conv_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda"); dy = torch.randn(B, Cout, Lout, device="cuda")
    dw = torch.empty(Cout, Cin, k, device="cuda")
    fn = lambda: ops.conv_wgrad(x, dy, dw, s, p, transposed=False)
    for _ in range(6): fn()          # warm: the first configuration of a process otherwise reads ~2 points low
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * Lout * Cin * Cout * k
    print(f"Lin {Lin} frames out {Lout}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s ({fl / ms / 1e9 / 157.3 * 100:.1f} %)", flush=True)
