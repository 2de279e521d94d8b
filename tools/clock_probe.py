#!/usr/bin/env python3
"""Dev tool: in-kernel clock under load (s_memtime / s_memrealtime x 100 MHz, MI355X_MICROARCH.md 'DVFS give-back' item 6)
of the U0-forward kernel in the three operand modes.  Needs a diagnostic build:
    hipcc ... -DPG_ABL=8 -c conv_raw.hip -o build/conv_raw_clk.o ; link with the other objects into tools/abl/lib_clk.so"""
import os, sys
sys.path.insert(0, "unet-phasegen_amd"); sys.path.insert(0, ".")
from phasegen import _lib
_lib.LIB_PATH = os.path.abspath("tools/abl/lib_clk.so")
import torch
from phasegen import ops
ops.set_conv_schedule(2)
B, Cin, Cout, k, s, p, Lin = 64, 4096, 2048, 32, 2, 16, 129
x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cin, Cout, k, device="cuda") * 0.02
y = torch.empty(B, Cout, ops.convt_out_len(Lin, k, s, p), device="cuda")
ws = ops.conv_workspace(x.device)
for prec in ("fp32", "bf16x3", "bf16"):
    ops.set_conv_precision(prec)
    import time
    t0 = time.time()
    while time.time() - t0 < 2.5:                       # >= 2 s of back-to-back launches
        for _ in range(5): ops.conv_fwd(x, w, y, s, p, transposed=True)
        torch.cuda.synchronize()
    ops.conv_fwd(x, w, y, s, p, transposed=True); torch.cuda.synchronize()
    d = ws[:32].view(torch.int64).cpu().tolist()
    print(prec, "in-kernel clock: %.3f GHz  (cycles %d over %.2f ms)" % (d[0] / d[1] * 0.1, d[0], d[1] / 1e5))
