#!/usr/bin/env python3
"""Dev tool: U0 / U1 / D1 / D0 forward on the bf16-resident kernel, product build (argument 0) or an ablation build
(tools/abl/build_habl.sh; argument n = 1..6).  One library per process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
n = sys.argv[1] if len(sys.argv) > 1 else "0"
n = 0 if n == "0" else n
from phasegen import _lib
if n:
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl", "libphasegen_" + (("habl" + n) if n.isdigit() else n) + ".so")
import torch
import bench
from phasegen import ops
from phasegen.unet import LAYERS, frame_plan
C, L, B = 1024, 256, 64
L1, L2, L3, L4 = frame_plan(L)
geo = {"U0": (4 * C, 2 * C, 32, L1), "U1": (4 * C, 2 * C, 8, L2), "D1": (2 * C, 2 * C, 8, L1), "D0": (C, 2 * C, 32, L)}
fl = bench.conv_flops(C, L, B)
out = []
for name, (Cin, Cout, k, Lin) in geo.items():
    _, kind, s, p = LAYERS[name]
    tr = kind == "t"
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda") * 0.02
    y = torch.empty(B, Cout, Lout, device="cuda")
    xh = ops.h_alloc(B, Cin, Lin, "cuda"); ops.cast_rows_bf16(x, xh); wh = ops.shadow_weights(w, tr, s)
    fn = lambda: ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y)
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out.append(f"{name} {ms:.3f} ms {fl[name]/ms/1e9:.0f} TF")
print(f"habl={n}: " + " | ".join(out), flush=True)
