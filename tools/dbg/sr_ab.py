#!/usr/bin/env python3
"""dev tool: conv_raw3's tile order (super-row height R, schedule bits 15-16) on the k = 32 and k = 8 F / T launches at the bench shape.
   python tools/dbg/sr_ab.py [time|once R]   -- `once R`: a single launch per layer with that R (for rocprofv3 --pmc FETCH_SIZE)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from phasegen import ops
from phasegen.unet import LAYERS, frame_plan
C, L, B = 1024, 256, 64
L1, L2, L3, L4 = frame_plan(L)
geo = {"D0": (C, 2 * C, 32, L), "D1": (2 * C, 2 * C, 8, L1), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
fl = bench.conv_flops(C, L, B)
cases = [("U0", "dgrad"), ("D0", "fwd"), ("U0", "fwd"), ("U1", "fwd"), ("U1", "dgrad"), ("D1", "fwd")]
mode = sys.argv[1] if len(sys.argv) > 1 else "time"
for name, ps in cases:
    Cin, Cout, k, Lin = geo[name]
    _, kind, s, p = LAYERS[name]
    tr = kind == "t"
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda")
    w = torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda") * 0.02
    y = torch.empty(B, Cout, Lout, device="cuda"); dy = torch.randn(B, Cout, Lout, device="cuda"); dx = torch.empty_like(x)
    def run(sched):
        if ps == "fwd": ops.conv_fwd(x, w, y, s, p, transposed=tr, schedule=sched)
        else: ops.conv_dgrad(dy, w, dx, s, p, transposed=tr, ref=x, mask=2, schedule=sched)
    if mode == "once":
        R = int(sys.argv[2]); code = {1: 1, 2: 2, 4: 3}[R]
        run(code << 15); torch.cuda.synchronize()
        continue
    out = []
    for R, code in ((1, 1), (2, 2), (4, 3)):
        sched = code << 15
        run(sched); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run(sched)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out.append(f"R={R}: {ms:7.3f} ms ({fl[name] / ms / 1e9 / 157.3:.3f})")
    print(f"{name}.{ps:6s}", "   ".join(out), flush=True)
