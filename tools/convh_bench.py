#!/usr/bin/env python3
"""Dev tool: forward convs at the bench shape, fp32-tensor kernels in bf16 operand mode vs the bf16-resident kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from phasegen import ops
from phasegen.unet import LAYERS, frame_plan
C, L, B = 1024, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 64
L1, L2, L3, L4 = frame_plan(L)
geo = {"D0": (C, 2 * C, 32, L), "D1": (2 * C, 2 * C, 8, L1), "D2": (2 * C, 2 * C, 8, L2), "D3": (2 * C, 4 * C, 4, L3),
       "U2": (4 * C, 2 * C, 8, L3), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
fl = bench.conv_flops(C, L, B)
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = [0.0, 0.0]
for name, (Cin, Cout, k, Lin) in geo.items():
    _, kind, s, p = LAYERS[name]
    tr = kind == "t"
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda") * 0.02
    y = torch.empty(B, Cout, Lout, device="cuda")
    xh = ops.h_alloc(B, Cin, Lin, "cuda"); ops.cast_rows_bf16(x, xh); wh = ops.shadow_weights(w, tr, s)
    yh = ops.h_alloc(B, Cout, Lout, "cuda")
    a = timeit(lambda: ops.conv_fwd(x, w, y, s, p, transposed=tr, precision="bf16"))
    b = timeit(lambda: ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y))
    c = timeit(lambda: ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, yh=yh, yh_act=2))
    tot[0] += a; tot[1] += b
    print(f"{name}: bf16 operand mode {a:7.3f} ms {fl[name]/a/1e9:7.1f} TF | resident (fp32 out) {b:7.3f} ms {fl[name]/b/1e9:7.1f} TF ({fl[name]/b/1e9/2516.6*100:.0f} % of 2.5 PF) | resident (bf16 out) {c:7.3f} ms", flush=True)
print(f"sum: {tot[0]:.2f} -> {tot[1]:.2f} ms")
