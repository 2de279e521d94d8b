#!/usr/bin/env python3
"""Dev tool: the bf16-resident forward convs at the bench shape under each tile family (pg_convh_args.schedule bits 5-6):
128 x 256 on 4 waves (two workgroups per CU), 128 x 512 and 256 x 256 on 8 waves (one per CU).  Interleaved rounds in ONE
process (cdna_hip_programming.md rule 24); prints the median per layer and family, TFLOP/s and the fraction of 2.5 PF."""
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
       "U3": (4 * C, 2 * C, 5, L4), "U2": (4 * C, 2 * C, 8, L3), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
fl = bench.conv_flops(C, L, B)
FAM = {"256x256w4": 4096, "256x256w4/1tile": 4097, "128x256": 32, "128x512": 64, "256x256": 96, "auto": 0, "128x256/1tile": 33, "128x512/1tile": 65, "256x256/1tile": 97}
def once(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = {k: 0.0 for k in FAM}
for name, (Cin, Cout, k, Lin) in geo.items():
    _, kind, s, p = LAYERS[name]
    tr = kind == "t"
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda") * 0.02
    y = torch.empty(B, Cout, Lout, device="cuda")
    xh = ops.h_alloc(B, Cin, Lin, "cuda"); ops.cast_rows_bf16(x, xh); wh = ops.shadow_weights(w, tr, s)
    fns = {f: (lambda sc=sc: ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y, schedule=sc)) for f, sc in FAM.items()}
    ts = {f: [] for f in FAM}
    for f in FAM: fns[f](); fns[f]()
    torch.cuda.synchronize()
    for rnd in range(5):
        for f in FAM: ts[f].append(once(fns[f]))
    line = f"{name}:"
    for f in FAM:
        t = sorted(ts[f])[len(ts[f]) // 2]
        tot[f] += t
        line += f"  {f} {t:6.3f} ms {fl[name] / t / 1e9:6.0f} TF ({fl[name] / t / 1e9 / 2516.6 * 100:4.1f} %)"
    print(line, flush=True)
print("sum: " + "  ".join(f"{f} {t:.3f} ms" for f, t in tot.items()), f"| all conv FLOPs {sum(fl.values()) / 1e12:.3f} T")
