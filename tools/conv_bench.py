#!/usr/bin/env python3
"""Per-layer conv timing at the bench shape (dev tool): python tools/conv_bench.py [layer.pass ...] [--reps N]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from phasegen import ops  # noqa: E402
from phasegen.unet import LAYERS, frame_plan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("what", nargs="*", default=["U0.fwd", "U0.dgrad", "U0.wgrad", "D0.fwd", "D1.fwd", "U1.fwd", "U3.fwd"])
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--lib", default=None)
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--act", type=int, default=2)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--schedule", type=int, default=0)
ap.add_argument("--oversub", type=int, default=4)
ap.add_argument("--frames", type=int, default=256)
a = ap.parse_args()
if a.lib:
    from phasegen import _lib
    _lib.LIB_PATH = os.path.abspath(a.lib)
C, L, B = 1024, a.frames, a.batch
ops.set_conv_precision(a.precision)
if a.schedule & 0x70: ops._tls.schedule = a.schedule      # bits the test hook refuses (16 = the data-parallel 'contended' split): dev tool only
else: ops.set_conv_schedule(a.schedule)
ops.set_conv_oversubscribe(a.oversub)
L1, L2, L3, L4 = frame_plan(L)
geo = {"D0": (C, 2 * C, 32, L), "D1": (2 * C, 2 * C, 8, L1), "D2": (2 * C, 2 * C, 8, L2), "D3": (2 * C, 4 * C, 4, L3),
       "U3": (4 * C, 2 * C, 5, L4), "U2": (4 * C, 2 * C, 8, L3), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
fl = bench.conv_flops(C, L, B)
for item in a.what:
    name, ps = item.split(".")
    Cin, Cout, k, Lin = geo[name]
    _, kind, s, p = LAYERS[name]
    tr = kind == "t"
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda")
    w = torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda") * 0.02
    y = torch.empty(B, Cout, Lout, device="cuda")
    dy = torch.randn(B, Cout, Lout, device="cuda")
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    fn = {"fwd": lambda: ops.conv_fwd(x, w, y, s, p, x_act=a.act, transposed=tr),
          "dgrad": lambda: ops.conv_dgrad(dy, w, dx, s, p, transposed=tr, ref=x, mask=2),
          "wgrad": lambda: ops.conv_wgrad(x, dy, dw, s, p, x_act=a.act, transposed=tr)}[ps]
    fn(); torch.cuda.synchronize()
    if a.stamps:
        ops.set_conv_schedule(1)
        ws = ops.conv_workspace(x.device); ws[:64].zero_(); fn(); torch.cuda.synchronize()
        c = ws[:32].view(torch.int64).cpu().tolist()
        nsl = max(c[3], 1)
        print(f"{item}: per wave-slab cycles  issue {c[0]/nsl:.0f}  read+mfma {c[1]/nsl:.0f}  barrier {c[2]/nsl:.0f}  total {(c[0]+c[1]+c[2])/nsl:.0f}  (wave-slabs {nsl})")
        ops.set_conv_schedule(0)
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print(f"{item:10s} {ms:8.3f} ms  {fl[name] / ms / 1e9:7.1f} TFLOP/s  ({fl[name] / ms / 1e9 / 157.3 * 100:.0f}%)", flush=True)
