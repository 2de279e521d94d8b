"""dev: A/B of conv_h2 variant libraries on U0 / U1 / D0 / D1 forward (bf16-resident), interleaved in ONE process is not possible
across libraries (one CDLL per process): run each library in its own process on the same box, same data, several rounds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
from phasegen import _lib
name = sys.argv[1] if len(sys.argv) > 1 else ""
if name:
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl", "libphasegen_" + name + ".so")
import torch
import bench
from phasegen import ops
from phasegen.unet import LAYERS, frame_plan
C, L, B = 1024, 256, 64
L1, L2, L3, L4 = frame_plan(L)
geo = {"D0": (C, 2 * C, 32, L), "D1": (2 * C, 2 * C, 8, L1), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
fl = bench.conv_flops(C, L, B)
torch.manual_seed(0)
out = []
for nm, (Cin, Cout, k, Lin) in geo.items():
    _, kind, s, p = LAYERS[nm]
    tr = kind == "t"
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda") * 0.02
    y = torch.empty(B, Cout, Lout, device="cuda")
    xh = ops.h_alloc(B, Cin, Lin, "cuda"); ops.cast_rows_bf16(x, xh); wh = ops.shadow_weights(w, tr, s)
    fn = lambda: ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y, schedule=64)
    for _ in range(3): fn()
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    t = sorted(ts)[3]
    out.append(f"{nm} {t:.3f} ms {fl[nm] / t / 1e9:5.0f} TF")
print(f"{name or 'product':10s}", " | ".join(out), flush=True)
