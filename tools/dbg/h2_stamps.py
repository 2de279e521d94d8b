"""dev: s_memtime phases per iteration of conv_h2 (build: hipcc -DPG_ABL=7 conv_h2.hip, linked into tools/abl/libphasegen_h2s.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
from phasegen import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "abl", "libphasegen_" + (sys.argv[1] if len(sys.argv) > 1 else "h2s") + ".so")
import torch
from phasegen import ops
from phasegen.unet import LAYERS, frame_plan
C, L, B = 1024, 256, 64
L1, L2, L3, L4 = frame_plan(L)
geo = {"D0": (C, 2 * C, 32, L), "D1": (2 * C, 2 * C, 8, L1), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
for name, (Cin, Cout, k, Lin) in geo.items():
    _, kind, s, p = LAYERS[name]
    tr = kind == "t"
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda") * 0.02
    y = torch.empty(B, Cout, Lout, device="cuda")
    xh = ops.h_alloc(B, Cin, Lin, "cuda"); ops.cast_rows_bf16(x, xh); wh = ops.shadow_weights(w, tr, s)
    for fam, sc in (("128x512", 64), ("256x256", 96)):
        ws = ops.conv_workspace(xh.device)
        for _ in range(2):
            ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y, schedule=sc)
        torch.cuda.synchronize()
        ws[:32].zero_()
        torch.cuda.synchronize()
        ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y, schedule=sc)
        torch.cuda.synchronize()
        v = ws[:32].view(torch.int64).cpu().tolist()
        slabs = max(v[3], 1)
        print(f"{name} {fam}: per wave and slab: gather issue {v[0] / slabs:7.1f}  fragments+MFMA {v[1] / slabs:7.1f}  barrier {v[2] / slabs:7.1f} cycles "
              f"(sum {sum(v[:3]) / slabs:7.1f}; 16 MFMAs = 512 pipe cycles, two waves share a SIMD)", flush=True)
