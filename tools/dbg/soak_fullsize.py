"""Dev: launch-to-launch bit-identity of the one-wave-per-SIMD kernels at the bench shape (64 x 1024 x 256): every fp32 forward / dgrad
layer on conv_raw3 and every bf16-resident forward layer on conv_h3, N launches each (default 20), all compared with the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
from phasegen import ops
from phasegen.unet import LAYERS, frame_plan
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C, L, B = 1024, 256, 64
L1, L2, L3, L4 = frame_plan(L)
geo = {"D0": (C, 2 * C, 32, L), "D1": (2 * C, 2 * C, 8, L1), "D2": (2 * C, 2 * C, 8, L2), "D3": (2 * C, 4 * C, 4, L3),
       "U3": (4 * C, 2 * C, 5, L4), "U2": (4 * C, 2 * C, 8, L3), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
bad = 0
for name, (Cin, Cout, k, Lin) in geo.items():
    _, kind, s, p = LAYERS[name]
    tr = kind == "t"
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda") * 0.02
    dy = torch.randn(B, Cout, Lout, device="cuda")
    xh = ops.h_alloc(B, Cin, Lin, "cuda"); ops.cast_rows_bf16(x, xh); wh = ops.shadow_weights(w, tr, s)
    first = None
    for it in range(N):
        y = torch.empty(B, Cout, Lout, device="cuda"); dx = torch.empty_like(x); yb = torch.empty(B, Cout, Lout, device="cuda")
        ops.conv_fwd(x, w, y, s, p, transposed=tr)
        ops.conv_dgrad(dy, w, dx, s, p, transposed=tr)
        ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=yb)
        if first is None: first = (y, dx, yb)
        else:
            ok = torch.equal(y, first[0]), torch.equal(dx, first[1]), torch.equal(yb, first[2])
            if not all(ok): bad += 1; print(name, "launch", it, "differs (fwd, dgrad, bf16 fwd):", ok, flush=True)
    print(name, "ok" if not bad else "MISMATCHES", flush=True)
print("bit-identical over", N, "launches" if not bad else f"launches: {bad} mismatching launches")
sys.exit(1 if bad else 0)
