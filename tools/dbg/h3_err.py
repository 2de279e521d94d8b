"""Dev: error map of the one-wave-per-SIMD conv_h family on one geometry (per 32 x 32 block of the GEMM view)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd"))
import torch, torch.nn.functional as F
from phasegen import ops
geoms = [(False, c, 64, 8, 1, 2, 65, 3) for c in (8, 16, 24, 32, 40, 48, 64)]
for tr, Cin, Cout, k, s, p, Lin, B in geoms:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, Lin, generator=g); w = torch.randn(*((Cin, Cout, k) if tr else (Cout, Cin, k)), generator=g) * 0.1
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    want = (F.conv_transpose1d if tr else F.conv1d)(xb.double(), wb.double(), stride=s, padding=p)
    Lout = want.shape[2]
    xh = ops.h_alloc(B, Cin, Lin, "cuda"); ops.cast_rows_bf16(x.cuda(), xh); wh = ops.shadow_weights(w.cuda(), tr, s)
    for sched in (4097,):
        y = torch.full((B, Cout, Lout), float("nan"), device="cuda")
        ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y, schedule=sched)
        err = (y.cpu().double() - want).abs()            # (B, Cout, Lout)
        print((tr, Cin, Cout, k, s, p, Lin, B), "sched", sched, "max err", float(err.max()), "nan", int(torch.isnan(y).sum()))
        if float(err.max()) > 1e-3 or torch.isnan(y).any():
            bad = (err > 1e-3) | torch.isnan(y.cpu())
            print("  bad per sample:", bad.sum((1, 2)).tolist())
            print("  bad per 32-row block:", [int(bad[:, i:i + 32].sum()) for i in range(0, Cout, 32)])
            cols = bad.permute(1, 0, 2).reshape(Cout, -1)       # columns n = b * Lout + t
            print("  bad per 32-col block:", [int(cols[:, j:j + 32].sum()) for j in range(0, cols.shape[1], 32)])
            print("  bad per row mod 8:", [int(bad[:, i::8].sum()) for i in range(8)])
