"""Dev: error map of the fp32 one-wave-per-SIMD conv kernels (conv_raw3.hip) against torch on the CPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd"))
import torch, torch.nn.functional as F
from phasegen import ops
geoms = [(False, 64, 160, 32, 2, 16, 128, 2), (False, 64, 16, 32, 2, 16, 128, 2), (False, 8, 160, 32, 2, 16, 128, 2), (False, 64, 160, 32, 2, 16, 24, 1),
         (False, 16, 160, 32, 2, 16, 128, 2), (False, 24, 160, 32, 2, 16, 128, 2)]
for tr, Cin, Cout, k, s, p, Lin, B in geoms:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, Lin, generator=g); w = torch.randn(*((Cin, Cout, k) if tr else (Cout, Cin, k)), generator=g) * 0.1
    want = (F.conv_transpose1d if tr else F.conv1d)(x.double(), w.double(), stride=s, padding=p)
    Lout = want.shape[2]
    for sched in (1, 2, 0x2000 | 1):
        y = torch.full((B, Cout, Lout), float("nan"), device="cuda")
        ops.conv_fwd(x.cuda(), w.cuda(), y, s, p, transposed=tr, schedule=sched)
        err = (y.cpu().double() - want).abs()
        print((tr, Cin, Cout, k, s, p, Lin, B), "sched", hex(sched), "max err", float(err.max()), "nan", int(torch.isnan(y).sum()), flush=True)
        if float(err.max()) > 1e-3 or torch.isnan(y).any():
            bad = (err > 1e-3) | torch.isnan(y.cpu())
            print("  bad per sample:", bad.sum((1, 2)).tolist())
            print("  bad per 32-row block:", [int(bad[:, i:i + 32].sum()) for i in range(0, Cout, 32)])
            cols = bad.permute(1, 0, 2).reshape(Cout, -1)
            print("  bad per 32-col block:", [int(cols[:, j:j + 32].sum()) for j in range(0, cols.shape[1], 32)])
