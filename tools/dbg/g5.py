"""dev: where does the raw k = 5 wgrad differ from torch?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "unet-phasegen_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, torch.nn.functional as F
from oracle import unet_ref  # noqa
from phasegen import ops, detgen
def rnd(seed, *shape): return torch.from_numpy(detgen.uniform(seed, shape, -1.0, 1.0))
for (tr, Cin, Cout, k, s, p, Lin, B) in [(True, 72, 140, 5, 2, 1, 30, 9), (True, 8, 60, 5, 2, 1, 30, 2), (False, 60, 8, 5, 2, 1, 61, 2)]:
    x = rnd(1, B, Cin, Lin); w = rnd(2, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose1d(xr, wr, stride=s, padding=p) if tr else F.conv1d(xr, wr, stride=s, padding=p)
    dy = rnd(3, *yr.shape); yr.backward(dy)
    for sched in (1, 5):
        dw = torch.full(w.shape, float("nan"), device="cuda")
        ops.conv_wgrad(x.cuda(), dy.cuda(), dw, s, p, transposed=tr, schedule=sched)
        err = (dw.cpu() - wr.grad).abs()
        big = err > 1e-3 * wr.grad.abs().max()
        print((tr, Cin, Cout, k, s, p, Lin, B), "sched", sched, "max err", float(err.max()), "bad", int(big.sum()), "of", big.numel())
        if big.any():
            q_axis = 1 if tr else 1
            print("  bad per m (first 8):", big.sum(dim=(1, 2))[:8].tolist(), " per q:", big.sum(dim=(0, 2)).tolist()[:64], " per j:", big.sum(dim=(0, 1)).tolist())
