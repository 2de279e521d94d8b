#!/usr/bin/env python3
"""BatchNorm kernels alone at the bench shapes (dev tool): cold operands (8 rotating buffer sets, > the 256 MB Infinity Cache)
and warm (one set reused), against the algorithmic bytes -- forward: 1 read + 1 write (+ the second activated copy where the
layer stores one), backward: 2 reads + 1 write."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
from phasegen import ops

def timeit(fns, reps):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

B, C = 64, 2048
for L in (256, 129, 126, 61):
    sets = []
    for _ in range(8):
        x = torch.randn(B, C, L, device="cuda"); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
        sets.append((x, y, dy, dx))
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); sm = torch.empty(C, device="cuda"); si = torch.empty(C, device="cuda")
    dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    fw = [lambda s=s: ops.bn_fwd(s[0], s[1], g, b, sm, si) for s in sets]
    bw = [lambda s=s: ops.bn_bwd(s[0], s[2], s[3], g, sm, si, dg, db) for s in sets]
    nbytes = B * C * L * 4
    for name, fns, mult in (("fwd", fw, 2), ("bwd", bw, 3)):
        cold = timeit(fns, 40); warm = timeit(fns[:1], 40)
        print(f"bn_{name} (64, 2048, {L}): cold {cold:6.1f} us = {mult*nbytes/cold/1e6:5.2f} TB/s   warm {warm:6.1f} us = {mult*nbytes/warm/1e6:5.2f} TB/s")
