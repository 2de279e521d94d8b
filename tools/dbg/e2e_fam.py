"""Dev: the bf16-resident U-Net forward of the e2e bench (64 signals x 256 frames) with every conv forced to one tile family."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from phasegen import ops
from phasegen.model import UNetModel
torch.manual_seed(0)
model = UNetModel(1024, 2048, gpu_ids=[0], precision="bf16")
x = torch.randn(64, 1024, 256, device="cuda") * 0.5
fwd = lambda: model.engine.forward(x, update_stats=False, inference=True)
fl = bench.conv_flops(1024, 256, 64)
res = {}
for rnd in range(3):
    for name, sc in (("auto", 0), ("256x256w4", 4096), ("128x512", 64), ("128x256", 32), ("256x256", 96)):
        ops._tls.schedule = sc
        model.engine.plans.clear() if hasattr(model.engine, "plans") else None
        ms = bench._timed(torch, None, 1, fwd, 2, 10) * 1e3
        res.setdefault(name, []).append(ms)
ops._tls.schedule = 0
for name, v in res.items():
    print(f"{name}: {sorted(v)[1]:.3f} ms  ({sum(fl.values()) / sorted(v)[1] / 1e9:.0f} TF on the conv FLOPs)")
for name, sc in (("auto", 0), ("256x256w4", 4096)):
    ops._tls.schedule = sc
    ks, by = bench.kernel_pass(torch, ops, fwd, 3, fl, 2516.6, 1.0)
    print(name, {k.replace("conv_", ""): round(v["ms_per_step"], 4) for k, v in by.items() if "conv_h" in k})
ops._tls.schedule = 0
