#!/usr/bin/env python3
"""dev tool: bench.py's demo_clip leg alone (demo.py's timed region at batch 1, fp32 and bf16-resident)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from phasegen.model import UNetModel
torch.manual_seed(0)
m = UNetModel(1024, 2048, gpu_ids=[0])
r = bench.measure_demo_clip(torch, m, 1024)
for k in ("fp32", "bf16"):
    v = r[k]
    print(k, v["ms_per_clip"], "ms/clip; convs", v["conv_launches_ms"], "ms;", v["weight_TBps"], "TB/s of weights;", json.dumps(v["by_layer_ms"]))
