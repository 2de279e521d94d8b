#!/usr/bin/env python3
"""How do the conv schedules behave when another kernel holds part of the chip (as RCCL does during DP backward)?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from phasegen import ops
spin = ctypes.CDLL(os.path.join(ROOT, "tools", "spin", "libspin.so"))
spin.pg_dev_spin.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, device="cuda")
side = torch.cuda.Stream()
C, B = 1024, 64
fl = bench.conv_flops(C, 256, B)
x = torch.randn(B, 4 * C, 129, device="cuda"); w = torch.randn(4 * C, 2 * C, 32, device="cuda") * 0.02
dy = torch.randn(B, 2 * C, 256, device="cuda"); dx = torch.empty_like(x); y = torch.empty_like(dy)
cases = {"U0.dgrad (stream-K when auto)": lambda: ops.conv_dgrad(dy, w, dx, 2, 16, transposed=True),
         "U0.fwd (balanced: plain when auto)": lambda: ops.conv_fwd(x, w, y, 2, 16, transposed=True)}
for hold in (0, 16, 48):
    for label, mode, over in (("auto x4", 0, 4), ("stream-K x4", 2, 4), ("stream-K x8", 2, 8), ("tile-per-wg", 1, 1)):
        ops.set_conv_schedule(mode); ops.set_conv_oversubscribe(over)
        out = []
        for name, fn in cases.items():
            fn(); torch.cuda.synchronize()
            if hold:
                with torch.cuda.stream(side):
                    spin.pg_dev_spin(hold, int(2.4e9 * 0.6), sink.data_ptr(), side.cuda_stream)   # ~0.6 s
                torch.cuda._sleep(int(2e6))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): fn()
            e1.record(); e1.synchronize()
            out.append(e0.elapsed_time(e1) / 3)
            torch.cuda.synchronize()
        print(f"held workgroup slots {hold:3d}  {label:12s}  U0.dgrad {out[0]:7.2f} ms   U0.fwd {out[1]:7.2f} ms", flush=True)
ops.set_conv_schedule(0); ops.set_conv_oversubscribe(1)
