#!/usr/bin/env python3
"""What does a collective's kernel cost the data-parallel backward?  (model.py:40-41 replaced by RCCL.)

Every conv launch of backward (the 15 dgrad / wgrad launches that can overlap a bucket's all-reduce), at the headline shape, while an
RCCL-shaped kernel (tools/spin/spin.hip: 256 threads, <= 113 VGPRs, 32 KB LDS) holds 0 / 16 / 32 / 64 CUs.  Three things such a
kernel can do with the SIMDs it sits on are measured: "rccl+mem" streams memory the whole time (what a collective's data movers do: few
VALU instructions, many outstanding loads / stores), "rccl" spins on dependent-free VALU work (the worst case: every issue slot the
conv wave on the same SIMD does not take is taken), "fat" does the same with 194 VGPRs (no conv_raw3 workgroup fits beside it) --
under the work-split policies the engine can choose from:

    auto                   schedule 0: what a single-GPU step runs
    contended              PG_SCHED_CONTENDED: always the fine stream-K split (no whole-tile / hybrid grids)
    contended+no_raw3      ... and the two-waves-per-SIMD kernels (212-236 registers per wave) instead of conv_raw3 (one wave per SIMD,
                           352-372 registers: a 113-register wave still fits beside it, a 194-register one does not)
    no_raw3                the two-waves-per-SIMD kernels with the automatic grids

One run per cell (backward twice clean, once with an event pair per launch).  Writes gpurun_out/r04_contention.json; the per-layer
policy the table implies is computed at the end (`policy`) and is what phasegen.unet.CONTENDED_SCHEDULE holds.

    python tools/contention.py [--batch 64] [--frames 256] [--rows 0,16:rccl,...]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import bench  # noqa: E402
from hold import Hold  # noqa: E402
from phasegen import _lib, ops  # noqa: E402
from phasegen.model import UNetModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--channels", type=int, default=1024)
ap.add_argument("--rows", default="0,16:rccl+mem,32:rccl+mem,64:rccl+mem,16:rccl,32:rccl,32:fat",
                help="held workgroups : hold-kernel shape (tools/hold.py: rccl = 113 VGPRs + 32 KB LDS, rccl+mem = that plus a memory "
                     "stream, fat = 194 VGPRs: no conv_raw3 workgroup fits beside it)")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_contention.json"))
a = ap.parse_args()

C, L, B = a.channels, a.frames, a.batch
POLICIES = {"auto": 0, "contended": _lib.SCHED_CONTENDED, "contended+no_raw3": _lib.SCHED_CONTENDED | _lib.SCHED_NO_RAW3,
            "no_raw3": _lib.SCHED_NO_RAW3, "engine": None, "engine(wgrad auto)": None}
# "engine": engine.contended = True -- the per-launch table phasegen.unet.CONTENDED_DGRAD / _WGRAD a data-parallel Trainer runs
# (fine split for the long dgrads, one tile per workgroup for the wgrads); "engine(wgrad auto)": the same with automatic wgrad grids
torch.manual_seed(0)
model = UNetModel(C, 2 * C, gpu_ids=[0])
eng = model.engine
batch = bench.synthetic_batch(torch, B, C, L, 1)
pred = eng.forward(batch[:, 0])
dpred = torch.empty_like(pred)
ops.loss_fwd_bwd(pred, batch, dpred)
fl = bench.conv_flops(C, L, B)
hold = Hold()
cells = []


def backward():
    eng.backward(dpred)            # plain order (wgrad, bucket ready, dgrad): what a data-parallel rank enqueues


import phasegen.unet as unet_mod  # noqa: E402
WGRAD_DEFAULT = dict(unet_mod.CONTENDED_WGRAD)


def set_policy(name):
    """Returns the schedule word to install as the thread default for this policy."""
    eng.contended = name.startswith("engine")
    unet_mod.CONTENDED_WGRAD.update({k: (0 if name == "engine(wgrad auto)" else WGRAD_DEFAULT[k]) for k in WGRAD_DEFAULT})
    return 0 if eng.contended else POLICIES[name]


def measure(sched):
    """Enqueue and time on the MAIN stream only; nothing here may wait for the device as a whole (torch.cuda.synchronize() would
    wait for the hold kernel on its side stream: the first version of this tool measured an idle chip that way)."""
    main = torch.cuda.current_stream()
    with ops.conv_options(schedule=sched):
        backward()
        main.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        backward(); backward()
        e1.record()
        e1.synchronize()
        clean = e0.elapsed_time(e1) / 2
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        backward()
        main.synchronize()
        ops.set_timer(None)
    return clean, timer


def per_launch(timer):          # (after the hold kernel has gone: KernelTimer.summary() synchronises the device)
    per = {k: round(ms, 4) for k, (n, ms) in sorted(timer.summary().items())}
    return per, {k: timer.plans.get(k, "?") for k in per}


rows = []
for item in a.rows.split(","):
    h, _, shape = item.partition(":")
    rows.append((int(h), shape or "rccl"))
for held, shape in rows:
    for name in POLICIES:
        sched = set_policy(name)
        with ops.conv_options(schedule=sched):
            backward()                              # warm (workspaces, first-launch costs) before anything is held
        torch.cuda.synchronize()
        if held:
            hold.start(held, shape=shape, max_us=12_000_000)
        t0 = time.time()
        clean, timer = measure(sched)
        busy_s = time.time() - t0
        info = hold.stop() if held else {"cus_held": 0}
        per, kern = per_launch(timer)
        cell = {"held_workgroups": held, "hold_shape": shape if held else "-", "hold": info, "policy": name,
                "schedule": sched, "backward_ms": round(clean, 3), "conv_launch_ms_sum": round(sum(v for k, v in per.items() if not k.startswith("hbm:")), 3),
                "launch_ms": per, "plan": kern, "wall_s": round(time.time() - t0, 2)}
        if held and (info["held_ms_min"] < busy_s * 1e3 * 0.95 or info["held_ms_max"] > 11900):
            cell["warning"] = "the hold kernel did not cover exactly the measurement (left early, or ran into its time limit)"
        cells.append(cell)
        print(f"held {held:3d} {cell['hold_shape']:9s} ({info['cus_held']:3d} CUs, {info.get('held_ms_min', 0):7.1f} ms"
              f"{', %.0f GB/s streamed' % info['streamed_GBps'] if 'streamed_GBps' in info else ''})  {name:18s}  backward {clean:8.2f} ms   "
              f"convs {sum(per.values()):8.2f} ms", flush=True)

# the policy each hold shape implies: per launch, the schedule with the smallest time summed over that shape's rows
eng.contended = False
unet_mod.CONTENDED_WGRAD.update(WGRAD_DEFAULT)
labels = sorted(k for k in cells[0]["launch_ms"] if not k.startswith("hbm:"))
policy, table = {}, {}
for lab in labels:
    table[lab] = {f"{c['policy']}@{c['held_workgroups']}{c['hold_shape'] if c['held_workgroups'] else ''}": c["launch_ms"][lab] for c in cells}
for shape in sorted({c["hold_shape"] for c in cells if c["held_workgroups"]}):
    policy[shape] = {}
    for lab in labels:
        tot = {}
        for c in cells:
            if c["held_workgroups"] and c["hold_shape"] == shape:
                tot[c["policy"]] = tot.get(c["policy"], 0.0) + c["launch_ms"][lab]
        best = min(tot, key=tot.get)
        policy[shape][lab] = {"best": best, "gain_vs_auto_pct": round(100 * (tot["auto"] - tot[best]) / tot["auto"], 2)}
free = {c["policy"]: c["backward_ms"] for c in cells if not c["held_workgroups"]}
summary = [{"held": c["held_workgroups"], "shape": c["hold_shape"], "policy": c["policy"], "backward_ms": c["backward_ms"],
            "vs_free_auto": round(c["backward_ms"] / free["auto"], 4)} for c in cells] if "auto" in free else []
out = {"what": __doc__.split("\n\n")[0], "shape": {"batch": B, "channels": C, "frames": L},
       "hold_kernels": {"rccl": {"threads": 256, "vgprs": 113, "lds_bytes": 32768}, "rccl+mem": {"threads": 256, "vgprs": 93, "lds_bytes": 32768,
                        "stream": "16 B per lane read + write"}, "fat": {"threads": 256, "vgprs": 194, "lds_bytes": 32768},
                        "source": "tools/spin/spin.hip"},
       "device": torch.cuda.get_device_name(0), "summary": summary, "policy": policy, "by_launch": table, "cells": cells}
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(out, open(a.out, "w"), indent=1)
print("policy:", json.dumps(policy))
