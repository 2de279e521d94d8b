#!/usr/bin/env python3
"""dev tool: which HW_ID fields tell two co-resident workgroups of a CU apart?  (hold kernel, 512 workgroups of 256 threads)"""
import os, sys, ctypes, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from hold import Hold
h = Hold()
h.start(512, shape="rccl", lds=64 * 1024, max_us=200000)
import time; time.sleep(0.01)
blocks = h.blocks
h.lib.pg_dev_hold_release(h.ctl); h.side.synchronize()
rec = np.zeros((blocks, 4), dtype=np.uint32)
h.lib.pg_dev_hold_records(h.ctl, blocks, rec.ctypes.data_as(ctypes.c_void_p))
hw, xcc = rec[:, 0], rec[:, 1] & 15
per = collections.defaultdict(list)
for i in range(blocks):
    w = int(hw[i])
    per[(int(xcc[i]), (w >> 13) & 7, (w >> 12) & 1, (w >> 8) & 15)].append(((w >> 16) & 15, w & 15, (w >> 4) & 3, i))
print("CUs", len(per), "workgroups per CU", collections.Counter(len(v) for v in per.values()))
print("TG_ID sets", collections.Counter(tuple(sorted(t[0] for t in v)) for v in per.values()).most_common(8))
print("WAVE_ID sets (wave 0 of each workgroup)", collections.Counter(tuple(sorted(t[1] for t in v)) for v in per.values()).most_common(8))
print("pairs of workgroup indices on a CU (first 8)", [tuple(t[3] for t in v) for v in list(per.values())[:8]])
