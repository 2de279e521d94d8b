"""dev tool (not product): hold part of the chip with a collective-shaped kernel while something else is measured.

    h = Hold(); h.start(32); ...launch and time the work on the current stream...; info = h.stop()

The kernel (tools/spin/spin.hip, pg_dev_hold): `blocks` workgroups of 256 threads, resident until the host raises a flag in coherent
host memory or `max_us` have passed.  shape "rccl": 113 VGPRs (<= 128: four such waves fit a SIMD) and `lds` bytes of LDS, ALU spin;
"rccl+mem": the same and every workgroup streams memory (read + write) the whole time; "fat": 194 VGPRs -- no one-wave-per-SIMD
conv workgroup (372 registers) fits beside it.  stop() returns what the kernel recorded: distinct CUs it sat on, how long its
workgroups stayed (must cover the measurement), bytes streamed.
Used by tools/contention.py and bench.py's dp_equivalent_held32 leg."""
import ctypes
import os
import subprocess
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "spin", "spin.hip")
LIB = os.path.join(HERE, "spin", "libspin.so")
SHAPES = {"rccl": 0, "rccl+mem": 1, "fat": 2}


def build():
    """hipcc tools/spin/spin.hip -> tools/spin/libspin.so (cross-compiles without a GPU; called by __graft_entry__.build())."""
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", SRC, "-o", LIB])
    return LIB


class Hold:
    def __init__(self, device=None, traffic_bytes=1 << 30):
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} missing: run __graft_entry__.build() (or tools/hold.py's build())")
        lib = self.lib = ctypes.CDLL(LIB)
        lib.pg_dev_hold_ctl.restype = ctypes.c_void_p
        lib.pg_dev_hold_reset.argtypes = [ctypes.c_void_p]
        lib.pg_dev_hold_release.argtypes = [ctypes.c_void_p]
        lib.pg_dev_hold_started.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.pg_dev_hold_records.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.pg_dev_hold.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p,
                                    ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            self.ctl = lib.pg_dev_hold_ctl()
        if not self.ctl:
            raise RuntimeError("hold: hipHostMalloc of the control block failed")
        self.sink = torch.zeros(4, device=self.device)
        # a HIGH-PRIORITY stream: HIP multiplexes streams onto a few hardware queues per priority class, and a normal-priority stream
        # of the measured program (the Trainer's Adam side stream) that lands in the hold kernel's queue waits behind it for seconds --
        # first seen as a 937 ms "step" that was really 4 s of queueing (bench_r04_c); another class, another queue
        self.side = torch.cuda.Stream(self.device, priority=-1)
        self.traffic_bytes = traffic_bytes
        self.buf = None
        self.blocks = 0
        self.shape = None

    def start(self, blocks, shape="rccl", lds=32 * 1024, max_us=3_000_000):
        """Launch the hold kernel on a side stream and return once all its workgroups are resident."""
        if not 0 < blocks <= 1024:
            raise ValueError("hold: 1..1024 workgroups")
        if shape == "rccl+mem" and self.buf is None:
            self.buf = torch.zeros(self.traffic_bytes, device=self.device, dtype=torch.uint8)
        torch.cuda.synchronize(self.device)
        self.lib.pg_dev_hold_reset(self.ctl)
        rc = self.lib.pg_dev_hold(blocks, SHAPES[shape], lds, self.ctl, max_us, self.buf.data_ptr() if self.buf is not None else None,
                                  self.traffic_bytes if self.buf is not None else 0, self.sink.data_ptr(), self.side.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"pg_dev_hold failed: {rc}")
        self.blocks, self.shape, self.t_start = blocks, shape, time.time()
        while self.lib.pg_dev_hold_started(self.ctl, blocks) != blocks:      # plain host reads of coherent memory: no stream involved
            if time.time() - self.t_start > 2.0:
                self.stop()
                raise RuntimeError("hold: workgroups did not become resident within 2 s")
            time.sleep(0.0005)

    def stop(self):
        """Raise the flag (a host store), wait for the hold kernel to drain; returns what it recorded."""
        if not self.blocks:
            return {"cus_held": 0}
        t_rel = time.time()
        self.lib.pg_dev_hold_release(self.ctl)
        self.side.synchronize()
        rec = np.zeros((self.blocks, 4), dtype=np.uint32)
        self.lib.pg_dev_hold_records(self.ctl, self.blocks, rec.ctypes.data_as(ctypes.c_void_p))
        hw, xcc = rec[:, 0], rec[:, 1] & 15
        cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        cus = len({(int(x), int(e), int(s), int(c)) for x, e, s, c in zip(xcc, se, sh, cu)})
        held_ms = rec[:, 2].astype(np.float64) / 1e5                        # 100 MHz ticks -> ms
        out = {"cus_held": cus, "workgroups": self.blocks, "shape": self.shape, "held_ms_min": float(held_ms.min()),
               "held_ms_max": float(held_ms.max()), "host_ms_start_to_release": (t_rel - self.t_start) * 1e3,
               "drain_ms": (time.time() - t_rel) * 1e3}
        if self.shape == "rccl+mem":      # one loop iteration = 16 x 256 lanes x 16 B read + written
            out["streamed_GBps"] = float((rec[:, 3].astype(np.float64) * 16 * 256 * 32).sum() / (held_ms.mean() * 1e-3) / 1e9)
        self.blocks = 0
        return out


if __name__ == "__main__":
    print(build())
