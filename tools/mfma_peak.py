#!/usr/bin/env python3
"""Dev tool: the MFMA rate this chip SUSTAINS on v_mfma_f32_32x32x16_bf16 -- (1) a register-only loop (no LDS, no memory: what power
and clocks leave of the 2516.6 TFLOP/s data-sheet figure), at one and two waves per SIMD; (2) the vendor GEMM (torch.matmul = hipBLASLt,
bf16 in, fp32 accumulate) at the GEMM shapes of the bench's conv layers -- the two yardsticks DESIGN.md section 4.4c prices the
bf16-resident conv kernels against.  Builds tools/dbg/mfma_peak.hip into /tmp at run time (hipcc is on the GPU box)."""
import ctypes, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch

so = os.path.join(tempfile.gettempdir(), "mfma_peak.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools/dbg/mfma_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.mfma_peak.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)] * 2
for random, what in ((0, "constant operands"), (1, "random operands in [-1, 1), 4 register sets in rotation")):
    for threads, label in ((256, "1 wave/SIMD"), (512, "2 waves/SIMD")):
        for iters in (20000, 200000):
            tf, mhz = ctypes.c_double(), ctypes.c_double()
            rc = lib.mfma_peak(256 * (2 if threads == 256 else 1) * 2, threads, iters, 5, random, ctypes.byref(tf), ctypes.byref(mhz))
            print(f"register-only MFMA loop, {what}, {label}, {iters} x 4 MFMAs per wave: rc={rc} {tf.value:7.1f} TFLOP/s ({tf.value / 2516.6 * 100:4.1f} % of 2516.6), shader clock {mhz.value:6.0f} MHz", flush=True)
lib.mfma_peak_f32.argtypes = lib.mfma_peak.argtypes
for random, what in ((0, "constant operands"), (1, "random operands in [-1, 1), 8 register sets in rotation")):
    for threads, label in ((256, "1 wave/SIMD"), (512, "2 waves/SIMD")):
        tf, mhz = ctypes.c_double(), ctypes.c_double()
        rc = lib.mfma_peak_f32(256 * (2 if threads == 256 else 1) * 2, threads, 100000, 5, random, ctypes.byref(tf), ctypes.byref(mhz))
        print(f"register-only v_mfma_f32_32x32x2_f32 loop, {what}, {label}: rc={rc} {tf.value:7.1f} TFLOP/s ({tf.value / 157.3 * 100:4.1f} % of 157.3), shader clock {mhz.value:6.0f} MHz", flush=True)
lib.mfma_peak_f32_16.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_double)] * 2
for random in (0, 1):
    tf, mhz = ctypes.c_double(), ctypes.c_double()
    rc = lib.mfma_peak_f32_16(256, 20000, 5, random, ctypes.byref(tf), ctypes.byref(mhz))
    print(f"register-only v_mfma_f32_32x32x2_f32 loop, 16 accumulators (256 registers, AGPRs), 1 wave/SIMD, {'random' if random else 'constant'} operands: rc={rc} {tf.value:7.1f} TFLOP/s ({tf.value / 157.3 * 100:4.1f} % of 157.3), shader clock {mhz.value:6.0f} MHz", flush=True)

import bench
from phasegen import ops
from phasegen.unet import LAYERS, frame_plan
C, L, B = 1024, 256, 64
L1, L2, L3, L4 = frame_plan(L)
geo = {"D0": (C, 2 * C, 32, L), "D1": (2 * C, 2 * C, 8, L1), "D2": (2 * C, 2 * C, 8, L2), "D3": (2 * C, 4 * C, 4, L3),
       "U3": (4 * C, 2 * C, 5, L4), "U2": (4 * C, 2 * C, 8, L3), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
def t_of(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, (Cin, Cout, k, Lin) in (geo.items() if len(sys.argv) > 1 and sys.argv[1] == "gemm" else ()):
    _, kind, s, p = LAYERS[name]
    tr = kind == "t"
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    # the layer as a dense GEMM with the same useful FLOPs: rows x K x columns (a transposed conv computes k / s taps per output)
    M, K, N = Cout, Cin * (k // s if tr else k) if not (tr and k == 5) else Cin * 5 // 2, B * Lout
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    bt = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    t_nn = t_of(lambda: torch.matmul(a, b)); t_nt = t_of(lambda: torch.matmul(a, bt.t()))
    fl = 2.0 * M * K * N
    print(f"{name}: GEMM {M} x {K} x {N} bf16 -> bf16: NN {t_nn:6.3f} ms {fl / t_nn / 1e9:6.0f} TF ({fl / t_nn / 1e9 / 2516.6 * 100:4.1f} %)   NT {t_nt:6.3f} ms {fl / t_nt / 1e9:6.0f} TF ({fl / t_nt / 1e9 / 2516.6 * 100:4.1f} %)", flush=True)
