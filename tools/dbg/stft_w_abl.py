#!/usr/bin/env python3
"""dev tool: stft_w_kernel at the e2e shape on variant libraries (tools/abl/build_one.sh w<n> stft -DPG_W_ABL=<n>), one process each."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
    import torch
    from phasegen import _lib
    if sys.argv[2] != "product":
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl", f"libphasegen_{sys.argv[2]}.so")
    from phasegen import ops
    nsig, frames, n_fft, hop = 64, 256, 2048, 512
    y = torch.randn(nsig, hop * (frames - 1), device="cuda") * 0.1
    out = torch.empty(nsig, 2, n_fft // 2, frames, device="cuda")
    for polar in (False, True):
        fn = lambda: ops.stft(y, n_fft, hop, polar=polar, out=out)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{sys.argv[2]:10s} stft polar={int(polar)}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
else:
    for name in ["product"] + sys.argv[1:]:
        subprocess.call([sys.executable, os.path.abspath(__file__), "--one", name])
