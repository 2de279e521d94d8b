"""CPU: the C-ABI library loads and exports every symbol include/phasegen.h declares (no compute calls without a
GPU); argument validation that happens before any launch; host-side planning logic."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "phasegen.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    from phasegen import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 21 and "pg_conv1d_fwd" in names and "pg_istft" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in phasegen.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "ctypes table and header disagree"
    assert lib.pg_version() == 400


def test_struct_sizes_match_the_header_layout():
    from phasegen import _lib
    assert ctypes.sizeof(_lib.ConvArgs) == 8 * 4 + 8 + 8 + 4 + 4 + 8 + 4 * 16 + 8 + 8 + 8 + 8 + 24 + 16 + 8   # 208
    assert _lib.ConvArgs.precision.offset == 52 and _lib.ConvArgs.schedule.offset == 148    # the two former pad words
    assert ctypes.sizeof(_lib.StftArgs) == 80 and ctypes.sizeof(_lib.IstftArgs) == 88
    assert ctypes.sizeof(_lib.MomentsArgs) == 40
    assert ctypes.sizeof(_lib.BnArgs) == 232 and _lib.BnArgs.num_batches_tracked.offset == 224     # 0.4: counter appended
    assert ctypes.sizeof(_lib.AdamArgs) == 8 + 4 * 8 + 5 * 8 + 8
    assert ctypes.sizeof(_lib.LossArgs) == 16 + 4 * 8 + 8 + 8


def test_argument_errors_are_reported_without_a_gpu():
    """Validation runs on the host before any launch, so bad calls fail identically here and on the GPU box."""
    from phasegen import _lib
    lib = _lib.load()
    a = _lib.ConvArgs()
    assert lib.pg_conv1d_fwd(ctypes.byref(a), None) == -2                 # PG_ERR_SHAPE: zero dims
    assert b"non-positive" in lib.pg_last_error_string()
    a.B, a.Cin, a.Cout, a.Lin, a.Lout, a.k, a.stride, a.pad = 1, 8, 16, 24, 99, 32, 2, 16
    assert lib.pg_conv1d_fwd(ctypes.byref(a), None) == -2                 # Lout inconsistent
    assert b"Lout" in lib.pg_last_error_string()
    a.Lout = 13
    assert lib.pg_conv1d_fwd(ctypes.byref(a), None) == -1                 # PG_ERR_NULL: no pointers
    s = _lib.StftArgs()
    s.n_signals, s.n_samples, s.n_fft, s.hop, s.n_frames = 1, 1000, 1000, 250, 5
    s.y = s.out = 1
    assert lib.pg_stft(ctypes.byref(s), None) == -4                       # PG_ERR_UNSUPPORTED: n_fft not a power of two
    ad = _lib.AdamArgs()
    ad.n, ad.p, ad.g, ad.m, ad.v, ad.step = 4, 16, 16, 16, 16, 0
    assert lib.pg_adam_step(ctypes.byref(ad), None) == -2                 # step is 1-based


def test_abi_keeps_no_process_wide_state():
    """'keeps no global mutable state => re-entrant and thread-safe per stream'.  Precision, schedule and the
    FFT schedule are struct fields; no setter is exported and no translation unit defines a mutable namespace-scope int."""
    from phasegen import _lib
    lib = _lib.load()
    for gone in ("pg_conv_set_precision", "pg_conv_set_schedule", "pg_conv_set_oversubscribe", "pg_stft_set_mode"):
        assert not hasattr(lib, gone), gone
    csrc = os.path.join(ROOT, "unet-phasegen_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            for ln in open(os.path.join(csrc, f)):
                assert not re.match(r"^(static\s+)?(int|bool|long|unsigned|float)\s+g_\w+\s*(=|;)", ln), (f, ln)
    assert "PHASEGEN_CONV_PRECISION" not in open(os.path.join(ROOT, "unet-phasegen_amd", "phasegen", "_lib.py")).read()


def _u0_args(lib, precision=0, schedule=0):
    from phasegen import _lib
    a = _lib.ConvArgs()
    a.B, a.Cin, a.Cout, a.Lin, a.Lout, a.k, a.stride, a.pad = 64, 4096, 2048, 129, 256, 32, 2, 16
    a.x = a.w = a.y = a.dy = a.dx = a.dw = 4096                      # never dereferenced: describe launches nothing
    a.x_bs = a.dx_bs = 4096 * 129
    a.y_bs = a.dy_bs = 2048 * 256
    a.workspace, a.workspace_bytes = 4096, lib.pg_workspace_bytes_conv()
    a.precision, a.schedule = precision, schedule
    return a


def test_per_call_knobs_are_validated_and_steer_the_launch_plan():
    """pg_conv_describe is a pure function of the arguments (runs without a GPU): the per-call precision / schedule fields
    pick the kernel, and bad values are refused with PG_ERR_*."""
    from phasegen import _lib, ops
    lib = _lib.load()
    d = ops.conv_describe(_u0_args(lib), _lib.OP_CONVT1D_FWD)
    # fp32 F / T: the one-wave-per-SIMD kernels (256 x 256 tiles, one workgroup per CU): 512 tiles = 2 whole tiles per CU
    assert d.startswith("conv_raw3_kernel<32, 2, true, false>|") and "grid=512|tiles=512" in d and "split=0" in d
    dc = ops.conv_describe(_u0_args(lib, schedule=_lib.SCHED_CONTENDED), _lib.OP_CONVT1D_FWD)
    assert "grid=1024" in dc and "split=1" in dc                       # data-parallel backward: keep the fine split
    dw = ops.conv_describe(_u0_args(lib), _lib.OP_CONVT1D_WGRAD)
    assert "grid=2048|tiles=8192" in dw and "split=0" in dw            # 4 whole tiles per workgroup: no fixup either
    # 64 x 65 = 4160 columns = 16 full 256-wide tiles + 64: the tail goes to a second launch of the tall-tile kernel (0.4), the 512 full
    # tiles run two whole tiles per CU; without the tail launch 528 tiles = 2 x 256 whole + 16 split over 256 more
    dd = ops.conv_describe(_u0_args(lib), _lib.OP_CONVT1D_DGRAD)
    assert dd.startswith("conv_raw3_kernel<32, 2, false, false>|grid=512|tiles=512|") and "split=0" in dd and dd.endswith("|tail=conv_raw_kernel<32, 2, false, 0, 1>,grid=512")
    dd = ops.conv_describe(_u0_args(lib, schedule=_lib.SCHED_NO_COLSPLIT), _lib.OP_CONVT1D_DGRAD)
    assert dd.startswith("conv_raw3_kernel<32, 2, false, false>|") and "grid=768|tiles=528" in dd and "split=1" in dd and "whole=512" in dd
    assert "whole=0" in ops.conv_describe(_u0_args(lib, schedule=_lib.SCHED_CONTENDED | _lib.SCHED_NO_COLSPLIT), _lib.OP_CONVT1D_DGRAD)
    # bit 13: the two-waves-per-SIMD raw kernels (128 x 256 tiles, two workgroups per CU) as before round 3
    d2 = ops.conv_describe(_u0_args(lib, schedule=_lib.SCHED_NO_RAW3), _lib.OP_CONVT1D_FWD)
    assert d2.startswith("conv_raw_kernel<32, 2, true, 0, 2>|") and "grid=1024|tiles=1024" in d2 and "split=0" in d2
    dd2 = ops.conv_describe(_u0_args(lib, schedule=_lib.SCHED_NO_RAW3), _lib.OP_CONVT1D_DGRAD)     # 1056 = 2 x 512 whole + 32 split over 512 more
    assert "grid=1536|tiles=1056" in dd2 and "split=1" in dd2 and "whole=1024" in dd2
    assert ops.conv_describe(_u0_args(lib, precision=1), _lib.OP_CONVT1D_FWD).startswith("conv_raw_kernel<32, 2, true, 1, 2>|")
    assert ops.conv_describe(_u0_args(lib, precision=2), _lib.OP_CONVT1D_WGRAD).startswith("conv_g_raw_kernel<32, 2, 2>|")
    assert ops.conv_describe(_u0_args(lib, schedule=_lib.SCHED_NO_RAW), _lib.OP_CONVT1D_DGRAD).startswith("conv_f_kernel<0, 0, 0>|")
    one = ops.conv_describe(_u0_args(lib, schedule=_lib.SCHED_TILE_PER_WG), _lib.OP_CONVT1D_FWD)
    assert "split=0" in one and "grid=512|tiles=512" in one
    g4 = ops.conv_describe(_u0_args(lib, schedule=_lib.SCHED_FORCE_STREAMK | _lib.SCHED_NO_COLSPLIT), _lib.OP_CONVT1D_DGRAD)
    g1 = ops.conv_describe(_u0_args(lib, schedule=_lib.SCHED_FORCE_STREAMK | _lib.SCHED_NO_COLSPLIT | (1 << 8)), _lib.OP_CONVT1D_DGRAD)   # oversubscribe factor 1
    assert "grid=1024" in g4 and "grid=256" in g1
    buf = ctypes.create_string_buffer(256)
    assert lib.pg_conv_describe(ctypes.byref(_u0_args(lib, precision=3)), 3, buf, 256) == -4      # PG_ERR_UNSUPPORTED
    assert lib.pg_conv_describe(ctypes.byref(_u0_args(lib, schedule=3)), 3, buf, 256) == -2       # PG_ERR_SHAPE
    assert lib.pg_conv_describe(ctypes.byref(_u0_args(lib, schedule=9 << 8)), 3, buf, 256) == -2
    assert lib.pg_conv_describe(ctypes.byref(_u0_args(lib)), 7, buf, 256) == -4
    assert lib.pg_conv1d_fwd(ctypes.byref(_u0_args(lib, precision=5)), None) != 0


def test_resident_geometry_query_runs_on_the_host():
    """pg_conv_fwd_h_supported: the window-fit rule of the bf16-resident kernels, answered without pointers or a GPU."""
    from phasegen import detgen, ops
    shp = detgen.conv_shapes(1024)
    assert ops.conv_fwd_h_supported(64, shp[detgen.K_D0], 256, 2, 16, False)
    assert ops.conv_fwd_h_supported(64, shp[detgen.K_U3], 30, 2, 1, True)          # k = 5 through the 4-taps-per-phase shadow
    assert not ops.conv_fwd_h_supported(64, shp[detgen.K_D0], 24, 2, 16, False)    # 20 samples per tile: windows do not fit
    assert not ops.conv_fwd_h_supported(4, (30, 22, 8), 64, 2, 1, False)           # 32-deep slab = 4 channels x 8 taps: Cin % 4


def test_thread_defaults_are_thread_local():
    """ops.set_conv_precision / set_conv_schedule are per-THREAD Python defaults that are passed per call (pg_conv_args)."""
    import threading
    from phasegen import ops
    ops.set_conv_precision("bf16")
    ops.set_conv_schedule(4)
    seen = {}

    def other():
        seen["p"], seen["s"] = ops._tls.precision, ops._tls.schedule
    t = threading.Thread(target=other)
    t.start()
    t.join()
    try:
        assert (ops._tls.precision, ops._tls.schedule) == (1, 4) and seen == {"p": 0, "s": 0}
        with ops.conv_options(precision="fp32", schedule=0):
            assert (ops._tls.precision, ops._tls.schedule) == (0, 0)
        assert (ops._tls.precision, ops._tls.schedule) == (1, 4)
    finally:
        ops.set_conv_precision("fp32")
        ops.set_conv_schedule(0)


def test_frame_plan_and_arena_layout():
    from phasegen import detgen
    from phasegen.unet import ALIGN, frame_plan
    assert frame_plan(128) == (65, 62, 29, 14) and frame_plan(256) == (129, 126, 61, 30)
    for L in range(2, 300):
        ok = (L % 8 == 0 and L >= 24)
        if ok:
            frame_plan(L)
        else:
            with pytest.raises(ValueError):
                frame_plan(L)
    # arena: reference parameter order, every parameter ALIGN-float aligned, layers contiguous for bucketing
    shapes = detgen.conv_shapes(1024)
    total = sum(int.__mul__(*shp[:2]) * shp[2] for shp in shapes.values()) + 12 * 2048
    assert total == 612392960                                              # 612 392 960 parameters
    assert ALIGN % 4 == 0


def test_engine_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from phasegen.model import UNetModel
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        UNetModel(8, 16)


def test_no_kernel_in_the_library_uses_scratch():
    """the generic (k, s) fallback conv_t_kernel<0, 0, *> carried 296 B of scratch (73 spilled VGPRs) and
    conv_g_raw_kernel<4, 2, 1> 2 spills.  Read every gfx950 code object out of the shipped .so and require, from the kernel
    descriptors' notes, .private_segment_fixed_size == 0 and .vgpr_spill_count == 0 for every kernel (no GPU needed)."""
    import shutil
    import subprocess
    import tempfile
    from phasegen import _lib
    tools = "/opt/rocm/lib/llvm/bin"
    objdump, readelf = os.path.join(tools, "llvm-objdump"), os.path.join(tools, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        import pytest
        pytest.skip("ROCm's llvm-objdump / llvm-readelf not installed")
    with tempfile.TemporaryDirectory() as td:
        so = shutil.copy(_lib.LIB_PATH, td)                       # --offloading unbundles NEXT TO its input: never in the tree
        subprocess.run([objdump, "--offloading", so], cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [f for f in sorted(os.listdir(td)) if f.endswith("gfx950")]
        assert len(cos) >= 9                                      # one code object per translation unit
        kernels, bad = 0, []
        for f in cos:
            notes = subprocess.run([readelf, "--notes", os.path.join(td, f)], check=True, capture_output=True, text=True).stdout
            name = None
            for ln in notes.splitlines():
                ln = ln.strip()
                if ln.startswith(".name:"):                   # (kernel ARGUMENTS carry .name: lines too: remember, do not count)
                    name = ln.split(":", 1)[1].strip()
                elif ln.startswith(".symbol:") and ln.endswith(".kd"):
                    kernels += 1                              # one kernel descriptor per kernel
                elif ln.startswith((".private_segment_fixed_size:", ".vgpr_spill_count:", ".sgpr_spill_count:")):
                    k, v = ln.split(":")
                    if int(v) != 0 and not k.startswith(".sgpr"):  # SGPR spills go to VGPR lanes, not to memory
                        bad.append((name, k, int(v)))
        assert kernels >= 100 and not bad, bad
