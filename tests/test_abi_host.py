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
    assert lib.pg_version() == 100


def test_struct_sizes_match_the_header_layout():
    from phasegen import _lib
    assert ctypes.sizeof(_lib.ConvArgs) == 8 * 4 + 8 + 8 + 4 + 4 + 8 + 4 * 16 + 8 + 8 + 8 + 8 + 24 + 16   # 208
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
