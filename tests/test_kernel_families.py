"""CPU: which kernel family does the automatic choice (schedule = 0) reach, for every op x (k, stride) x size class the library serves?

pg_conv_describe / pg_conv_fwd_h_describe are pure functions of the call's arguments, so the whole map is computed without a GPU.  The
test pins the SET of families the automatic choice can reach -- a family nothing reaches is dead weight (round 3 carried three such:
conv_g3, conv_h's 128 x 256 and conv_h2's eight-wave tiles; removed in round 4) -- and prints the table DESIGN.md section 4.1 holds."""
import ctypes
import re

import pytest


def _args(_lib, B, Cin, Cout, Lin, k, s, p, tr, precision=0, schedule=0):
    a = _lib.ConvArgs()
    a.B, a.Cin, a.Cout, a.Lin, a.k, a.stride, a.pad = B, Cin, Cout, Lin, k, s, p
    a.Lout = (Lin - 1) * s - 2 * p + k if tr else (Lin + 2 * p - k) // s + 1
    a.x = a.w = a.y = a.dy = a.dx = a.dw = 4096                      # never dereferenced: describe launches nothing
    a.x_bs = a.dx_bs = Cin * Lin
    a.y_bs = a.dy_bs = Cout * a.Lout
    a.workspace, a.workspace_bytes = 4096, _lib.load().pg_workspace_bytes_conv()
    a.precision, a.schedule = precision, schedule
    return a


# the U-Net's layers at width C (name, transposed, Cin, Cout, k, s, p, frames in at L = 256 / 128)
def layers(C, L):
    from phasegen.unet import frame_plan
    L1, L2, L3, L4 = frame_plan(L)
    return [("D0", False, C, 2 * C, 32, 2, 16, L), ("D1", False, 2 * C, 2 * C, 8, 1, 2, L1), ("D2", False, 2 * C, 2 * C, 8, 2, 1, L2),
            ("D3", False, 2 * C, 4 * C, 4, 2, 1, L3), ("U3", True, 4 * C, 2 * C, 5, 2, 1, L4), ("U2", True, 4 * C, 2 * C, 8, 2, 1, L3),
            ("U1", True, 4 * C, 2 * C, 8, 1, 2, L2), ("U0", True, 4 * C, 2 * C, 32, 2, 16, L1)]


def family(desc):
    name = desc.split("|")[0]
    m = re.match(r"(conv_[a-z0-9_]+)_kernel<(.*)>", name)
    fam = m.group(1)
    if fam == "conv_raw":
        fam += "(tall 256x128)" if m.group(2).split(",")[-1].strip() == "1" else "(128x256)"
    return fam


SIZE_CLASSES = [("bench: batch 64 x 256 frames, C = 1024", 1024, 256, 64), ("configs[1]: batch 32 x 256", 1024, 256, 32),
                ("reference default: batch 16 x 128", 1024, 128, 16), ("demo clip: batch 1 x 128", 1024, 128, 1),
                ("1024-FFT variant: C = 512, batch 64 x 256", 512, 256, 64), ("goldens: C = 16, batch 3 x 24", 16, 24, 3),
                ("many short clips: C = 64, batch 64 x 24", 64, 24, 64)]


def sweep():
    from phasegen import _lib, ops
    ops_ = (("fwd", lambda tr: _lib.OP_CONVT1D_FWD if tr else _lib.OP_CONV1D_FWD), ("dgrad", lambda tr: _lib.OP_CONVT1D_DGRAD if tr else _lib.OP_CONV1D_DGRAD),
            ("wgrad", lambda tr: _lib.OP_CONVT1D_WGRAD if tr else _lib.OP_CONV1D_WGRAD))
    table = {}
    for label, C, L, B in SIZE_CLASSES:
        for name, tr, Cin, Cout, k, s, p, Lin in layers(C, L):
            for prec, pname in ((0, "fp32"), (1, "bf16"), (2, "bf16x3")):
                for opname, opf in ops_:
                    if name == "D0" and opname == "dgrad":
                        continue
                    d = ops.conv_describe(_args(_lib, B, Cin, Cout, Lin, k, s, p, tr, precision=prec), opf(tr))
                    table[(label, pname, name, opname)] = family(d)
    # generic geometry (not one of the network's five (k, s) pairs)
    for opname, opf in ops_:
        d = ops.conv_describe(_args(_lib, 4, 24, 40, 50, 7, 3, 2, False), opf(False))
        table[("generic (k, s) = (7, 3)", "fp32", "-", opname)] = family(d)
    return table


def test_every_family_is_reached_by_some_automatic_choice_and_only_those_exist():
    table = sweep()
    reached = set(table.values())
    assert reached == {"conv_raw3", "conv_raw(128x256)", "conv_raw(tall 256x128)", "conv_g_raw", "conv_g_ps", "conv_f", "conv_t", "conv_g"}, reached
    # who takes what at the bench shape, fp32 (the headline): every F / T layer on one wave per SIMD, wgrads on the per-sample-slab / flat-K kernels
    bench = "bench: batch 64 x 256 frames, C = 1024"
    assert all(table[(bench, "fp32", l, o)] == "conv_raw3" for l in ("D0", "D1", "D2", "D3", "U3", "U2", "U1", "U0") for o in ("fwd", "dgrad") if (l, o) != ("D0", "dgrad"))
    assert {table[(bench, "fp32", l, "wgrad")] for l in ("D0", "U0")} == {"conv_g_raw"} and {table[(bench, "fp32", l, "wgrad")] for l in ("D1", "D2", "D3", "U3", "U2", "U1")} == {"conv_g_ps"}
    # the bf16 operand modes stay on the two-waves-per-SIMD raw kernels; batch 1 takes the tall tile; generic geometry and windows that do not fit: im2col
    assert table[(bench, "bf16", "U0", "fwd")] == "conv_raw(128x256)" and table[("demo clip: batch 1 x 128", "fp32", "D1", "fwd")] == "conv_raw(tall 256x128)"
    assert table[("generic (k, s) = (7, 3)", "fp32", "-", "fwd")] == "conv_f" and table[("many short clips: C = 64, batch 64 x 24", "fp32", "D3", "fwd")] in ("conv_f", "conv_raw(tall 256x128)", "conv_raw3", "conv_raw(128x256)")
    # the table itself (pytest -s prints it; DESIGN.md section 4.1 holds a copy)
    rows = {}
    for (label, prec, layer, op), fam in sorted(table.items()):
        rows.setdefault((label, prec, fam), []).append(f"{layer}.{op}")
    for (label, prec, fam), items in sorted(rows.items()):
        print(f"{label:45s} {prec:7s} {fam:24s} {' '.join(items)}")


def test_the_resident_forward_has_one_family():
    from phasegen import _lib, ops
    a = _lib.ConvhArgs()
    a.B, a.Cin, a.Cout, a.Lin, a.k, a.stride, a.pad, a.transposed = 64, 1024, 2048, 256, 32, 2, 16, 0
    a.Lout = 129
    a.x_pitch = ops.h_pitch(256)
    a.x_bs = 1024 * a.x_pitch
    a.x = a.w = a.y = 4096
    a.y_bs = 2048 * 129
    a.workspace, a.workspace_bytes = 4096, _lib.load().pg_workspace_bytes_conv()
    assert ops.conv_fwd_h_describe(a).startswith("conv_h3_kernel<32, 2, false>|")
    a.schedule = 64
    buf = ctypes.create_string_buffer(256)
    assert _lib.load().pg_conv_fwd_h_describe(ctypes.byref(a), buf, 256) == _lib.ERR_UNSUPPORTED


def test_fixup_form_follows_segments_per_split_tile():
    """The wide fixup (four workgroups per 32 x 32 block, segments summed four abreast) is taken from 8 segments per split tile on --
    demo.py's single clip: 8-16 tiles over 512 workgroups -- and never at the bench shapes (2-3 segments per split tile): the order in
    which a tile's segments are added is a function of (grid, tiles) alone."""
    from phasegen import _lib, ops

    def fields(d):
        return dict(kv.split("=", 1) for kv in d.split("|")[1:])

    for name, tr, Cin, Cout, k, s, p, Lin in layers(1024, 128):
        f = fields(ops.conv_describe(_args(_lib, 1, Cin, Cout, Lin, k, s, p, tr), _lib.OP_CONVT1D_FWD if tr else _lib.OP_CONV1D_FWD))
        assert f["split"] == "1" and f["fixup"] == "wide" and int(f["grid"]) >= 8 * (int(f["tiles"]) - int(f["whole"])), (name, f)
    seen = set()
    for name, tr, Cin, Cout, k, s, p, Lin in layers(1024, 256):
        for op in ((_lib.OP_CONVT1D_FWD, _lib.OP_CONVT1D_DGRAD, _lib.OP_CONVT1D_WGRAD) if tr else (_lib.OP_CONV1D_FWD, _lib.OP_CONV1D_WGRAD)):
            f = fields(ops.conv_describe(_args(_lib, 64, Cin, Cout, Lin, k, s, p, tr), op))
            seen.add(f["fixup"])
    assert seen <= {"none", "plain"}, seen


def test_column_tail_launch_policy():
    """A conv_raw3 problem whose columns end <= 128 past a full 256-wide tile hands that tail to a second launch of the tall-tile
    kernel where the cost model prices the tail under the extra tile column: the bench shape's 64 x 129 = 8256 columns
    (full tiles: a whole number per CU), the reference's own 16 x 65 = 1040; never where the tail is half a tile (64 x 126 = 8064, 64 x 30),
    never under one-tile-per-workgroup, and PG_SCHED_NO_COLSPLIT turns it off."""
    from phasegen import _lib, ops
    plan = {}
    for name, tr, Cin, Cout, k, s, p, Lin in layers(1024, 256):
        for opn, op in (("fwd", _lib.OP_CONVT1D_FWD if tr else _lib.OP_CONV1D_FWD), ("dgrad", _lib.OP_CONVT1D_DGRAD if tr else _lib.OP_CONV1D_DGRAD)):
            if (name, opn) != ("D0", "dgrad"):
                plan[name + "." + opn] = ops.conv_describe(_args(_lib, 64, Cin, Cout, Lin, k, s, p, tr), op)
    tails = sorted(n for n, d in plan.items() if "|tail=" in d)
    # (D2.fwd / U2.dgrad -- 64 x 61 frames: 120 / 240 full tiles -- would split into unaligned ranges over 256 CUs: left whole)
    assert tails == ["D0.fwd", "D1.dgrad", "U0.dgrad", "U1.fwd"], tails
    for n in tails:
        assert plan[n].startswith("conv_raw3_kernel<") and "|tail=conv_raw_kernel<" in plan[n] and ", 0, 1>,grid=" in plan[n].split("|tail=")[1], plan[n]
    assert "|split=0|" in plan["D0.fwd"] and "|split=0|" in plan["U0.dgrad"]      # 256 / 512 full tiles: whole tiles per workgroup, no fixup
    d0 = ("D0", False, 1024, 2048, 32, 2, 16, 256)
    for sched in (_lib.SCHED_NO_COLSPLIT, _lib.SCHED_TILE_PER_WG, _lib.SCHED_NO_TALL):
        assert "|tail=" not in ops.conv_describe(_args(_lib, 64, *d0[2:4], d0[7], *d0[4:7], False, schedule=sched), _lib.OP_CONV1D_FWD), sched
    ref = ops.conv_describe(_args(_lib, 16, 1024, 2048, 128, 32, 2, 16, False), _lib.OP_CONV1D_FWD)
    assert ref.startswith("conv_raw3_kernel<32, 2, false, false>|grid=256|tiles=32|") and "|tail=conv_raw_kernel<32, 2, false, 0, 1>" in ref, ref
    lib = _lib.load()
    buf = ctypes.create_string_buffer(256)
    assert lib.pg_conv_describe(ctypes.byref(_args(_lib, 64, 1024, 2048, 256, 32, 2, 16, False, schedule=0x60000)), _lib.OP_CONV1D_FWD, buf, 256) == -2
