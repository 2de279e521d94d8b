"""GPU parity at the FULL BASELINE sizes (C = 1024, frames 256, batch 64) through size-independent checks -- the oracle
cannot run a batch-64 step in test time, so every conv layer of the U-Net is checked at its real geometry by

  1. float64 spot values: a hundred output elements of fwd, dgrad and wgrad recomputed on the host DIRECTLY from the
     inputs (dot products of up to 65 536 terms) -- exact parity on sampled positions, incl. the padded edges;
  2. adjoint identities that tie the three passes together without any reference:
        <conv(x, w), dy> == <x_act, dgrad(dy, w)> == <w, wgrad(x_act, dy)>      (to fp32 rounding);
  3. the U-Net training step itself: two identical steps are bit-identical (fixed accumulation order everywhere) and a
     step with every workgroup schedule forced (one tile per workgroup / im2col kernels under stream-K) agrees to rounding.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
C, L, B = 1024, 256, 64
#          name  transposed Cin    Cout   k   s  p   Lin  act
LAYERS = [("D0", False, C, 2 * C, 32, 2, 16, 256, 0), ("D1", False, 2 * C, 2 * C, 8, 1, 2, 129, 1),
          ("D2", False, 2 * C, 2 * C, 8, 2, 1, 126, 1), ("D3", False, 2 * C, 4 * C, 4, 2, 1, 61, 1),
          ("U3", True, 4 * C, 2 * C, 5, 2, 1, 30, 2), ("U2", True, 4 * C, 2 * C, 8, 2, 1, 61, 2),
          ("U1", True, 4 * C, 2 * C, 8, 1, 2, 126, 2), ("U0", True, 4 * C, 2 * C, 32, 2, 16, 129, 2)]


@pytest.fixture(params=[0, 1, 6, 0x2000, 0x4000], ids=["auto", "tile-per-wg", "im2col+stream-k", "raw-2-waves-per-simd", "raw-1-wave-per-simd-everywhere"])
def schedule(request):
    from phasegen import ops
    ops.set_conv_schedule(request.param)
    yield request.param
    ops.set_conv_schedule(0)


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_layer_at_full_size(layer, schedule):
    check_layer(layer, bf16=False)


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_layer_at_full_size_bf16_operands(layer):
    """The same checks with pg_conv_args.precision = PG_PREC_BF16.  x, w and dy are made bf16-representable, so the operand rounding
    only acts on LeakyReLU outputs (0.2 x is not representable) and the host recomputation -- on the rounded activated
    operand -- is exact for the raw-window kernels and for the fp32 fallback passes alike."""
    from phasegen import ops
    ops.set_conv_precision("bf16")
    try:
        check_layer(layer, bf16=True)
    finally:
        ops.set_conv_precision("fp32")


@pytest.mark.parametrize("layer", [LAYERS[0], LAYERS[2], LAYERS[4], LAYERS[7]], ids=["D0", "D2", "U3", "U0"])
def test_conv_layer_at_full_size_bf16x3_split(layer):
    """pg_conv_args.precision = PG_PREC_BF16X3 at the real geometry: the same float64 spot values (UNROUNDED operands, 1e-4 of max-abs)
    and adjoint identities as the fp32 path."""
    from phasegen import ops
    ops.set_conv_precision("bf16x3")
    try:
        check_layer(layer, bf16=False)
    finally:
        ops.set_conv_precision("fp32")


# the reference's own batch (train.py:14-15: 16 x 128 frames): 16 x 65 = 1040 columns = 4 full 256-wide tiles + a 16-column tail launch
LAYERS_REF_DEFAULT = [("D0", False, C, 2 * C, 32, 2, 16, 128, 0), ("D1", False, 2 * C, 2 * C, 8, 1, 2, 65, 1),
                      ("U1", True, 4 * C, 2 * C, 8, 1, 2, 62, 2), ("U0", True, 4 * C, 2 * C, 32, 2, 16, 65, 2)]


@pytest.mark.parametrize("layer", LAYERS_REF_DEFAULT, ids=[l[0] for l in LAYERS_REF_DEFAULT])
@pytest.mark.parametrize("sched", [0, 0x20000], ids=["auto(column tail launch)", "no-column-split"])
def test_conv_layer_at_the_reference_default_batch(layer, sched):
    """The four layers whose forward or dgrad has 1040 columns at batch 16 x 128 frames, with the tail launch (automatic) and without."""
    from phasegen import ops
    ops.set_conv_schedule(sched)
    try:
        check_layer(layer, bf16=False, B=16)
    finally:
        ops.set_conv_schedule(0)


def check_layer(layer, bf16, B=B):
    from phasegen import ops
    name, tr, Cin, Cout, k, s, p, Lin, act = layer
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    g = torch.Generator(device="cuda").manual_seed(sum(map(ord, name)))
    rb = (lambda t: t.to(torch.bfloat16).float()) if bf16 else (lambda t: t)
    x = rb(torch.randn(B, Cin, Lin, device="cuda", generator=g))
    w = rb(torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda", generator=g) * 0.02)
    dy = rb(torch.randn(B, Cout, Lout, device="cuda", generator=g))
    y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
    ops.conv_fwd(x, w, y, s, p, x_act=act, transposed=tr)
    ops.conv_dgrad(dy, w, dx, s, p, transposed=tr)
    ops.conv_wgrad(x, dy, dw, s, p, x_act=act, transposed=tr)
    xa = rb(torch.nn.functional.leaky_relu(x, 0.2) if act == 1 else (torch.relu(x) if act == 2 else x))
    # --- adjoint identities (float64 reductions of the device tensors)
    d1 = float((y.double() * dy.double()).sum())
    d2 = float((xa.double() * dx.double()).sum())
    d3 = float((w.double() * dw.double()).sum())
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(d1 - d2) < 2e-6 * scale and abs(d1 - d3) < 2e-6 * scale, (name, d1, d2, d3)
    # --- float64 spot values straight from the inputs
    rng = np.random.default_rng(7)
    xh, wh, dyh = xa.cpu().double().numpy(), w.cpu().double().numpy(), dy.cpu().double().numpy()
    yh, dxh, dwh = y.cpu().numpy(), dx.cpu().numpy(), dw.cpu().numpy()
    ymax, dxmax, dwmax = np.abs(yh).max(), np.abs(dxh).max(), np.abs(dwh).max()
    edge_t = [0, 1, Lout - 1, Lout // 2]
    for n in range(48):                                         # y[b,o,t]
        b, o = int(rng.integers(B)), int(rng.integers(Cout))
        t = edge_t[n % 4] if n < 16 else int(rng.integers(Lout))
        if n % 6 == 5:                                          # the last columns of the flattened (sample, frame) axis: the column tail launch
            b, t = B - 1, Lout - 1 - int(rng.integers(min(Lout, 64)))
        acc = 0.0
        for j in range(k):
            if tr:
                if (t + p - j) % s:
                    continue
                i = (t + p - j) // s
                if 0 <= i < Lin:
                    acc += float(wh[:, o, j] @ xh[b, :, i])
            else:
                i = s * t + j - p
                if 0 <= i < Lin:
                    acc += float(wh[o, :, j] @ xh[b, :, i])
        assert abs(yh[b, o, t] - acc) < 1e-4 * ymax, (name, "fwd", b, o, t, yh[b, o, t], acc)
    for n in range(32):                                         # dx[b,c,i] (grad wrt the activated operand)
        b, c = int(rng.integers(B)), int(rng.integers(Cin))
        i = [0, Lin - 1][n % 2] if n < 8 else int(rng.integers(Lin))
        if n % 6 == 5:
            b, i = B - 1, Lin - 1 - int(rng.integers(min(Lin, 64)))
        acc = 0.0
        for j in range(k):
            if tr:
                t = s * i + j - p
                if 0 <= t < Lout:
                    acc += float(wh[c, :, j] @ dyh[b, :, t])
            else:
                if (i + p - j) % s:
                    continue
                t = (i + p - j) // s
                if 0 <= t < Lout:
                    acc += float(wh[:, c, j] @ dyh[b, :, t])
        assert abs(dxh[b, c, i] - acc) < 1e-4 * dxmax, (name, "dgrad", b, c, i, dxh[b, c, i], acc)
    for n in range(24):                                         # dw: conv (o,c,j) / convT (c,o,j)
        o, c = int(rng.integers(Cout)), int(rng.integers(Cin))
        j = [0, k - 1][n % 2] if n < 8 else int(rng.integers(k))
        acc = 0.0
        if tr:
            for i in range(Lin):
                t = s * i + j - p
                if 0 <= t < Lout:
                    acc += float(xh[:, c, i] @ dyh[:, o, t])
            got = dwh[c, o, j]
        else:
            for t in range(Lout):
                i = s * t + j - p
                if 0 <= i < Lin:
                    acc += float(xh[:, c, i] @ dyh[:, o, t])
            got = dwh[o, c, j]
        assert abs(got - acc) < 1e-4 * dwmax, (name, "wgrad", o, c, j, got, acc)


def test_full_size_step_is_reproducible_and_schedule_independent():
    from phasegen import ops
    from phasegen.model import UNetModel
    from phasegen.trainer import Trainer
    g = torch.Generator(device="cuda").manual_seed(3)
    batch = torch.stack([torch.rand(B, C, L, device="cuda", generator=g) * 3,
                         (torch.rand(B, C, L, device="cuda", generator=g) * 2 - 1) * torch.pi], dim=1).contiguous()
    results = []
    for mode in (0, 0, 1, 6):              # automatic twice, one tile per workgroup, im2col kernels under stream-K
        torch.manual_seed(11)
        m = UNetModel(C, 2 * C)
        tr = Trainer(m)
        ops.set_conv_schedule(mode)
        losses = tr.step(batch).clone()
        ops.set_conv_schedule(0)
        results.append((losses, m.engine.arena.grad.clone()))
        del tr, m
        torch.cuda.empty_cache()
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])   # bit-reproducible
    # Other schedules only change the ORDER of fp32 summation.  Losses agree to 1e-5; the gradient of a ReLU/LeakyReLU
    # network is not continuous in such perturbations (a pre-activation within rounding of 0 flips its mask, and six
    # batch-norms amplify it): measured 4e-3 of the gradient norm at this size, while every layer taken alone matches
    # float64 spot values to 1e-4 under every schedule (test above).  Bound it loosely; a real indexing bug is O(1).
    # (The tight check of full-width gradients lives in tests/test_fullwidth_gpu.py: against the oracle with the device's
    # ReLU sign patterns frozen, where the same gradients agree to 8.4e-6.)
    for losses, grad in results[2:]:
        assert torch.allclose(losses, results[0][0], rtol=1e-5)
        num = float((grad.double() - results[0][1].double()).norm())
        den = float(results[0][1].double().norm())
        assert num < 2e-2 * den, (num, den)
    assert torch.isfinite(results[0][0]).all() and 0.5 < float(results[0][0][0]) < 10


def test_full_size_step_bf16_operands_tracks_fp32():
    """Config 5's arithmetic (bf16 MFMA operands, fp32 accumulate, fp32 master weights) on the full-size step: losses
    within 1 % of the fp32 step; gradient within 25 % of its norm (measured 11 %: at this random initialisation the gradient is
    so sensitive to ReLU-mask flips that re-ordering the fp32 sums alone moves it by 0.4 %, see the test above, and operand
    rounding is a 2^-9 perturbation per product).  Not the 1e-4 parity path; every layer taken alone is exact to 1e-4
    against the rounded-operand recomputation (test_conv_layer_at_full_size_bf16_operands)."""
    from phasegen import ops
    from phasegen.model import UNetModel
    from phasegen.trainer import Trainer
    g = torch.Generator(device="cuda").manual_seed(3)
    batch = torch.stack([torch.rand(16, C, L, device="cuda", generator=g) * 3,
                         (torch.rand(16, C, L, device="cuda", generator=g) * 2 - 1) * torch.pi], dim=1).contiguous()
    res = []
    for prec in ("fp32", "bf16"):
        torch.manual_seed(11)
        m = UNetModel(C, 2 * C)
        tr = Trainer(m)
        ops.set_conv_precision(prec)
        try:
            losses = tr.step(batch).clone()
        finally:
            ops.set_conv_precision("fp32")
        res.append((losses, m.engine.arena.grad.clone()))
        del tr, m
        torch.cuda.empty_cache()
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-2), (res[0][0], res[1][0])
    num = float((res[0][1].double() - res[1][1].double()).norm()), float(res[0][1].double().norm())
    assert num[0] < 0.25 * num[1], num
