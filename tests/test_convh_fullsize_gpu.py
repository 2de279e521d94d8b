"""GPU: the bf16-RESIDENT forward kernels (csrc/conv_h3.hip, BASELINE configs[4]) pinned at the REAL geometry -- batch 64, C = 1024,
256 frames, every one of the eight layers (the small-size oracle comparisons live in tests/test_convh_gpu.py):

  1. each layer alone, under the three work-split schedules: a hundred output values recomputed on the host in float64 from
     the SAME bf16 operands (products of bf16 values are exact in fp32, so only the summation order differs: 2e-5 of max-abs),
     the padded edges included, and the bf16 output copies equal to the fp32 result activated and rounded once;
  2. the engine's resident forward as a CHAIN: every conv is checked against float64 spot values computed from the bf16 input
     tensor the DEVICE fed it (the previous layers' own outputs) and the bf16 shadow of its weight -- so an error in any
     layer's kernel at this geometry (window slots depend on B and L) shows in that layer, not as a 2e-2 whole-network drift.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_ref  # noqa: F401  (disables oneDNN, see oracle/unet_ref.py)

pytestmark = pytest.mark.gpu
C, L, B = 1024, 256, 64
#          name  transposed Cin    Cout   k   s  p   Lin
LAYERS = [("D0", False, C, 2 * C, 32, 2, 16, 256), ("D1", False, 2 * C, 2 * C, 8, 1, 2, 129),
          ("D2", False, 2 * C, 2 * C, 8, 2, 1, 126), ("D3", False, 2 * C, 4 * C, 4, 2, 1, 61),
          ("U3", True, 4 * C, 2 * C, 5, 2, 1, 30), ("U2", True, 4 * C, 2 * C, 8, 2, 1, 61),
          ("U1", True, 4 * C, 2 * C, 8, 1, 2, 126), ("U0", True, 4 * C, 2 * C, 32, 2, 16, 129)]


def spot_fwd(xh, wh, tr, k, s, p, Lin, b, o, t):
    """y[b, o, t] of conv / conv_transpose in float64 from host arrays xh (B, Cin, >= Lin) and wh (weight layout)."""
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
    return acc


def sample_points(rng, n, Cout, Lout):
    edge = [0, 1, Lout - 1, Lout - 2, Lout // 2]
    pts = []
    for i in range(n):
        t = edge[i % 5] if i < 25 else int(rng.integers(Lout))
        pts.append((int(rng.integers(B)), int(rng.integers(Cout)), t))
    # the corners of the problem: first / last sample, first / last channel
    pts += [(0, 0, 0), (B - 1, Cout - 1, Lout - 1), (0, Cout - 1, 0), (B - 1, 0, Lout - 1)]
    return pts


@pytest.mark.parametrize("sched", [0, 1, 2], ids=["auto", "tile-per-wg", "stream-k"])
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_h_layer_at_full_size(layer, sched):
    from phasegen import ops
    name, tr, Cin, Cout, k, s, p, Lin = layer
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    g = torch.Generator(device="cuda").manual_seed(sum(map(ord, name)) + 1)
    x = torch.randn(B, Cin, Lin, device="cuda", generator=g).to(torch.bfloat16).float()
    w = (torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), device="cuda", generator=g) * 0.02).to(torch.bfloat16).float()
    xh = ops.h_alloc(B, Cin, Lin, "cuda")
    ops.cast_rows_bf16(x, xh)                                   # exact: x is bf16-representable
    wsh = ops.shadow_weights(w, tr, s)                          # exact cast (+ zero taps for k = 5)
    assert ops.conv_fwd_h_supported(B, tuple(w.shape), Lin, s, p, tr)
    y = torch.full((B, Cout, Lout), float("nan"), device="cuda")
    yh, yh2 = ops.h_alloc(B, Cout, Lout, "cuda"), ops.h_alloc(B, Cout, Lout, "cuda")
    ops.conv_fwd_h(xh, Lin, wsh, tuple(w.shape), s, p, transposed=tr, y=y, yh=yh, yh_act=ops.ACT_LEAKY, yh2=yh2, yh2_act=ops.ACT_RELU,
                   schedule=sched)
    assert bool(torch.isfinite(y).all())
    ymax = float(y.abs().max())
    xc, wc, yc = x.cpu().double().numpy(), w.cpu().double().numpy(), y.cpu().numpy()
    rng = np.random.default_rng(5)
    for (b, o, t) in sample_points(rng, 100, Cout, Lout):
        want = spot_fwd(xc, wc, tr, k, s, p, Lin, b, o, t)
        assert abs(yc[b, o, t] - want) < 2e-5 * ymax, (name, sched, b, o, t, yc[b, o, t], want)
    # bf16 copies: the fp32 result, activated, rounded once (RNE); row tails untouched (zero)
    assert torch.equal(yh[:, :, :Lout], F.leaky_relu(y, 0.2).to(torch.bfloat16))
    assert torch.equal(yh2[:, :, :Lout], torch.relu(y).to(torch.bfloat16))
    assert float(yh[:, :, Lout:].abs().max()) == 0.0 and float(yh2[:, :, Lout:].abs().max()) == 0.0


def test_resident_forward_chain_at_full_size():
    """Engine forward on the bf16-resident kernels at B = 64, C = 1024, L = 256: every conv against float64 spot values of ITS
    OWN device inputs (bf16 activations as the previous layers wrote them, bf16 weight shadows)."""
    from phasegen import ops
    from phasegen.model import UNetModel
    from phasegen.unet import LAYERS as ENG_LAYERS
    torch.manual_seed(21)
    m = UNetModel(C, 2 * C, precision="bf16")
    eng = m.engine
    g = torch.Generator(device="cuda").manual_seed(22)
    x = torch.log1p(torch.randn(B, C, L, device="cuda", generator=g).abs() * 3)
    assert eng.resident_ok(B, L)
    out = eng.forward(x, update_stats=False, inference=True)
    assert bool(torch.isfinite(out).all())
    f = eng.plans[("h", B, L)]["fwd"]
    h = 2 * C
    # layer -> (bf16 input tensor, frames, fp32 raw output or None, bf16 output + the activation it was stored with)
    io = {"D0": (f["x0"], 256, None, (f["l0"], 0.2)), "D1": (f["l0"], 129, f["c1"], None), "D2": (f["l1"], 126, f["c2"], None),
          "D3": (f["l2"], 61, None, (f["d3"], 0.0)), "U3": (f["d3"], 30, f["r3"], None), "U2": (f["cat2"], 61, f["r2"], None),
          "U1": (f["cat1"], 126, f["r1"], None), "U0": (f["cat0"], 129, f["r0"], None)}
    rng = np.random.default_rng(9)
    for name, tr, Cin, Cout, k, s, p, Lin in LAYERS:
        xin, Lx, raw, hout = io[name]
        assert Lx == Lin and xin.shape[1] == Cin
        key = ENG_LAYERS[name][0]
        wc = eng.arena.p(key).to(torch.bfloat16).cpu().double().numpy()       # the shadow is the RNE cast of the master weight
        xc = xin.cpu().double().numpy()
        Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
        if raw is not None:
            got, ymax = raw.cpu().numpy(), float(raw.abs().max())
        else:
            got, ymax = hout[0][:, :, :Lout].float().cpu().numpy(), float(hout[0].float().abs().max())
        for (b, o, t) in sample_points(rng, 40, Cout, Lout):
            want = spot_fwd(xc, wc, tr, k, s, p, Lin, b, o, t)
            if raw is not None:
                assert abs(got[b, o, t] - want) < 2e-5 * ymax, (name, b, o, t, got[b, o, t], want)
            else:                       # only the activated bf16 copy exists: one bf16 rounding (2^-9 relative) on top
                wa = max(want, hout[1] * want)
                assert abs(got[b, o, t] - wa) < 2 ** -8 * abs(wa) + 2e-5 * ymax, (name, b, o, t, got[b, o, t], wa)
