"""GPU parity of every conv / BN / loss / Adam kernel against stock fp32 torch CPU ops (the oracle's building
blocks: F.conv1d, F.conv_transpose1d and their autograd), through the C ABI.  Tolerance: 1e-4 relative to the
tensor's max-abs (BASELINE.json: "within 1e-4 rel fp32"); observed errors are ~1e-6."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_ref  # noqa: F401  (disables oneDNN: see the bug note in oracle/unet_ref.py)
from phasegen import detgen

pytestmark = pytest.mark.gpu
TOL = 1e-4


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def act_cpu(x, act):
    return F.leaky_relu(x, 0.2) if act == 1 else (F.relu(x) if act == 2 else x)


def rnd(seed, *shape):
    return torch.from_numpy(detgen.uniform(seed, shape, -1.0, 1.0))


# (transposed, Cin, Cout, k, s, p, Lin, B): every layer geometry of the U-Net at small and mid channel counts,
# plus ragged sizes that leave partial tiles in M, N and K and a geometry that uses the runtime-(k,s) fallback.
GEOMS = [
    (False, 8, 16, 32, 2, 16, 24, 1), (False, 16, 16, 8, 1, 2, 13, 3), (False, 16, 16, 8, 2, 1, 10, 3),
    (False, 16, 32, 4, 2, 1, 3, 2), (True, 32, 16, 5, 2, 1, 1, 2), (True, 32, 16, 8, 2, 1, 3, 3),
    (True, 32, 16, 8, 1, 2, 10, 3), (True, 32, 16, 32, 2, 16, 13, 1),
    (False, 64, 160, 32, 2, 16, 128, 2), (False, 160, 136, 8, 1, 2, 65, 3), (False, 136, 130, 8, 2, 1, 62, 2),
    (False, 130, 260, 4, 2, 1, 29, 3), (True, 260, 130, 5, 2, 1, 14, 3), (True, 264, 132, 8, 2, 1, 29, 2),
    (True, 264, 132, 8, 1, 2, 62, 2), (True, 200, 140, 32, 2, 16, 65, 2),
    (False, 24, 40, 7, 3, 2, 50, 2), (True, 24, 40, 7, 3, 2, 17, 2),
    # k = 5, s = 2 beyond the U-Net's own use (transposed, even channel counts): as a Conv1d, with odd channel counts (the F
    # form then stays on the im2col kernel), with enough columns for several tiles and samples per tile
    (False, 34, 20, 5, 2, 1, 31, 3), (False, 33, 70, 5, 2, 2, 17, 2), (True, 36, 17, 5, 2, 1, 9, 2), (True, 72, 140, 5, 2, 1, 30, 9),
    # frame counts at which the wgrad takes the per-sample-slab kernel (padding a sample to whole 16-frame slabs costs <= 7 %):
    # 61 and 30 frames of dy / x for the stride-2 families, several samples, ragged channel counts
    (False, 40, 70, 8, 2, 1, 126, 3), (False, 36, 72, 4, 2, 1, 61, 5), (True, 72, 40, 8, 2, 1, 61, 3), (False, 24, 136, 8, 1, 2, 33, 4),
    # sized for the one-wave-per-SIMD fp32 kernels (conv_raw3.hip: 256 x 256 tiles; rows just under a multiple of 256, more than one
    # column tile, several samples per tile): every (k, s) pair and form they cover
    (False, 32, 250, 8, 2, 1, 130, 5), (True, 48, 120, 8, 2, 1, 70, 4), (True, 64, 128, 32, 2, 16, 65, 3), (False, 16, 250, 32, 2, 16, 200, 3),
    (True, 32, 250, 8, 1, 2, 100, 3), (False, 32, 500, 4, 2, 1, 160, 4), (False, 48, 230, 8, 1, 2, 90, 4),
    # ... and sized for 256-row tiles (the removed one-wave-per-SIMD wgrad, conv_g3.hip, was tested on them): ~250 rows of P; samples of 16 cf + 1 / + 2 frames at a batch of 16 / 32
    # (the rem last frames of 16 samples form "leftover" slabs), and frame counts whose padding to whole slabs costs <= 7 % (0, 3.2, 6.7 %)
    (False, 16, 250, 32, 2, 16, 64, 16), (False, 24, 250, 32, 2, 16, 66, 32), (True, 250, 16, 32, 2, 16, 33, 16),
    (False, 40, 250, 8, 1, 2, 34, 3), (False, 40, 250, 8, 2, 1, 125, 3), (False, 70, 250, 4, 2, 1, 62, 5), (True, 250, 24, 8, 2, 1, 31, 4),
    (False, 16, 250, 8, 1, 2, 35, 2),
    # k = 5 at stride 2 on the one-wave-per-SIMD kernels (a virtual k = 8 whose virtual taps' MFMAs are never issued): forward and dgrad
    # of both layer kinds with ~250 GEMM rows on the side that takes them
    (True, 48, 125, 5, 2, 1, 30, 6), (False, 32, 250, 5, 2, 2, 61, 4), (False, 125, 40, 5, 2, 2, 61, 4), (True, 250, 40, 5, 2, 1, 30, 5),
    # few columns, long K (a single clip through wide layers): one or two tiles split into tens to hundreds of segments -- the wide
    # fixup (conv_igemm.hip, fixup_wide) in every form: F, T stride 1, T stride 2 (phase-major rows), and a wgrad over 18 000 frames
    (False, 512, 250, 32, 2, 16, 24, 1), (True, 768, 120, 8, 1, 2, 20, 1), (True, 1024, 120, 8, 2, 1, 14, 2), (True, 640, 130, 32, 2, 16, 9, 1),
    (False, 600, 300, 4, 2, 1, 30, 1), (False, 8, 16, 8, 1, 2, 3000, 6),
]


@pytest.fixture(params=[1, 2, 5, 6, 10, 128 | 2, 0x2000 | 1, 0x2000 | 2, 0x4000 | 1, 0x4000 | 2, 0x40000, 0x40000 | 0x4000 | 2],
                ids=["raw/tile-per-wg", "raw/stream-k", "im2col/tile-per-wg", "im2col/stream-k", "raw-wide-only/stream-k",
                     "flat-K-wgrad/stream-k", "raw-2-waves-per-simd/tile-per-wg", "raw-2-waves-per-simd/stream-k",
                     "raw-1-wave-per-simd-everywhere/tile-per-wg", "raw-1-wave-per-simd-everywhere/stream-k",
                     "column-tail-launch/auto", "column-tail-launch/1-wave-everywhere/stream-k"])
def schedule(request):
    """Run the conv tests under both work decompositions (one whole tile per workgroup; the persistent stream-K split
    with partial tiles through the workspace + fixup kernel, which the library otherwise only picks for tile counts
    that quantise badly over the CUs) and with the raw-window F/T kernels enabled or disabled (bit 2), so the im2col
    kernels they normally replace stay covered; bit 3 keeps the small problems of this file on the wide 128 x 256 raw tile
    (they otherwise take the tall 256 x 128 one), so both tile shapes see every geometry; bit 7 keeps the wgrad on the flat-K kernel
    where it would take the per-sample-slab one; bit 18 hands the columns past the last full 256-wide tile of a conv_raw3 problem to a
    second launch of the tall-tile kernel wherever the geometry allows (the automatic choice does so only where its cost model says the
    tail is cheaper than another tile column -- never on problems of this file's size); bit 13 keeps the fp32 F / T problems that conv_raw3.hip covers on the older
    two-waves-per-SIMD raw kernels, bit 14 puts every problem they cover on them (the F form of k = 32 is otherwise left out)."""
    from phasegen import ops
    ops.set_conv_schedule(request.param)
    yield request.param
    ops.set_conv_schedule(0)


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv_fwd_dgrad_wgrad(geom, act, schedule):
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    x = rnd(1, B, Cin, Lin)
    w = rnd(2, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xa = act_cpu(xr, act)
    xa.retain_grad()
    yr = F.conv_transpose1d(xa, wr, stride=s, padding=p) if tr else F.conv1d(xa, wr, stride=s, padding=p)
    dy = rnd(3, *yr.shape)
    yr.backward(dy)

    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    y = torch.full(yr.shape, float("nan"), device="cuda")
    ops.conv_fwd(xd, wd, y, s, p, x_act=act, transposed=tr)
    assert relerr(y, yr) < TOL

    # dgrad wrt the (activated) operand the conv read: autograd's grad of xa
    dx = torch.full(x.shape, float("nan"), device="cuda")
    ops.conv_dgrad(dyd, wd, dx, s, p, transposed=tr)
    assert relerr(dx, xa.grad) < TOL
    # fused epilogue: (+ add) * act'(ref)  == grad wrt the pre-activation tensor plus a skip gradient
    if act:
        add = rnd(4, *x.shape)
        dx2 = torch.full(x.shape, float("nan"), device="cuda")
        ops.conv_dgrad(dyd, wd, dx2, s, p, transposed=tr, add=add.cuda(), ref=xd, mask=act)
        slope = 0.2 if act == 1 else 0.0
        want = (xa.grad + add) * torch.where(x > 0, torch.ones_like(x), torch.full_like(x, slope))
        assert relerr(dx2, want) < TOL

    dw = torch.full(w.shape, float("nan"), device="cuda")
    ops.conv_wgrad(xd, dyd, dw, s, p, x_act=act, transposed=tr)
    assert relerr(dw, wr.grad) < TOL


def bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("mode", [1, 6], ids=["raw/tile-per-wg", "im2col/stream-k"])
def test_conv_bf16_operand_mode(geom, act, mode):
    """pg_conv_args.precision = PG_PREC_BF16 (BASELINE config 5): operands rounded to bf16 (RNE) AFTER the fused activation, fp32
    accumulate, in every conv kernel.  Oracle = float64 autograd of the same conv on the rounded tensors, so only the
    accumulation order differs: 2e-5 (against the unrounded fp32 oracle the same outputs are ~3e-3 off)."""
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    x = rnd(1, B, Cin, Lin)
    w = rnd(2, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    conv = (lambda a, b: F.conv_transpose1d(a, b, stride=s, padding=p)) if tr else (lambda a, b: F.conv1d(a, b, stride=s, padding=p))
    xa = bf16_round(act_cpu(x, act)).double().requires_grad_(True)
    wr = bf16_round(w).double().requires_grad_(True)
    yr = conv(xa, wr)
    dy = rnd(3, *yr.shape)
    yr.backward(bf16_round(dy).double())
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    y, dx, dw = torch.empty(yr.shape, device="cuda"), torch.empty(x.shape, device="cuda"), torch.empty(w.shape, device="cuda")
    try:
        ops.set_conv_schedule(mode)
        ops.set_conv_precision("bf16")
        ops.conv_fwd(xd, wd, y, s, p, x_act=act, transposed=tr)
        ops.conv_dgrad(dyd, wd, dx, s, p, transposed=tr)
        ops.conv_wgrad(xd, dyd, dw, s, p, x_act=act, transposed=tr)
    finally:
        ops.set_conv_precision("fp32")
        ops.set_conv_schedule(0)
    assert relerr(y, yr.detach()) < 2e-5 and relerr(dx, xa.grad) < 2e-5 and relerr(dw, wr.grad) < 2e-5


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("mode", [1, 6], ids=["raw/tile-per-wg", "im2col/stream-k"])
def test_conv_bf16x3_split_mode_meets_the_fp32_bound(geom, mode):
    """pg_conv_args.precision = PG_PREC_BF16X3: fp32 operands split hi + lo into bf16 pairs, three bf16 MFMA products, fp32 accumulate.
    Checked against the UNROUNDED float64 convolution: the dropped lo*lo term and the split residuals are <= 2^-18 of a
    product, so the result sits ~5e-6 from exact -- 20x inside the 1e-4 parity bound (fp32 MFMA: ~1e-6)."""
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    act = 1
    x = rnd(1, B, Cin, Lin)
    w = rnd(2, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    xa = act_cpu(x, act).double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = F.conv_transpose1d(xa, wr, stride=s, padding=p) if tr else F.conv1d(xa, wr, stride=s, padding=p)
    dy = rnd(3, *yr.shape)
    yr.backward(dy.double())
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    y, dx, dw = torch.empty(yr.shape, device="cuda"), torch.empty(x.shape, device="cuda"), torch.empty(w.shape, device="cuda")
    try:
        ops.set_conv_schedule(mode)
        ops.set_conv_precision("bf16x3")
        ops.conv_fwd(xd, wd, y, s, p, x_act=act, transposed=tr)
        ops.conv_dgrad(dyd, wd, dx, s, p, transposed=tr)
        ops.conv_wgrad(xd, dyd, dw, s, p, x_act=act, transposed=tr)
    finally:
        ops.set_conv_precision("fp32")
        ops.set_conv_schedule(0)
    errs = relerr(y, yr.detach()), relerr(dx, xa.grad), relerr(dw, wr.grad)
    assert max(errs) < 2e-5, errs


def test_conv_on_concat_slices(schedule):
    """Operands given as channel slices of a wider buffer (batch stride != C*L), as the U-Net concat does."""
    from phasegen import ops
    B, C, L = 3, 24, 29
    cat = rnd(5, B, 2 * C, L)
    w = rnd(6, 2 * C, 16, 8) * 0.1                       # ConvTranspose1d(2C -> 16, k8, s2, p1)
    yr = F.conv_transpose1d(F.relu(cat), w, stride=2, padding=1)
    catd = torch.empty(B, 2 * C, L, device="cuda")
    catd[:, :C] = cat[:, :C].cuda()
    catd[:, C:] = cat[:, C:].cuda()
    y = torch.empty(yr.shape, device="cuda")
    ops.conv_fwd(catd, w.cuda(), y, 2, 1, x_act=2, transposed=True)
    assert relerr(y, yr) < TOL
    w2 = rnd(7, 16, C, 8) * 0.1                          # Conv1d(C -> 16) reading only the first half
    y2r = F.conv1d(F.leaky_relu(cat[:, :C], 0.2), w2, stride=1, padding=2)
    y2 = torch.empty(y2r.shape, device="cuda")
    ops.conv_fwd(catd[:, :C], w2.cuda(), y2, 1, 2, x_act=1)
    assert relerr(y2, y2r) < TOL
    out = torch.zeros(B, 32, y2r.shape[2], device="cuda")  # ... and writing into the second half of a wider buffer
    ops.conv_fwd(catd[:, :C], w2.cuda(), out[:, 16:], 1, 2, x_act=1)
    assert relerr(out[:, 16:], y2r) < TOL and float(out[:, :16].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(1, 16, 13), (3, 48, 62), (64, 40, 129), (2, 8, 300)])
def test_bn_fwd_bwd(shape):
    from phasegen import ops
    B, C, L = shape
    x = rnd(8, B, C, L) * 2 + 0.3
    g = torch.from_numpy(detgen.uniform(9, (C,), 0.5, 1.5))
    b = torch.from_numpy(detgen.uniform(10, (C,), -0.5, 0.5))
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    yr = F.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    dy = rnd(11, B, C, L)
    yr.backward(dy)
    xd = x.cuda()
    y = torch.empty_like(xd)
    sm, si = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    rmd, rvd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    ops.bn_fwd(xd, y, g.cuda(), b.cuda(), sm, si, rmd, rvd)
    assert relerr(y, yr) < TOL
    assert relerr(rmd, rm) < TOL and relerr(rvd, rv) < TOL
    dx = torch.empty_like(xd)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_bwd(xd, dy.cuda(), dx, g.cuda(), sm, si, dg, db)
    assert relerr(dx, xr.grad) < TOL
    assert relerr(dg, gr.grad) < TOL and relerr(db, br.grad) < TOL


@pytest.mark.parametrize("shape", [(1, 8, 24), (3, 16, 64), (5, 33, 77)])
def test_loss_fwd_bwd(shape):
    from phasegen import ops
    B, C, L = shape
    pred = (rnd(12, B, 2 * C, L) * 4).requires_grad_(True)
    batch = torch.from_numpy(detgen.make_batch(B, C, L, seed=3))
    loss, ang, mag = unet_ref.phase_loss(pred, batch)
    loss.backward()
    dpred = torch.empty(B, 2 * C, L, device="cuda")
    out = ops.loss_fwd_bwd(pred.detach().cuda(), batch.cuda(), dpred)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.allclose(got, [loss.item(), ang.item(), mag.item()], rtol=1e-5)
    assert relerr(dpred, pred.grad) < TOL


def test_adam_matches_torch_optim():
    from phasegen import ops
    n = 100003
    p0 = torch.from_numpy(detgen.uniform(13, (n,), -0.1, 0.1))
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    p, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = torch.from_numpy(detgen.uniform(20 + step, (n,), -1e-2, 1e-2))
        g[::7] = 0.0
        pr.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.cuda(), m, v, step)
    assert relerr(p, pr) < 1e-6
    st = opt.state[pr]
    assert relerr(m, st["exp_avg"]) < 1e-6 and relerr(v, st["exp_avg_sq"]) < 1e-6


def test_bad_arguments_raise():
    from phasegen import ops
    x = torch.zeros(2, 8, 24, device="cuda")
    w = torch.zeros(16, 8, 32, device="cuda")
    with pytest.raises(ValueError):
        ops.conv_fwd(x, w, torch.zeros(2, 16, 99, device="cuda"), 2, 16)        # wrong Lout
    with pytest.raises(ValueError):
        ops.conv_fwd(x.transpose(1, 2), w, torch.zeros(2, 16, 13, device="cuda"), 2, 16)  # not frame-contiguous
    with pytest.raises(RuntimeError):
        ops.adam_step(x.view(-1), x.view(-1), x.view(-1), x.view(-1), 0)        # step is 1-based


def test_stream_k_is_bit_identical_run_to_run_and_matches_plain_closely():
    """The stream-K split sums partial tiles in a fixed order: two runs are bit-identical; against the one-tile-per-
    workgroup schedule only the k-summation order differs (fp32 rounding)."""
    from phasegen import ops
    B, Cin, Cout, k, s, p, Lin = 5, 96, 200, 8, 2, 1, 61
    x, w = rnd(41, B, Cin, Lin).cuda(), (rnd(42, Cout, Cin, k) * 0.1).cuda()
    Lout = ops.conv_out_len(Lin, k, s, p)
    outs = []
    for mode in (2, 2, 1):
        ops.set_conv_schedule(mode)
        y = torch.empty(B, Cout, Lout, device="cuda")
        ops.conv_fwd(x, w, y, s, p, x_act=1)
        outs.append(y.clone())
    ops.set_conv_schedule(0)
    assert torch.equal(outs[0], outs[1])
    assert relerr(outs[0], outs[2]) < 1e-5


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("geom", GEOMS, ids=[f"{'T' if g[0] else 'C'}{g[1]}-{g[2]}-k{g[3]}s{g[4]}" for g in GEOMS])
def test_wgrad_with_fused_adam_equals_wgrad_then_adam(geom, schedule, precision):
    """pg_conv_args.adam: the Adam update of a conv weight in the epilogue of its wgrad kernel (GEMM kernel for whole tiles,
    fixup kernel for stream-K split tiles; raw-window and im2col kernels; every operand precision) leaves dw, p, exp_avg and
    exp_avg_sq BIT-identical to the plain wgrad followed by pg_adam_step -- at step 1 (zero state) and at step 3."""
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    Lout = ops.convt_out_len(Lin, k, s, p) if tr else ops.conv_out_len(Lin, k, s, p)
    wshape = (Cin, Cout, k) if tr else (Cout, Cin, k)
    x, dy = rnd(1, B, Cin, Lin).cuda(), rnd(2, B, Cout, Lout).cuda()
    for step, grad_scale in ((1, 1.0), (3, 0.5)):
        w0 = rnd(3, *wshape).cuda()
        m0 = rnd(4, *wshape).cuda() * (0.01 if step > 1 else 0.0)
        v0 = rnd(5, *wshape).cuda().abs() * (1e-4 if step > 1 else 0.0)
        dw_ref = torch.empty_like(w0)
        ops.conv_wgrad(x, dy, dw_ref, s, p, x_act=1, transposed=tr, precision=precision)
        w_ref, m_ref, v_ref = w0.clone(), m0.clone(), v0.clone()
        ops.adam_step(w_ref.view(-1), dw_ref.view(-1), m_ref.view(-1), v_ref.view(-1), step, lr=1e-3, grad_scale=grad_scale)
        w, m, v, dw = w0.clone(), m0.clone(), v0.clone(), torch.empty_like(w0)
        ad = ops.adam_args(w, m, v, step, lr=1e-3, grad_scale=grad_scale)
        ops.conv_wgrad(x, dy, dw, s, p, x_act=1, transposed=tr, precision=precision, adam=ad)
        assert torch.equal(dw, dw_ref)
        assert torch.equal(w, w_ref) and torch.equal(m, m_ref) and torch.equal(v, v_ref)
        assert not torch.equal(w, w0)


def test_fused_adam_argument_errors():
    from phasegen import ops
    x, dy = rnd(1, 2, 8, 24).cuda(), rnd(2, 2, 16, 12).cuda()
    w = rnd(3, 16, 8, 4).cuda()
    m, v, dw = torch.zeros_like(w), torch.zeros_like(w), torch.empty_like(w)
    with pytest.raises(RuntimeError, match="1-based"):
        ops.conv_wgrad(x, dy, dw, 2, 1, adam=ops.adam_args(w, m, v, 0))
    with pytest.raises(RuntimeError, match="alias"):
        ops.conv_wgrad(x, dy, dw, 2, 1, adam=ops.adam_args(dw, m, v, 1))
    with pytest.raises(ValueError, match="shape"):
        ops.conv_wgrad(x, dy, dw, 2, 1, adam=ops.adam_args(w[:8].contiguous(), m[:8].contiguous(), v[:8].contiguous(), 1))
    # the library itself (a C caller has no Python wrapper in front of it) refuses p / m / v of another size
    import ctypes
    from phasegen import _lib
    a = ops._conv_args(False, 2, 8, 16, 24, 4, 2, 1, x.device)
    a.x, a.x_bs = ops._act3(x, "x")
    a.dy, a.dy_bs = ops._act3(dy, "dy")
    a.dw = dw.data_ptr()
    ad = ops.adam_args(w, m, v, 1)
    ad.n = w.numel() - 4
    a.adam = ctypes.addressof(ad)
    before = w.clone()
    assert _lib.load().pg_conv1d_wgrad(ctypes.byref(a), ops._stream()) == _lib.ERR_SHAPE
    torch.cuda.synchronize()
    assert torch.equal(w, before)


def _random_geoms(n, seed):
    """Seeded sweep over geometries the fixed list does not hold: every (k, stride) kernel family plus the runtime-(k, s)
    fallback, paddings up to k - 1, frame counts from one tile column to several samples per tile, channel counts that leave
    partial slabs and partial tiles."""
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        tr = bool(rs.randint(2))
        k, s = [(32, 2), (8, 1), (8, 2), (4, 2), (5, 2), (3, 1), (7, 3), (6, 2), (1, 1), (16, 4)][rs.randint(10)]
        p = int(rs.randint(0, k))
        Cin, Cout = int(rs.choice([3, 8, 17, 40, 64, 130])), int(rs.choice([5, 16, 33, 72, 136]))
        Lin, B = int(rs.choice([1, 2, 7, 16, 31, 64, 129])), int(rs.randint(1, 5))
        Lout = (Lin - 1) * s - 2 * p + k if tr else (Lin + 2 * p - k) // s + 1
        if Lout < 1 or (not tr and Lin + 2 * p < k):
            continue
        out.append((tr, Cin, Cout, k, s, p, Lin, B))
    return out


@pytest.mark.parametrize("geom", _random_geoms(48, 20261004), ids=lambda g: f"{'T' if g[0] else 'C'}{g[1]}-{g[2]}-k{g[3]}s{g[4]}p{g[5]}-L{g[6]}-B{g[7]}")
def test_conv_random_geometries_vs_torch(geom):
    """forward / dgrad / wgrad of 48 seeded random geometries (automatic schedule) against fp32 torch on the CPU."""
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    x = rnd(11, B, Cin, Lin)
    w = rnd(12, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose1d(xr, wr, stride=s, padding=p) if tr else F.conv1d(xr, wr, stride=s, padding=p)
    dy = rnd(13, *yr.shape)
    yr.backward(dy)
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    y = torch.full(yr.shape, float("nan"), device="cuda")
    ops.conv_fwd(xd, wd, y, s, p, transposed=tr)
    dx = torch.full(x.shape, float("nan"), device="cuda")
    ops.conv_dgrad(dyd, wd, dx, s, p, transposed=tr)
    dw = torch.full(w.shape, float("nan"), device="cuda")
    ops.conv_wgrad(xd, dyd, dw, s, p, transposed=tr)
    assert relerr(y, yr) < TOL and relerr(dx, xr.grad) < TOL and relerr(dw, wr.grad) < TOL


def _random_geoms_one_wave(n, seed):
    """Seeded sweep sized for the one-wave-per-SIMD fp32 kernels (conv_raw3.hip; schedule bit 14 forces them wherever they cover the problem): GEMM rows
    just under a multiple of 256 on the side that matters, channel counts that give whole 16-deep slabs, frame counts from less than a
    column tile to several samples per tile, every (k, stride) pair they cover, paddings up to k - 1."""
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        tr = bool(rs.randint(2))
        k, s = [(32, 2), (8, 1), (8, 2), (4, 2), (5, 2)][rs.randint(5)]
        p = int(rs.randint(0, k))
        big, small = int(rs.choice([200, 236, 250, 256, 470, 500])), int(rs.choice([8, 16, 24, 32, 48]))
        # conv: rows of fwd = Cout, of dgrad (T form) = Cin * s; convT: rows of fwd (T form) = Cout * s, of dgrad = Cin
        Cin, Cout = (small, big) if rs.randint(2) else (big // (s if not tr else 1), small * 4)
        Lin, B = int(rs.choice([17, 30, 33, 61, 64, 100, 129])), int(rs.choice([1, 2, 3, 5, 16]))
        Lout = (Lin - 1) * s - 2 * p + k if tr else (Lin + 2 * p - k) // s + 1
        if Lout < 1 or (not tr and Lin + 2 * p < k) or B * max(Lin, Lout) * max(Cin, Cout) > 3_000_000:
            continue
        out.append((tr, Cin, Cout, k, s, p, Lin, B))
    return out


@pytest.mark.parametrize("sched", [0, 0x4000 | 2, 0x4000 | (3 << 15), 0x4000 | (2 << 15) | 2, (1 << 15) | 1, 0x40000 | 0x4000, 0x40000 | 2],
                         ids=["auto", "one-wave-everywhere/stream-k", "super-rows-of-4", "super-rows-of-2/stream-k", "row-major/tile-per-wg",
                              "column-tail-launch/one-wave-everywhere", "column-tail-launch/stream-k"])
@pytest.mark.parametrize("geom", _random_geoms_one_wave(32, 20261005), ids=lambda g: f"{'T' if g[0] else 'C'}{g[1]}-{g[2]}-k{g[3]}s{g[4]}p{g[5]}-L{g[6]}-B{g[7]}")
def test_conv_random_geometries_one_wave_kernels(geom, sched):
    """forward / dgrad / wgrad (with an input activation on the window operand and the fused dgrad epilogue) of 32 seeded random
    geometries sized for the one-wave-per-SIMD kernels, under the automatic schedule and with those kernels forced wherever they
    cover the problem, against fp32 torch on the CPU."""
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    x = rnd(21, B, Cin, Lin)
    w = rnd(22, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xa = act_cpu(xr, 1)
    xa.retain_grad()
    yr = F.conv_transpose1d(xa, wr, stride=s, padding=p) if tr else F.conv1d(xa, wr, stride=s, padding=p)
    dy = rnd(23, *yr.shape)
    yr.backward(dy)
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    ops.set_conv_schedule(sched)
    try:
        y = torch.full(yr.shape, float("nan"), device="cuda")
        ops.conv_fwd(xd, wd, y, s, p, x_act=1, transposed=tr)
        dx = torch.full(x.shape, float("nan"), device="cuda")
        add = rnd(24, *x.shape)
        ops.conv_dgrad(dyd, wd, dx, s, p, transposed=tr, add=add.cuda(), ref=xd, mask=1)
        dw = torch.full(w.shape, float("nan"), device="cuda")
        ops.conv_wgrad(xd, dyd, dw, s, p, x_act=1, transposed=tr)
    finally:
        ops.set_conv_schedule(0)
    want_dx = (xa.grad + add) * torch.where(x > 0, torch.ones_like(x), torch.full_like(x, 0.2))
    assert relerr(y, yr) < TOL and relerr(dx, want_dx) < TOL and relerr(dw, wr.grad) < TOL


@pytest.mark.parametrize("geom", [(True, 96, 250, 32, 2, 16, 129, 4), (False, 64, 500, 8, 1, 2, 126, 6), (True, 128, 125, 8, 2, 1, 61, 8),
                                  (False, 48, 250, 4, 2, 1, 62, 16), (False, 32, 250, 32, 2, 16, 256, 16)],
                         ids=["T-k32", "F-k8s1", "T-k8s2", "F-k4", "F-k32"])
def test_one_wave_kernels_are_race_free_by_repetition(geom):
    """The one-wave-per-SIMD kernels order their LDS traffic with counted vmcnt waits, one raw s_barrier per slab and loop-carried
    asm reads -- nothing the compiler checks.  A misplaced wait shows as results that change from launch to launch (the DMA sometimes
    lands first): 40 launches each of forward / dgrad / wgrad under the forced stream-K split (partial tiles, fixup) and with one tile per
    workgroup must be bit-identical to the first, which itself is checked against torch."""
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    x = rnd(31, B, Cin, Lin)
    w = rnd(32, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose1d(xr, wr, stride=s, padding=p) if tr else F.conv1d(xr, wr, stride=s, padding=p)
    dy = rnd(33, *yr.shape)
    yr.backward(dy)
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    for sched in (0x4000 | 2, 0x4000 | 1):
        ops.set_conv_schedule(sched)
        try:
            first = None
            for it in range(int(os.environ.get("PG_RACE_REPS", "40"))):       # (a one-off soak: PG_RACE_REPS=400)
                y = torch.empty(yr.shape, device="cuda"); dx = torch.empty(x.shape, device="cuda"); dw = torch.empty(w.shape, device="cuda")
                ops.conv_fwd(xd, wd, y, s, p, transposed=tr)
                ops.conv_dgrad(dyd, wd, dx, s, p, transposed=tr)
                ops.conv_wgrad(xd, dyd, dw, s, p, transposed=tr)
                if first is None:
                    first = (y, dx, dw)
                    assert relerr(y, yr) < TOL and relerr(dx, xr.grad) < TOL and relerr(dw, wr.grad) < TOL
                else:
                    assert torch.equal(y, first[0]) and torch.equal(dx, first[1]) and torch.equal(dw, first[2]), (sched, it)
        finally:
            ops.set_conv_schedule(0)


def _bn_cases():
    # (B, C, L, channel offset of the views inside wider buffers, extra channels of those buffers)
    return [(64, 24, 256, 0, 0), (64, 16, 256, 8, 16), (64, 16, 126, 0, 0), (64, 16, 126, 3, 5), (64, 12, 129, 0, 0), (64, 12, 61, 1, 2),
            (7, 20, 128, 0, 4), (5, 9, 4, 0, 0), (3, 7, 2, 2, 2), (1, 6, 1000, 0, 0), (16, 8, 1024, 0, 0), (2, 5, 8192, 0, 0), (3, 4, 6000, 0, 0),
            (1, 3, 2, 0, 0), (33, 10, 30, 5, 0),
            # single clips (demo.py): one or two units per thread -- the smallest instantiation of each unit width
            (1, 10, 64, 0, 0), (1, 6, 66, 0, 0), (1, 6, 66, 1, 0), (2, 5, 65, 0, 0), (1, 7, 130, 1, 0), (3, 5, 128, 0, 4)]


@pytest.mark.parametrize("case", _bn_cases(), ids=lambda c: "B%d-C%d-L%d-off%d+%d" % c)
def test_bn_flat_walk_variants(case):
    """The register-resident BatchNorm kernels pick 16-byte, 8-byte or 4-byte units from the frame count and the alignment of
    every tensor, and walk the channel flat: frame counts divisible by 4 / by 2 / odd, channel VIEWS of wider buffers (the
    concat buffers of the U-Net: batch stride != C * L, starts that are or are not 16-byte aligned), second activated output,
    B * L from 1 to 16 384 values per channel, and past that the three-pass fallback -- against F.batch_norm and its autograd."""
    from phasegen import ops
    B, C, L, off, extra = case
    x = rnd(8, B, C, L) * 2 + 0.3
    g = torch.from_numpy(detgen.uniform(9, (C,), 0.5, 1.5))
    b = torch.from_numpy(detgen.uniform(10, (C,), -0.5, 0.5))
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.batch_norm(xr, None, None, gr, br, training=True, momentum=0.1, eps=1e-5)
    dy = rnd(11, B, C, L)
    yr.backward(dy)
    Cw = C + off + extra

    def view_of(t=None, fill=float("nan")):
        buf = torch.full((B, Cw, L), fill, device="cuda")
        v = buf[:, off:off + C]
        if t is not None:
            v.copy_(t.cuda())
        return buf, v

    _, xv = view_of(x)
    ybuf, yv = view_of()
    y2buf, y2v = view_of()
    sm, si = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_fwd(xv, yv, g.cuda(), b.cuda(), sm, si, y_act=ops.ACT_LEAKY, y2=y2v, y2_act=ops.ACT_RELU)
    assert relerr(yv, F.leaky_relu(yr, 0.2)) < TOL and relerr(y2v, F.relu(yr)) < TOL
    if off or extra:                                                   # nothing outside the view was written
        rest = torch.cat([ybuf[:, :off], ybuf[:, off + C:]], 1)
        assert bool(torch.isnan(rest).all())
    _, dyv = view_of(dy)
    dxbuf, dxv = view_of()
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_bwd(xv, dyv, dxv, g.cuda(), sm, si, dg, db)
    assert relerr(dxv, xr.grad) < TOL and relerr(dg, gr.grad) < TOL and relerr(db, br.grad) < TOL
    if off or extra:
        assert bool(torch.isnan(torch.cat([dxbuf[:, :off], dxbuf[:, off + C:]], 1)).all())
