"""GPU: the bf16-RESIDENT forward convolutions (csrc/conv_h3.hip, BASELINE configs[4]) against a float64 convolution of the
SAME bf16 operands (products of bf16 values are exact in fp32, so only the accumulation order differs: 2e-5 of max-abs), at
every (k, stride) geometry of the U-Net, at ragged sizes (partial tiles in M and N, several samples per tile, odd frame counts
whose last bf16 pair is half padding) and under both work decompositions; plus the helper kernels (weight shadow, row cast,
BatchNorm's bf16 outputs)."""
import os
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_ref  # noqa: F401  (disables oneDNN: see oracle/unet_ref.py)
from phasegen import detgen

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(seed, *shape):
    return torch.from_numpy(detgen.uniform(seed, shape, -1.0, 1.0))


# (transposed, Cin, Cout, k, s, p, Lin, B): Cin must be a multiple of 32 / min(taps per phase, 32)
GEOMS = [
    (False, 8, 16, 32, 2, 16, 24, 1), (False, 16, 16, 8, 1, 2, 13, 3), (False, 16, 24, 8, 2, 1, 10, 3), (False, 16, 32, 4, 2, 1, 7, 2),
    (True, 32, 16, 8, 2, 1, 3, 3), (True, 32, 16, 8, 1, 2, 10, 3), (True, 32, 16, 32, 2, 16, 13, 1),
    (False, 64, 160, 32, 2, 16, 128, 2), (False, 160, 136, 8, 1, 2, 65, 3), (False, 136, 130, 8, 2, 1, 62, 2), (False, 136, 260, 4, 2, 1, 29, 3),
    (True, 264, 132, 8, 2, 1, 29, 2), (True, 264, 132, 8, 1, 2, 62, 2), (True, 200, 140, 32, 2, 16, 65, 2),
    (False, 64, 128, 32, 2, 16, 256, 5), (True, 128, 64, 32, 2, 16, 129, 5), (True, 64, 96, 8, 1, 2, 126, 6), (False, 64, 96, 8, 2, 1, 126, 6),
    (True, 32, 16, 5, 2, 1, 1, 2), (True, 264, 130, 5, 2, 1, 14, 3), (True, 128, 96, 5, 2, 1, 30, 9),      # k = 5: shadow padded to 4 taps per phase
    # few columns, long K (a single clip): one or two tiles in tens to hundreds of segments -> the wide fixup of conv_h3.hip
    (False, 512, 250, 32, 2, 16, 24, 1), (True, 768, 120, 8, 1, 2, 20, 1), (True, 1024, 300, 8, 2, 1, 14, 2), (False, 608, 300, 4, 2, 1, 30, 1),
]


# schedule word: bits 0-1 work split (0 auto, 1 one tile per workgroup, 2 stream-K).  One tile family since ABI 0.4 (256 x 256 on 4
# waves, one per SIMD: conv_h3.hip); bits 5-6, which selected the removed families, are refused (test below)
SCHEDS = [0, 1, 2]


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("sched", SCHEDS, ids=["auto", "tile-per-wg", "stream-k"])
def test_conv_fwd_h_vs_float64_of_the_bf16_operands(geom, sched):
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    x = rnd(11, B, Cin, Lin)
    w = rnd(12, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    want = (F.conv_transpose1d if tr else F.conv1d)(xb.double(), wb.double(), stride=s, padding=p)
    Lout = want.shape[2]
    xh = ops.h_alloc(B, Cin, Lin, "cuda")
    ops.cast_rows_bf16(x.cuda(), xh)
    assert torch.equal(xh[:, :, :Lin].cpu(), xb) and float(xh[:, :, Lin:].abs().max()) == 0.0
    wh = ops.shadow_weights(w.cuda(), tr, s)
    y = torch.full((B, Cout, Lout), float("nan"), device="cuda")
    yh, yh2 = ops.h_alloc(B, Cout, Lout, "cuda"), ops.h_alloc(B, Cout, Lout, "cuda")
    ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y, yh=yh, yh_act=ops.ACT_LEAKY, yh2=yh2, yh2_act=ops.ACT_RELU, schedule=sched)
    assert relerr(y, want) < 2e-5
    # the bf16 copies are the activated fp32 result rounded once (RNE); their row tails were never written
    yc = y.cpu()
    assert torch.equal(yh[:, :, :Lout].cpu(), F.leaky_relu(yc, 0.2).to(torch.bfloat16))
    assert torch.equal(yh2[:, :, :Lout].cpu(), F.relu(yc).to(torch.bfloat16))
    assert float(yh[:, :, Lout:].abs().max()) == 0.0 and float(yh2[:, :, Lout:].abs().max()) == 0.0
    # a bf16 output alone (no fp32 result) is allowed
    yh3 = ops.h_alloc(B, Cout, Lout, "cuda")
    ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, yh=yh3, schedule=sched)
    assert torch.equal(yh3[:, :, :Lout].cpu(), yc.to(torch.bfloat16))


def test_weight_shadow_layouts():
    from phasegen import ops
    w = rnd(5, 6, 4, 8)                                      # ConvTranspose1d (Cin=6, Cout=4, k=8), stride 2: 4 taps per phase
    sh = ops.shadow_weights(w.cuda(), True, 2).cpu().view(4 * 2, 6 * 4)
    wb = w.to(torch.bfloat16)
    for o in range(4):
        for phi in range(2):
            for q in range(6):
                for jj in range(4):
                    assert sh[o * 2 + phi, q * 4 + jj] == wb[q, o, 2 * (3 - jj) + phi]
    w5 = rnd(7, 3, 2, 5)                                     # k = 5, stride 2: 3 and 2 real taps per phase, stored as 4 (zeros)
    sh5 = ops.shadow_weights(w5.cuda(), True, 2).cpu().view(2 * 2, 3 * 4)
    for o in range(2):
        for phi in range(2):
            for q in range(3):
                for jj in range(4):
                    j = 2 * (3 - jj) + phi
                    assert sh5[o * 2 + phi, q * 4 + jj] == (w5.to(torch.bfloat16)[q, o, j] if j < 5 else 0)
    w2 = rnd(6, 5, 3, 8)                                     # Conv1d (Cout=5, Cin=3, k=8): a cast
    assert torch.equal(ops.shadow_weights(w2.cuda(), False, 1).cpu(), w2.to(torch.bfloat16).reshape(-1))


def test_removed_tile_families_are_refused():
    """ABI 0.4 removed the 128 x 256 / 128 x 512 / eight-wave 256 x 256 tiles (no automatic choice reached them): schedule bits 5-6 are
    an error now, bit 12 (the surviving family) is accepted and changes nothing."""
    from phasegen import ops
    B, Cin, Cout, k, s, p, Lin = 1, 8, 16, 32, 2, 16, 24
    w = rnd(3, Cout, Cin, k)
    wh = ops.shadow_weights(w.cuda(), False, s)
    xh = ops.h_alloc(B, Cin, Lin, "cuda")
    y = torch.empty(B, Cout, ops.conv_out_len(Lin, k, s, p), device="cuda")
    for bad in (32, 64, 96, 32 | 1):
        with pytest.raises(RuntimeError, match="removed"):
            ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, y=y, schedule=bad)
    ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, y=y, schedule=4096)
    y0 = y.clone()
    ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, y=y, schedule=0)
    assert torch.equal(y, y0)


def test_unsupported_geometries_are_refused():
    from phasegen import ops
    x = ops.h_alloc(1, 16, 30, "cuda")                       # k = 7, s = 3: no bf16-resident kernel
    w = torch.zeros(16 * 8 * 3 * 4, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="not covered"):
        ops.conv_fwd_h(x, 30, w, (16, 8, 7), 3, 2, transposed=True, yh=ops.h_alloc(1, 8, 90, "cuda"))
    x2 = ops.h_alloc(1, 3, 24, "cuda")                       # Cin = 3 with 8 taps: 4 channels per slab needed
    with pytest.raises(RuntimeError, match="not covered"):
        ops.conv_fwd_h(x2, 24, torch.zeros(16 * 3 * 8, device="cuda", dtype=torch.bfloat16), (16, 3, 8), 1, 2, yh=ops.h_alloc(1, 16, 21, "cuda"))


def test_activation_without_the_zero_head_is_refused():
    """ABI 0.3 contract: the kernels read PG_H_HEAD zero elements in front of x.  A tensor that sits at the very start of its
    allocation (plain torch.zeros instead of ops.h_alloc) must be refused on the host -- the C side cannot see it."""
    from phasegen import ops
    B, Cin, Cout, k, s, p, Lin = 1, 8, 16, 32, 2, 16, 24
    w = rnd(3, Cout, Cin, k)
    wh = ops.shadow_weights(w.cuda(), False, s)
    Lout = ops.conv_out_len(Lin, k, s, p)
    y = torch.empty(B, Cout, Lout, device="cuda")
    plain = torch.zeros(B, Cin, ops.h_pitch(Lin), device="cuda", dtype=torch.bfloat16)      # right pitch and tails, NO head
    with pytest.raises(ValueError, match="h_alloc"):
        ops.conv_fwd_h(plain, Lin, wh, tuple(w.shape), s, p, y=y)
    good = ops.h_alloc(B, Cin, Lin, "cuda")
    ops.conv_fwd_h(good, Lin, wh, tuple(w.shape), s, p, y=y)                                  # same call with the head: accepted
    cat = ops.h_alloc(B, 2 * Cin, Lin, "cuda")
    ops.conv_fwd_h(cat[:, Cin:], Lin, wh, tuple(w.shape), s, p, y=y)                          # a channel slice has the previous row's tail
    torch.cuda.synchronize()


def test_bn_fwd_bf16_outputs():
    from phasegen import ops
    B, Cc, L = 5, 24, 61
    x = rnd(21, B, Cc, L)
    gamma, beta = rnd(22, Cc) + 1.5, rnd(23, Cc)
    y = torch.empty(B, Cc, L, device="cuda")
    yh, yh2 = ops.h_alloc(B, Cc, L, "cuda"), ops.h_alloc(B, Cc, L, "cuda")
    sm, si = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    ops.bn_fwd(x.cuda(), y, gamma.cuda(), beta.cuda(), sm, si, yh=yh, yh_act=ops.ACT_LEAKY, yh2=yh2, yh2_act=ops.ACT_RELU)
    want = F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5)
    assert relerr(y, want) < 1e-5
    assert torch.equal(yh[:, :, :L].cpu(), F.leaky_relu(y.cpu(), 0.2).to(torch.bfloat16)) and float(yh[:, :, L:].abs().max()) == 0.0
    assert torch.equal(yh2[:, :, :L].cpu(), F.relu(y.cpu()).to(torch.bfloat16))
    ops.bn_fwd(x.cuda(), None, gamma.cuda(), beta.cuda(), sm, si, yh=yh2)                # bf16 output only
    assert torch.equal(yh2[:, :, :L].cpu(), y.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("C,L,B", [(16, 64, 3), (32, 128, 2), (1024, 256, 4)], ids=["C16", "C32", "C1024-L256-B4"])
def test_resident_forward_equals_the_bf16_operand_forward(C, L, B):
    """The engine's bf16-resident inference forward (activations stored as bf16 by the producers, bf16 weight shadows) feeds
    the matrix cores exactly the operand values of the fp32-tensor kernels' bf16 mode (which round act(x) and w to bf16 when
    fragments are loaded): per layer the two differ only by fp32 accumulation order (2e-5, tests above).  Through the whole
    network a 1e-6 difference in a pre-rounding value occasionally lands on the other side of a bf16 rounding boundary (a
    0.4 % step in that element), which the next layers average over thousands of terms: 1e-4 at small widths, measured
    5e-3 at C = 1024 (bound 2e-2; the fp32 forward is 3e-2 away from both).  BatchNorm statistics included."""
    from phasegen.model import UNetModel
    if C <= 32:
        pn = detgen.make_params(C, seed=0)
        m = UNetModel(C, 2 * C, precision="bf16").load_numpy(pn)
    else:
        torch.manual_seed(9)
        m = UNetModel(C, 2 * C, precision="bf16")
    x = torch.from_numpy(detgen.make_batch(B, C, L, seed=2)[:, 0].copy()).cuda()
    eng = m.engine
    assert eng.resident_ok()
    ref = eng.forward(x, update_stats=False).clone()                      # bf16 operand mode on fp32 tensors
    got = eng.forward(x, update_stats=False, inference=True)              # bf16-resident kernels
    assert eng.cur is None                                                # nothing kept for a backward
    tol = 1e-4 if C <= 32 else 2e-2
    assert relerr(got, ref) < tol
    with pytest.raises(RuntimeError, match="before forward"):
        eng.backward(torch.zeros_like(got))
    # parameters changed through the supported paths invalidate the weight shadows
    v0 = eng.arena.version
    if C <= 32:
        m.load_numpy(detgen.make_params(C, seed=3))
        assert eng.arena.version > v0
        ref2 = eng.forward(x, update_stats=False).clone()
        assert relerr(eng.forward(x, update_stats=False, inference=True), ref2) < 1e-4 and relerr(ref2, ref) > 1e-2
    # running statistics are updated by the resident forward exactly as by the other one
    rm0 = {k: v.clone() for k, v in eng.arena.buffers.items()}
    eng.forward(x, update_stats=True, inference=True)
    rm1 = {k: v.clone() for k, v in eng.arena.buffers.items()}
    for k in rm0:
        eng.arena.buffers[k].copy_(rm0[k])
    eng.forward(x, update_stats=True)
    for k, v in eng.arena.buffers.items():
        assert relerr(v.float(), rm1[k].float()) < tol, k


def _random_h_geoms(n, seed):
    import numpy as np
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        tr = bool(rs.randint(2))
        k, s, p = [(32, 2, 16), (8, 1, 2), (8, 2, 1), (4, 2, 1), (5, 2, 1)][rs.randint(5)]
        if (k == 5 and not tr) or (k == 4 and tr):
            continue
        taps = (k + s - 1) // s if tr else k
        taps = 1 << (taps - 1).bit_length()
        nq = 32 // min(taps, 32)
        Cin = nq * int(rs.randint(1, 24))
        Cout = int(rs.choice([8, 24, 64, 130, 200]))
        Lin, B = int(rs.choice([3, 9, 14, 30, 61, 64, 126, 129])), int(rs.randint(1, 7))
        if not tr and (Lin + 2 * p - k) // s + 1 < 1:
            continue
        out.append((tr, Cin, Cout, k, s, p, Lin, B))
    return out


@pytest.mark.parametrize("geom", _random_h_geoms(32, 4102026), ids=lambda g: f"{'T' if g[0] else 'C'}{g[1]}-{g[2]}-k{g[3]}s{g[4]}-L{g[6]}-B{g[7]}")
def test_conv_fwd_h_random_geometries(geom):
    """32 seeded random problems of the five (k, stride) families: where pg_conv_fwd_h_supported says yes the result matches the
    float64 convolution of the bf16 operands; where it says no the call refuses (never a wrong result)."""
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    x = rnd(31, B, Cin, Lin)
    w = rnd(32, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    want = (F.conv_transpose1d if tr else F.conv1d)(x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), stride=s, padding=p)
    xh = ops.h_alloc(B, Cin, Lin, "cuda")
    ops.cast_rows_bf16(x.cuda(), xh)
    wh = ops.shadow_weights(w.cuda(), tr, s)
    y = torch.full(tuple(want.shape), float("nan"), device="cuda")
    if ops.conv_fwd_h_supported(B, tuple(w.shape), Lin, s, p, tr):
        ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y)
        assert relerr(y, want) < 2e-5
        # ... with the stream-K split forced (partial tiles through the workspace + the fixup kernel) and one tile per workgroup
        for sched in (2, 1):
            y.fill_(float("nan"))
            ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y, schedule=sched)
            assert relerr(y, want) < 2e-5, sched
    else:
        with pytest.raises(RuntimeError, match="not covered"):
            ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y)


@pytest.mark.parametrize("geom", [(True, 128, 250, 32, 2, 16, 129, 4), (False, 64, 500, 8, 1, 2, 126, 6), (True, 256, 125, 8, 2, 1, 61, 8),
                                  (False, 64, 250, 4, 2, 1, 62, 16), (False, 32, 250, 32, 2, 16, 256, 16)],
                         ids=["T-k32", "F-k8s1", "T-k8s2", "F-k4", "F-k32"])
def test_conv_h3_is_race_free_by_repetition(geom):
    """conv_h3.hip (the default family) orders its LDS traffic with counted vmcnt waits, a raw s_barrier per stage group and
    loop-carried asm reads -- nothing the compiler checks.  A misplaced wait shows as results that change from launch to launch:
    40 launches under the forced stream-K split and with one tile per workgroup must be bit-identical to the first, which itself is
    checked against the float64 convolution of the bf16 operands."""
    from phasegen import ops
    tr, Cin, Cout, k, s, p, Lin, B = geom
    x = rnd(41, B, Cin, Lin)
    w = rnd(42, *((Cin, Cout, k) if tr else (Cout, Cin, k))) * 0.1
    want = (F.conv_transpose1d if tr else F.conv1d)(x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), stride=s, padding=p)
    assert ops.conv_fwd_h_supported(B, tuple(w.shape), Lin, s, p, tr)
    xh = ops.h_alloc(B, Cin, Lin, "cuda")
    ops.cast_rows_bf16(x.cuda(), xh)
    wh = ops.shadow_weights(w.cuda(), tr, s)
    for sched in (2, 1):
        first = None
        for it in range(int(os.environ.get("PG_RACE_REPS", "40"))):       # (a one-off soak: PG_RACE_REPS=400)
            y = torch.empty(tuple(want.shape), device="cuda")
            ops.conv_fwd_h(xh, Lin, wh, tuple(w.shape), s, p, transposed=tr, y=y, schedule=sched)
            if first is None:
                first = y
                assert relerr(y, want) < 2e-5
            else:
                assert torch.equal(y, first), (sched, it)
