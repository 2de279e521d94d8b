"""GPU parity of the STFT / polar / ISTFT kernels and of the data/utils surface against the oracle and the goldens.

  * frame indexing: BIT-EXACT against oracle.signal_ref.frame_indices (BASELINE.json: "bit-exact on STFT frame indexing")
  * polar: against fixture G4 (output of the imported reference data.py), exact on the branch-cut edge cases
  * stft / istft values: fp32, tolerance 2e-5 relative to the tensor max (parity vs librosa itself is UNPINNED:
    oracle/signal_ref.py restates librosa's published definition and is cross-checked vs torch.stft/istft on CPU)
"""
import os

import numpy as np
import pytest
import torch

from oracle import signal_ref
from phasegen import detgen

pytestmark = pytest.mark.gpu


def relmax(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


@pytest.mark.parametrize("n,n_fft,hop", [(65024, 2048, 512), (65280, 1024, 256), (64000, 1024, 256), (184, 32, 8),
                                          (1000, 64, 16), (40, 64, 16), (130560, 2048, 512)])
def test_frame_indexing_bit_exact(n, n_fft, hop):
    from phasegen import ops
    idx = ops.stft_frame_index(n, n_fft, hop).cpu().numpy()
    want = signal_ref.frame_indices(n, n_fft, hop)
    assert idx.shape == want.shape == (1 + n // hop, n_fft)
    assert np.array_equal(idx.astype(np.int64), want)


def test_stft_of_a_ramp_recovers_the_index_map():
    """Same contract through the real kernel: with the window divided out, frame t of a ramp is the index map."""
    from phasegen import ops
    n, n_fft, hop = 1000, 64, 16
    y = torch.arange(n, dtype=torch.float32, device="cuda")
    S = ops.stft(y, n_fft, hop).cpu().numpy()[0]                      # (2, 32, frames) DC dropped
    want = signal_ref.chunk_and_stft(np.arange(n, dtype=np.float32), n_fft, hop)
    assert S.shape == want.shape == (2, 32, 1 + n // hop)
    assert relmax(S, want) < 2e-5


@pytest.mark.parametrize("n,n_fft,hop,nsig", [(65024, 2048, 512, 2), (65280, 1024, 256, 3), (184, 32, 8, 1), (4096, 4096, 1024, 1)])
def test_stft_and_fused_polar_vs_oracle(n, n_fft, hop, nsig):
    from phasegen import ops
    y = np.stack([detgen.make_clip(n, seed=20 + i) for i in range(nsig)])
    yd = torch.from_numpy(y).cuda()
    S = ops.stft(yd, n_fft, hop)
    want = np.stack([signal_ref.chunk_and_stft(y[i], n_fft, hop) for i in range(nsig)])
    assert tuple(S.shape) == want.shape
    assert relmax(S, want) < 2e-5
    P = ops.stft(yd, n_fft, hop, polar=True).cpu().numpy()
    wp = signal_ref.get_spec_and_angle(want)
    assert relmax(P[:, 0], wp[:, 0]) < 2e-5
    # angles: compare on the unit circle (wrap at +-pi) and only where the magnitude is not numerically zero
    big = np.abs(want[:, 0] + 1j * want[:, 1]) > 1e-3 * np.max(np.abs(want))
    d = np.angle(np.exp(1j * (P[:, 1] - wp[:, 1])))
    assert np.max(np.abs(d[big])) < 2e-3
    P2 = ops.polar(S).cpu().numpy()                                   # two-kernel path == fused path
    assert np.array_equal(P2, P)


def test_polar_vs_reference_golden_exact_edges(golden_dir):
    from phasegen.data import get_spec_and_angle
    g = np.load(os.path.join(golden_dir, "polar_g4.npz"))
    out = get_spec_and_angle(g["input"])
    assert out.dtype == np.float32 and out.shape == g["output"].shape
    assert np.max(np.abs(out - g["output"])) < 2e-6
    # branch cut and zeros exactly as the reference (incl. the -0.0 imaginary -> +pi quirk of data.py:40)
    for f in range(5):
        assert out[0, 1, 0, f] == g["output"][0, 1, 0, f], f
        assert out[0, 0, 0, f] == pytest.approx(g["output"][0, 0, 0, f], abs=1e-7)
    mag_only = get_spec_and_angle(g["input"], use_exp=False)
    assert np.allclose(mag_only[:, 0], np.expm1(g["output"][:, 0].astype(np.float64)), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bins,frames,hop,nsig", [(1024, 128, 512, 2), (512, 256, 256, 2), (16, 24, 8, 3), (32, 9, 16, 1)])
def test_istft_vs_oracle(bins, frames, hop, nsig):
    from phasegen import ops
    re = detgen.normal(31, (nsig, bins, frames))
    im = detgen.normal(32, (nsig, bins, frames))
    for norm in (False, True):
        y = ops.istft(torch.from_numpy(re).cuda(), torch.from_numpy(im).cuda(), hop, mode=1, normalize=norm).cpu().numpy()
        for i in range(nsig):
            S = np.concatenate([np.zeros((1, frames), np.complex64), (re[i] + 1j * im[i]).astype(np.complex64)], 0)
            w = signal_ref.istft(S, hop)
            if norm:
                w = w / np.max(np.abs(w))
            assert y[i].shape == w.shape == (hop * (frames - 1),)
            assert relmax(y[i], w) < 2e-5


def test_polar_arithmetic_against_float64():
    """pg_fastmath.h (round 4): |z| by scaled squares, log1p through v_log_f32 with the rounding of 1 + x divided out, atan2 as one
    v_rcp_f32 + a 9-term odd polynomial.  Against float64 over 60 decades of magnitude, all four quadrants, both axes, signed zeros."""
    from phasegen import ops
    rng = np.random.default_rng(7)
    n = 1 << 16
    mag = 10.0 ** rng.uniform(-30, 30, n)
    ang = rng.uniform(-np.pi, np.pi, n)
    re, im = (mag * np.cos(ang)).astype(np.float32), (mag * np.sin(ang)).astype(np.float32)
    # moderate magnitudes (what an STFT holds), the axes, signed zeros, ties |re| == |im|
    m2 = 10.0 ** rng.uniform(-4, 4, n)
    re2, im2 = (m2 * np.cos(ang)).astype(np.float32), (m2 * np.sin(ang)).astype(np.float32)
    special = np.array([(0, 0), (-0.0, 0), (0, -0.0), (-0.0, -0.0), (1, 0), (-1, 0), (0, 1), (0, -1), (-1, -0.0), (1, -0.0), (3, 3), (-3, 3),
                        (-3, -3), (3, -3), (1e-38, 1e-38), (-1e-40, 1e-40), (1e38, -1e38), (2.5, 1e-30), (-2.5, 1e-30), (1e-30, 2.5)], np.float32)
    re = np.concatenate([re, re2, special[:, 0]])
    im = np.concatenate([im, im2, special[:, 1]])
    pad = (-len(re)) % 8
    re, im = np.concatenate([re, np.ones(pad, np.float32)]), np.concatenate([im, np.ones(pad, np.float32)])
    d = np.stack([re, im])[None, :, None, :]                                    # (1, 2, 1, n)
    out = ops.polar(torch.from_numpy(np.ascontiguousarray(d)).cuda()).cpu().numpy()[0, :, 0]
    z = d[0, 0, 0] + d[0, 1, 0] * 1j                 # data.py:40's own arithmetic, complex64 (an imaginary part of -0.0 becomes +0.0)
    want_ang = np.angle(z.astype(np.complex128))
    want_mag = np.log1p(np.abs(z.astype(np.complex128)))
    finite = np.isfinite(want_mag)
    assert np.max(np.abs(out[1] - want_ang)) < 5e-7
    rel = np.abs(out[0][finite] - want_mag[finite]) / np.maximum(want_mag[finite], 1e-37)
    assert np.max(rel) < 6e-7, float(np.max(rel))
    k = len(special)
    sp = out[1][2 * n:2 * n + k]
    # origin (also with a -0.0 real part: data.py:40's complex arithmetic turns it into +0.0) and the axes are exact; (-1, -0.0) is the
    # reference's quirk: the imaginary -0.0 becomes +0.0, so the angle is +pi, not -pi
    assert sp[0] == 0.0 and sp[1] == 0.0 and sp[4] == 0.0 and sp[5] == np.float32(np.pi) and sp[8] == np.float32(np.pi)
    assert sp[6] == np.float32(np.pi / 2) and sp[7] == np.float32(-np.pi / 2)


@pytest.mark.parametrize("bins,frames,hop,nsig", [(1024, 256, 512, 3), (512, 64, 256, 2), (64, 36, 32, 2), (32, 44, 16, 1)])
def test_istft_synthesis_arithmetic(bins, frames, hop, nsig):
    """(1) mode 0 = (exp(m) - 1) e^{j phi} on v_exp_f32 and the polynomial sincos (pg_fastmath.h) equals mode 1 fed the same
    spectrum formed in float64; (2) the two transform schedules agree; (3) peak normalisation really ends at a peak of 1."""
    from phasegen import ops
    m = np.abs(detgen.normal(41, (nsig, bins, frames))).astype(np.float32) * 2.0
    phi = (detgen.normal(42, (nsig, bins, frames)) * 4.0).astype(np.float32)          # beyond (-pi, pi]: the network's output is unbounded
    zr = (np.expm1(m.astype(np.float64)) * np.cos(phi.astype(np.float64))).astype(np.float32)
    zi = (np.expm1(m.astype(np.float64)) * np.sin(phi.astype(np.float64))).astype(np.float32)
    for norm in (False, True):
        y0 = ops.istft(torch.from_numpy(m).cuda(), torch.from_numpy(phi).cuda(), hop, mode=0, normalize=norm).cpu().numpy()
        y1 = ops.istft(torch.from_numpy(zr).cuda(), torch.from_numpy(zi).cuda(), hop, mode=1, normalize=norm).cpu().numpy()
        assert relmax(y0, y1) < 1e-5
        y2 = ops.istft(torch.from_numpy(zr).cuda(), torch.from_numpy(zi).cuda(), hop, mode=1, normalize=norm, single_frame=True).cpu().numpy()
        assert relmax(y1, y2) < 1e-5
        if norm:
            assert np.allclose(np.abs(y1).max(axis=1), 1.0, atol=1e-6)


@pytest.mark.parametrize("n,n_fft,hop,nsig", [(66100, 2048, 512, 3), (10007, 512, 128, 2), (3001, 256, 50, 1), (97, 32, 8, 5)])
def test_batched_and_single_frame_transforms_agree(n, n_fft, hop, nsig):
    """The two transform schedules (4 frames per workgroup through the half-length radix-4 real FFT / one frame per
    workgroup, full-length radix-2) follow the same framing; frame counts here are not multiples of 4 and hops are not
    multiples of anything in particular, so the ragged last group and the unaligned loads are covered."""
    from phasegen import ops
    y = torch.from_numpy(np.stack([detgen.make_clip(n, seed=70 + i) for i in range(nsig)])).cuda()
    try:
        ops.set_stft_mode(1)
        S1 = ops.stft(y, n_fft, hop)
        lm = torch.log1p(torch.sqrt(S1[:, 0] ** 2 + S1[:, 1] ** 2)).contiguous()
        ph = torch.atan2(S1[:, 1], S1[:, 0]).contiguous()
        r1 = ops.istft(S1[:, 0].contiguous(), S1[:, 1].contiguous(), hop, mode=1, normalize=False)
        e1 = ops.istft(lm, ph, hop, mode=0, normalize=True)
        ops.set_stft_mode(0)
        S0 = ops.stft(y, n_fft, hop)
        r0 = ops.istft(S1[:, 0].contiguous(), S1[:, 1].contiguous(), hop, mode=1, normalize=False)
        e0 = ops.istft(lm, ph, hop, mode=0, normalize=True)
    finally:
        ops.set_stft_mode(0)
    want = np.stack([signal_ref.chunk_and_stft(y[i].cpu().numpy(), n_fft, hop) for i in range(nsig)])
    assert relmax(S0, want) < 2e-5 and relmax(S1, want) < 2e-5
    assert relmax(r0, r1.cpu().numpy()) < 2e-5 and relmax(e0, e1.cpu().numpy()) < 5e-5


def test_generate_audio_and_fused_synthesis_vs_golden(golden_dir):
    from phasegen import audio
    g = np.load(os.path.join(golden_dir, "demo_g5.npz"))
    C = 16
    hyb = signal_ref.hybrid_spectrum(g["polar"][0], g["pred"][:C])
    a = audio.generate_audio(hyb, 16000, 8, is_stft=True)               # demo.py:40 call shape
    assert a.dtype == np.float32 and a.shape == g["audio"].shape
    assert relmax(a, g["audio"]) < 5e-5 and abs(np.max(np.abs(a)) - 1) < 1e-6
    b = audio.generate_audio(g["spec"], 16000, 8)                       # is_stft=False: [re; im] planes
    assert relmax(b, signal_ref.generate_audio(g["spec"], 8)) < 5e-5
    lm = torch.from_numpy(g["polar"][0][None]).cuda()
    ph = torch.from_numpy(g["pred"][None, :C]).cuda()
    c = audio.synthesize(lm, ph, 8).cpu().numpy()[0]                    # demo.py:39-40 fused on device
    assert relmax(c, g["audio"]) < 5e-5
    silent = audio.generate_audio(np.zeros((2, 16, 24), np.float32), 16000, 8)
    assert np.all(silent == 0)
    with pytest.raises(ValueError):
        bad = g["spec"].copy(); bad[0, 3, 3] = np.inf
        audio.generate_audio(bad, 16000, 8)


def test_stft_istft_round_trip_full_size():
    """Size-independent property at the BASELINE shapes: ISTFT(STFT(y)) == y away from the clip edges."""
    from phasegen import ops
    for n, n_fft, hop in ((65024, 2048, 512), (65280, 1024, 256)):
        y = torch.from_numpy(np.stack([detgen.make_clip(n, seed=40 + i) for i in range(4)])).cuda()
        S = ops.stft(y, n_fft, hop)
        r = ops.istft(S[:, 0].contiguous(), S[:, 1].contiguous(), hop, mode=1, normalize=False)
        # the DC bin is dropped by the pipeline, so compare against the DC-free signal: remove per-frame means ~ use interior
        S0 = signal_ref.stft(y[0].cpu().numpy(), n_fft, hop); S0[0] = 0
        w = signal_ref.istft(S0, hop)
        assert relmax(r[0], w) < 5e-5


def test_chunk_and_stft_pads_tail_like_reference():
    from phasegen import audio
    y = detgen.make_clip(300, seed=50)
    out = audio.chunk_and_stft(y[None], 200, 184, 32, 8).cpu().numpy()   # only 100 samples left: zero-padded to 184
    chunk = np.zeros(184, np.float32); chunk[:100] = y[200:300]
    assert relmax(out[0], signal_ref.chunk_and_stft(chunk, 32, 8)) < 2e-5


def test_loader_semantics(tmp_path):
    from phasegen.data import get_fft_npy_loader
    d = detgen.normal(60, (5, 2, 16, 24))
    p = str(tmp_path / "Pop_audio_train.npy")
    np.save(p, d)
    torch.manual_seed(0)
    loader = get_fft_npy_loader([p, str(tmp_path / "missing.npy")], [0, 1], batch_size=2, precon=True)
    batches = list(loader)
    assert len(loader) == 3 and [b[0].shape[0] for b in batches] == [2, 2, 1]        # short last batch kept (train.py:38 drops it)
    x, lab = batches[0]
    assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape[1:]) == (2, 16, 24) and tuple(lab.shape) == (2, 1)
    assert float(lab.abs().max()) == 0.0                                              # zip truncation -> label 0 (train.py:18-20)
    allx = torch.cat([b[0] for b in batches]).cpu().numpy()
    want = signal_ref.get_spec_and_angle(d).astype(np.float32)
    order = [int(np.argmin([np.abs(allx[i] - want[j]).max() for j in range(5)])) for i in range(5)]
    assert sorted(order) == [0, 1, 2, 3, 4]                                           # a permutation of the clips
    assert np.max(np.abs(allx - want[order])) < 2e-6
    first = loader.__iter__().__next__()[0]                                           # demo.py:28 access pattern
    assert first.shape[0] == 2
    r0 = get_fft_npy_loader(p, batch_size=2, precon=True, rank=0, world=2, seed=7)
    r1 = get_fft_npy_loader(p, batch_size=2, precon=True, rank=1, world=2, seed=7)
    n0 = sum(b[0].shape[0] for b in r0); n1 = sum(b[0].shape[0] for b in r1)
    assert (n0, n1) == (2, 2)            # clips r::W of one shared permutation cut to whole global batches (2 x 2 of 5 clips)
    with pytest.raises(AssertionError):
        get_fft_npy_loader([str(tmp_path / "nope.npy")])


@pytest.mark.parametrize("n_fft,hop,frames,n_iter", [(64, 16, 32, 4), (2048, 512, 128, 3), (1024, 256, 24, 2)])
def test_griffin_lim_vs_oracle(n_fft, hop, frames, n_iter):
    """Row N1 (utils.py:85-134), including the reference's quirk of inverting the DC-dropped matrix with n_fft - 2 points
    (2046 for 2048): on device that inverse transform is a 1x1 convolution with a synthesis matrix on the MFMA kernel."""
    from phasegen import audio
    n = hop * (frames - 1)
    y = detgen.make_clip(n, seed=60)
    mag = np.abs(np.delete(signal_ref.stft(y, n_fft, hop), 0, axis=0)).astype(np.float32)
    init = detgen.normal(61, (n,)).astype(np.float64)
    want_a, want_s, want_l = signal_ref.griffin_lim(mag, n_fft, hop, n_iter, init)
    got_a, got_s, got_l = audio.griffin_lim(mag, n_fft, hop, n_iter, init=init)
    assert got_a.shape == want_a.shape == (n,) and got_s.shape == want_s.shape
    # Tolerance: each iteration is stft -> angle -> istft; the device's inverse is an fp32 GEMM against a 2046-point synthesis
    # matrix rounded to fp32 (the oracle works in float64), and angle() of small bins amplifies ~1e-6 errors, which compound over
    # the iterations.  Measured on MI355X: audio 2e-6 .. 7e-6, spectrum 4e-6 .. 3.4e-5 for these cases; bound 2e-4 (round 1
    # allowed 2e-3).
    ea, es = relmax(got_a, want_a), float(np.max(np.abs(got_s - want_s)) / np.max(np.abs(want_s)))
    print(f"\ngriffin_lim n_fft={n_fft} iters={n_iter}: audio {ea:.1e} spec {es:.1e} loss {abs(got_l - want_l) / abs(want_l):.1e}")
    assert ea < 2e-4 and es < 2e-4
    assert abs(got_l - want_l) < 1e-5 * abs(want_l)
    assert abs(np.max(np.abs(got_a)) - 1.0) < 1e-6
    a2, _, _ = audio.griffin_lim(mag, n_fft, hop, n_iter, seed=7)      # seeded random start: reproducible
    a3, _, _ = audio.griffin_lim(mag, n_fft, hop, n_iter, seed=7)
    assert np.array_equal(a2, a3)


@pytest.mark.parametrize("n_fft,hop,frames,n_iter,n_clips", [(64, 16, 32, 5, 5), (2048, 512, 128, 3, 3), (256, 64, 40, 2, 70)])
def test_griffin_lim_batch_equals_the_clips_one_by_one(n_fft, hop, frames, n_iter, n_clips):
    """n clips cost the launches of one (STFT, projection, the 2046-point inverse GEMM and the overlap-add all
    carry the clip axis), and clip c of the batch is BIT-IDENTICAL to the clip run alone -- no kernel reduces across clips.
    70 clips exercise the 64-clip chunking of the overlap-add's peak words."""
    import torch
    from phasegen import audio
    n = hop * (frames - 1)
    mags = np.stack([np.abs(np.delete(signal_ref.stft(detgen.make_clip(n, seed=300 + c), n_fft, hop), 0, axis=0)) for c in range(n_clips)])
    mags = torch.from_numpy(mags.astype(np.float32)).cuda()
    init = torch.from_numpy(np.stack([detgen.normal(400 + c, (n,)) for c in range(n_clips)]).astype(np.float32))
    a, s, l = audio.griffin_lim_batch(mags, n_fft, hop, n_iter, init=init)
    assert tuple(a.shape) == (n_clips, n) and tuple(s.shape) == (n_clips, 2, n_fft // 2, frames) and tuple(l.shape) == (n_clips,)
    for c in sorted({0, 1, n_clips // 2, n_clips - 1}):
        a1, s1, l1 = audio.griffin_lim_batch(mags[c:c + 1], n_fft, hop, n_iter, init=init[c:c + 1])
        assert torch.equal(a[c], a1[0]) and torch.equal(s[c], s1[0]) and torch.equal(l[c], l1[0])
    assert float((a.abs().amax(dim=1) - 1).abs().max()) < 1e-6
    # seeded starts: clip c of a batch draws with seed + c, so a batch reproduces single-clip calls with those seeds
    b, _, _ = audio.griffin_lim_batch(mags[:2], n_fft, hop, n_iter, seed=11)
    b1, _, _ = audio.griffin_lim(mags[1].cpu().numpy(), n_fft, hop, n_iter, seed=12)
    assert np.array_equal(b[1].cpu().numpy(), b1)


@pytest.mark.parametrize("n_fft,hop,t_slice,single", [(64, 16, 496, 0), (64, 16, 496, 1), (2048, 512, 65024, 0), (256, 64, 1001, 0)])
def test_stft_reads_chunks_in_place(n_fft, hop, t_slice, single):
    """pg_stft_args.chunk_start / chunk_row (row N2, preproc_mdb.py:84-97): the STFT of chunks read in place from a
    (channels, samples) source -- odd starts, a chunk that runs off the end (zero tail, :86-88), a chunk that begins at the
    last sample -- equals, BIT FOR BIT, the STFT of the same chunks gathered and zero-padded on the host."""
    from phasegen import ops
    a_len = 3 * t_slice + 777
    src = np.stack([detgen.make_clip(a_len, seed=60), detgen.make_clip(a_len, seed=61)])
    starts = [0, 1, t_slice, 2 * t_slice + 333, 3 * t_slice, a_len - 5, a_len - 1, 12345 % a_len]
    rows = [0, 1, 1, 0, 1, 0, 1, 0]
    padded = np.concatenate([src, np.zeros((2, t_slice), np.float32)], axis=1)
    gathered = np.stack([padded[r, s:s + t_slice] for s, r in zip(starts, rows)])
    want = ops.stft(torch.from_numpy(gathered).cuda(), n_fft, hop, single_frame=single)
    got = ops.stft(torch.from_numpy(src).cuda(), n_fft, hop, single_frame=single, chunk_len=t_slice,
                   chunk_start=torch.tensor(starts, dtype=torch.int64, device="cuda"),
                   chunk_row=torch.tensor(rows, dtype=torch.int32, device="cuda"))
    assert got.shape == want.shape == (8, 2, n_fft // 2, 1 + t_slice // hop)
    assert torch.equal(got, want)
    ref = signal_ref.chunk_and_stft(gathered[3], n_fft, hop)                       # and one chunk against the oracle
    assert np.max(np.abs(got[3].cpu().numpy() - ref)) < 2e-5 * np.max(np.abs(ref))
    mono = ops.stft(torch.from_numpy(src[0]).cuda(), n_fft, hop, chunk_len=t_slice, single_frame=single,
                    chunk_start=torch.tensor([starts[3]], dtype=torch.int64, device="cuda"))       # chunk_row NULL -> row 0
    assert torch.equal(mono[0], got[3])
    with pytest.raises(TypeError):
        ops.stft(torch.from_numpy(src).cuda(), n_fft, hop, chunk_len=t_slice, chunk_start=torch.tensor(starts, device="cuda").int())


@pytest.mark.parametrize("n,off", [(1 << 20, 0), (1000003, 0), (77, 0), (999999, 1), (64 * 2 * 1024 * 128, 0)])
def test_standardize_matches_numpy(n, off):
    """preproc_mdb.py:182 `(x - x.mean()) / x.std()`: moments against float64 numpy to 1e-12, the result against the float32
    formula with those moments to one rounding; odd sizes and a 4-byte-aligned (not 16) start take the scalar path."""
    from phasegen import ops
    x = (detgen.normal(70 + off, (n + off,)) * 3.0 + 1.5).astype(np.float32)
    base = torch.from_numpy(x).cuda()
    view = base[off:]
    stats = ops.standardize_(view).cpu().numpy()
    x64 = x[off:].astype(np.float64)
    assert abs(stats[0] - x64.mean()) < 1e-12 * max(1.0, abs(x64.mean())) * 10 and abs(stats[1] - x64.std()) < 1e-11 * x64.std()
    want = (x[off:] - np.float32(stats[0])) / np.float32(stats[1])
    assert np.max(np.abs(view.cpu().numpy() - want)) <= 2.4e-7 * np.max(np.abs(want))
    if off:
        assert float(base[0]) == float(x[0])                                     # nothing before the view was touched
    npy = (x[off:] - x[off:].mean()) / x[off:].std()                              # numpy's own float32 reduction
    assert np.max(np.abs(view.cpu().numpy() - npy)) < 5e-6 * np.max(np.abs(npy))


def test_preproc_chunker_matches_reference_algorithm(tmp_path):
    """Row N2 (preproc_mdb.py:66-97,174-196): same chunk starts, zero-padded tails, STFT layout, whole-array
    normalisation and shuffled split as a numpy restatement driven by the same Generator."""
    from phasegen import preproc
    n_fft, hop, rsr, sec = 64, 16, 1000, 0.496                  # t_slice = 496 samples -> 32 frames
    t_slice = int(sec * rsr)
    tracks = [detgen.make_clip(1300, seed=80), detgen.make_clip(700, seed=81)]
    train, val = preproc.build_dataset(tracks, sec, rsr, n_fft, hop, n_random=2, n_val=3, seed=5, out_dir=str(tmp_path))
    # numpy restatement of preproc_mdb.py:66-97 written out HERE (not through phasegen.preproc): one aligned chunk every
    # t_slice samples, each followed by n_random crops starting at randint(0, a_len - t_slice // 1.3); a tail shorter than
    # t_slice is zero-padded (:86-88).  The reference draws from the global np.random state; the product takes a Generator,
    # so the restatement draws from an identically seeded one in the same order.
    rng = np.random.default_rng(5)
    chunks, n_starts = [], 0
    for a in tracks:
        a_len = len(a)
        bnd = a_len - t_slice // 1.3                            # preproc_mdb.py:70 (float floor-division, kept as is)
        starts = []
        for i in range(0, a_len, t_slice):                      # :73
            starts.append(i)
            for _ in range(2):                                  # n_random = 2, :77-80
                starts.append(int(rng.integers(0, bnd)))
        n_starts += len(starts)
        for st in starts:
            c = a[st:st + t_slice]
            if len(c) < t_slice:
                c = np.pad(c, (0, t_slice - len(c)), "constant")
            chunks.append(signal_ref.chunk_and_stft(c, n_fft, hop))
    assert n_starts == (3 + 2) * 3 and starts[0] == 0 and starts[3] == t_slice      # 1300 -> 3 aligned chunks, 700 -> 2
    x = np.asarray(chunks, dtype=np.float32)
    x = (x - x.mean()) / x.std()
    idx = np.linspace(0, len(x) - 1, len(x), dtype=int)
    rng.shuffle(idx)
    assert val.shape == (3, 2, 32, 32) and train.shape == (len(x) - 3, 2, 32, 32) and train.dtype == np.float32
    assert np.max(np.abs(val - x[idx][:3])) < 5e-5 and np.max(np.abs(train - x[idx][3:])) < 5e-5
    assert os.path.exists(tmp_path / "Pop_audio_train.npy")
    from phasegen.data import get_fft_npy_loader                       # and the loader consumes what preproc wrote
    ld = get_fft_npy_loader([str(tmp_path / "Pop_audio_train.npy")], batch_size=4, precon=True)
    assert next(iter(ld))[0].shape == (4, 2, 32, 32)


def _random_stft_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        n_fft = int(2 ** rs.randint(5, 13))                      # 32 .. 4096
        hop = int(rs.choice([n_fft // 4, n_fft // 2, n_fft // 8, max(1, n_fft // 4 - 3), 50]))
        frames = int(rs.randint(2, 40))
        n_samp = hop * (frames - 1) + int(rs.randint(0, hop))
        if n_samp <= n_fft // 2 or n_samp > 200000:
            continue
        out.append((n_samp, n_fft, hop, int(rs.randint(1, 4))))
    return out


@pytest.mark.parametrize("n,n_fft,hop,nsig", _random_stft_cases(20, 7))
def test_stft_istft_random_sizes_vs_oracle(n, n_fft, hop, nsig):
    """20 seeded random (samples, n_fft, hop) combinations -- every power-of-two transform length, hops that are not n_fft / 4,
    sample counts that are not multiples of the hop, 2 .. 40 frames: STFT against the oracle in both transform schedules, and
    ISTFT of that spectrum against the oracle's ISTFT (librosa's conventions; values "parity unpinned", see the file header)."""
    from phasegen import ops
    y = np.stack([detgen.make_clip(n, seed=300 + i) for i in range(nsig)])
    want = np.stack([signal_ref.chunk_and_stft(y[i], n_fft, hop) for i in range(nsig)])
    yd = torch.from_numpy(y).cuda()
    for single in (0, 1):
        S = ops.stft(yd, n_fft, hop, single_frame=single)
        assert tuple(S.shape) == want.shape and relmax(S, want) < 2e-5
    frames = want.shape[3]
    if frames >= 2 and hop * 4 <= n_fft * 2:                     # the overlap-add needs a positive window sum everywhere it keeps
        back = ops.istft(S[:, 0].contiguous(), S[:, 1].contiguous(), hop, mode=1, normalize=False).cpu().numpy()
        for i in range(nsig):
            Z = np.concatenate([np.zeros((1, frames), np.complex64), (want[i, 0] + 1j * want[i, 1]).astype(np.complex64)], 0)
            w = signal_ref.istft(Z, hop)
            assert back[i].shape == w.shape and relmax(back[i], w) < 5e-5
