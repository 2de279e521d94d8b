"""BASELINE configs[4] as a test: waveform -> STFT fused with log1p|z| / angle -> U-Net forward -> ISTFT of
(exp(m) - 1) e^{j phi_pred}  (preproc_mdb.py:84-97, data.py:39-47, model.py forward, demo.py:36-40), every stage on the
device through the C ABI; and the re-entrancy contract of the ABI: two host threads on two streams, one
running fp32 and one bf16 convolutions at the same time, each checked against its own oracle.

STFT / ISTFT VALUES are "parity unpinned" against librosa (oracle/signal_ref.py restates its published definition; see its
header): what is pinned here is the composition against that restatement + the pinned U-Net oracle at a small n_fft, and,
at the full 2048 / 512 size, the size-independent round-trip property.
"""
import threading

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import signal_ref, unet_ref
from phasegen import detgen

pytestmark = pytest.mark.gpu


def relmax(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def device_chain(model, wav, n_fft, hop):
    from phasegen import audio, ops
    C = n_fft // 2
    polar = ops.stft(wav, n_fft, hop, polar=True)
    pred = model.engine.forward(polar[:, 0], update_stats=False, inference=True)    # precision bf16 -> bf16-resident kernels
    out = audio.synthesize(polar[:, 0], pred[:, :C], hop)
    return polar, pred, out


def test_e2e_chain_small_vs_oracle():
    """n_fft 32 / hop 8 -> C = 16 bins, 64 frames, 3 signals: the whole chain against signal_ref + unet_ref in fp32."""
    from phasegen.model import UNetModel
    n_fft, hop, L, nsig = 32, 8, 64, 3
    C, n = n_fft // 2, hop * (L - 1)
    y = np.stack([detgen.make_clip(n, seed=90 + i) for i in range(nsig)])
    pn = detgen.make_params(C, seed=0)
    model = UNetModel(C, 2 * C, precision="fp32").load_numpy(pn)
    polar, pred, out = device_chain(model, torch.from_numpy(y).cuda(), n_fft, hop)
    # oracle
    S = np.stack([signal_ref.chunk_and_stft(y[i], n_fft, hop) for i in range(nsig)])
    P = signal_ref.get_spec_and_angle(S).astype(np.float32)
    with torch.no_grad():
        want_pred = unet_ref.unet_forward(unet_ref.to_torch(pn), torch.from_numpy(P[:, 0].copy())).numpy()
    assert tuple(polar.shape) == P.shape == (nsig, 2, C, L)
    assert relmax(polar[:, 0], P[:, 0]) < 2e-5
    assert relmax(pred, want_pred) < 2e-4                      # 1e-4 forward parity + the 2e-5 of its input
    for i in range(nsig):
        want = signal_ref.generate_audio(signal_ref.hybrid_spectrum(P[i, 0], want_pred[i, :C]), hop, is_stft=True)
        assert out[i].shape == want.shape == (n,)
        assert relmax(out[i], want) < 1e-3                     # e^{j phi}: phase errors of 2e-4 x |phi| up to ~10 rad
        assert abs(float(out[i].abs().max()) - 1.0) < 1e-6
    # the same chain with bf16 MFMA operands (configs[4]'s arithmetic) stays close to the fp32 chain
    model_b = UNetModel(C, 2 * C, precision="bf16").load_numpy(pn)
    _, pred_b, out_b = device_chain(model_b, torch.from_numpy(y).cuda(), n_fft, hop)
    assert relmax(pred_b, pred) < 5e-2 and bool(torch.isfinite(out_b).all())


def test_e2e_chain_full_size_properties():
    """2048-FFT / 512-hop, 2 stereo clips = 4 signals of 130 560 samples -> (4, 1024, 256): size-independent properties.
      (1) with the TRUE angle in place of the network's, ISTFT(STFT(y)) returns the DC-free signal, peak-normalised;
      (2) the bf16-operand forward (configs[4]) stays within 3e-2 of the fp32 forward at C = 1024;
      (3) outputs are finite, peak 1, hop * (frames - 1) samples long."""
    from phasegen import audio, ops
    from phasegen.model import UNetModel
    n_fft, hop, n, nsig = 2048, 512, 255 * 512, 4
    C, L = n_fft // 2, 1 + n // hop
    y = np.stack([detgen.make_clip(n, seed=95 + i) for i in range(nsig)])
    wav = torch.from_numpy(y).cuda()
    polar = ops.stft(wav, n_fft, hop, polar=True)
    assert tuple(polar.shape) == (nsig, 2, C, L) and L == 256
    rt = audio.synthesize(polar[:, 0], polar[:, 1].contiguous(), hop).cpu().numpy()
    S0 = signal_ref.stft(y[0], n_fft, hop)
    S0[0] = 0                                                   # the pipeline drops the DC bin (preproc_mdb.py:93)
    w = signal_ref.istft(S0, hop)
    assert relmax(rt[0], w / np.max(np.abs(w))) < 2e-4
    torch.manual_seed(5)
    m32 = UNetModel(C, 2 * C, precision="fp32")
    m16 = UNetModel(C, 2 * C, precision="bf16")
    m16.engine.arena.flat.copy_(m32.engine.arena.flat)         # an in-place torch write: the bf16 shadows notice it by themselves
    p32 = m32.engine.forward(polar[:, 0], update_stats=False).clone()
    _, p16, out = device_chain(m16, wav, n_fft, hop)
    assert relmax(p16, p32) < 3e-2
    assert tuple(out.shape) == (nsig, hop * (L - 1)) and bool(torch.isfinite(out).all())
    assert float((out.abs().amax(dim=1) - 1).abs().max()) < 1e-6


def test_bf16_shadows_follow_in_place_parameter_writes_by_torch():
    """the bf16 weight shadows of the resident inference forward were keyed on the hand-bumped arena.version only, so
    an in-place write torch makes through a parameter view -- torch.optim.Adam(model.parameters()).step(), the reference's own
    optimiser (train.py:26), or p.data.copy_ -- left them stale.  They are now also keyed on arena.flat._version."""
    from phasegen.model import UNetModel
    C, L, B = 32, 64, 3
    torch.manual_seed(11)
    model = UNetModel(C, 2 * C, precision="bf16")
    x = torch.from_numpy(detgen.make_batch(B, C, L, seed=4)).cuda()
    with torch.no_grad():
        y0 = model.forward(x[:, 0]).clone()                    # builds the shadows
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)        # stock torch optimiser on the arena's parameter views
    pred = model.forward(x[:, 0])
    (pred * pred).mean().backward()
    opt.step()
    with torch.no_grad():
        y1 = model.forward(x[:, 0]).clone()                    # must multiply by the UPDATED weights
    fresh = UNetModel(C, 2 * C, precision="bf16")
    fresh.engine.arena.flat.copy_(model.engine.arena.flat)
    for k, v in model.engine.arena.buffers.items():
        fresh.engine.arena.buffers[k].copy_(v)
    with torch.no_grad():
        want = fresh.forward(x[:, 0])
    assert torch.equal(y1, want)
    assert relmax(y1, y0) > 1e-3                               # the step did move the output: a stale shadow would have given y0
    p0 = next(iter(model.parameters()))
    with torch.no_grad():
        p0.mul_(0.5)                                           # a plain in-place write under no_grad
        y2 = model.forward(x[:, 0]).clone()
        p0.detach().mul_(2.0)                                  # ... and one through a detached alias: back to y1's weights
        y3 = model.forward(x[:, 0]).clone()
    assert relmax(y2, y1) > 1e-3 and torch.equal(y3, y1)
    # (`p.data` hands out a tensor with its OWN version counter: a write through it is invisible to torch and needs arena.touch())
    with torch.no_grad():
        p0.data.mul_(0.5)
        model.engine.arena.touch()
        assert torch.equal(model.forward(x[:, 0]), y2)


@pytest.mark.parametrize("B,L,update", [(1, 128, False), (5, 64, True)])
def test_resident_forward_graph_replay_equals_eager(B, L, update):
    """engine.graphs: the bf16-resident inference forward behind the input cast is captured into a HIP graph on its second call at
    a shape and replayed afterwards.  Replays must be bit-identical to the eager launch sequence, follow new inputs and new
    weights (the shadows are rebuilt in place), and keep counting BatchNorm statistics when asked to."""
    from phasegen.model import UNetModel
    C = 64
    torch.manual_seed(3)
    m = UNetModel(C, 2 * C, precision="bf16")
    eng = m.engine
    xs = [torch.from_numpy(detgen.make_batch(B, C, L, seed=40 + i))[:, 0].contiguous().cuda() for i in range(4)]
    want = [eng.forward(x, update_stats=False, inference=True).clone() for x in xs]
    nb0 = int(eng.arena.buffers[detgen.BN_KEYS[0] + ".num_batches_tracked"])
    eng.graphs = True
    got = [eng.forward(x, update_stats=update, inference=True).clone() for x in xs]       # eager, capture + replay, replay, replay
    assert all(torch.equal(g, w) for g, w in zip(got, want))
    from phasegen import ops
    graph, held_ws = eng.plans[("graph", B, L, update, ops.current_schedule(), eng.precision)]     # key = everything a capture freezes
    assert isinstance(graph, torch.cuda.CUDAGraph) and held_ws.numel() > 0                          # the workspace lives with the graph
    ops.release_workspaces()                                   #  dropping the caches must not pull memory from under the graph
    assert int(eng.arena.buffers[detgen.BN_KEYS[0] + ".num_batches_tracked"]) == nb0 + (4 if update else 0)
    with torch.no_grad():
        next(iter(m.parameters())).mul_(0.5)                   # new weights: the replay must see the rebuilt shadows
    y_new = eng.forward(xs[0], update_stats=update, inference=True).clone()
    eng.graphs = False
    assert torch.equal(y_new, eng.forward(xs[0], update_stats=False, inference=True)) and not torch.equal(y_new, want[0])


def test_resident_forward_falls_back_where_windows_do_not_fit():
    """Many very short samples per 256-column tile (B = 64 clips of 24 frames) exceed the bf16-resident kernels' window
    slots: the engine must notice BEFORE launching anything (pg_conv_fwd_h_supported) and run the fp32-tensor kernels with
    bf16 operands instead, not raise half-way through with running statistics already updated."""
    from phasegen import ops
    from phasegen.model import UNetModel
    C, B, L = 16, 64, 24
    pn = detgen.make_params(C, seed=3)
    m = UNetModel(C, 2 * C, precision="bf16").load_numpy(pn)
    assert m.engine.resident_ok() and not m.engine.resident_ok(B, L) and m.engine.resident_ok(4, 64)
    x = torch.from_numpy(detgen.make_batch(B, C, L, seed=4)[:, 0].copy()).cuda()
    got = m.engine.forward(x, update_stats=False, inference=True)
    with torch.no_grad():
        want = unet_ref.unet_forward(unet_ref.to_torch(pn), x.cpu()).numpy()
    assert relmax(got, want) < 5e-2 and bool(torch.isfinite(got).all())
    shp = detgen.conv_shapes(1024)
    assert ops.conv_fwd_h_supported(64, shp[detgen.K_U0], 129, 2, 16, True)
    assert not ops.conv_fwd_h_supported(64, shp[detgen.K_U0], 13, 2, 16, True)


def conv_oracle64(x, w, k, s, p, tr, bf16):
    """float64 convolution of the operands as the kernel sees them (bf16 mode: rounded RNE first)."""
    r = (lambda t: t.to(torch.bfloat16).double()) if bf16 else (lambda t: t.double())
    return (F.conv_transpose1d if tr else F.conv1d)(r(x), r(w), stride=s, padding=p)


def test_two_threads_two_streams_two_precisions():
    """The ABI keeps no process-wide state (include/phasegen.h): precision and schedule travel in pg_conv_args and every
    stream has its own stream-K workspace.  Two host threads, each on its own HIP stream, run a stream-K-split convolution
    40 times concurrently -- one in fp32, the other with bf16 operands -- and each result matches ITS oracle (the fp32
    result differs from the bf16 oracle by far more than the tolerance, so a leaked precision would be caught)."""
    from phasegen import ops
    B, Cin, Cout, k, s, p, Lin = 4, 512, 256, 8, 2, 1, 61          # an up-conv (T form); schedule 2 forces the split + fixup
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, Lin, generator=g)
    w = torch.randn(Cin, Cout, k, generator=g) * 0.05
    want = {"fp32": conv_oracle64(x, w, k, s, p, True, False), "bf16": conv_oracle64(x, w, k, s, p, True, True)}
    assert relmax(want["fp32"], want["bf16"]) > 1e-3
    xd, wd = x.cuda(), w.cuda()
    torch.cuda.synchronize()
    results, errors = {}, []
    start = threading.Barrier(2)

    def run(prec):
        try:
            st = torch.cuda.Stream()
            y = torch.empty(B, Cout, ops.convt_out_len(Lin, k, s, p), device="cuda")
            ops.set_conv_precision(prec)                           # THIS thread's default; the other thread keeps its own
            with torch.cuda.stream(st):
                start.wait()
                for _ in range(40):
                    ops.conv_fwd(xd, wd, y, s, p, transposed=True, schedule=2)
            st.synchronize()
            results[prec] = y.cpu()
        except Exception as e:                                      # noqa: BLE001
            errors.append((prec, repr(e)))

    ts = [threading.Thread(target=run, args=(prec,)) for prec in ("fp32", "bf16")]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert relmax(results["fp32"], want["fp32"]) < 2e-5
    assert relmax(results["bf16"], want["bf16"]) < 2e-5
    assert ops._tls.precision == 0                                  # the main thread's default was never touched
    assert len({k for k in ops._conv_ws.d if k[0].type == "cuda"}) >= 2   # one stream-K workspace per stream


def test_tensor_on_another_device_is_refused_not_faulted():
    """launching on the current device with a tensor of another device would be a GPU memory fault.  With one
    GPU the mismatch cannot be built from real tensors; the guard itself is exercised through a stand-in."""
    from phasegen import ops

    class Fake:
        device = torch.device("cuda", torch.cuda.current_device() + 1)
    with pytest.raises(ValueError, match="current device"):
        ops._on_current_device(Fake(), "x")


def test_integration_md_stub_is_a_working_binding():
    """INTEGRATION.md §B shows the ctypes stub a maintainer of the reference would write against include/phasegen.h.  The
    block is executed verbatim here (so the document cannot drift from the header) and its conv1d_fwd is compared with
    stock fp32 torch on the CPU: nn.Conv1d(...)(LeakyReLU(0.2)(x)) of model.py:77-80."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# pg_stub\.py.*?)```", text, flags=re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)                                   # the stub loads the library by its path relative to the repository root
    try:
        exec(compile(block, "INTEGRATION.md:pg_stub", "exec"), ns)
    finally:
        os.chdir(cwd)
    import ctypes
    assert ctypes.sizeof(ns["ConvArgs"]) == 208
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 16, 64, generator=g)
    w = torch.randn(32, 16, 8, generator=g) * 0.1
    got = ns["conv1d_fwd"](x.cuda(), w.cuda(), stride=2, pad=1, act=1)          # act 1 = LeakyReLU(0.2) on the input, fused
    want = F.conv1d(F.leaky_relu(x, 0.2), w, stride=2, padding=1)
    assert got.shape == want.shape and relmax(got, want) < 1e-5
    with pytest.raises(RuntimeError):
        ns["conv1d_fwd"](x.cuda(), w.cuda(), stride=2, pad=1, precision=7)         # PG_ERR_UNSUPPORTED -> RuntimeError with the message
