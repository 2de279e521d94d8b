"""Thin host-side wrappers: torch (ROCm) tensors in, one libphasegen C-ABI call out.

PyTorch is plumbing here (device memory + the current HIP stream); every op below is a hand-written gfx950
kernel behind ``include/phasegen.h``.  Activation tensors are (B, channels, frames) views whose frame stride is 1
and channel stride is ``frames``; the batch stride is free, so the two halves of a U-Net concat buffer are passed
as plain slices ``buf[:, :n]`` / ``buf[:, n:]`` without a copy.
"""
import contextlib
import ctypes as C
import threading

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, SCHED_CONTENDED, SCHED_TILE_PER_WG  # noqa: F401  (re-exported)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_current_device(t, name):
    """Kernels are launched on the CURRENT device's current stream: a tensor that lives on another GPU would make the
    kernel dereference foreign pointers (a GPU memory fault that aborts the process) -- raise instead."""
    if t.device.index != torch.cuda.current_device():
        raise ValueError(f"{name}: tensor is on {t.device} but the current device is cuda:{torch.cuda.current_device()} "
                         "(wrap the call in torch.cuda.device(...))")


class KernelTimer:
    """HIP-event timing of labelled launches on the stream they are launched on (bench.py's roofline leg), plus each
    label's launch plan (pg_conv_describe: the kernel symbol rocprofv3 will report for it)."""

    def __init__(self):
        self.records = {}
        self.plans = {}
        self.bytes = {}          # label -> algorithmic HBM bytes of ONE launch (the HBM-bound kernels: BatchNorm, loss, Adam)

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.records.items()}


_timer = None
_cur_label = None


def set_timer(t):
    global _timer
    _timer = t


def conv_describe(a, op):
    """pg_conv_describe: 'kernel<...>|grid=G|tiles=T|slabs=S|split=0/1|whole=W|fixup=none/plain/wide' for a filled ConvArgs (nothing is launched)."""
    buf = C.create_string_buffer(256)
    _lib.check(_lib.load().pg_conv_describe(C.byref(a), op, buf, 256), "conv_describe")
    return buf.value.decode()


def _note_plan(a, op):
    if _timer is not None and _cur_label is not None and _cur_label not in _timer.plans:
        _timer.plans[_cur_label] = conv_describe(a, op)


class timed:
    """``with ops.timed(label[, nbytes])``: one HIP-event pair around the launches inside, on the stream they go to (only while a
    KernelTimer is installed; otherwise free).  ``nbytes``: algorithmic HBM bytes of the launch, for the HBM-side rooflines."""

    def __init__(self, label, nbytes=None):
        self.label = label
        self.nbytes = nbytes

    def __enter__(self):
        global _cur_label
        if _timer is not None:
            _cur_label = self.label
            if self.nbytes is not None:
                _timer.bytes[self.label] = self.nbytes
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        global _cur_label
        if _timer is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _timer.records.setdefault(self.label, []).append((self.a, b))
            _cur_label = None


def _act3(t, name):
    """(ptr, batch_stride) of a (B, C, L) fp32 device view with L-contiguous channels."""
    if t is None:
        return None, 0
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3):
        raise ValueError(f"{name}: expected a 3-D float32 device tensor, got {tuple(t.shape)} {t.dtype} {t.device}")
    _on_current_device(t, name)
    B, Cc, L = t.shape
    if t.stride(2) != 1 or (Cc > 1 and t.stride(1) != L):
        raise ValueError(f"{name}: frames must be contiguous and channel stride == frames (strides {t.stride()})")
    return t.data_ptr(), (t.stride(0) if B > 1 else Cc * L)


def _dense(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous float32 device tensor")
    _on_current_device(t, name)
    return t.data_ptr()


def _dense_as(t, dtype, name):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous {dtype} device tensor")
    _on_current_device(t, name)
    return t.data_ptr()


def conv_out_len(Lin, k, s, p):
    return (Lin + 2 * p - k) // s + 1


def convt_out_len(Lin, k, s, p):
    return (Lin - 1) * s - 2 * p + k


class _StreamCache:
    """Scratch buffers keyed by (device, stream handle), bounded: at most ``cap`` streams per cache, least recently used evicted.
    A buffer is allocated while its stream is the CURRENT one, so torch's caching allocator ties the block to that stream: an
    evicted (or released) buffer goes back to that stream's pool, and whoever reuses the block is ordered after the stream's
    pending work -- also when a stream handle is recycled after the evicted entry's stream died."""

    def __init__(self, cap=4):
        self.cap = cap
        self.d = {}

    def get(self, device, nbytes_fn, at_least=0):
        device = torch.device(device)
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        buf = self.d.pop(key, None)                     # re-inserted below: dicts keep insertion order -> LRU
        if buf is None or buf.numel() < at_least:
            buf = torch.empty(max(nbytes_fn(), at_least), device=device, dtype=torch.uint8)
        self.d[key] = buf
        while len(self.d) > self.cap:
            self.d.pop(next(iter(self.d)))
        return buf

    def clear(self):
        self.d.clear()


_conv_ws = _StreamCache()
_caches = [_conv_ws]


def _ws_key(device):
    """Workspaces are owned by one stream at a time: key the caches by (device, current stream)."""
    device = torch.device(device)
    return (device, torch.cuda.current_stream(device).cuda_stream)


def conv_workspace(device):
    """Scratch for the conv kernels' stream-K schedule (pg_workspace_bytes_conv(), 512 MiB), one per (device, STREAM):
    launches on one stream are ordered and may share it, launches on different streams may overlap and must not.  At most four
    streams per device hold one at a time (least recently used is dropped); ``release_workspaces()`` drops them all."""
    return _conv_ws.get(device, _lib.load().pg_workspace_bytes_conv)


def release_workspaces():
    """Drop every cached scratch buffer (conv stream-K, loss, ISTFT, overlap-add, moments): they are re-created on demand."""
    for c in _caches:
        c.clear()


# ---- per-call knobs ---------------------------------------------------------------------------------------------
# The C ABI keeps no state: MFMA operand precision and work-split schedule are fields of pg_conv_args, the FFT schedule
# a field of pg_stft_args / pg_istft_args.  Every op below takes them as keyword arguments; when omitted, the calling
# THREAD's defaults apply (a fresh thread starts at fp32 / automatic), so two threads can drive two streams in different
# precisions.  The engine passes its own precision explicitly.
_PRECISIONS = {"fp32": 0, "f32": 0, "bf16": 1, "bf16x3": 2, 0: 0, 1: 1, 2: 2}


class _Defaults(threading.local):
    precision = 0
    schedule = 0
    stft_single = 0


_tls = _Defaults()


def precision_code(mode):
    if mode not in _PRECISIONS:
        raise ValueError(f"conv precision {mode!r}: expected fp32 / bf16 / bf16x3 (or 0 / 1 / 2)")
    return _PRECISIONS[mode]


def set_conv_precision(mode):
    """This thread's default MFMA operand mode.  0 / "fp32": fp32 operands (the parity path).  1 / "bf16": operands rounded
    to bf16 at fragment load, fp32 accumulate, fp32 tensors and master weights (BASELINE config 5).  2 / "bf16x3": split."""
    _tls.precision = precision_code(mode)


def current_schedule():
    """This thread's default pg_conv_args.schedule word."""
    return _tls.schedule


def set_conv_schedule(mode):
    """This thread's default schedule bits (test hook).  Bits 0-1: 0 automatic, 1 one tile per workgroup, 2 force the stream-K
    split; bit 2 (value 4): im2col kernels instead of the raw-window ones; bit 3 (value 8): never the tall 256 x 128 raw tile;
    bit 7 (value 128): the wgrad keeps the flat-K raw kernel where it would take the per-sample-slab one; bit 13 (value 0x2000):
    never the one-wave-per-SIMD fp32 kernels (conv_raw3.hip), i.e. the two-waves-per-SIMD raw kernels everywhere; bit 14 (0x4000):
    those kernels wherever they cover the problem, also where the automatic choice keeps the older ones (F form of k = 32); bits 15-16:
    their tile order (1 << 15: row-major, 2 << 15 / 3 << 15: super-rows of 2 / 4 tile rows); bit 17 (0x20000): never split the columns past the
    last full 256-wide tile off into a tail launch, bit 18 (0x40000): always, where the geometry allows (the automatic choice prices it)."""
    if mode < 0 or (mode & ~0x7e0ff) or (mode & 0x60000) == 0x60000 or (mode & 3) == 3 or (mode & 0x70) or (mode & 0x6000) == 0x6000:
        raise ValueError("conv schedule: bad mode")
    _tls.schedule = (_tls.schedule & 0xf00) | mode


def set_conv_oversubscribe(factor):
    """Stream-K grid = factor x resident workgroup slots (1..8).  Use > 1 when other kernels (RCCL collectives) share the chip."""
    if not 1 <= factor <= 8:
        raise ValueError("conv oversubscribe: factor must be 1..8")
    _tls.schedule = (_tls.schedule & ~0xf00) | (factor << 8)


def set_stft_mode(single_frame):
    """0: 4 frames per workgroup through the half-length radix-4 real FFT (default); 1: one frame per workgroup, radix-2."""
    _tls.stft_single = int(bool(single_frame))


@contextlib.contextmanager
def conv_options(precision=None, schedule=None):
    """Scoped thread defaults: ``with ops.conv_options(precision="bf16"): ...``."""
    old = (_tls.precision, _tls.schedule)
    if precision is not None:
        _tls.precision = precision_code(precision)
    if schedule is not None:
        _tls.schedule = schedule
    try:
        yield
    finally:
        _tls.precision, _tls.schedule = old


def _conv_args(transposed, B, Cin, Cout, Lin, k, s, p, device=None, precision=None, schedule=None):
    a = _lib.ConvArgs()
    a.precision = _tls.precision if precision is None else precision_code(precision)
    a.schedule = _tls.schedule if schedule is None else schedule
    a.B, a.Cin, a.Cout, a.Lin, a.k, a.stride, a.pad = B, Cin, Cout, Lin, k, s, p
    a.Lout = convt_out_len(Lin, k, s, p) if transposed else conv_out_len(Lin, k, s, p)
    if device is not None:
        ws = conv_workspace(device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    return a


def _geom(transposed, w, k=None):
    if transposed:
        Cin, Cout, kk = w.shape
    else:
        Cout, Cin, kk = w.shape
    return Cin, Cout, kk


def conv_fwd(x, w, y, stride, pad, x_act=ACT_NONE, transposed=False, y_act=ACT_NONE, y2=None, y2_act=ACT_NONE,
             precision=None, schedule=None):
    """y = y_act(conv(x_act(x), w)) (nn.Conv1d, model.py:77) or conv_transpose (model.py:88) -- writes into ``y``;
    optionally a second copy ``y2 = y2_act(conv(...))`` (pre-activated tensors for the consumers)."""
    Cin, Cout, k = _geom(transposed, w)
    B, _, Lin = x.shape
    a = _conv_args(transposed, B, Cin, Cout, Lin, k, stride, pad, x.device, precision, schedule)
    if tuple(x.shape) != (B, Cin, Lin) or tuple(y.shape) != (B, Cout, a.Lout):
        raise ValueError(f"conv_fwd: shapes x{tuple(x.shape)} w{tuple(w.shape)} y{tuple(y.shape)} inconsistent (Lout {a.Lout})")
    a.x, a.x_bs = _act3(x, "x")
    a.y, a.y_bs = _act3(y, "y")
    a.w = _dense(w, "w")
    a.x_act, a.y_act, a.y2_act = x_act, y_act, y2_act
    if y2 is not None:
        if y2.shape != y.shape:
            raise ValueError("conv_fwd: y2 must have y's shape")
        a.y2, a.y2_bs = _act3(y2, "y2")
    lib = _lib.load()
    _note_plan(a, _lib.OP_CONVT1D_FWD if transposed else _lib.OP_CONV1D_FWD)
    fn = lib.pg_convt1d_fwd if transposed else lib.pg_conv1d_fwd
    _lib.check(fn(C.byref(a), _stream()), "convt1d_fwd" if transposed else "conv1d_fwd")
    return y


def conv_dgrad(dy, w, dx, stride, pad, transposed=False, add=None, ref=None, mask=ACT_NONE, precision=None, schedule=None):
    """dx = dgrad(dy, w) [+ add] [* act'(ref)] -- grad wrt the tensor the forward op read."""
    Cin, Cout, k = _geom(transposed, w)
    B, _, Lin = dx.shape
    a = _conv_args(transposed, B, Cin, Cout, Lin, k, stride, pad, dx.device, precision, schedule)
    if tuple(dy.shape) != (B, Cout, a.Lout) or tuple(dx.shape) != (B, Cin, Lin):
        raise ValueError(f"conv_dgrad: shapes dy{tuple(dy.shape)} w{tuple(w.shape)} dx{tuple(dx.shape)} inconsistent")
    a.dy, a.dy_bs = _act3(dy, "dy")
    a.dx, a.dx_bs = _act3(dx, "dx")
    a.w = _dense(w, "w")
    if add is not None:
        if add.shape != dx.shape:
            raise ValueError("conv_dgrad: add must have dx's shape")
        a.dx_add, a.dx_add_bs = _act3(add, "add")
    if ref is not None:
        if ref.shape != dx.shape:
            raise ValueError("conv_dgrad: ref must have dx's shape")
        a.dx_ref, a.dx_ref_bs = _act3(ref, "ref")
        a.dx_mask = mask
    lib = _lib.load()
    _note_plan(a, _lib.OP_CONVT1D_DGRAD if transposed else _lib.OP_CONV1D_DGRAD)
    fn = lib.pg_convt1d_dgrad if transposed else lib.pg_conv1d_dgrad
    _lib.check(fn(C.byref(a), _stream()), "dgrad")
    return dx


def adam_args(p, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """pg_adam_args for ``conv_wgrad(..., adam=)``: the weight ``p`` and its exp_avg / exp_avg_sq, all shaped like dw."""
    a = _lib.AdamArgs()
    a.n = p.numel()
    a.p, a.m, a.v = _dense(p, "p"), _dense(m, "m"), _dense(v, "v")
    a.lr, a.beta1, a.beta2, a.eps, a.grad_scale, a.step = lr, beta1, beta2, eps, grad_scale, step
    return a


def conv_wgrad(x, dy, dw, stride, pad, x_act=ACT_NONE, transposed=False, precision=None, schedule=None, adam=None):
    """dw = wgrad(act(x), dy), overwriting ``dw`` (same layout as the weight).  ``adam`` (from ``adam_args``): the Adam update
    of this weight runs in the kernel's epilogue from the gradient it stores -- bit-identical to ``adam_step`` afterwards; the
    caller must have enqueued every reader of the old weight (the layer's dgrad) before this call."""
    Cin, Cout, k = _geom(transposed, dw)
    B, _, Lin = x.shape
    a = _conv_args(transposed, B, Cin, Cout, Lin, k, stride, pad, x.device, precision, schedule)
    if tuple(x.shape) != (B, Cin, Lin) or tuple(dy.shape) != (B, Cout, a.Lout):
        raise ValueError(f"conv_wgrad: shapes x{tuple(x.shape)} dy{tuple(dy.shape)} dw{tuple(dw.shape)} inconsistent")
    a.x, a.x_bs = _act3(x, "x")
    a.dy, a.dy_bs = _act3(dy, "dy")
    a.dw = _dense(dw, "dw")
    a.x_act = x_act
    if adam is not None:
        if adam.n != dw.numel():
            raise ValueError("conv_wgrad: fused adam tensors must have dw's shape")
        a.adam = C.addressof(adam)
    lib = _lib.load()
    _note_plan(a, _lib.OP_CONVT1D_WGRAD if transposed else _lib.OP_CONV1D_WGRAD)
    fn = lib.pg_convt1d_wgrad if transposed else lib.pg_conv1d_wgrad
    _lib.check(fn(C.byref(a), _stream()), "wgrad")
    return dw


# ---- bf16-resident forward path (BASELINE configs[4]; csrc/conv_h3.hip) ----------------------------------------------------
H_HEAD = 32      # PG_H_HEAD: zero elements the bf16-resident kernels may read in front of a tensor's first row
H_TAIL = 40      # zero tail of every row that covers each layer of the U-Net (pg_conv_fwd_h_supported checks a layer's need)


def h_pitch(L):
    """Row pitch (elements) of a bf16 activation tensor with L frames: 16-byte rows and a zero tail of at least H_TAIL elements
    (the kernels gather row windows as unchecked 16-byte pieces: the convolution's zero padding is read from the tails)."""
    return (L + H_TAIL + 7) // 8 * 8


def h_alloc(B, Cc, L, device):
    """Zero-filled bf16 (B, C, pitch) tensor with H_HEAD zero elements in front of it (a view into a slightly larger buffer):
    producers only ever write frames [0, L), so the zero tails -- and the head -- stay zero."""
    pitch = h_pitch(L)
    flat = torch.zeros(H_HEAD + B * Cc * pitch, device=device, dtype=torch.bfloat16)
    return flat[H_HEAD:].view(B, Cc, pitch)


def _h3(t, L, name):
    """(ptr, batch stride, pitch) of a bf16 (B, C, pitch) device view whose rows are `pitch` apart (pitch = size of dim 2)."""
    if t is None:
        return None, 0, 0
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 3 and t.stride(2) == 1):
        raise ValueError(f"{name}: expected a 3-D bfloat16 device tensor with contiguous rows, got {tuple(t.shape)} {t.dtype}")
    _on_current_device(t, name)
    pitch = t.stride(1) if t.shape[1] > 1 else t.shape[2]
    if pitch < L or (pitch & 1):
        raise ValueError(f"{name}: row pitch {pitch} must be even and >= {L}")
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else t.shape[1] * pitch), pitch


def conv_fwd_h_supported(B, w_shape, Lin, stride, pad, transposed):
    """True if the bf16-resident kernels cover this layer at this batch / frame count (pg_conv_fwd_h_supported; host only)."""
    if transposed:
        Cin, Cout, k = w_shape
    else:
        Cout, Cin, k = w_shape
    a = _lib.ConvhArgs()
    a.B, a.Cin, a.Cout, a.Lin, a.k, a.stride, a.pad, a.transposed = B, Cin, Cout, Lin, k, stride, pad, int(transposed)
    a.Lout = convt_out_len(Lin, k, stride, pad) if transposed else conv_out_len(Lin, k, stride, pad)
    a.x_pitch = h_pitch(Lin)
    a.x_bs = Cin * a.x_pitch
    return bool(_lib.load().pg_conv_fwd_h_supported(C.byref(a)))


def shadow_weights(w, transposed, stride, out=None):
    """bf16 shadow of a conv weight in the bf16-resident kernels' GEMM layout (pg_shadow_weights)."""
    Cin, Cout, k = _geom(transposed, w)
    n = _lib.load().pg_shadow_elems(Cin, Cout, k, stride, int(transposed))
    if out is None or out.numel() != n:
        out = torch.empty(n, device=w.device, dtype=torch.bfloat16)
    _lib.check(_lib.load().pg_shadow_weights(_dense(w, "w"), C.c_void_p(out.data_ptr()), Cin, Cout, k, stride, int(transposed), _stream()),
               "shadow_weights")
    return out


def cast_rows_bf16(x, out, act=ACT_NONE):
    """fp32 (B, C, L) -> bf16 (B, C, pitch) rows, activation applied, tails zeroed."""
    a = _lib.CastArgs()
    a.B, a.C, a.L = x.shape
    a.act = act
    a.x, a.x_bs = _act3(x, "x")
    a.y, a.y_bs, a.pitch = _h3(out, x.shape[2], "out")
    _lib.check(_lib.load().pg_cast_rows_bf16(C.byref(a), _stream()), "cast_rows_bf16")
    return out


def conv_fwd_h(xh, Lin, wh, w_shape, stride, pad, transposed=False, y=None, yh=None, yh_act=ACT_NONE, yh2=None, yh2_act=ACT_NONE,
               schedule=None):
    """bf16-resident forward conv (pg_conv_fwd_h): xh bf16 (B, Cin, pitch) holding Lin frames, wh the layer's bf16 shadow
    (``w_shape`` = shape of the fp32 master weight).  Outputs: fp32 y (B, Cout, Lout) and / or bf16 yh / yh2 (stored activated)."""
    if transposed:
        Cin, Cout, k = w_shape
    else:
        Cout, Cin, k = w_shape
    B = xh.shape[0]
    a = _lib.ConvhArgs()
    a.B, a.Cin, a.Cout, a.Lin, a.k, a.stride, a.pad, a.transposed = B, Cin, Cout, Lin, k, stride, pad, int(transposed)
    a.Lout = convt_out_len(Lin, k, stride, pad) if transposed else conv_out_len(Lin, k, stride, pad)
    a.schedule = _tls.schedule if schedule is None else schedule
    if xh.shape[1] != Cin:
        raise ValueError(f"conv_fwd_h: x has {xh.shape[1]} channels, weight expects {Cin}")
    a.x, a.x_bs, a.x_pitch = _h3(xh, Lin + 1, "xh")
    # v0.3 layout contract: the kernels gather the window of row (0, 0) from up to PG_H_HEAD elements IN FRONT of the tensor (the
    # convolution's left padding is read, not synthesised).  The C side only sees a pointer; here the storage is visible: a tensor
    # that starts closer than H_HEAD elements to the beginning of its allocation (plain torch.zeros(B, C, pitch) instead of
    # ops.h_alloc) would make the kernel read in front of the allocation -- foreign bytes as padding, or a GPU memory fault.
    if xh.storage_offset() < H_HEAD:
        raise ValueError(f"conv_fwd_h: xh starts {xh.storage_offset()} elements into its allocation; the bf16-resident kernels read "
                         f"{H_HEAD} zero elements in front of it -- allocate activations with ops.h_alloc (or pass a channel slice)")
    if wh.dtype != torch.bfloat16 or wh.numel() != _lib.load().pg_shadow_elems(Cin, Cout, k, stride, int(transposed)):
        raise ValueError("conv_fwd_h: wh must be the bf16 shadow of the layer's weight (ops.shadow_weights)")
    _on_current_device(wh, "wh")
    a.w = wh.data_ptr()
    if y is not None:
        if tuple(y.shape) != (B, Cout, a.Lout):
            raise ValueError(f"conv_fwd_h: y{tuple(y.shape)} should be {(B, Cout, a.Lout)}")
        a.y, a.y_bs = _act3(y, "y")
    if yh is not None:
        a.yh, a.yh_bs, a.yh_pitch = _h3(yh, a.Lout, "yh")
        a.yh_act = yh_act
    if yh2 is not None:
        a.yh2, a.yh2_bs, a.yh2_pitch = _h3(yh2, a.Lout, "yh2")
        a.yh2_act = yh2_act
    ws = conv_workspace(xh.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    if _timer is not None and _cur_label is not None and _cur_label not in _timer.plans:
        _timer.plans[_cur_label] = conv_fwd_h_describe(a)
    _lib.check(_lib.load().pg_conv_fwd_h(C.byref(a), _stream()), "conv_fwd_h")
    return a.Lout


def conv_fwd_h_describe(a):
    """pg_conv_fwd_h_describe: 'conv_h3_kernel<...>|grid=G|tiles=T|slabs=S|split=0/1|whole=W|fixup=none/plain/wide' for a filled ConvhArgs."""
    buf = C.create_string_buffer(256)
    _lib.check(_lib.load().pg_conv_fwd_h_describe(C.byref(a), buf, 256), "conv_fwd_h_describe")
    return buf.value.decode()


def bn_fwd(x, y, gamma, beta, save_mean, save_invstd, running_mean=None, running_var=None, eps=1e-5, momentum=0.1,
           y_act=ACT_NONE, y2=None, y2_act=ACT_NONE, yh=None, yh_act=ACT_NONE, yh2=None, yh2_act=ACT_NONE, num_batches_tracked=None):
    """Train-mode BatchNorm forward (model.py:81,83).  ``num_batches_tracked``: the layer's int64 counter (0-d device tensor),
    incremented by the kernel itself -- no extra launch."""
    a = _lib.BnArgs()
    a.y_act, a.y2_act = y_act, y2_act
    if y2 is not None:
        a.y2, a.y2_bs = _act3(y2, "y2")
    a.B, a.C, a.L = x.shape
    a.eps, a.momentum = eps, momentum
    a.x, a.x_bs = _act3(x, "x")
    if yh is not None:
        a.yh, a.yh_bs, a.yh_pitch = _h3(yh, x.shape[2], "yh")
        a.yh_act = yh_act
    if yh2 is not None:
        a.yh2, a.yh2_bs, a.yh2_pitch = _h3(yh2, x.shape[2], "yh2")
        a.yh2_act = yh2_act
    if y is not None:
        a.y, a.y_bs = _act3(y, "y")
    a.gamma, a.beta = _dense(gamma, "gamma"), _dense(beta, "beta")
    a.save_mean, a.save_invstd = _dense(save_mean, "save_mean"), _dense(save_invstd, "save_invstd")
    if running_mean is not None:
        a.running_mean, a.running_var = _dense(running_mean, "running_mean"), _dense(running_var, "running_var")
    if num_batches_tracked is not None:
        a.num_batches_tracked = _dense_as(num_batches_tracked, torch.int64, "num_batches_tracked")
    _lib.check(_lib.load().pg_bn_fwd(C.byref(a), _stream()), "bn_fwd")
    return y


def bn_bwd(x, dy, dx, gamma, save_mean, save_invstd, dgamma, dbeta):
    a = _lib.BnArgs()
    a.B, a.C, a.L = x.shape
    a.x, a.x_bs = _act3(x, "x")
    a.dy, a.dy_bs = _act3(dy, "dy")
    a.dx, a.dx_bs = _act3(dx, "dx")
    a.gamma = _dense(gamma, "gamma")
    a.save_mean, a.save_invstd = _dense(save_mean, "save_mean"), _dense(save_invstd, "save_invstd")
    a.dgamma, a.dbeta = _dense(dgamma, "dgamma"), _dense(dbeta, "dbeta")
    _lib.check(_lib.load().pg_bn_bwd(C.byref(a), _stream()), "bn_bwd")
    return dx


_loss_ws = _StreamCache()
_caches.append(_loss_ws)


def loss_fwd_bwd(pred, batch, dpred=None, losses=None, mag_weight=0.2):
    """train.py:45-60 fused with its gradient.  Returns the 3-float device tensor [loss, ang, mag]."""
    B, C2, L = pred.shape
    Cc = C2 // 2
    if tuple(batch.shape) != (B, 2, Cc, L):
        raise ValueError(f"loss: batch {tuple(batch.shape)} does not match pred {tuple(pred.shape)}")
    a = _lib.LossArgs()
    a.B, a.C, a.L, a.mag_weight = B, Cc, L, mag_weight
    a.pred, a.batch = _dense(pred, "pred"), _dense(batch, "batch")
    if dpred is not None:
        a.dpred = _dense(dpred, "dpred")
    if losses is None:
        losses = torch.empty(3, device=pred.device, dtype=torch.float32)
    a.losses = _dense(losses, "losses")
    lib = _lib.load()
    need = lib.pg_workspace_bytes_loss(C.byref(a))
    ws = _loss_ws.get(pred.device, lambda: need, need)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(lib.pg_loss_fwd_bwd(C.byref(a), _stream()), "loss_fwd_bwd")
    return losses


def adam_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, thin=False):
    a = _lib.AdamArgs()
    a.thin = int(bool(thin))
    a.n = p.numel()
    a.p, a.g, a.m, a.v = _dense(p, "p"), _dense(g, "g"), _dense(m, "m"), _dense(v, "v")
    a.lr, a.beta1, a.beta2, a.eps, a.grad_scale, a.step = lr, beta1, beta2, eps, grad_scale, step
    _lib.check(_lib.load().pg_adam_step(C.byref(a), _stream()), "adam_step")


def fill(t, value):
    _lib.check(_lib.load().pg_fill(C.c_void_p(_dense(t, "t")), t.numel(), value, _stream()), "fill")
    return t


def stft(y, n_fft, hop, polar=False, out=None, single_frame=None, chunk_start=None, chunk_row=None, chunk_len=None):
    """(n_signals, n_samples) -> (n_signals, 2, n_fft/2, 1 + n_samples // hop); preproc_mdb.py:84-97 (+ data.py:39-47).

    Chunked source (preproc_mdb.py:66-97): with ``chunk_start`` (int64 device tensor) signal s is the ``chunk_len`` samples of
    row ``chunk_row[s]`` (int32 device tensor; default row 0) of y (rows, samples) that begin at chunk_start[s]; samples past
    the end of the row read as zero -- no gathered or zero-padded copy of the audio is made."""
    if y.dim() == 1:
        y = y[None]
    a = _lib.StftArgs()
    if chunk_start is None:
        n_sig, n_samp = y.shape
    else:
        if chunk_start.dtype != torch.int64 or (chunk_row is not None and chunk_row.dtype != torch.int32):
            raise TypeError("chunk_start must be int64 and chunk_row int32")
        n_sig, n_samp = chunk_start.numel(), int(chunk_len)
        if chunk_row is not None and chunk_row.numel() != n_sig:
            raise ValueError("chunk_row needs one entry per chunk")
        a.chunk_start = _dense_as(chunk_start, torch.int64, "chunk_start")
        a.chunk_row = _dense_as(chunk_row, torch.int32, "chunk_row") if chunk_row is not None else None
        a.src_len, a.src_stride = y.shape[1], y.shape[1]
    nf = 1 + n_samp // hop
    if out is None:
        out = torch.empty(n_sig, 2, n_fft // 2, nf, device=y.device, dtype=torch.float32)
    elif tuple(out.shape) != (n_sig, 2, n_fft // 2, nf):
        raise ValueError(f"stft: out must be {(n_sig, 2, n_fft // 2, nf)}, got {tuple(out.shape)}")
    a.n_signals, a.n_samples, a.n_fft, a.hop, a.n_frames, a.polar = n_sig, n_samp, n_fft, hop, nf, int(polar)
    a.single_frame = _tls.stft_single if single_frame is None else int(bool(single_frame))
    a.y, a.out = _dense(y, "y"), _dense(out, "out")
    _lib.check(_lib.load().pg_stft(C.byref(a), _stream()), "stft")
    return out


_moments_ws = _StreamCache()
_caches.append(_moments_ws)


def standardize_(x):
    """preproc_mdb.py:182 in place on a dense float32 tensor: x = (x - x.mean()) / x.std() over the WHOLE array (population
    std, moments reduced in double).  Returns the (mean, std) device tensor (2 doubles)."""
    lib = _lib.load()
    ws = _moments_ws.get(x.device, lib.pg_workspace_bytes_moments)
    stats = torch.empty(2, dtype=torch.float64, device=x.device)
    a = _lib.MomentsArgs()
    a.n, a.x, a.stats = x.numel(), _dense(x, "x"), stats.data_ptr()
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(lib.pg_moments(C.byref(a), _stream()), "moments")
    _lib.check(lib.pg_standardize(C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(stats.data_ptr()), _stream()), "standardize")
    return stats


def stft_frame_index(n_samples, n_fft, hop, device="cuda"):
    nf = 1 + n_samples // hop
    idx = torch.empty(nf, n_fft, device=device, dtype=torch.int32)
    _lib.check(_lib.load().pg_stft_frame_index(n_samples, n_fft, hop, nf, C.c_void_p(idx.data_ptr()), _stream()), "stft_frame_index")
    return idx


def polar(d, out=None, use_exp=True):
    """data.py:39-47: (N, 2, bins, frames) [re; im] -> [log1p|z| (or |z|); angle]."""
    if out is None:
        out = torch.empty_like(d)
    a = _lib.PolarArgs()
    a.n_items, a.inner = d.shape[0], d[0, 0].numel()
    a.inp, a.out = _dense(d, "d"), _dense(out, "out")
    a.use_exp = int(bool(use_exp))
    _lib.check(_lib.load().pg_polar(C.byref(a), _stream()), "polar")
    return out


_istft_ws = _StreamCache()
_caches.append(_istft_ws)


def istft(a_t, b_t, hop, mode=0, normalize=True, single_frame=None):
    """(n, bins, frames) x2 -> (n, hop*(frames-1)).  mode 0: (logmag, phase) per demo.py:39; mode 1: (re, im).
    utils.py:34-42: zero DC row, librosa.istft, peak normalisation."""
    n, bins, nf = a_t.shape
    audio = torch.empty(n, hop * (nf - 1), device=a_t.device, dtype=torch.float32)
    lib = _lib.load()
    for s0 in range(0, n, 64):
        s1 = min(n, s0 + 64)
        a = _lib.IstftArgs()
        a.n_signals, a.bins, a.n_frames, a.hop, a.mode, a.normalize = s1 - s0, bins, nf, hop, mode, int(normalize)
        a.single_frame = _tls.stft_single if single_frame is None else int(bool(single_frame))
        a.a, a.a_bs = _act3(a_t[s0:s1], "a")
        a.b, a.b_bs = _act3(b_t[s0:s1], "b")
        a.audio = audio[s0:s1].data_ptr()
        need = lib.pg_workspace_bytes_istft(C.byref(a))
        ws = _istft_ws.get(a_t.device, lambda: need, need)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        _lib.check(lib.pg_istft(C.byref(a), _stream()), "istft")
    return audio


def gl_project(S, mag, x, spec_out=None):
    """utils.py:122-124: new_spec = mag * exp(1j * angle(S)) -> GEMM operand x (2*bins-2, frames) (+ [re; im] copy).
    Batched: S (n, 2, bins, frames), mag (n, bins, frames), x (n, 2*bins-2, frames), spec_out (n, 2, bins, frames)."""
    a = _lib.GlArgs()
    if mag.dim() == 3:
        a.n, a.bins, a.frames = mag.shape
        if tuple(S.shape) != (a.n, 2, a.bins, a.frames) or tuple(x.shape) != (a.n, 2 * a.bins - 2, a.frames):
            raise ValueError("gl_project: batched shapes inconsistent")
    else:
        a.bins, a.frames = mag.shape
    a.S, a.mag, a.x = _dense(S, "S"), _dense(mag, "mag"), _dense(x, "x")
    if spec_out is not None:
        if spec_out.shape != S.shape:
            raise ValueError("gl_project: spec_out must have S's shape")
        a.spec_out = _dense(spec_out, "spec_out")
    _lib.check(_lib.load().pg_gl_project(C.byref(a), _stream()), "gl_project")


_ola_ws = _StreamCache()
_caches.append(_ola_ws)


def ola_nt(frames_nt, hop, audio, normalize=False):
    """Overlap-add of (n_fft, frames)-major windowed frames (any even n_fft) into ``audio`` (hop * (frames - 1),); batched:
    frames (n, n_fft, frames) -> audio (n, hop * (frames - 1)), n <= 64, each clip normalised by its own peak."""
    a = _lib.OlaArgs()
    if frames_nt.dim() == 3:
        a.n, a.n_fft, a.frames = frames_nt.shape
        if tuple(audio.shape) != (a.n, hop * (a.frames - 1)):
            raise ValueError("ola_nt: audio must be (n, hop * (frames - 1))")
    else:
        a.n_fft, a.frames = frames_nt.shape
    a.hop, a.normalize = hop, int(normalize)
    a.fr, a.audio = _dense(frames_nt, "frames"), _dense(audio, "audio")
    ws = _ola_ws.get(audio.device, lambda: 256)
    a.workspace, a.workspace_bytes = ws.data_ptr(), 256
    _lib.check(_lib.load().pg_ola_nt(C.byref(a), _stream()), "ola_nt")
    return audio
