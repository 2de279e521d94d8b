"""ctypes binding of libphasegen.so (the C ABI declared in include/phasegen.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared object is missing or a
symbol cannot be resolved, importing the ops raises; if a call returns non-zero, ``check`` raises RuntimeError
with ``pg_last_error_string()`` (the reference's error convention is Python exceptions).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libphasegen.so")
ABI_VERSION = 400     # include/phasegen.h PG_VERSION these struct layouts were written against (argument structs grow between minor versions)

ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2
PREC_FP32, PREC_BF16, PREC_BF16X3 = 0, 1, 2           # pg_conv_args.precision
OK, ERR_NULL, ERR_SHAPE, ERR_ALIGN, ERR_UNSUPPORTED, ERR_WORKSPACE = 0, -1, -2, -3, -4, -5   # return codes (PG_ERR_*)
OP_CONV1D_FWD, OP_CONV1D_DGRAD, OP_CONV1D_WGRAD, OP_CONVT1D_FWD, OP_CONVT1D_DGRAD, OP_CONVT1D_WGRAD = range(6)   # pg_conv_describe
SCHED_AUTO, SCHED_TILE_PER_WG, SCHED_FORCE_STREAMK, SCHED_NO_RAW, SCHED_NO_TALL, SCHED_CONTENDED = 0, 1, 2, 4, 8, 16   # pg_conv_args.schedule bits
SCHED_NO_PS = 128      # wgrad: keep the flat-K raw kernel (no per-sample slabs)
SCHED_NO_RAW3 = 0x2000  # fp32 F / T: never the one-wave-per-SIMD kernels (conv_raw3.hip)
SCHED_ALL_RAW3 = 0x4000  # ... those kernels wherever they cover the problem (also the F form of k = 32, which auto leaves on the older ones)
SCHED_NO_COLSPLIT = 0x20000  # fp32 F / T on conv_raw3: keep the columns past the last full 256-wide tile in the same launch (no tail launch)
SCHED_COLSPLIT = 0x40000     # ... or always hand them to the tail launch where the geometry allows (tests; the automatic choice prices the tail)
# (pg_convh_args.schedule bits 5-6 selected tile families that ABI 0.4 removed; bit 12, the surviving conv_h3 family, is a no-op)

c_float_p = C.c_void_p  # device pointers travel as integers


class ConvArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("Lin", C.c_int32), ("Lout", C.c_int32),
                ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("x", C.c_void_p), ("x_bs", C.c_int64), ("x_act", C.c_int32), ("precision", C.c_int32),
                ("w", C.c_void_p),
                ("y", C.c_void_p), ("y_bs", C.c_int64),
                ("dy", C.c_void_p), ("dy_bs", C.c_int64),
                ("dx", C.c_void_p), ("dx_bs", C.c_int64),
                ("dx_add", C.c_void_p), ("dx_add_bs", C.c_int64),
                ("dx_ref", C.c_void_p), ("dx_ref_bs", C.c_int64),
                ("dx_mask", C.c_int32), ("schedule", C.c_int32),
                ("dw", C.c_void_p), ("y_act", C.c_int32), ("y2_act", C.c_int32), ("y2", C.c_void_p), ("y2_bs", C.c_int64),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("adam", C.c_void_p)]        # const pg_adam_args*: optimiser step fused into the wgrad epilogue (or NULL)


class BnArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("L", C.c_int32), ("eps", C.c_float), ("momentum", C.c_float),
                ("x", C.c_void_p), ("x_bs", C.c_int64), ("y", C.c_void_p), ("y_bs", C.c_int64),
                ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("dy", C.c_void_p), ("dy_bs", C.c_int64), ("dx", C.c_void_p), ("dx_bs", C.c_int64),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("y_act", C.c_int32), ("y2_act", C.c_int32), ("y2", C.c_void_p), ("y2_bs", C.c_int64),
                ("yh", C.c_void_p), ("yh_bs", C.c_int64), ("yh_pitch", C.c_int32), ("yh_act", C.c_int32),
                ("yh2", C.c_void_p), ("yh2_bs", C.c_int64), ("yh2_pitch", C.c_int32), ("yh2_act", C.c_int32),
                ("num_batches_tracked", C.c_void_p)]


class ConvhArgs(C.Structure):
    """pg_convh_args: bf16-resident forward conv / transposed conv (include/phasegen.h)."""
    _fields_ = [("B", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("Lin", C.c_int32), ("Lout", C.c_int32),
                ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("transposed", C.c_int32), ("schedule", C.c_int32),
                ("x", C.c_void_p), ("x_bs", C.c_int64), ("x_pitch", C.c_int32), ("_pad0", C.c_int32),
                ("w", C.c_void_p), ("y", C.c_void_p), ("y_bs", C.c_int64),
                ("yh", C.c_void_p), ("yh_bs", C.c_int64), ("yh_pitch", C.c_int32), ("yh_act", C.c_int32),
                ("yh2", C.c_void_p), ("yh2_bs", C.c_int64), ("yh2_pitch", C.c_int32), ("yh2_act", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


class CastArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("L", C.c_int32), ("pitch", C.c_int32), ("act", C.c_int32), ("_pad0", C.c_int32),
                ("x", C.c_void_p), ("x_bs", C.c_int64), ("y", C.c_void_p), ("y_bs", C.c_int64)]


class LossArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("L", C.c_int32), ("mag_weight", C.c_float),
                ("pred", C.c_void_p), ("batch", C.c_void_p), ("dpred", C.c_void_p), ("losses", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


class AdamArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("grad_scale", C.c_double), ("step", C.c_int32), ("thin", C.c_int32)]


class StftArgs(C.Structure):
    _fields_ = [("n_signals", C.c_int32), ("n_samples", C.c_int32), ("n_fft", C.c_int32), ("hop", C.c_int32),
                ("n_frames", C.c_int32), ("polar", C.c_int32), ("single_frame", C.c_int32), ("_pad0", C.c_int32),
                ("y", C.c_void_p), ("out", C.c_void_p),
                ("chunk_start", C.c_void_p), ("chunk_row", C.c_void_p), ("src_len", C.c_int64), ("src_stride", C.c_int64)]


class MomentsArgs(C.Structure):
    _fields_ = [("n", C.c_int64), ("x", C.c_void_p), ("stats", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_int64)]


class PolarArgs(C.Structure):
    _fields_ = [("n_items", C.c_int64), ("inner", C.c_int64), ("inp", C.c_void_p), ("out", C.c_void_p),
                ("use_exp", C.c_int32), ("_pad0", C.c_int32)]


class IstftArgs(C.Structure):
    _fields_ = [("n_signals", C.c_int32), ("bins", C.c_int32), ("n_frames", C.c_int32), ("hop", C.c_int32),
                ("mode", C.c_int32), ("normalize", C.c_int32), ("single_frame", C.c_int32), ("_pad0", C.c_int32),
                ("a", C.c_void_p), ("a_bs", C.c_int64), ("b", C.c_void_p), ("b_bs", C.c_int64),
                ("audio", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


class GlArgs(C.Structure):
    _fields_ = [("bins", C.c_int32), ("frames", C.c_int32), ("S", C.c_void_p), ("mag", C.c_void_p), ("x", C.c_void_p),
                ("spec_out", C.c_void_p), ("n", C.c_int32), ("_pad0", C.c_int32)]


class OlaArgs(C.Structure):
    _fields_ = [("n_fft", C.c_int32), ("frames", C.c_int32), ("hop", C.c_int32), ("normalize", C.c_int32),
                ("fr", C.c_void_p), ("audio", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("n", C.c_int32), ("_pad0", C.c_int32)]


# every symbol include/phasegen.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "pg_conv1d_fwd": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "pg_conv1d_dgrad": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "pg_conv1d_wgrad": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "pg_convt1d_fwd": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "pg_convt1d_dgrad": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "pg_convt1d_wgrad": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "pg_conv_describe": (C.c_int, [C.POINTER(ConvArgs), C.c_int32, C.c_char_p, C.c_int32]),
    "pg_conv_fwd_h": (C.c_int, [C.POINTER(ConvhArgs), C.c_void_p]),
    "pg_conv_fwd_h_supported": (C.c_int, [C.POINTER(ConvhArgs)]),
    "pg_workspace_bytes_moments": (C.c_int64, []),
    "pg_moments": (C.c_int, [C.POINTER(MomentsArgs), C.c_void_p]),
    "pg_standardize": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pg_shadow_elems": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pg_shadow_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pg_cast_rows_bf16": (C.c_int, [C.POINTER(CastArgs), C.c_void_p]),
    "pg_workspace_bytes_conv": (C.c_int64, []),
    "pg_bn_fwd": (C.c_int, [C.POINTER(BnArgs), C.c_void_p]),
    "pg_bn_bwd": (C.c_int, [C.POINTER(BnArgs), C.c_void_p]),
    "pg_workspace_bytes_loss": (C.c_int64, [C.POINTER(LossArgs)]),
    "pg_loss_fwd_bwd": (C.c_int, [C.POINTER(LossArgs), C.c_void_p]),
    "pg_adam_step": (C.c_int, [C.POINTER(AdamArgs), C.c_void_p]),
    "pg_stft": (C.c_int, [C.POINTER(StftArgs), C.c_void_p]),
    "pg_stft_frame_index": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pg_polar": (C.c_int, [C.POINTER(PolarArgs), C.c_void_p]),
    "pg_workspace_bytes_istft": (C.c_int64, [C.POINTER(IstftArgs)]),
    "pg_istft": (C.c_int, [C.POINTER(IstftArgs), C.c_void_p]),
    "pg_gl_project": (C.c_int, [C.POINTER(GlArgs), C.c_void_p]),
    "pg_ola_nt": (C.c_int, [C.POINTER(OlaArgs), C.c_void_p]),
    "pg_fill": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "pg_conv_fwd_h_describe": (C.c_int, [C.POINTER(ConvhArgs), C.c_char_p, C.c_int32]),
    "pg_version": (C.c_int, []),
    "pg_last_error_string": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libphasegen.so and bind every declared symbol.  Raises if anything is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C unet-phasegen_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback for the phasegen hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.pg_version() // 100 != ABI_VERSION // 100:
        raise RuntimeError(f"{LIB_PATH} is ABI {lib.pg_version()}, this binding was written against {ABI_VERSION}: rebuild the library "
                           "(`make -C unet-phasegen_amd/csrc`)")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pg_last_error_string()
        raise RuntimeError(f"libphasegen {what} failed: {msg.decode() if msg else rc}")
