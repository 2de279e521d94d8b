/* phasegen.h -- C ABI of libphasegen.so: the MI355X (gfx950) hot path of UNet-PhaseGen.
 *
 * The reference (LemonATsu/UNet-PhaseGen) is pure Python and has no FFI of its own; what it binds on
 * this path is torch.nn.Conv1d / ConvTranspose1d / BatchNorm / (Leaky)ReLU / cat (model.py:77-113),
 * MSELoss + Adam (train.py:26-28,45-62), np.abs/np.angle (data.py:39-47) and librosa.stft / istft
 * (preproc_mdb.py:93, utils.py:40).  Each entry point below replaces one of those and cites it.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, scalars; no torch types.  All tensors fp32.
 *   - activations are (B, channels, frames), frames fastest; a tensor is given as {ptr, batch_stride}
 *     (elements) so a channel slice of a wider buffer (the U-Net's concat halves) is passed without copy.
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library allocates nothing and keeps
 *     no mutable global state besides a thread-local last-error string: every knob (MFMA operand precision,
 *     work-split schedule, FFT schedule) is a field of the call's argument struct, so calls on different
 *     streams / threads are independent.  A workspace belongs to ONE stream at a time.
 *   - every call is asynchronous on the hipStream_t passed as `void* stream`; no internal syncs.
 *   - return 0 on success, <0 = PG_ERR_* (bad argument), >0 = hipError_t from the launch.
 */
#ifndef PHASEGEN_H
#define PHASEGEN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_VERSION 400 /* 0.4.0: pg_bn_args.num_batches_tracked; (0.3.0: pg_conv_fwd_h reads its zero padding from the rows' tails and a
                        * PG_H_HEAD zero head; 0.2.0: knobs in the argument structs).  Argument structs GROW between minor versions
                        * (new fields are appended, and a zero / NULL there means the old behaviour), so their SIZE is part of the ABI:
                        * a caller compiled against an older header passes a shorter struct and the library would read past its end.
                        * Callers check pg_version() / 100 == PG_VERSION / 100 at load time and are rebuilt when it differs. */

enum { PG_OK = 0, PG_ERR_NULL = -1, PG_ERR_SHAPE = -2, PG_ERR_ALIGN = -3, PG_ERR_UNSUPPORTED = -4,
       PG_ERR_WORKSPACE = -5 };

/* activation fused on operand load / derivative mask fused in a dgrad epilogue */
enum { PG_ACT_NONE = 0, PG_ACT_LEAKY02 = 1 /* nn.LeakyReLU(0.2), model.py:80 */, PG_ACT_RELU = 2 /* model.py:82 */ };

/* One descriptor serves the six convolution entry points.  Geometry is always that of the FORWARD op:
 *   Conv1d          x (B,Cin,Lin) * w (Cout,Cin,k)  -> y (B,Cout,Lout),  Lout = (Lin + 2 pad - k)/stride + 1
 *   ConvTranspose1d x (B,Cin,Lin) * w (Cin,Cout,k)  -> y (B,Cout,Lout),  Lout = (Lin - 1) stride - 2 pad + k
 * bias is never used (model.py:65-69: use_bias == False under BatchNorm). */
struct pg_adam_args;
typedef struct pg_conv_args {
    int32_t B, Cin, Cout, Lin, Lout, k, stride, pad;
    const float* x;  int64_t x_bs;   /* forward input; read by *_fwd and *_wgrad                           */
    int32_t x_act;                   /* PG_ACT_* applied to x as it is read (the in-place (Leaky)ReLU that  */
                                     /*   precedes the conv in model.py:91,96,103 is never materialised)    */
    int32_t precision;               /* PG_PREC_*: operand precision of the MFMA contraction (below)        */
    const float* w;                  /* weights (layout above); read by *_fwd and *_dgrad                  */
    float* y;        int64_t y_bs;   /* forward output; written by *_fwd                                   */
    const float* dy; int64_t dy_bs;  /* grad wrt y; read by *_dgrad and *_wgrad                            */
    float* dx;       int64_t dx_bs;  /* grad wrt the tensor x was read from; written by *_dgrad            */
    const float* dx_add; int64_t dx_add_bs; /* optional (NULL): added to the dgrad result before masking (skip grad) */
    const float* dx_ref; int64_t dx_ref_bs; /* optional (NULL): dx *= act'(dx_ref) with act = dx_mask              */
    int32_t dx_mask;
    int32_t schedule;                /* 0 = automatic; otherwise PG_SCHED_* bits (tests and measurements)   */
    float* dw;                       /* grad wrt w, same layout as w; OVERWRITTEN by *_wgrad (beta = 0:     */
                                     /*   optim.zero_grad(), train.py:41, is folded into the write)         */
    int32_t y_act; int32_t y2_act;   /* *_fwd only: activation applied to the result as it is STORED (producers can */
    float* y2;       int64_t y2_bs;  /*   hand consumers pre-activated tensors); y2 = optional second copy with its */
                                     /*   own activation (the skip tensor is read as LeakyReLU by the next down conv */
                                     /*   and as ReLU by the up path, model.py:80,82,113)                           */
    void* workspace; int64_t workspace_bytes; /* optional scratch (pg_workspace_bytes_conv()): lets tile counts   */
                                     /*   that quantise badly over the CUs be split evenly (stream-K);      */
                                     /*   contents are garbage between calls; NULL = always one tile per WG */
    const struct pg_adam_args* adam; /* *_wgrad only, optional (NULL): the optimiser step of THIS weight fused into the   */
                                     /*   wgrad epilogue (train.py:61-62 in one kernel): adam->p / m / v are the weight,   */
                                     /*   exp_avg and exp_avg_sq tensors (layout of w), updated from the gradient value the */
                                     /*   epilogue stores to dw -- same arithmetic as pg_adam_step, bit for bit; adam->g   */
                                     /*   and adam->n are ignored.  Callers order the layer's dgrad (which reads w) BEFORE */
                                     /*   this call on the stream.  Not for data-parallel runs (reduce first).             */
} pg_conv_args;
int64_t pg_workspace_bytes_conv(void);
/* pg_conv_args.precision (BASELINE config 5): PG_PREC_FP32 (0, default) = fp32 operands, v_mfma_f32_32x32x2_f32, the 1e-4
 * parity path; PG_PREC_BF16 = operands rounded to bf16 (RNE, after the fused activation) at fragment load,
 * v_mfma_f32_32x32x16_bf16, fp32 accumulate -- tensors and master weights in HBM stay fp32; PG_PREC_BF16X3 = every fp32
 * operand is split exactly into hi + lo bf16 parts and each product is taken as hi*hi' + hi*lo' + lo*hi' on the bf16 pipe
 * (three MFMAs at 1/16 of the fp32 cost; ~5e-6 from exact against fp32 MFMA's ~1e-6: inside the 1e-4 parity bound, not fp32). */
enum { PG_PREC_FP32 = 0, PG_PREC_BF16 = 1, PG_PREC_BF16X3 = 2 };
/* pg_conv_args.schedule: bits 0-1 work split (0 automatic, 1 one tile per workgroup, 2 force stream-K); bit 2: no raw-window
 * kernels (im2col kernels); bit 3: no tall (256 x 128) raw tile; bit 4: other kernels (RCCL collectives) share the chip --
 * keep the fine stream-K split even where the tile count quantises perfectly; bits 8-11: stream-K grid = that many x the
 * resident workgroup slots (1..8; 0 = default 4; > 1 bounds the tail when other kernels such as RCCL's share the chip); bit 7: the
 * wgrad keeps the flat-K kernel; bit 13: never the one-wave-per-SIMD fp32 F / T kernels (conv_raw3.hip); bit 14: those kernels wherever they cover the problem;
 * bits 15-16: their tile order (1: row-major, 2 / 3: super-rows of 2 / 4 tile rows; 0: automatic) -- measurements only.
 * PG_SCHED_NO_COLSPLIT (bit 17): the conv_raw3 launches keep a short column tail (<= 128 columns past the last full 256-wide tile)
 * instead of handing it to a second launch of the tall-tile kernel (pg_conv_describe shows the second launch as |tail=...);
 * PG_SCHED_COLSPLIT (bit 18): the tail launch wherever the geometry allows, whatever the cost model says (tests). */
enum { PG_SCHED_AUTO = 0, PG_SCHED_TILE_PER_WG = 1, PG_SCHED_FORCE_STREAMK = 2, PG_SCHED_NO_RAW = 4, PG_SCHED_NO_TALL = 8,
       PG_SCHED_CONTENDED = 16, PG_SCHED_NO_PS = 128, PG_SCHED_NO_RAW3 = 0x2000, PG_SCHED_ALL_RAW3 = 0x4000, PG_SCHED_NO_COLSPLIT = 0x20000, PG_SCHED_COLSPLIT = 0x40000 };
#define PG_SCHED_OVERSUB(f) (((f) & 15) << 8)

/* nn.Conv1d forward / backward (model.py:77-78; autograd of train.py:61) */
int pg_conv1d_fwd(const pg_conv_args* a, void* stream);
int pg_conv1d_dgrad(const pg_conv_args* a, void* stream);
int pg_conv1d_wgrad(const pg_conv_args* a, void* stream);
/* nn.ConvTranspose1d forward / backward (model.py:88-89,94-95,101-102) */
int pg_convt1d_fwd(const pg_conv_args* a, void* stream);
int pg_convt1d_dgrad(const pg_conv_args* a, void* stream);
int pg_convt1d_wgrad(const pg_conv_args* a, void* stream);
/* The launch plan of one of the six calls above WITHOUT launching it (measurement aid; pure function of the arguments):
 * buf receives "kernel<template args>|grid=G|tiles=T|slabs=S|split=0/1|whole=W|fixup=none/plain/wide", the kernel named as rocprofv3 reports it. */
enum { PG_OP_CONV1D_FWD = 0, PG_OP_CONV1D_DGRAD = 1, PG_OP_CONV1D_WGRAD = 2,
       PG_OP_CONVT1D_FWD = 3, PG_OP_CONVT1D_DGRAD = 4, PG_OP_CONVT1D_WGRAD = 5 };
int pg_conv_describe(const pg_conv_args* a, int32_t op, char* buf, int32_t buflen);

/* ---- bf16-RESIDENT forward convolutions (BASELINE configs[4]: "bf16 MFMA convs") ----------------------------------------
 * Same two forward ops, but the operands already live in HBM as bf16: activations (B, C, pitch) whose rows are `pitch`
 * elements apart (pitch and batch stride even, pitch > L, elements [L, pitch) of every row ZERO -- allocate zero-filled once;
 * producers never write the tail), weights as a bf16 shadow of the fp32 master weights in the kernels' GEMM layout
 * (pg_shadow_weights, rebuilt whenever the masters change).
 * Zero-padding contract of x (v0.3): the kernels gather row windows as unchecked 16-byte pieces, so the convolution's zero
 * padding is READ from memory: (1) every row's zero tail must cover what a window reaches in front of the NEXT row and behind
 * this row's last frame (pg_conv_fwd_h_supported checks the layer's need; 40 elements cover every layer of the U-Net), and
 * (2) the PG_H_HEAD elements (64 bytes) in front of x must be readable and zero -- for a channel slice of a larger tensor
 * they are the previous channel's tail; a tensor of its own is allocated with that many zero elements in front.  v_mfma_f32_32x32x16_bf16, fp32 accumulate.  Outputs: the fp32 result as is (y, optional: what
 * pg_bn_fwd normalises) and / or up to two bf16 copies stored already activated for the next layer.  Geometries: the U-Net's
 * (k, stride) pairs with Cin a multiple of 32 / min(taps per phase, 32); others return PG_ERR_UNSUPPORTED (use pg_conv*_fwd). */
#define PG_H_HEAD 32
typedef struct pg_convh_args {
    int32_t B, Cin, Cout, Lin, Lout, k, stride, pad;
    int32_t transposed;              /* 0: nn.Conv1d forward (model.py:77-78), 1: nn.ConvTranspose1d forward (model.py:88-102) */
    int32_t schedule;                /* work-split bits of pg_conv_args.schedule (0-1, 4, 8-11).  One tile family since 0.4 (256 x 256 on 4
                                      * waves, one per SIMD); bits 5-6, which selected the removed ones, return PG_ERR_UNSUPPORTED */
    const uint16_t* x; int64_t x_bs; /* bf16 (B, Cin, x_pitch), batch stride in elements */
    int32_t x_pitch; int32_t _pad0;
    const uint16_t* w;               /* pg_shadow_weights output for this layer */
    float* y; int64_t y_bs;          /* optional fp32 (B, Cout, Lout) */
    uint16_t* yh; int64_t yh_bs; int32_t yh_pitch; int32_t yh_act;       /* optional bf16 (B, Cout, yh_pitch) = PG_ACT(result) */
    uint16_t* yh2; int64_t yh2_bs; int32_t yh2_pitch; int32_t yh2_act;   /* optional second bf16 copy with its own activation */
    void* workspace; int64_t workspace_bytes;   /* pg_workspace_bytes_conv() */
} pg_convh_args;
int pg_conv_fwd_h(const pg_convh_args* a, void* stream);
int pg_conv_fwd_h_supported(const pg_convh_args* a);   /* 1 / 0: geometry (sizes, strides, pitches; pointers ignored) is covered */
int pg_conv_fwd_h_describe(const pg_convh_args* a, char* buf, int32_t buflen);   /* launch plan without launching, as pg_conv_describe */
/* bf16 shadow of one conv layer's weights: Conv1d (Cout, Cin, k) -> [o][(q, j)] (a cast); ConvTranspose1d (Cin, Cout, k) ->
 * [(o, phase)][(q, tap)] with the taps of a phase in the order the gather-form kernel reads them, ceil(k / stride) taps per
 * phase rounded up to a power of two with zero weights (k = 5, stride 2: 4).  wh holds pg_shadow_elems(...) elements. */
int64_t pg_shadow_elems(int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t transposed);
int pg_shadow_weights(const float* w, uint16_t* wh, int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t transposed, void* stream);
/* fp32 (B, C, L) -> bf16 (B, C, pitch) with PG_ACT applied and the row tails zeroed (the network input of the bf16 path). */
typedef struct pg_cast_args { int32_t B, C, L, pitch, act, _pad0; const float* x; int64_t x_bs; uint16_t* y; int64_t y_bs; } pg_cast_args;
int pg_cast_rows_bf16(const pg_cast_args* a, void* stream);

/* Train-mode batch norm over (B, L) per channel (model.py:81,83 applied to 3-D tensors; eps 1e-5, momentum
 * 0.1; biased variance normalises, unbiased variance goes to running_var). */
typedef struct pg_bn_args {
    int32_t B, C, L; float eps, momentum;
    const float* x;  int64_t x_bs;       /* raw conv output                                  */
    float* y;        int64_t y_bs;       /* fwd: normalised + affine output                  */
    const float* gamma; const float* beta;
    float* save_mean; float* save_invstd;     /* (C) written by fwd, read by bwd             */
    float* running_mean; float* running_var;  /* (C) updated in place by fwd; may be NULL    */
    const float* dy; int64_t dy_bs;      /* bwd: grad wrt y                                  */
    float* dx;       int64_t dx_bs;      /* bwd: grad wrt x                                  */
    float* dgamma; float* dbeta;         /* bwd: (C), overwritten                            */
    int32_t y_act; int32_t y2_act;       /* fwd: activation applied as y is stored; optional second output y2 with  */
    float* y2;       int64_t y2_bs;      /*   its own activation (same purpose as in pg_conv_args)                  */
    /* fwd, bf16-resident path: optional bf16 outputs (B, C, pitch) with their own activations (tails stay untouched);  */
    /* y may then be NULL                                                                                               */
    uint16_t* yh;  int64_t yh_bs;  int32_t yh_pitch;  int32_t yh_act;
    uint16_t* yh2; int64_t yh2_bs; int32_t yh2_pitch; int32_t yh2_act;
    int64_t* num_batches_tracked;        /* fwd, optional (NULL): the layer's nn.BatchNorm counter, incremented by one on the device */
} pg_bn_args;
int pg_bn_fwd(const pg_bn_args* a, void* stream);
int pg_bn_bwd(const pg_bn_args* a, void* stream);

/* train.py:45-60: loss = MSE(cos p, cos th) + MSE(sin p, sin th) + mag_weight * MSE(m, logmag), fused with its
 * gradient wrt pred.  pred (B,2C,L) = [phase ; magnitude]; batch (B,2,C,L) = [logmag ; angle].
 * losses[0..2] = {loss, ang, mag}.  Deterministic two-stage reduction through `workspace`. */
typedef struct pg_loss_args {
    int32_t B, C, L; float mag_weight;
    const float* pred; const float* batch;
    float* dpred;      /* may be NULL (evaluation only) */
    float* losses;     /* 3 floats, device */
    void* workspace; int64_t workspace_bytes;
} pg_loss_args;
int64_t pg_workspace_bytes_loss(const pg_loss_args* a);
int pg_loss_fwd_bwd(const pg_loss_args* a, void* stream);

/* torch.optim.Adam.step with defaults (train.py:27,62) over one flat parameter arena: p -= lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps).
 * grad_scale multiplies g first (1/world for data-parallel averaging); step is 1-based. */
typedef struct pg_adam_args {
    int64_t n; float* p; const float* g; float* m; float* v;
    double lr, beta1, beta2, eps, grad_scale;   /* Python floats, rounded to fp32 exactly where torch rounds them */
    int32_t step;
    int32_t thin;    /* 0: full-rate streaming launch (the chip is idle); 1: at most one small workgroup per CU, <= 32 VGPRs,  */
                     /*    non-temporal accesses -- for running BESIDE the MFMA-bound conv kernels of backward on another stream */
} pg_adam_args;
int pg_adam_step(const pg_adam_args* a, void* stream);

/* librosa.stft(y, n_fft, hop) + DC drop + [re; im] stacking (preproc_mdb.py:84-97), optionally fused with
 * data.py:39-47 (polar = 1 -> out = [log1p|z| ; angle z]).  y (n_signals, n_samples) -> out (n_signals, 2, n_fft/2, n_frames),
 * n_frames = 1 + n_samples / hop, reflect padding n_fft/2, periodic Hann.  n_fft: power of two in [32, 4096]. */
typedef struct pg_stft_args {
    int32_t n_signals, n_samples, n_fft, hop, n_frames, polar;
    int32_t single_frame;   /* transform schedule: 0 (default) = 4 frames per workgroup, real FFT through an n_fft/2-point radix-4 */
    int32_t _pad0;          /*   transform (n_fft <= 2048; longer always take the other path); 1 = one frame per workgroup,       */
    const float* y; float* out; /* full-length radix-2.  Same framing; they differ in fp32 rounding only (tests, measurements).   */
    /* Chunked source (preproc_mdb.py:66-97 without a gathered copy).  chunk_start == NULL: signal s is row s of y
     * (n_signals, n_samples).  Otherwise signal s is the n_samples-long chunk of source row chunk_row[s] (NULL: row 0) of
     * y (rows, src_stride) that begins at sample chunk_start[s] (0 <= start); samples at or beyond src_len read as zero --
     * the reference's zero-padded tail (preproc_mdb.py:86-88).  Both arrays are device pointers of n_signals entries. */
    const int64_t* chunk_start; const int32_t* chunk_row; int64_t src_len; int64_t src_stride;
} pg_stft_args;
int pg_stft(const pg_stft_args* a, void* stream);
/* The integer framing map alone (bit-exact contract): idx[t, k] = sample index of tap k of frame t. */
int pg_stft_frame_index(int32_t n_samples, int32_t n_fft, int32_t hop, int32_t n_frames, int32_t* idx, void* stream);

/* data.py:39-47 on its own: in (N,2,...) [re; im] -> out (N,2,...) [log1p|z| ; angle]; inner = bins*frames. */
typedef struct pg_polar_args { int64_t n_items; int64_t inner; const float* in; float* out;
                               int32_t use_exp; /* data.py:39 use_exp: 1 -> log1p|z|, 0 -> |z| */ int32_t _pad0; } pg_polar_args;
int pg_polar(const pg_polar_args* a, void* stream);

/* demo.py:39 + utils.py:34-42: z = (exp(logmag) - 1) e^{j phase} (mode 0) or re + j im (mode 1); zero DC row prepended;
 * librosa.istft (irfft x Hann, overlap-add, / window-sum-square, trim n_fft/2); optional peak normalisation.
 * a, b: (n_signals, bins, n_frames) with the given batch strides; audio (n_signals, hop * (n_frames - 1)). */
typedef struct pg_istft_args {
    int32_t n_signals, bins, n_frames, hop, mode, normalize;
    int32_t single_frame; int32_t _pad0;   /* transform schedule, as in pg_stft_args */
    const float* a; int64_t a_bs; const float* b; int64_t b_bs;
    float* audio;
    void* workspace; int64_t workspace_bytes;
} pg_istft_args;
int64_t pg_workspace_bytes_istft(const pg_istft_args* a);
int pg_istft(const pg_istft_args* a, void* stream);

/* Griffin-Lim building blocks (utils.py:112-134).  pg_gl_project: S (2,bins,frames) = [re; im] of the current estimate's
 * STFT and the target magnitudes mag (bins,frames) -> new_spec = mag * exp(j angle(S)) written as (2,bins,frames) to
 * spec_out (may be NULL) and, laid out as the operand of the inverse-DFT GEMM, to x (2*bins-2, frames): rows 0..bins-1
 * real parts, rows bins.. imaginary parts of bins 1..bins-2 (DC/Nyquist imaginary parts are ignored by an irfft).
 * pg_ola_nt: overlap-add of windowed frames given as (n_fft, frames) [n][t], any even n_fft (the reference inverts the
 * DC-dropped matrix, n_fft = 2*(bins-1) = 2046), / window-sum-square, trim n_fft/2, optional peak normalisation. */
typedef struct pg_gl_args { int32_t bins, frames; const float* S; const float* mag; float* x; float* spec_out;
                            int32_t n, _pad0; } pg_gl_args;       /* n clips (0 = 1): every tensor gains a leading clip axis */
int pg_gl_project(const pg_gl_args* a, void* stream);
typedef struct pg_ola_args { int32_t n_fft, frames, hop, normalize; const float* fr; float* audio; void* workspace; int64_t workspace_bytes;
                             int32_t n, _pad0; } pg_ola_args;     /* n clips (0 = 1, at most 64): fr (n, n_fft, frames), audio (n, len) */
int pg_ola_nt(const pg_ola_args* a, void* stream);   /* workspace: 256 bytes */

/* preproc_mdb.py:182: x = (x - x.mean()) / x.std() over the WHOLE array (population std).  pg_moments reduces in double and
 * writes stats[0] = mean, stats[1] = std (device doubles); pg_standardize applies them in place.  Two launches + one
 * elementwise pass: 4 B read for the moments, 8 B for the update, per element. */
typedef struct pg_moments_args { int64_t n; const float* x; double* stats; void* workspace; int64_t workspace_bytes; } pg_moments_args;
int64_t pg_workspace_bytes_moments(void);
int pg_moments(const pg_moments_args* a, void* stream);
int pg_standardize(float* x, int64_t n, const double* stats, void* stream);

/* small helpers the training step needs on device */
int pg_fill(float* p, int64_t n, float value, void* stream);

int pg_version(void);
const char* pg_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* PHASEGEN_H */
