// conv_igemm.hip -- the six 1-D convolution passes of the U-Net as three implicit-GEMM kernels on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
//   F ("forward-shaped"):  Y[b,m,t]  = sum_{q,j}               W[m][q][j] * act(X[b,q,s*t+j-p])
//        = nn.Conv1d forward (model.py:77-78)            and nn.ConvTranspose1d dgrad
//   T ("transposed"):      Y[b,m,tau] = sum_{q,j: s*i+j-p=tau} W[q][m][j] * act(X[b,q,i])
//        = nn.ConvTranspose1d forward (model.py:88-102)  and nn.Conv1d dgrad
//        computed in gather form: output phase phi = (tau+p) mod s only sees taps j = s*jj + phi, so the GEMM
//        rows are (m,phi) pairs, K = (q,jj), N = (b,u) with tau = s*u + phi - p.  No col2im scatter, no atomics.
//   G ("gradient of W"):   dW[m][q][j] = sum_{b,i} actP(P[b,m,i]) * actQ(Q[b,q,s*i+j-p])
//        = wgrad of both (conv: P=dy, Q=x; convT: P=x, Q=dy); beta = 0 write (zero_grad folded in).
//
// Two generations of kernels live here, selected per launch by the host (launch()):
//
//  * RAW-WINDOW kernels (conv_raw_kernel F/T, conv_g_raw_kernel) -- the fast path for every layer geometry of the U-Net
//    except k = 5: workgroup tile 128 (M) x 256 (N), 4 waves of 64 x 128 (128 accumulator registers, 2 waves/SIMD).
//    The weight / P tile is gathered by LDS-DMA into a swizzled K-contiguous image; the ACTIVATION operand is staged
//    as raw row windows (every element once) and the im2col overlap is resolved when fragments are read.  In-kernel
//    stamps showed that global->LDS bytes per MFMA is what limits these kernels (LDS-DMA sustains ~5 B/clk per CU for
//    these gathers): raw windows + a tile that is wide on the activation side cut those bytes by ~60 %.
//  * IM2COL kernels (conv_f/t/g_kernel) -- 256 x 128 tile, both operands gathered element by element by LDS-DMA with
//    per-lane source addresses (im2col, phase split, zero padding and the XOR swizzle all live in the address).  They
//    serve k = 5, generic (k, s) and shapes whose windows do not fit, and stay covered by the tests (schedule bit 2).
//
// Common to both: operands reach LDS through buffer_load ... lds (no staging registers, no ds_write; out-of-range lanes
// write 0.0, which implements conv padding, tile edges and K tails); (Leaky)ReLU in front of every conv is applied
// branch-free on the MFMA fragments, so the in-place activations (model.py:80,82) and torch.cat (model.py:113) are
// never materialised; phase order per slab is pinned with sched_barrier(0): gathers for slab s+1, then fragment reads
// + MFMAs of slab s, then one __syncthreads() whose vmcnt(0) therefore sits behind the matrix work; double-buffered
// LDS.  Work decomposition is a persistent stream-K split over (tile, slab) with a deterministic fixup kernel (below).
// Each output element is accumulated in a fixed order: results are bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "phasegen.h"
#include "pg_common.h"

namespace {

constexpr int WMB = 4;                    // 32-row MFMA blocks per wave along M: wave tile (32*WMB) x 64
constexpr int BM = 64 * WMB, BN = 128, BK = 16, NT = 256;   // workgroup tile 256 x 128, waves 2 (M) x 2 (N)
constexpr int TILE_A = BM * BK, TILE_B = BN * BK;   // floats per operand tile (16 KB + 8 KB)
constexpr int STAGE = TILE_A + TILE_B;    // one LDS stage; two stages = 48 KB
constexpr int AE = BM / 16;               // dword gather pieces per thread and slab for the A tile (B tile: 8)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct IgemmParams {
    const float* x; long x_bs;       // B-operand source (F,T: input activations; G: the "Q" tensor)
    const float* w;                  // F,T: weights (A operand)
    const float* pt; long pt_bs;     // G: the "P" tensor (A operand), (B, M, LP)
    float* y; long y_bs;             // F,T: output activations; G: dW
    const float* add; long add_bs;   // optional epilogue addend (same shape as y)
    const float* ref; long ref_bs;   // optional epilogue mask source (same shape as y)
    unsigned x_bytes, w_bytes, pt_bytes;   // extents for the buffer descriptors (hardware bounds check)
    int B, Q, M, Lx, Ly, k, s, p;    // Q: channels of x; M: output channels (F,T) / channels of P (G)
    int act_x, act_p, mask_mode;
    int U, u_off;                    // T: positions per phase, first u
    int LP; float inv_LP;            // G: frames of P and 1/LP
    int a_vec;                       // F: weight rows may be read as aligned float4
    int tilesM, tilesN;
    float* y2; long y2_bs; float y_slope, y2_slope;   // F,T fwd: activation on store, optional second output
    float* ws;                       // stream-K partial-tile workspace: [grid][2][64][256] floats (or NULL)
    int nslab;                       // K slabs per tile
};

// Activations are applied branch-free on the MFMA fragments as max(v, slope*v) -- exact for 0 <= slope <= 1:
// slope 1 = identity, 0.2 = LeakyReLU(0.2) (model.py:80), 0 = ReLU (model.py:82).  A runtime switch per element would
// make hipcc branch around every gathered value; callers test `slope != 1` once per slab (wave-uniform).
__host__ __device__ __forceinline__ float act_slope(int act) {
    return act == PG_ACT_LEAKY02 ? 0.2f : (act == PG_ACT_RELU ? 0.0f : 1.0f);
}
__device__ __forceinline__ float act_apply(float v, float slope) { return fmaxf(v, slope * v); }

// Operand gathers go through buffer descriptors: a lane whose element is padding / out of the tile / past K gets
// the offset OOB and the hardware returns 0.0 -- no exec-masked branch around the load, no 64-bit address math.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int OOB = 0x7ffffff0;
__device__ __forceinline__ rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, int elem_off, bool ok) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, ok ? elem_off * 4 : OOB, 0, 0));
}

// XCD-aware, bijective remap of the linear workgroup id: hardware deals consecutive ids round-robin over the
// 8 XCDs; give every XCD a contiguous run of tiles (same weight panel => private-L2 hits).  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, local = bid >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

template <int MB, int NB> struct AccT { f32x16 c[MB][NB]; };   // MB x NB blocks of 32x32 per wave
using Acc = AccT<WMB, 2>;       // im2col kernels: wave tile 128 x 64
using AccR = AccT<2, 4>;        // raw-window kernels: wave tile 64 x 128

// ---- LDS tile image --------------------------------------------------------------------------------------------
// One operand tile = 128 rows x 16 k, UNPADDED (64-B rows), 16-B chunks XOR-swizzled by (row>>2)&3: element (r,k) sits
// at dword r*16 + ((k>>2) ^ ((r>>2)&3))*4 + (k&3).  Unpadded because LDS-DMA writes 64 consecutive dwords per wave
// instruction; swizzled so the fragment ds_read_b128 (lanes = 32 consecutive rows, same logical chunk) is
// conflict-free.  The swizzle is applied on the DMA *source* side: instruction e of wave w fills dwords
// [(4e+w)*64, +64), i.e. lane L carries row 16e + 4w + (L>>4) and logical chunk ((L>>2)&3) ^ w -- so a thread owns ONE
// k column (kt) and eight rows, and its (channel, tap) decode is done once per slab.
__device__ __forceinline__ int dma_kt(int lane, int w) { return ((((lane >> 2) & 3) ^ w) << 2) | (lane & 3); }
__device__ __forceinline__ int dma_row(int lane, int w, int e) { return 16 * e + 4 * w + (lane >> 4); }

// 16 bytes per lane: one wave instruction fills 16 rows x 64 B.  Lane L lands on 16-B chunk (4e+w)*64 + L of the tile
// image: row (4e+w)*16 + (L>>2), physical chunk L&3, i.e. logical chunk (L&3) ^ ((L>>4)&3) of that row.
__device__ __forceinline__ int dma16_row(int lane, int w, int e) { return (4 * e + w) * 16 + (lane >> 2); }
__device__ __forceinline__ int dma16_kc(int lane) { return ((lane & 3) ^ ((lane >> 4) & 3)) << 2; }
#ifndef PG_ABL
#define PG_ABL 0
#endif
#if PG_ABL == 4     /* dev ablation: identical instruction stream, every gather address folded into a 1 KB window */
#define PG_ADDR(x) ((x) & 0x3f0)
#else
#define PG_ADDR(x) (x)
#endif
__device__ __forceinline__ void dma16(rsrc_t r, float* lds_wave_uniform, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_wave_uniform, 16, PG_ADDR(byte_off), 0, 0, 0);
}
__device__ __forceinline__ void dma4(rsrc_t r, float* lds_wave_uniform, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_wave_uniform, 4, PG_ADDR(byte_off), 0, 0, 0);
}

// bf16 operand mode (pg_conv_set_precision(1)): the same fragments -- lane (row, h) already holds k = 8h .. 8h+7 of the
// slab, which is exactly the operand layout of v_mfma_f32_32x32x16_bf16 -- are rounded to bf16 (RNE, v_cvt_pk_bf16_f32)
// after the activation and one MFMA replaces eight; accumulation stays fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 to_bf16x8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    bf16x8 r;
    r[0] = (__bf16)v0; r[1] = (__bf16)v1; r[2] = (__bf16)v2; r[3] = (__bf16)v3;
    r[4] = (__bf16)v4; r[5] = (__bf16)v5; r[6] = (__bf16)v6; r[7] = (__bf16)v7;
    return r;
}
__device__ __forceinline__ bf16x8 to_bf16x8(const f32x4 (&v)[2]) { return to_bf16x8(v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]); }
__device__ __forceinline__ bf16x8 to_bf16x8(const float (&v)[8]) { return to_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]); }
template <int MB, int NB>
__device__ __forceinline__ void mfma_bf16(const bf16x8 (&A)[MB], const bf16x8 (&B)[NB], AccT<MB, NB>& acc) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[j], acc.c[i][j], 0, 0, 0);
}
// Split mode (pg_conv_set_precision(2), "bf16x3"): every fp32 operand is written as hi + lo with hi = bf16(x) and
// lo = bf16(x - hi) (the subtraction is exact), and the product is taken as hi*hi' + hi*lo' + lo*hi' on the bf16 pipe:
// three MFMAs at 1/16 of the fp32 cost each.  Dropped: lo*lo' and the two representation residuals, each <= 2^-18 of the
// product, i.e. a relative error of ~1e-5 per product against fp32's 6e-8 -- inside the 1e-4 parity bound, NOT fp32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_bf16(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    u32x4 H, L;                                      // pairwise: one v_cvt_pk per two values, shifts/masks to widen hi back
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 x = {v[2 * i], v[2 * i + 1]};
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        const unsigned P = __builtin_bit_cast(unsigned, h);
        const f32x2 hf = {__uint_as_float(P << 16), __uint_as_float(P & 0xffff0000u)};
        const bf16x2 l = __builtin_convertvector(x - hf, bf16x2);
        H[i] = P;
        L[i] = __builtin_bit_cast(unsigned, l);
    }
    hi = __builtin_bit_cast(bf16x8, H);
    lo = __builtin_bit_cast(bf16x8, L);
}
template <int PM, int MB, int NB>
__device__ __forceinline__ void mfma_low(const float (&a)[MB][8], const float (&b)[NB][8], AccT<MB, NB>& acc) {
    if (PM == 1) {
        bf16x8 A[MB], B[NB];
#pragma unroll
        for (int i = 0; i < MB; ++i) A[i] = to_bf16x8(a[i]);
#pragma unroll
        for (int j = 0; j < NB; ++j) B[j] = to_bf16x8(b[j]);
        mfma_bf16<MB, NB>(A, B, acc);
    } else {
        // column block by column block, so the split of block j+1 (VALU) can run under the six MFMAs of block j; the same
        // accumulator is touched every MB MFMAs (small terms first)
        bf16x8 Ah[MB], Al[MB], Bh[2], Bl[2];
#pragma unroll
        for (int i = 0; i < MB; ++i) split_bf16(a[i], Ah[i], Al[i]);
        split_bf16(b[0], Bh[0], Bl[0]);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j + 1 < NB) split_bf16(b[j + 1], Bh[(j + 1) & 1], Bl[(j + 1) & 1]);
#pragma unroll
            for (int i = 0; i < MB; ++i) acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[i], Bh[j & 1], acc.c[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MB; ++i) acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[i], Bl[j & 1], acc.c[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MB; ++i) acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[i], Bh[j & 1], acc.c[i][j], 0, 0, 0);
            if (j + 1 < NB) {
#pragma unroll
                for (int g = 0; g < 3 * MB; ++g) {       // interleave: one MFMA, then a slice of the next block's split
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, (21 + 3 * MB - 1) / (3 * MB), 0);
                }
            }
        }
    }
}
template <int N>
__device__ __forceinline__ void flatten(const f32x4 (&v)[N][2], float (&o)[N][8]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[i][4 * c + e] = v[i][c][e];
}
template <int PM>
__device__ __forceinline__ void mfma_low_2x4(const f32x4 (&a)[2][2], const float (&b)[4][8], AccR& acc) {
    float af[2][8];
    flatten<2>(a, af);
    mfma_low<PM, 2, 4>(af, b, acc);
}

// One BK=16 slab: (2*WMB + 4) x ds_read_b128 (swizzled), optional activation on the fragments, 8 k-pairs x 2*WMB MFMA.
template <int BF>
__device__ __forceinline__ void mma_slab(const float* __restrict__ As, const float* __restrict__ Bs,
                                         int lane, int wm, int wn, float slopeA, float slopeB, Acc& acc) {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 2) & 3;
    const float* ap = As + (wm * (WMB * 32) + r) * BK;
    const float* bp = Bs + (wn * 64 + r) * BK;
    f32x4 a[WMB][2], b[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i = 0; i < WMB; ++i) a[i][c] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BK + (((2 * h + c) ^ sw) << 2));
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i][c] = *reinterpret_cast<const f32x4*>(bp + i * 32 * BK + (((2 * h + c) ^ sw) << 2));
    }
    if (slopeA != 1.0f) {
#pragma unroll
        for (int i = 0; i < WMB; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int v = 0; v < 4; ++v) a[i][c][v] = act_apply(a[i][c][v], slopeA);
    }
    if (slopeB != 1.0f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int v = 0; v < 4; ++v) b[i][c][v] = act_apply(b[i][c][v], slopeB);
    }
    if (BF) {
        float af[WMB][8], bf[2][8];
        flatten<WMB>(a, af);
        flatten<2>(b, bf);
        mfma_low<BF, WMB, 2>(af, bf, acc);
        return;
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int i = 0; i < WMB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk >> 2][kk & 3], b[j][kk >> 2][kk & 3], acc.c[i][j], 0, 0, 0);
}

// Fused dgrad epilogue: v = (acc + add) * act'(ref).  A missing addend / mask source is an EMPTY descriptor (every
// load returns 0) and slope 1, so the same branch-free code serves all combinations.
struct Epi {
    rsrc_t radd, rref; float slope; bool fused;
    __device__ __forceinline__ Epi(const IgemmParams& p, unsigned ybytes)
        : radd(make_rsrc(p.add, p.add ? ybytes_of(p.add_bs, p, ybytes) : 0u)),
          rref(make_rsrc(p.ref, (p.ref && p.mask_mode) ? ybytes_of(p.ref_bs, p, ybytes) : 0u)),
          slope((p.ref && p.mask_mode) ? act_slope(p.mask_mode) : 1.0f), fused(p.add || (p.ref && p.mask_mode)) {}
    static __device__ __forceinline__ unsigned ybytes_of(long bs, const IgemmParams& p, unsigned ybytes) {
        return (unsigned)(((long)(p.B - 1) * bs) * 4) + ybytes;
    }
    __device__ __forceinline__ float operator()(float v, int off_add, int off_ref) const {
        v += bload(radd, off_add, true);
        return v * (bload(rref, off_ref, true) > 0.f ? 1.0f : slope);
    }
};

// ---- work decomposition (stream-K) -------------------------------------------------------------------------------
// The launch is a grid of G workgroups over the linearised (tile, slab) space of tiles*nslab units; workgroup g owns
// the contiguous range [lo(g), lo(g+1)).  With G == tiles every workgroup owns exactly one whole tile (the plain
// data-parallel GEMM).  With G == resident workgroup slots (host picks that when the tile count quantises badly over
// 256 CUs, e.g. 1040 tiles) every CU gets the same number of MFMAs: a range then starts / ends inside tiles, those
// segments leave their accumulators in the workspace (slot 0 = the range's first segment, slot 1 = its last) and the
// fixup kernel adds a tile's segments in ascending workgroup order and runs the epilogue.  No atomics, no flags, no
// inter-workgroup ordering assumption: results are bit-reproducible.
struct Split { int total, q, r; };
__device__ __host__ __forceinline__ Split make_split(int tiles, int nslab, int G) {
    Split sp; sp.total = tiles * nslab; sp.q = sp.total / G; sp.r = sp.total - sp.q * G; return sp;
}
__device__ __host__ __forceinline__ int split_lo(const Split& sp, int g) { return g < sp.r ? g * (sp.q + 1) : sp.r * (sp.q + 1) + (g - sp.r) * sp.q; }
__device__ __host__ __forceinline__ int split_owner(const Split& sp, int x) {
    const int big = sp.r * (sp.q + 1);
    return x < big ? x / (sp.q + 1) : sp.r + (x - big) / sp.q;
}

constexpr int ACC_REGS = 128;             // accumulator registers per thread in both tile configurations (8 blocks x 16)
template <int MB, int NB>
__device__ __forceinline__ void store_partial(float* ws, int g, int slot, const AccT<MB, NB>& acc, int tid) {
    static_assert(MB * NB * 16 == ACC_REGS, "partial-tile slots are sized for 8 blocks per wave");
    float* dst = ws + ((long)(g * 2 + slot) * ACC_REGS) * NT + tid;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((i * NB + j) * 16 + r) * NT] = acc.c[i][j][r];
}

// ---- epilogues (shared by the GEMM kernels and the fixup kernels) -------------------------------------------------
// acc reg r of block (i,j): row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31; wave (wm, wn) owns rows
// wm*32*MB + ..., cols wn*32*NB + ...
template <int S, int MB, int NB>
__device__ __forceinline__ void epilogue_f(const IgemmParams& p, const AccT<MB, NB>& acc, int m0, int n0, int lane, int wm, int wn) {
    const int Ntot = p.B * p.Ly;
    const Epi ep(p, (unsigned)((long)p.M * p.Ly * 4));
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * (NB * 32) + j * 32 + (lane & 31);
        if (n >= Ntot) continue;
        const int b = n / p.Ly, t = n - b * p.Ly;
        float* yb = p.y + (long)b * p.y_bs;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (MB * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    const int off = m * p.Ly + t;
                    float v = acc.c[i][j][r];
                    if (ep.fused) v = ep(v, b * (int)p.add_bs + off, b * (int)p.ref_bs + off);
                    yb[off] = act_apply(v, p.y_slope);
                    if (p.y2) p.y2[(long)b * p.y2_bs + off] = act_apply(v, p.y2_slope);
                }
            }
    }
}

template <int S, int MB, int NB>
__device__ __forceinline__ void epilogue_t(const IgemmParams& p, const AccT<MB, NB>& acc, int m0, int n0, int lane, int wm, int wn) {
    const int s = S ? S : p.s;
    const int Ntot = p.B * p.U, Mrows = p.M * s;
    const Epi ep(p, (unsigned)((long)p.M * p.Ly * 4));
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * (NB * 32) + j * 32 + (lane & 31);
        if (n >= Ntot) continue;
        const int b = n / p.U, u = n - b * p.U + p.u_off;
        float* yb = p.y + (long)b * p.y_bs;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = m0 + wm * (MB * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int o = mr / s, phi = mr - o * s, tau = s * u + phi - p.p;
                if (mr < Mrows && tau >= 0 && tau < p.Ly) {
                    const int off = o * p.Ly + tau;
                    float v = acc.c[i][j][r];
                    if (ep.fused) v = ep(v, b * (int)p.add_bs + off, b * (int)p.ref_bs + off);
                    yb[off] = act_apply(v, p.y_slope);
                    if (p.y2) p.y2[(long)b * p.y2_bs + off] = act_apply(v, p.y2_slope);
                }
            }
    }
}

template <int S, int MB, int NB>
__device__ __forceinline__ void epilogue_g(const IgemmParams& p, const AccT<MB, NB>& acc, int m0, int n0, int lane, int wm, int wn) {
    const int Ntot = p.Q * p.k;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * (NB * 32) + j * 32 + (lane & 31);
        if (n >= Ntot) continue;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (MB * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) p.y[(long)m * Ntot + n] = acc.c[i][j][r];
            }
    }
}

// Body shared by the three GEMM kernels.  SETUP computes this thread's per-row gather constants for tile (m0, n0);
// ISSUE enqueues the 16 LDS-DMA gathers of one slab (8 per operand per thread) into the LDS buffer (As, Bs): there are
// no staging registers and no ds_write.  The only wait is the vmcnt(0) that __syncthreads() carries, and it sits AFTER
// the slab's 32 MFMAs (phase order pinned with sched_barrier: hipcc otherwise hoists the register-only MFMAs above the
// gather issue), so gather latency is covered by matrix work.  buf^1 is refilled while buf is read: its previous
// readers all passed the barrier that ended the last iteration.
#if PG_ABL == 7   /* dev-only: s_memtime stamps around the three phases of a slab; sums go to p.ws (u64 x 4) */
#define PG_STAMP(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); if (i) st_sum[i - 1] += t_ - st_prev; st_prev = t_; }
#define PG_STAMP_FLUSH if (lane == 0) { for (int i_ = 0; i_ < 3; ++i_) atomicAdd((unsigned long long*)p.ws + i_, st_sum[i_]); \
                                        atomicAdd((unsigned long long*)p.ws + 3, (unsigned long long)(se - sb)); }
#define PG_STAMP_DECL unsigned long long st_sum[3] = {0, 0, 0}, st_prev = 0;
#else
#define PG_STAMP(i)
#define PG_STAMP_FLUSH
#define PG_STAMP_DECL
#endif
#define PG_BODY(SETUP, ISSUE, ...)   /* variadic tail = the epilogue call (its template arguments contain commas) */                                                             \
    const int tid = threadIdx.x, lane = tid & 63;                                                   \
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv >> 1, wn = wv & 1;             \
    const int kw = KW ? KW : p.k, s = S ? S : p.s;                                                  \
    const int kt = dma_kt(lane, wv);                                                                \
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];                                   \
    const int g = xcd_remap(blockIdx.x, gridDim.x);                                                 \
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x);                           \
    int pos = split_lo(sp, g);                                                                      \
    const int pos_end = split_lo(sp, g + 1);                                                        \
    int slot = 0;                                                                                   \
    while (pos < pos_end) {                                                                         \
        const int tile = pos / p.nslab, sb = pos - tile * p.nslab;                                  \
        const int se = min(p.nslab, sb + (pos_end - pos));                                          \
        const int m0 = (tile / p.tilesN) * BM, n0 = (tile % p.tilesN) * BN;                         \
        SETUP                                                                                       \
        PG_STAMP_DECL                                                                               \
        Acc acc;                                                                                    \
        _Pragma("unroll") for (int i = 0; i < WMB; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc.c[i][j][r] = 0.f;                    \
        { float* const As = lds + wv * 64; float* const Bs = As + TILE_A; const int k0 = sb * BK; ISSUE } \
        __syncthreads();                                                                            \
        for (int sl = sb; sl < se; ++sl) {                                                          \
            const int cur = (sl - sb) & 1;                                                          \
            PG_STAMP(0)                                                                             \
            { float* const As = lds + (cur ^ 1) * STAGE + wv * 64; float* const Bs = As + TILE_A;   \
              const int k0 = (sl + 1) * BK;    /* past-the-end slab gathers only zeros */           \
              ISSUE }                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            PG_STAMP(1)                                                                             \
            mma_slab<BF>(lds + cur * STAGE, lds + cur * STAGE + TILE_A, lane, wm, wn, slopeA, slopeB, acc); \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            PG_STAMP(2)                                                                             \
            __syncthreads();                                                                        \
            PG_STAMP(3)                                                                             \
        }                                                                                           \
        PG_STAMP_FLUSH                                                                                           \
        if (sb == 0 && se == p.nslab) { __VA_ARGS__ }                                               \
        else store_partial(p.ws, g, slot, acc, tid);                                                \
        pos += se - sb;                                                                             \
        slot = 1;                                                                                   \
    }

constexpr int NEVER = 0x40000000;   // a "first valid tap/position" no index ever reaches: marks rows outside the tile
// Out-of-range byte offsets that replace per-element predicates.  Descriptors span < 0x7ffffff0 bytes, so with
//   FAR (rows outside the tile) = 0x80000000 and OOB (slabs past K) = 0x7ffffff0
// every sum {valid row + OOB, FAR + valid k offset, FAR + OOB} stays >= 0x7ffffff0 as an unsigned 32-bit value and
// never wraps back into range (FAR + FAR would: the two invalid cases therefore use different constants).
constexpr int FAR = (int)0x80000000u;

// ------------------------------------------------------------------------------------------------------------
// F kernel
// ------------------------------------------------------------------------------------------------------------
template <int KW, int S, int BF>
__global__ __launch_bounds__(NT, 2) void conv_f_kernel(const IgemmParams p) {
    const int Ktot = p.Q * (KW ? KW : p.k), Ntot = p.B * p.Ly;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);
    const float slopeA = 1.0f, slopeB = act_slope(p.act_x);
#define F_SETUP                                                                                       \
    int aoff[AE], xoff[8], jlo[8];     /* per-row constants (BYTE offsets) of this thread's A rows / 8 B rows */ \
    _Pragma("unroll") for (int e = 0; e < AE; ++e) {                                                  \
        const int m = m0 + dma_row(lane, wv, e);                                                      \
        aoff[e] = (!p.a_vec && m < p.M) ? (m * Ktot + kt) * 4 : FAR;                                  \
    }                                                                                                 \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                   \
        const int n = n0 + dma_row(lane, wv, e);                                                      \
        const bool nv = n < Ntot;                                                                     \
        const int b = nv ? n / p.Ly : 0, t = nv ? n - b * p.Ly : 0;                                   \
        xoff[e] = (b * (int)p.x_bs + s * t - p.p) * 4;  /* tap 0, channel 0 */                        \
        jlo[e] = nv ? p.p - s * t : NEVER;              /* taps with 0 <= j - jlo < Lx are inside the row */ \
    }                                                                                                 \
    int avoff[BM / 64];                /* dense weight rows as 16-B pieces (BM/64 per slab instead of BM/16) */ \
    _Pragma("unroll") for (int e = 0; e < BM / 64; ++e) {                                             \
        const int m = m0 + dma16_row(lane, wv, e);                                                    \
        avoff[e] = m < p.M ? (m * Ktot + dma16_kc(lane)) * 4 : FAR;                                   \
    }
#define F_ISSUE                                                                                       \
    { const int kk = k0 + kt, q = kk / kw; const bool kok = kk < Ktot;                                \
      const int j = kok ? kk - q * kw : -NEVER, xq = (q * p.Lx + j) * 4, ka = kok ? k0 * 4 : OOB;     \
      if (p.a_vec) {                                                                                  \
          const int kv = (k0 + dma16_kc(lane) < Ktot) ? k0 * 4 : OOB;                                 \
          _Pragma("unroll") for (int e = 0; e < BM / 64; ++e) dma16(rw, As + wv * 192 + e * 1024, avoff[e] + kv); \
      } else {                                                                                        \
          _Pragma("unroll") for (int e = 0; e < AE; ++e) dma4(rw, As + e * 256, aoff[e] + ka);        \
      }                                                                                               \
      _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                   \
          dma4(rx, Bs + e * 256, (unsigned)(j - jlo[e]) < (unsigned)p.Lx ? xoff[e] + xq : FAR);       \
    }
    PG_BODY(F_SETUP, F_ISSUE, epilogue_f<S, WMB, 2>(p, acc, m0, n0, lane, wm, wn);)
#undef F_SETUP
#undef F_ISSUE
}

// ------------------------------------------------------------------------------------------------------------
// T kernel.  GEMM rows m' = o*s + phi, K = (q, jj) with KJ = ceil(k/s) taps per phase, N = (b, u).
// ------------------------------------------------------------------------------------------------------------
template <int KW, int S, int BF>
__global__ __launch_bounds__(NT, 2) void conv_t_kernel(const IgemmParams p) {
    const int kw_ = KW ? KW : p.k, s_ = S ? S : p.s;
    const int KJ = (kw_ + s_ - 1) / s_;
    const int Ktot = p.Q * KJ, Ntot = p.B * p.U, Mrows = p.M * s_;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);
    const float slopeA = 1.0f, slopeB = act_slope(p.act_x);
    const int wq = p.M * kw_;                 // weight stride between input channels q
    /* every (q, jj) names a real tap when s divides k; otherwise (k5 s2) phase 1 has one tap fewer */ \
#define T_SETUP                                                                                       \
    int aoff[AE], xoff[8], ub[8];                                                                     \
    constexpr bool all_taps = KW != 0 && S != 0 && KW % (S ? S : 1) == 0;                             \
    _Pragma("unroll") for (int e = 0; e < AE; ++e) {                                                  \
        const int mr = m0 + dma_row(lane, wv, e);                                                     \
        const int o = mr / s, phi = mr - o * s;                                                       \
        aoff[e] = mr < Mrows ? (o * kw + phi) * 4 : FAR;     /* W[q][o][s*jj + phi] */                \
    }                                                                                                 \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                   \
        const int n = n0 + dma_row(lane, wv, e);                                                      \
        const bool nv = n < Ntot;                                                                     \
        const int b = nv ? n / p.U : 0, u = (nv ? n - b * p.U : 0) + p.u_off;                         \
        xoff[e] = (b * (int)p.x_bs + u) * 4;            /* X[b][q][u - jj] */                          \
        ub[e] = nv ? u : -NEVER;                        /* position u - jj must lie in [0, Lx) */      \
    }                                                                                                 \
    int phi_of[AE];                                                                                   \
    _Pragma("unroll") for (int e = 0; e < AE; ++e) { const int mr = m0 + dma_row(lane, wv, e); phi_of[e] = all_taps ? 0 : mr - (mr / s) * s; }
#define T_ISSUE                                                                                       \
    { const int kk = k0 + kt, q = kk / KJ; const bool kok = kk < Ktot;                                \
      const int jj = kok ? kk - q * KJ : NEVER, wo = kok ? (q * wq + s * jj) * 4 : OOB, xq = (q * p.Lx - jj) * 4; \
      _Pragma("unroll") for (int e = 0; e < AE; ++e)                                                  \
          dma4(rw, As + e * 256, (all_taps || s * jj + phi_of[e] < kw) ? aoff[e] + wo : FAR);         \
      _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                   \
          dma4(rx, Bs + e * 256, (unsigned)(ub[e] - jj) < (unsigned)p.Lx ? xoff[e] + xq : FAR);       \
    }
    PG_BODY(T_SETUP, T_ISSUE, epilogue_t<S, WMB, 2>(p, acc, m0, n0, lane, wm, wn);)
#undef T_SETUP
#undef T_ISSUE
}

// ------------------------------------------------------------------------------------------------------------
// G kernel.  dW[m][(q,j)] = sum over kk = (b,i) of actP(P[b,m,i]) * actQ(Q[b,q,s*i+j-p]);  Q tensor is p.x.
// ------------------------------------------------------------------------------------------------------------
// n / d for 0 <= n < 2^24 via the float reciprocal, exact after one correction step (branch-free selects).
__device__ __forceinline__ void divmod24(int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { r += d; --q; }
    if (r >= d) { r -= d; ++q; }
}

template <int KW, int S, int BF>
__global__ __launch_bounds__(NT, 2) void conv_g_kernel(const IgemmParams p) {
    const int Ntot = p.Q * (KW ? KW : p.k);
    const rsrc_t rp = make_rsrc(p.pt, p.pt_bytes), rx = make_rsrc(p.x, p.x_bytes);
    const float slopeA = act_slope(p.act_p), slopeB = act_slope(p.act_x);
    const int pbs = (int)p.pt_bs, xbs = (int)p.x_bs;
#define G_SETUP                                                                                       \
    int aoff[AE], xoff[8], jp[8];                                                                     \
    _Pragma("unroll") for (int e = 0; e < AE; ++e) {                                                  \
        const int m = m0 + dma_row(lane, wv, e);                                                      \
        aoff[e] = m < p.M ? m * p.LP * 4 : FAR;         /* P[b][m][i] */                               \
    }                                                                                                 \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                   \
        const int n = n0 + dma_row(lane, wv, e);                                                      \
        const bool nv = n < Ntot;                                                                     \
        const int q = nv ? n / kw : 0, j = nv ? n - q * kw : 0;                                       \
        xoff[e] = (q * p.Lx + j - p.p) * 4;             /* Q[b][q][s*i + j - p] */                     \
        jp[e] = nv ? j - p.p : -NEVER;                                                                \
    }
#define G_ISSUE                                                                                       \
    { int bb, ii; divmod24(k0 + kt, p.LP, p.inv_LP, bb, ii);                                          \
      const bool kok = bb < p.B; const int po = kok ? (bb * pbs + ii) * 4 : OOB, xo = (bb * xbs + s * ii) * 4; \
      const int si = kok ? s * ii : -NEVER;       /* with jp = -NEVER the sum is still far below 0 */                                                            \
      _Pragma("unroll") for (int e = 0; e < AE; ++e) dma4(rp, As + e * 256, aoff[e] + po);            \
      _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                   \
          dma4(rx, Bs + e * 256, (unsigned)(si + jp[e]) < (unsigned)p.Lx ? xoff[e] + xo : FAR);       \
    }
    PG_BODY(G_SETUP, G_ISSUE, epilogue_g<S, WMB, 2>(p, acc, m0, n0, lane, wm, wn);)
#undef G_SETUP
#undef G_ISSUE
}

// ================================================================================================================
// Raw-window variants of the F and T kernels: workgroup tile 128 (M) x 256 (N), wave tile 64 x 128.
//
// The stamps showed that what limits the im2col kernels is the number of bytes moved global -> LDS per MFMA (LDS-DMA
// sustains only a few B/clk per CU).  An im2col tile holds every activation element k/s times.  Here the activation
// operand is staged RAW: for each channel of a slab one contiguous window of the input row (every element once,
// zero-filled outside [0, Lx)), and the im2col overlap is resolved when the MFMA fragments are read: column c of the
// tile reads taps at window offset vcol(c) + tap (F) or vcol(c) - tap (T), vcol(c) = s'*c + 16*seg(c), where seg(c)
// counts the sample boundaries between column 0 and c (a 16-float gap per boundary keeps windows of different samples
// apart).  The tile is made wide on the activation side, where bytes are now cheap: per slab 8 KB of weights plus ~1-2 KB
// of activations feed 128x256x16 MACs -- 60 % fewer global->LDS bytes per MFMA than the 256x128 im2col tiling.
// Supported when the taps per channel in K order (F: k, T: k/s) are 4, 8, 16 or 32 and the windows fit RS floats;
// everything else (k = 5, generic) stays on the im2col kernels.
// ================================================================================================================
constexpr int RBM = 128, RBN = 256;       // raw-window workgroup tile
constexpr int RS2 = 768, RS1 = 384;       // floats reserved per channel window for column stride 2 / 1
constexpr int RG = 16;                    // gap between the windows of consecutive samples inside a tile
constexpr int RTILE_A = RBM * BK;         // weight tile, same swizzled image as above (8 KB)

// ds_read_b32-based B fragments: lane (column block jb, column r, half h) needs k = 8h .. 8h+7 of the slab, i.e.
// (channel qi, tap tau) = divmod(8h + i, TJ); the element lives at  qi*RS + bbase[jb] +/- tau.
struct RawFrags { f32x4 a[2][2]; float b[4][8]; };

template <int TJ, bool DESC, int RS>
__device__ __forceinline__ void raw_load_frags(const float* __restrict__ As, const float* __restrict__ Bw, int lane, int wm,
                                               const int (&bbase)[4], float slopeA, float slopeB, RawFrags& f) {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 2) & 3;
    const float* ap = As + (wm * 64 + r) * BK;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i) f.a[i][c] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BK + (((2 * h + c) ^ sw) << 2));
    // lane part of the index: TJ == 16 -> one channel, taps 8h + i;  TJ <= 8 -> channels (8/TJ)*h + i/TJ, taps i % TJ
    const int lanepart = (TJ == 16) ? (DESC ? -8 * h : 8 * h) : (8 / TJ) * h * RS;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        const float* bp = Bw + bbase[jb] + lanepart;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int qoff = (TJ == 16) ? 0 : (i / TJ) * RS, tau = (TJ == 16) ? i : (i % TJ);
            f.b[jb][i] = bp[qoff + (DESC ? -tau : tau)];
        }
    }
    if (slopeA != 1.0f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int v = 0; v < 4; ++v) f.a[i][c][v] = act_apply(f.a[i][c][v], slopeA);
    }
    if (slopeB != 1.0f) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int i = 0; i < 8; ++i) f.b[jb][i] = act_apply(f.b[jb][i], slopeB);
    }
}

template <int BF>
__device__ __forceinline__ void raw_mfma(const RawFrags& f, AccR& acc) {
    if (BF) { mfma_low_2x4<BF>(f.a, f.b, acc); return; }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][kk >> 2][kk & 3], f.b[j][kk], acc.c[i][j], 0, 0, 0);
}

template <int TJ, bool DESC, int RS, int BF>
__device__ __forceinline__ void mma_slab_raw(const float* __restrict__ As, const float* __restrict__ Bw, int lane, int wm,
                                             const int (&bbase)[4], float slopeA, float slopeB, AccR& acc) {
    RawFrags f;
    raw_load_frags<TJ, DESC, RS>(As, Bw, lane, wm, bbase, slopeA, slopeB, f);
    raw_mfma<BF>(f, acc);
}

// TKIND false: F (conv fwd / convT dgrad, taps ascend with stride s between columns)
// TKIND true : T (convT fwd / conv dgrad in gather form, unit column stride, taps descend)
template <int KW, int S, bool TKIND, int BF>
__global__ __launch_bounds__(NT, 2) void conv_raw_kernel(const IgemmParams p) {
    constexpr int KWP = TKIND ? KW / S : KW;          // taps per channel in K order
    constexpr int TJ = KWP < 16 ? KWP : 16, NQ = 16 / TJ;
    constexpr int SC = TKIND ? 1 : S;                 // window positions per column step
    constexpr int RS = SC == 1 ? RS1 : RS2;           // floats reserved per channel window
    constexpr int NPC = (RS + NT - 1) / NT;           // gather pieces per thread and window
    constexpr int STG = RTILE_A + NQ * RS;            // floats per LDS stage
    constexpr int SPB = (4 * STG * 4 <= 64 * 1024) ? 2 : 1;   // slabs per barrier (two when both stages still fit 64 KB)
    static_assert(KWP == 4 || KWP == 8 || KWP == 16 || KWP == 32, "raw-window kernels need 4/8/16/32 taps per channel");
    static_assert(!TKIND || KW % S == 0, "T raw kernel needs s | k");
    __shared__ __attribute__((aligned(16))) float lds[2 * SPB * STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv >> 1, wn = wv & 1;
    const int kt = dma_kt(lane, wv);
#if PG_ABL == 8   /* dev-only: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6) */
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int Lcol = TKIND ? p.U : p.Ly;              // columns (output positions) per sample
    const int Ktot = p.Q * KWP, Mrows = TKIND ? p.M * S : p.M;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);
    const float slopeA = 1.0f, slopeB = act_slope(p.act_x);
    const int wq = p.M * KW;                          // T: weight stride between input channels
    const int g = xcd_remap(blockIdx.x, gridDim.x);
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x);
    int pos = split_lo(sp, g);
    const int pos_end = split_lo(sp, g + 1);
    int slot = 0;
    while (pos < pos_end) {
        const int tile = pos / p.nslab, sb = pos - tile * p.nslab;
        const int se = min(p.nslab, sb + (pos_end - pos));
        const int m0 = (tile / p.tilesN) * RBM, n0 = (tile % p.tilesN) * RBN;
        const int b0 = n0 / Lcol, t0 = n0 - b0 * Lcol;            // sample / position of the tile's first column
        const int nseg = (t0 + RBN - 1) / Lcol + 1;
        const int rlen = SC * (RBN - 1) + TJ + RG * (nseg - 1);   // floats of a channel window that are ever read

        // --- weight-tile gather constants (BYTE offsets) --------------------------------------------------------
        int aoff[8], avoff[2];
        if (TKIND) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int mr = m0 + dma_row(lane, wv, e), o = mr / S, phi = mr - o * S;
                aoff[e] = mr < Mrows ? (o * KW + phi) * 4 : FAR;              // W[q][o][S*jj + phi]
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = m0 + dma_row(lane, wv, e);
                aoff[e] = (!p.a_vec && m < p.M) ? (m * Ktot + kt) * 4 : FAR;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int m = m0 + dma16_row(lane, wv, e);
                avoff[e] = m < p.M ? (m * Ktot + dma16_kc(lane)) * 4 : FAR;
            }
        }
        // --- window gather constants: thread owns window positions v = tid + 256 e --------------------------------
        int posb[NPC], rowb[NPC];
#pragma unroll
        for (int e = 0; e < NPC; ++e) {
            const int v = tid + 256 * e;
            int k = 0;                                            // segment (sample) this window position belongs to
            while (k + 1 < nseg && SC * ((k + 1) * Lcol - t0) + RG * (k + 1) <= v) ++k;
            const int cs = k ? k * Lcol - t0 : 0;                 // first column of the segment
            const int vl = v - (SC * cs + RG * k);                // position inside the segment's window
            const int tf = k ? 0 : t0;                            // frame index of that first column
            const int b = b0 + k;
            // F: memory position = s*t - p + tau;  T: u - tau with u = u_off + t, stored ascending from u - (TJ-1)
            posb[e] = b < p.B ? (TKIND ? p.u_off + tf - (TJ - 1) + vl : S * tf - p.p + vl) : -NEVER;
            rowb[e] = b * (int)p.x_bs * 4;
        }
        // --- fragment bases: window offset of each of this lane's 4 columns ----------------------------------------
        int bbase[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int c = wn * 128 + jb * 32 + (lane & 31);
            bbase[jb] = SC * c + RG * ((t0 + c) / Lcol) + (TKIND ? TJ - 1 : 0);
        }

        AccR acc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.c[i][j][r] = 0.f;

#define RAW_ISSUE(STAGE_PTR, K0)                                                                          \
    {   float* const As = (STAGE_PTR) + wv * 64; float* const Bw = (STAGE_PTR) + RTILE_A + wv * 64;       \
        const int k0 = (K0);                                                                              \
        const bool kok = k0 < Ktot;                                                                       \
        if (TKIND) {                                                                                      \
            const int kk = k0 + kt, q = kk / KWP, jj = kk - q * KWP;                                      \
            const int wo = kok ? (q * wq + S * jj) * 4 : OOB;                                             \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) dma4(rw, As + e * 256, aoff[e] + wo);           \
        } else if (p.a_vec) {                                                                             \
            const int kv = (k0 + dma16_kc(lane) < Ktot) ? k0 * 4 : OOB;                                   \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) dma16(rw, As + wv * 192 + e * 1024, avoff[e] + kv); \
        } else {                                                                                          \
            const int ka = (k0 + kt < Ktot) ? k0 * 4 : OOB;                                               \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) dma4(rw, As + e * 256, aoff[e] + ka);           \
        }                                                                                                 \
        const int q0 = k0 / KWP, tau0 = k0 - q0 * KWP;    /* tau0 != 0 only when a channel spans two slabs */ \
        _Pragma("unroll") for (int qi = 0; qi < NQ; ++qi) {                                               \
            const int qq = q0 + qi;                                                                       \
            const int qo = (kok && qq < p.Q) ? qq * p.Lx : -NEVER;                                        \
            _Pragma("unroll") for (int e = 0; e < NPC; ++e) {                                             \
                if (e * 256 + wv * 64 < rlen) {                                                           \
                    const int ps = posb[e] + (TKIND ? -tau0 : tau0);                                      \
                    const bool ok = (unsigned)ps < (unsigned)p.Lx && qo >= 0;                             \
                    dma4(rx, Bw + qi * RS + e * 256, ok ? rowb[e] + (qo + ps) * 4 : FAR);                 \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
    }

        PG_STAMP_DECL
        // Two 16-deep slabs per barrier when the stage pair fits (SPB = 2): each LDS stage holds two half-stages that are
        // gathered together and multiplied one after the other, halving the barrier (and gather-burst) rate.  Fragments
        // are still loaded 16 deep, so the register budget is unchanged.  The second half is skipped when it lies past
        // this segment's end (it belongs to the next workgroup's range, or past K where the gathers returned zeros).
        constexpr int SSTG = SPB * STG;
        static_assert(2 * SSTG * 4 <= 64 * 1024, "LDS budget");
#pragma unroll
        for (int hf = 0; hf < SPB; ++hf) RAW_ISSUE(lds + hf * STG, (sb + hf) * BK)
        __syncthreads();
        for (int sl = sb; sl < se; sl += SPB) {
            const int cur = ((sl - sb) / SPB) & 1;
            PG_STAMP(0)
#pragma unroll
            for (int hf = 0; hf < SPB; ++hf) RAW_ISSUE(lds + (cur ^ 1) * SSTG + hf * STG, (sl + SPB + hf) * BK)
            __builtin_amdgcn_sched_barrier(0);
            PG_STAMP(1)
            mma_slab_raw<TJ, TKIND, RS, BF>(lds + cur * SSTG, lds + cur * SSTG + RTILE_A, lane, wm, bbase, slopeA, slopeB, acc);
            if (SPB == 2 && sl + 1 < se)
                mma_slab_raw<TJ, TKIND, RS, BF>(lds + cur * SSTG + STG, lds + cur * SSTG + STG + RTILE_A, lane, wm, bbase, slopeA, slopeB, acc);
            __builtin_amdgcn_sched_barrier(0);
            PG_STAMP(2)
            __syncthreads();
            PG_STAMP(3)
        }
        PG_STAMP_FLUSH
#undef RAW_ISSUE
        if (sb == 0 && se == p.nslab) {
            if (TKIND) epilogue_t<S, 2, 4>(p, acc, m0, n0, lane, wm, wn);
            else epilogue_f<S, 2, 4>(p, acc, m0, n0, lane, wm, wn);
        } else store_partial(p.ws, g, slot, acc, tid);
        pos += se - sb;
        slot = 1;
    }
#if PG_ABL == 8
    if (tid == 0 && p.ws && blockIdx.x == gridDim.x / 2) {
        unsigned long long* d = (unsigned long long*)p.ws;
        d[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        d[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif
}

// ----------------------------------------------------------------------------------------------------------------
// Raw-window variant of the G (wgrad) kernel: dW[m][(q,j)] = sum_{k=(b,i)} P[b,m,i] * Q[b,q,s*i+j-p], tile 128 (m) x 256
// ((q,j) columns = 256/k whole channels).  Per slab of 16 consecutive (b,i) the columns of one channel are k shifted
// views of the SAME piece of Q's row: positions s*i0 - p + [0, 15 s + k).  That window is staged once per channel
// (LDS image [sub][channel][WLP], sub 1 only filled when the slab runs over the end of sample b into b+1) and the
// B fragment of column (q,j), slab element kl is read at  q*WLP + j + s*kl  (+ a wave-uniform shift for kl past the
// sample boundary).  For k = 32 that is 6.4x fewer bytes than the im2col tile.
// ----------------------------------------------------------------------------------------------------------------
template <int KW, int S> struct GRaw {
    static constexpr int WL = 15 * S + KW;                                   // window floats actually read
    static constexpr int WLP = (KW == 32) ? 64 : (KW == 8 ? (S == 1 ? 24 : 40) : 36);   // padded; keeps reads conflict-free
    static constexpr int NQT = RBN / KW;                                      // channels per tile
    static constexpr int SUB = NQT * WLP;                                     // floats per sub-window set
    static constexpr int NE = (SUB + NT - 1) / NT;                            // gather pieces per thread and sub-window
    static constexpr int STG = RTILE_A + 2 * SUB;                             // floats per LDS stage
    static_assert(WL <= WLP, "window does not fit its slot");
};

template <int KW, int S, int BF>
__global__ __launch_bounds__(NT, 2) void conv_g_raw_kernel(const IgemmParams p) {
    using C = GRaw<KW, S>;
    __shared__ __attribute__((aligned(16))) float lds[2 * C::STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv >> 1, wn = wv & 1;
    const int kt = dma_kt(lane, wv);
    const rsrc_t rp = make_rsrc(p.pt, p.pt_bytes), rx = make_rsrc(p.x, p.x_bytes);
    const float slopeA = act_slope(p.act_p), slopeB = act_slope(p.act_x);
    const int pbs4 = (int)p.pt_bs * 4, xbs4 = (int)p.x_bs * 4;
    const int g = xcd_remap(blockIdx.x, gridDim.x);
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x);
    int pos = split_lo(sp, g);
    const int pos_end = split_lo(sp, g + 1);
    int slot = 0;
    while (pos < pos_end) {
        const int tile = pos / p.nslab, sb = pos - tile * p.nslab;
        const int se = min(p.nslab, sb + (pos_end - pos));
        const int m0 = (tile / p.tilesN) * RBM, n0 = (tile % p.tilesN) * RBN;
        const int qbase = n0 / KW;

        int aoff[8];                                   // P[b][m][i]: byte offset of row m (this thread's k column added per slab)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int m = m0 + dma_row(lane, wv, e);
            aoff[e] = m < p.M ? m * p.LP * 4 : FAR;
        }
        int choff[C::NE], vv[C::NE];                   // window element owned by this thread: channel byte offset, v - pad
#pragma unroll
        for (int e = 0; e < C::NE; ++e) {
            const int idx = tid + NT * e, ql = idx / C::WLP, v = idx - ql * C::WLP;
            choff[e] = (idx < C::SUB && qbase + ql < p.Q) ? (qbase + ql) * p.Lx * 4 : FAR;
            vv[e] = v - p.p;
        }
        int bbase[4];                                  // fragment base of this lane's 4 columns
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int c = wn * 128 + jb * 32 + (lane & 31), qc = c / KW;
            bbase[jb] = qc * C::WLP + (c - qc * KW) + S * 8 * (lane >> 5);
        }
        AccR acc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.c[i][j][r] = 0.f;

        // wave-uniform (sample, frame) of the slab being GATHERED (one slab ahead of the one being multiplied)
        int gb, gi;
        { const int k0 = sb * BK; gb = k0 / p.LP; gi = k0 - gb * p.LP; }
        int kc_cur = 16;                               // first slab element that belongs to the next sample (16 = none)

#define GRAW_ISSUE(STAGE_PTR, K0)                                                                         \
    {   float* const As = (STAGE_PTR) + wv * 64; float* const Bw = (STAGE_PTR) + RTILE_A + wv * 64;       \
        const int k0 = (K0);                                                                              \
        { int bb, ii; divmod24(k0 + kt, p.LP, p.inv_LP, bb, ii);                                          \
          const int po = bb < p.B ? bb * pbs4 + ii * 4 : OOB;                                             \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) dma4(rp, As + e * 256, aoff[e] + po); }           \
        const int kc = p.LP - gi;                      /* elements of this slab left in sample gb */      \
        const int sb0 = gb < p.B ? gb * xbs4 : -NEVER, sb1 = (kc < 16 && gb + 1 < p.B) ? (gb + 1) * xbs4 : -NEVER; \
        _Pragma("unroll") for (int e = 0; e < C::NE; ++e) {                                               \
            if (e * NT + wv * 64 < C::SUB) {                                                              \
                const int ps = S * gi + vv[e];                                                            \
                dma4(rx, Bw + e * NT, ((unsigned)ps < (unsigned)p.Lx && sb0 >= 0) ? sb0 + choff[e] + ps * 4 : FAR); \
            }                                                                                             \
        }                                                                                                 \
        if (kc < 16) {                                 /* slab runs into the next sample: second sub-window */ \
            _Pragma("unroll") for (int e = 0; e < C::NE; ++e) {                                           \
                if (e * NT + wv * 64 < C::SUB) {                                                          \
                    const int ps = vv[e];                                                                 \
                    dma4(rx, Bw + C::SUB + e * NT, ((unsigned)ps < (unsigned)p.Lx && sb1 >= 0) ? sb1 + choff[e] + ps * 4 : FAR); \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        kc_next = kc < 16 ? kc : 16;                                                                      \
        gi += BK; if (gi >= p.LP) { gi -= p.LP; ++gb; }                                                   \
    }

        int kc_next;
        GRAW_ISSUE(lds, sb * BK)
        kc_cur = kc_next;
        __syncthreads();
        for (int sl = sb; sl < se; ++sl) {
            const int cur = (sl - sb) & 1;
            GRAW_ISSUE(lds + (cur ^ 1) * C::STG, (sl + 1) * BK)
            __builtin_amdgcn_sched_barrier(0);
            {   // fragments + MFMA for slab sl
                const float* As = lds + cur * C::STG;
                const float* Bw = As + RTILE_A;
                const int r = lane & 31, h = lane >> 5, sw = (r >> 2) & 3;
                const float* ap = As + (wm * 64 + r) * BK;
                f32x4 a[2][2];
                float b[4][8];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i][c] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BK + (((2 * h + c) ^ sw) << 2));
                if (kc_cur >= 16) {
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                        for (int i = 0; i < 8; ++i) b[jb][i] = Bw[bbase[jb] + S * i];
                } else {        // elements kl >= kc_cur live in the second sub-window, which starts at frame 0 of the next sample
                    const int shift = C::SUB - S * kc_cur;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int d = (8 * h + i >= kc_cur) ? shift : 0;
#pragma unroll
                        for (int jb = 0; jb < 4; ++jb) b[jb][i] = Bw[bbase[jb] + S * i + d];
                    }
                }
                if (slopeA != 1.0f) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int v = 0; v < 4; ++v) a[i][c][v] = act_apply(a[i][c][v], slopeA);
                }
                if (slopeB != 1.0f) {
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                        for (int i = 0; i < 8; ++i) b[jb][i] = act_apply(b[jb][i], slopeB);
                }
                if (BF) mfma_low_2x4<BF>(a, b, acc);
                else {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk >> 2][kk & 3], b[j][kk], acc.c[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            kc_cur = kc_next;
            __syncthreads();
        }
#undef GRAW_ISSUE
        if (sb == 0 && se == p.nslab) epilogue_g<S, 2, 4>(p, acc, m0, n0, lane, wm, wn);
        else store_partial(p.ws, g, slot, acc, tid);
        pos += se - sb;
        slot = 1;
    }
}

// ---- fixup: add the partial segments of every split tile in ascending workgroup order, then the epilogue ---------
template <int KIND, int MB, int NB>
__global__ __launch_bounds__(NT) void conv_fixup_kernel(const IgemmParams p, int G) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int tile = blockIdx.x;
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, G);
    const int first = tile * p.nslab, last = first + p.nslab - 1;
    const int g0 = split_owner(sp, first), g1 = split_owner(sp, last);
    if (g0 == g1 && split_lo(sp, g0) <= first && split_lo(sp, g0 + 1) > last) return;   // computed whole by one workgroup
    AccT<MB, NB> acc;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.c[i][j][r] = 0.f;
    for (int g = g0; g <= g1; ++g) {
        const int slot = (split_lo(sp, g) / p.nslab == tile) ? 0 : 1;     // the range's first segment, or its last
        const float* src = p.ws + ((long)(g * 2 + slot) * ACC_REGS) * NT + tid;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.c[i][j][r] += src[((i * NB + j) * 16 + r) * NT];
    }
    const int m0 = (tile / p.tilesN) * (64 * MB), n0 = (tile % p.tilesN) * (64 * NB);
    if (KIND == 0) epilogue_f<0, MB, NB>(p, acc, m0, n0, lane, wm, wn);
    else if (KIND == 1) epilogue_t<0, MB, NB>(p, acc, m0, n0, lane, wm, wn);
    else epilogue_g<0, MB, NB>(p, acc, m0, n0, lane, wm, wn);
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
enum Kind { KIND_F, KIND_T, KIND_G };
int g_bf16 = 0;         // pg_conv_set_precision: 0 fp32 MFMA, 1 bf16 operands, 2 bf16x3 split (all fp32 accumulate)

template <int KW, int S>
hipError_t launch_kind(Kind kind, const IgemmParams& p, int grid, hipStream_t st) {
#define PG_LAUNCH_KIND(PM)                                                                                      \
    switch (kind) {                                                                                             \
        case KIND_F: hipLaunchKernelGGL((conv_f_kernel<KW, S, PM>), dim3(grid), dim3(NT), 0, st, p); break;     \
        case KIND_T: hipLaunchKernelGGL((conv_t_kernel<KW, S, PM>), dim3(grid), dim3(NT), 0, st, p); break;     \
        case KIND_G: hipLaunchKernelGGL((conv_g_kernel<KW, S, PM>), dim3(grid), dim3(NT), 0, st, p); break;     \
    }
    if (g_bf16 == 1) { PG_LAUNCH_KIND(1) } else if (g_bf16 == 2) { PG_LAUNCH_KIND(2) } else { PG_LAUNCH_KIND(0) }
#undef PG_LAUNCH_KIND
    return hipGetLastError();
}

constexpr int WG_PER_CU = 2;                    // <= 256 VGPR+AGPR per lane -> 2 waves per SIMD; 48 KB LDS per workgroup
constexpr int MAX_STREAMK_WG = 2048;            // bound on the persistent grid (sizes the caller's workspace)
constexpr long WS_PER_WG = 2L * ACC_REGS * NT * 4;   // two partial tiles of 256x128 fp32 per workgroup

// schedule knobs (process-wide; set through pg_conv_set_schedule / pg_conv_set_oversubscribe)
int g_force_mode = 0;   // work split: 0 automatic, 1 one tile per workgroup, 2 force stream-K
int g_force_raw = 0;    // 1 = never use the raw-window kernels (exercise the im2col kernels)
int g_oversub = 4;      // stream-K grid = up to g_oversub x resident workgroup slots

int cu_count() { return pg_cu_count(); }

// Grid policy.  Default: a persistent stream-K grid of up to g_oversub (4) x the resident workgroup slots, each
// workgroup owning an equal contiguous range of the (tile, slab) space, plus the fixup launch.  Measured on MI355X
// (tools/contention.py): the oversubscribed split costs nothing on a free chip, removes tile-count quantisation, and --
// what matters for data-parallel training, where RCCL's collective kernels hold part of the chip during backward --
// degrades gracefully when slots are taken (16 of 512 slots held: 1x split 33 -> 58 ms, one-tile-per-workgroup 32 -> 43 ms,
// 4x split 33 -> 37 ms).  Small problems (less than 8 slabs per resident slot, or no workspace) run one tile per
// workgroup.  mode: 0 auto, 1 force one tile per workgroup, 2 force stream-K (tests).
int pick_grid(long tiles, int nslab, const IgemmParams& p, long ws_bytes, int mode) {
    const long total = tiles * (long)nslab;
    const long slots = (long)cu_count() * WG_PER_CU;
    long mult = total / (256 * slots);               // whole multiples of the slot count only (a ragged second wave is
    if (mult > g_oversub) mult = g_oversub;          // worse than none), and >= 256 slabs per workgroup so that partial-
    if (mult < 1) mult = 1;                          // tile traffic stays negligible
    long G = slots * mult;
    if (G > MAX_STREAMK_WG) G = (MAX_STREAMK_WG / slots) * slots;
    if (G > total) G = total;
    const bool can = p.ws && ws_bytes >= G * WS_PER_WG && total < 0x7fffffffL;
    if (mode == 1 || !can) return (int)tiles;
    if (mode == 2) return (int)G;
    return total >= slots * 8 ? (int)G : (int)tiles;
}

// raw-window kernels: supported (k, s) pairs and the window-length bound
bool raw_supported(Kind kind, const IgemmParams& p) {
    if (g_force_raw == 1) return false;
    int kwp, sc, lcol;
    if (kind == KIND_F) {
        if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 4 && p.s == 2))) return false;
        kwp = p.k; sc = p.s; lcol = p.Ly;
    } else if (kind == KIND_T) {
        if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2))) return false;
        kwp = p.k / p.s; sc = 1; lcol = p.U;
    } else {
        // a 16-element slab of (b, i) may run over at most ONE sample boundary in the raw-window wgrad kernel
        return p.LP >= 16 && ((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 4 && p.s == 2));
    }
    const int tj = kwp < 16 ? kwp : 16;
    const int nseg_max = (lcol - 1 + RBN - 1) / lcol + 1;
    return sc * (RBN - 1) + tj + RG * (nseg_max - 1) + (kind == KIND_T ? tj : 0) <= (sc == 1 ? RS1 : RS2);
}

template <int KW, int S, bool TK>
hipError_t launch_raw(const IgemmParams& p, int grid, hipStream_t st) {
    if (g_bf16 == 1) hipLaunchKernelGGL((conv_raw_kernel<KW, S, TK, 1>), dim3(grid), dim3(NT), 0, st, p);
    else if (g_bf16 == 2) hipLaunchKernelGGL((conv_raw_kernel<KW, S, TK, 2>), dim3(grid), dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((conv_raw_kernel<KW, S, TK, 0>), dim3(grid), dim3(NT), 0, st, p);
    return hipGetLastError();
}
template <int KW, int S>
hipError_t launch_g_raw(const IgemmParams& p, int grid, hipStream_t st) {
    if (g_bf16 == 1) hipLaunchKernelGGL((conv_g_raw_kernel<KW, S, 1>), dim3(grid), dim3(NT), 0, st, p);
    else if (g_bf16 == 2) hipLaunchKernelGGL((conv_g_raw_kernel<KW, S, 2>), dim3(grid), dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((conv_g_raw_kernel<KW, S, 0>), dim3(grid), dim3(NT), 0, st, p);
    return hipGetLastError();
}

int launch(Kind kind, IgemmParams& p, long rows, long cols, long Ktot, long ws_bytes, hipStream_t st) {
    const bool raw = raw_supported(kind, p);
    const int bm = raw ? RBM : BM, bn = raw ? RBN : BN;
    p.tilesM = (int)((rows + bm - 1) / bm);
    p.tilesN = (int)((cols + bn - 1) / bn);
    p.nslab = (int)((Ktot + BK - 1) / BK);
    const long tiles = (long)p.tilesM * p.tilesN;
    if (tiles <= 0 || tiles > 0x7fffffffL || p.nslab <= 0) return pg_fail(PG_ERR_SHAPE, "conv: empty or oversize grid");
    const int grid = pick_grid(tiles, p.nslab, p, ws_bytes, g_force_mode);
    hipError_t e;
    if (raw && kind == KIND_F) {
        if (p.k == 32) e = launch_raw<32, 2, false>(p, grid, st);
        else if (p.k == 8 && p.s == 1) e = launch_raw<8, 1, false>(p, grid, st);
        else if (p.k == 8) e = launch_raw<8, 2, false>(p, grid, st);
        else e = launch_raw<4, 2, false>(p, grid, st);
    } else if (raw && kind == KIND_T) {
        if (p.k == 32) e = launch_raw<32, 2, true>(p, grid, st);
        else if (p.s == 1) e = launch_raw<8, 1, true>(p, grid, st);
        else e = launch_raw<8, 2, true>(p, grid, st);
    } else if (raw) {
        if (p.k == 32) e = launch_g_raw<32, 2>(p, grid, st);
        else if (p.k == 8 && p.s == 1) e = launch_g_raw<8, 1>(p, grid, st);
        else if (p.k == 8) e = launch_g_raw<8, 2>(p, grid, st);
        else e = launch_g_raw<4, 2>(p, grid, st);
    }
    else if (p.k == 32 && p.s == 2) e = launch_kind<32, 2>(kind, p, grid, st);
    else if (p.k == 8 && p.s == 1) e = launch_kind<8, 1>(kind, p, grid, st);
    else if (p.k == 8 && p.s == 2) e = launch_kind<8, 2>(kind, p, grid, st);
    else if (p.k == 4 && p.s == 2) e = launch_kind<4, 2>(kind, p, grid, st);
    else if (p.k == 5 && p.s == 2) e = launch_kind<5, 2>(kind, p, grid, st);
    else e = launch_kind<0, 0>(kind, p, grid, st);
    if (e == hipSuccess && grid != tiles) {
        if (raw) {
            if (kind == KIND_F) hipLaunchKernelGGL((conv_fixup_kernel<0, 2, 4>), dim3((unsigned)tiles), dim3(NT), 0, st, p, grid);
            else if (kind == KIND_T) hipLaunchKernelGGL((conv_fixup_kernel<1, 2, 4>), dim3((unsigned)tiles), dim3(NT), 0, st, p, grid);
            else hipLaunchKernelGGL((conv_fixup_kernel<2, 2, 4>), dim3((unsigned)tiles), dim3(NT), 0, st, p, grid);
        } else switch (kind) {
            case KIND_F: hipLaunchKernelGGL((conv_fixup_kernel<0, WMB, 2>), dim3((unsigned)tiles), dim3(NT), 0, st, p, grid); break;
            case KIND_T: hipLaunchKernelGGL((conv_fixup_kernel<1, WMB, 2>), dim3((unsigned)tiles), dim3(NT), 0, st, p, grid); break;
            case KIND_G: hipLaunchKernelGGL((conv_fixup_kernel<2, WMB, 2>), dim3((unsigned)tiles), dim3(NT), 0, st, p, grid); break;
        }
        e = hipGetLastError();
    }
    if (e != hipSuccess) return pg_fail((int)e, hipGetErrorString(e));
    return PG_OK;
}

// bytes spanned by a (B, C, L) view with batch stride bs; 0 if it does not fit 31-bit buffer offsets
unsigned extent_bytes(long B, long bs, long C, long L) {
    const long e = ((B - 1) * bs + C * L) * 4;
    return (e > 0 && e < 0x7ffffff0L) ? (unsigned)e : 0u;
}

int check_geom(const pg_conv_args* a, bool transposed) {
    if (!a) return pg_fail(PG_ERR_NULL, "conv: null args");
    if (a->B <= 0 || a->Cin <= 0 || a->Cout <= 0 || a->Lin <= 0 || a->Lout <= 0 || a->k <= 0 || a->stride <= 0 || a->pad < 0)
        return pg_fail(PG_ERR_SHAPE, "conv: non-positive dimension");
    const long lo = transposed ? (long)(a->Lin - 1) * a->stride - 2L * a->pad + a->k
                               : ((long)a->Lin + 2L * a->pad - a->k) / a->stride + 1;
    if (lo != a->Lout) return pg_fail(PG_ERR_SHAPE, "conv: Lout inconsistent with Lin/k/stride/pad");
    if ((long)a->Cin * a->Cout * a->k * 4 >= 0x7ffffff0L) return pg_fail(PG_ERR_SHAPE, "conv: weight tensor exceeds 2 GiB");
    return PG_OK;
}

// fills the descriptor extents of the tensors a kernel gathers from; fails if one exceeds 31-bit byte offsets
int set_extents(IgemmParams& p, long xC, long xL, long ptC, long ptL) {
    p.x_bytes = extent_bytes(p.B, p.x_bs, xC, xL);
    if (!p.x_bytes) return pg_fail(PG_ERR_SHAPE, "conv: activation tensor exceeds 2 GiB (31-bit buffer offsets)");
    if (p.w) p.w_bytes = (unsigned)((long)p.M * p.Q * p.k * 4);
    if (p.pt) {
        p.pt_bytes = extent_bytes(p.B, p.pt_bs, ptC, ptL);
        if (!p.pt_bytes) return pg_fail(PG_ERR_SHAPE, "conv: activation tensor exceeds 2 GiB (31-bit buffer offsets)");
        if ((long)p.B * p.LP >= (1L << 24) - 64) return pg_fail(PG_ERR_UNSUPPORTED, "wgrad: B*L must stay below 2^24");
        p.inv_LP = 1.0f / (float)p.LP;
    }
    p.a_vec = p.w && (((long)p.Q * p.k) & 3) == 0 && ((uintptr_t)p.w & 15) == 0;
    return PG_OK;
}

}  // namespace

// nn.Conv1d forward: F kernel with M = Cout, Q = Cin.
extern "C" int pg_conv1d_fwd(const pg_conv_args* a, void* stream) {
    if (int e = check_geom(a, false)) return e;
    if (!a->x || !a->w || !a->y) return pg_fail(PG_ERR_NULL, "conv1d_fwd: x, w, y required");
    IgemmParams p = {};
    p.y_slope = act_slope(a->y_act); p.y2 = a->y2; p.y2_bs = a->y2_bs; p.y2_slope = act_slope(a->y2_act);
    p.x = a->x; p.x_bs = a->x_bs; p.w = a->w; p.y = a->y; p.y_bs = a->y_bs;
    p.B = a->B; p.Q = a->Cin; p.M = a->Cout; p.Lx = a->Lin; p.Ly = a->Lout; p.k = a->k; p.s = a->stride; p.p = a->pad;
    p.act_x = a->x_act;
    if (int e = set_extents(p, p.Q, p.Lx, 0, 0)) return e;
    p.ws = (float*)a->workspace;
    return launch(KIND_F, p, p.M, (long)p.B * p.Ly, (long)p.Q * p.k, a->workspace_bytes, (hipStream_t)stream);
}

// nn.ConvTranspose1d dgrad: dx[b,c,i] = sum_{o,j} w[c][o][j] dy[b,o,s*i+j-p]  -> F kernel with M = Cin, Q = Cout.
extern "C" int pg_convt1d_dgrad(const pg_conv_args* a, void* stream) {
    if (int e = check_geom(a, true)) return e;
    if (!a->dy || !a->w || !a->dx) return pg_fail(PG_ERR_NULL, "convt1d_dgrad: dy, w, dx required");
    IgemmParams p = {};
    p.y_slope = 1.0f; p.y2_slope = 1.0f;
    p.x = a->dy; p.x_bs = a->dy_bs; p.w = a->w; p.y = a->dx; p.y_bs = a->dx_bs;
    p.add = a->dx_add; p.add_bs = a->dx_add_bs; p.ref = a->dx_ref; p.ref_bs = a->dx_ref_bs;
    p.mask_mode = a->dx_ref ? a->dx_mask : 0;
    p.B = a->B; p.Q = a->Cout; p.M = a->Cin; p.Lx = a->Lout; p.Ly = a->Lin; p.k = a->k; p.s = a->stride; p.p = a->pad;
    if (int e = set_extents(p, p.Q, p.Lx, 0, 0)) return e;
    p.ws = (float*)a->workspace;
    return launch(KIND_F, p, p.M, (long)p.B * p.Ly, (long)p.Q * p.k, a->workspace_bytes, (hipStream_t)stream);
}

static int launch_t(IgemmParams& p, long ws_bytes, hipStream_t st) {
    // tau = s*u + phi - p >= 0 for some phi  <=>  u >= floor(p/s) at the latest; tau <= Ly-1 => u <= (Ly-1+p)/s
    p.u_off = p.p / p.s;
    const int u_max = (p.Ly - 1 + p.p) / p.s;
    p.U = u_max - p.u_off + 1;
    if (p.U <= 0) return pg_fail(PG_ERR_SHAPE, "convT: empty output");
    if (int e = set_extents(p, p.Q, p.Lx, 0, 0)) return e;
    return launch(KIND_T, p, (long)p.M * p.s, (long)p.B * p.U, (long)p.Q * ((p.k + p.s - 1) / p.s), ws_bytes, st);
}

// nn.ConvTranspose1d forward: T kernel with M = Cout, Q = Cin.
extern "C" int pg_convt1d_fwd(const pg_conv_args* a, void* stream) {
    if (int e = check_geom(a, true)) return e;
    if (!a->x || !a->w || !a->y) return pg_fail(PG_ERR_NULL, "convt1d_fwd: x, w, y required");
    IgemmParams p = {};
    p.y_slope = act_slope(a->y_act); p.y2 = a->y2; p.y2_bs = a->y2_bs; p.y2_slope = act_slope(a->y2_act);
    p.x = a->x; p.x_bs = a->x_bs; p.w = a->w; p.y = a->y; p.y_bs = a->y_bs;
    p.B = a->B; p.Q = a->Cin; p.M = a->Cout; p.Lx = a->Lin; p.Ly = a->Lout; p.k = a->k; p.s = a->stride; p.p = a->pad;
    p.act_x = a->x_act;
    p.ws = (float*)a->workspace;
    return launch_t(p, a->workspace_bytes, (hipStream_t)stream);
}

// nn.Conv1d dgrad: dx[b,c,u] = sum_{o,j,t: s*t+j-p=u} w[o][c][j] dy[b,o,t]  -> T kernel with M = Cin, Q = Cout.
extern "C" int pg_conv1d_dgrad(const pg_conv_args* a, void* stream) {
    if (int e = check_geom(a, false)) return e;
    if (!a->dy || !a->w || !a->dx) return pg_fail(PG_ERR_NULL, "conv1d_dgrad: dy, w, dx required");
    IgemmParams p = {};
    p.y_slope = 1.0f; p.y2_slope = 1.0f;
    p.x = a->dy; p.x_bs = a->dy_bs; p.w = a->w; p.y = a->dx; p.y_bs = a->dx_bs;
    p.add = a->dx_add; p.add_bs = a->dx_add_bs; p.ref = a->dx_ref; p.ref_bs = a->dx_ref_bs;
    p.mask_mode = a->dx_ref ? a->dx_mask : 0;
    p.B = a->B; p.Q = a->Cout; p.M = a->Cin; p.Lx = a->Lout; p.Ly = a->Lin; p.k = a->k; p.s = a->stride; p.p = a->pad;
    p.ws = (float*)a->workspace;
    return launch_t(p, a->workspace_bytes, (hipStream_t)stream);
}

// nn.Conv1d wgrad: dw[o][c][j] = sum_{b,t} dy[b,o,t] act(x)[b,c,s*t+j-p]  -> G with P = dy (M = Cout), Q = x.
extern "C" int pg_conv1d_wgrad(const pg_conv_args* a, void* stream) {
    if (int e = check_geom(a, false)) return e;
    if (!a->dy || !a->x || !a->dw) return pg_fail(PG_ERR_NULL, "conv1d_wgrad: dy, x, dw required");
    IgemmParams p = {};
    p.pt = a->dy; p.pt_bs = a->dy_bs; p.LP = a->Lout; p.act_p = PG_ACT_NONE;
    p.x = a->x; p.x_bs = a->x_bs; p.act_x = a->x_act; p.y = a->dw;
    p.B = a->B; p.Q = a->Cin; p.M = a->Cout; p.Lx = a->Lin; p.k = a->k; p.s = a->stride; p.p = a->pad;
    if (int e = set_extents(p, p.Q, p.Lx, p.M, p.LP)) return e;
    p.ws = (float*)a->workspace;
    return launch(KIND_G, p, p.M, (long)p.Q * p.k, (long)p.B * p.LP, a->workspace_bytes, (hipStream_t)stream);
}

// nn.ConvTranspose1d wgrad: dw[c][o][j] = sum_{b,i} act(x)[b,c,i] dy[b,o,s*i+j-p]  -> G with P = x (M = Cin), Q = dy.
extern "C" int pg_convt1d_wgrad(const pg_conv_args* a, void* stream) {
    if (int e = check_geom(a, true)) return e;
    if (!a->dy || !a->x || !a->dw) return pg_fail(PG_ERR_NULL, "convt1d_wgrad: dy, x, dw required");
    IgemmParams p = {};
    p.pt = a->x; p.pt_bs = a->x_bs; p.LP = a->Lin; p.act_p = a->x_act;
    p.x = a->dy; p.x_bs = a->dy_bs; p.act_x = PG_ACT_NONE; p.y = a->dw;
    p.B = a->B; p.Q = a->Cout; p.M = a->Cin; p.Lx = a->Lout; p.k = a->k; p.s = a->stride; p.p = a->pad;
    if (int e = set_extents(p, p.Q, p.Lx, p.M, p.LP)) return e;
    p.ws = (float*)a->workspace;
    return launch(KIND_G, p, p.M, (long)p.Q * p.k, (long)p.B * p.LP, a->workspace_bytes, (hipStream_t)stream);
}

// Workspace a caller should hand to the conv entry points (pg_conv_args.workspace) so that badly quantised tile counts
// can be balanced over all CUs (stream-K).  Without it every call falls back to one-tile-per-workgroup scheduling.
extern "C" int64_t pg_workspace_bytes_conv(void) { return (int64_t)MAX_STREAMK_WG * WS_PER_WG; }

// Test hook: 0 = automatic schedule, 1 = force one tile per workgroup, 2 = force stream-K (needs a workspace).
extern "C" int pg_conv_set_schedule(int mode) {
    // bits 0-1: 0 automatic split, 1 one tile per workgroup, 2 force stream-K;  bit 2: disable the raw-window kernels
    if (mode < 0 || mode > 7 || (mode & 3) == 3) return pg_fail(PG_ERR_SHAPE, "conv_set_schedule: bad mode");
    g_force_mode = mode & 3;
    g_force_raw = (mode >> 2) & 1;
    return PG_OK;
}

// Stream-K launches one workgroup per resident slot and gives each the same amount of work.  When other kernels hold part
// of the chip (RCCL's collective kernels during data-parallel backward), the workgroups that do not fit run as a second
// wave and the launch takes up to twice as long.  factor > 1 splits the work over factor x more, proportionally shorter
// workgroups, which bounds that tail at 1/factor of a workgroup's duration (at the price of more partial tiles).
extern "C" int pg_conv_set_precision(int32_t mode) {
    if (mode < 0 || mode > 2) return pg_fail(PG_ERR_UNSUPPORTED, "conv_set_precision: 0 (fp32), 1 (bf16 operands) or 2 (bf16x3 split)");
    g_bf16 = mode;
    return PG_OK;
}

extern "C" int pg_conv_set_oversubscribe(int factor) {
    if (factor < 1 || factor > 8) return pg_fail(PG_ERR_SHAPE, "conv_set_oversubscribe: factor must be 1..8");
    g_oversub = factor;
    return PG_OK;
}
