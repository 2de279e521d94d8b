// conv_common.h -- shared by the translation units of the convolution kernels (conv_igemm.hip = host side + fixup,
// conv_im2col.hip, conv_raw.hip, conv_raw_wgrad.hip): problem descriptor, LDS-DMA helpers, MFMA operand modes, stream-K
// split, epilogues.  Everything lives in an anonymous namespace (each unit gets its own copy); the only symbols with
// external linkage are the pgconv::launch_* functions declared at the end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "phasegen.h"
#include "pg_common.h"

namespace pgconv {

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
    int g_ps;                        // G: 1 = per-sample slabs (conv_g_ps_kernel): K = B * ceil(LP / 16) slabs of 16 frames of one sample
    int tn_stride;                   // columns between the origins of consecutive column tiles (= tile width; k = 5 wgrad: 255 of 256)
    float* y2; long y2_bs; float y_slope, y2_slope;   // F,T fwd: activation on store, optional second output
    float* ws;                       // stream-K partial-tile workspace: [grid][2][64][256] floats (or NULL)
    int nslab;                       // K slabs per tile
    int whole;                       // workgroups that own one whole tile each (hybrid split; 0 = even split)
    int sr;                          // tile order (conv_raw3): 0 / 1 = row-major; R > 1: column-major inside super-rows of R tile rows, so
    int n_lo;                        // first column of this launch (0 but for the tail launch of a column split, conv_igemm.hip launch())
                                     // that the 32 workgroups an XCD runs at a time cover R x 32/R tiles and share activation panels too
    // bf16-resident forward kernels (conv_h3.hip): x and w are bf16; rows of x are x_pitch elements apart (even, zero tail);
    // optional bf16 outputs (B, M, yh_pitch) stored already activated.  All NULL / 0 for the fp32-tensor kernels.
    int x_pitch;
    unsigned short* yh; long yh_bs; int yh_pitch; float yh_slope;
    unsigned short* yh2; long yh2_bs; int yh2_pitch; float yh2_slope;
    // G (wgrad) only: optional Adam update fused into the epilogue (pg_conv_args.adam): parameter / exp_avg / exp_avg_sq
    // tensors shaped like dW, updated from the gradient value the epilogue stores.  ad_p == NULL: plain wgrad.
    float* ad_p; float* ad_m; float* ad_v; PgAdamScalars ad;
};

}  // namespace pgconv
using pgconv::IgemmParams;

namespace {


constexpr int WMB = 4;                    // 32-row MFMA blocks per wave along M: wave tile (32*WMB) x 64
constexpr int BM = 64 * WMB, BN = 128, BK = 16, NT = 256;   // workgroup tile 256 x 128, waves 2 (M) x 2 (N)
constexpr int TILE_A = BM * BK, TILE_B = BN * BK;   // floats per operand tile (16 KB + 8 KB)
constexpr int STAGE = TILE_A + TILE_B;    // one LDS stage; two stages = 48 KB
constexpr int AE = BM / 16;               // dword gather pieces per thread and slab for the A tile (B tile: 8)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
// per-lane offset fixed for the tile + a wave-uniform (SGPR) offset that advances with the slab: no VALU per gather
__device__ __forceinline__ void dma4s(rsrc_t r, float* lds_wave_uniform, int lane_off, int uniform_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_wave_uniform, 4, lane_off, uniform_off, 0, 0);
}
__device__ __forceinline__ void dma16s(rsrc_t r, float* lds_wave_uniform, int lane_off, int uniform_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds_wave_uniform, 16, lane_off, uniform_off, 0, 0);
}

// bf16 operand mode (pg_conv_args.precision = PG_PREC_BF16): the same fragments -- lane (row, h) already holds k = 8h .. 8h+7 of the
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
// Split mode (PG_PREC_BF16X3, "bf16x3"): every fp32 operand is written as hi + lo with hi = bf16(x) and
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
// Hybrid (round 2): the first `w` workgroups own one WHOLE tile each (no partial tiles, no fixup work for them) and only the
// remaining tiles -- a tile count slightly above a multiple of the resident slots, e.g. 1056 = 2 x 512 + 32 -- are split evenly
// over the other G - w workgroups.  w = 0 is the plain even split.  Workgroup ids here are LOGICAL ids (logical_wg below).
struct Split { int total, q, r, w, wn, nslab; };
__device__ __host__ __forceinline__ Split make_split(int tiles, int nslab, int G, int w) {
    Split sp; sp.total = tiles * nslab; sp.w = w; sp.wn = w * nslab; sp.nslab = nslab;
    const int rest = sp.total - sp.wn, Gr = G - w;
    sp.q = Gr > 0 ? rest / Gr : 0; sp.r = rest - sp.q * Gr;
    return sp;
}
__device__ __host__ __forceinline__ int split_lo(const Split& sp, int g) {
    if (g < sp.w) return g * sp.nslab;
    const int gr = g - sp.w;
    return sp.wn + (gr < sp.r ? gr * (sp.q + 1) : sp.r * (sp.q + 1) + (gr - sp.r) * sp.q);
}
__device__ __host__ __forceinline__ int split_owner(const Split& sp, int x) {
    if (x < sp.wn) return x / sp.nslab;
    const int xr = x - sp.wn, big = sp.r * (sp.q + 1);
    return sp.w + (xr < big ? xr / (sp.q + 1) : sp.r + (xr - big) / sp.q);
}
// linear tile index -> (tile row, tile column)
__device__ __forceinline__ void tile_decode(const IgemmParams& p, int tile, int& tm, int& tn) {
    if (p.sr <= 1) { tm = tile / p.tilesN; tn = tile - tm * p.tilesN; return; }
    const int per = p.sr * p.tilesN, s = tile / per, r = tile - s * per;
    const int h = min(p.sr, p.tilesM - s * p.sr);             // (the last super-row may be shorter)
    tn = r / h; tm = s * p.sr + (r - tn * h);
}
// hardware workgroup id -> logical id: whole-tile workgroups come first in dispatch order (they are the long ones), each class is
// remapped so that every XCD gets a contiguous run of tiles / of remainder ranges
__device__ __forceinline__ int logical_wg(int bid, int G, int w) {
    return bid < w ? xcd_remap(bid, w) : w + xcd_remap(bid - w, G - w);
}

constexpr int ACC_REGS = 128;             // accumulator registers per thread in both tile configurations (8 blocks x 16)
template <int MB, int NB>
__device__ __forceinline__ void store_partial(float* ws, int g, int slot, const AccT<MB, NB>& acc, int tid, int nbv = NB) {
    static_assert(MB * NB * 16 == ACC_REGS, "partial-tile slots are sized for 8 blocks per wave");
    float* dst = ws + ((long)(g * 2 + slot) * ACC_REGS) * NT + tid;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < nbv) {           // (column blocks past the problem's last column are neither written nor, by the fixup, read)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((i * NB + j) * 16 + r) * NT] = acc.c[i][j][r];
            }
}

// fp32 -> bf16, round to nearest even (a plain cast: hipcc emits v_cvt_pk_bf16_f32, NaN stays NaN)
__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
// bf16 copies of a forward result (B, M, pitch), stored activated: what the bf16-resident kernels of the next layer read
__device__ __forceinline__ void store_h(const IgemmParams& p, int b, int m, int t, float v) {
    if (p.yh) p.yh[(long)b * p.yh_bs + (long)m * p.yh_pitch + t] = bf16_bits(act_apply(v, p.yh_slope));
    if (p.yh2) p.yh2[(long)b * p.yh2_bs + (long)m * p.yh2_pitch + t] = bf16_bits(act_apply(v, p.yh2_slope));
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
                    if (p.y) yb[off] = act_apply(v, p.y_slope);
                    if (p.y2) p.y2[(long)b * p.y2_bs + off] = act_apply(v, p.y2_slope);
                    store_h(p, b, m, t, v);
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
                    if (p.y) yb[off] = act_apply(v, p.y_slope);
                    if (p.y2) p.y2[(long)b * p.y2_bs + off] = act_apply(v, p.y2_slope);
                    store_h(p, b, o, tau, v);
                }
            }
    }
}

// T epilogue for the stride-2 kernels with the phase-major weight image (conv_raw_impl.h): row block i holds phase phi0 + i
// of the 32 output channels obase + (row inside the block); columns nbase + j * 32 + lane.
template <int MB, int NB>
__device__ __forceinline__ void epilogue_t_pm(const IgemmParams& p, const AccT<MB, NB>& acc, int obase, int nbase, int lane, int phi0) {
    const int Ntot = p.B * p.U;
    const Epi ep(p, (unsigned)((long)p.M * p.Ly * 4));
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = nbase + j * 32 + (lane & 31);
        if (n >= Ntot) continue;
        const int b = n / p.U, u = n - b * p.U + p.u_off;
        float* yb = p.y + (long)b * p.y_bs;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int tau = 2 * u + phi0 + i - p.p;
            if (tau < 0 || tau >= p.Ly) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = obase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (o < p.M) {
                    const int off = o * p.Ly + tau;
                    float v = acc.c[i][j][r];
                    if (ep.fused) v = ep(v, b * (int)p.add_bs + off, b * (int)p.ref_bs + off);
                    yb[off] = act_apply(v, p.y_slope);
                    if (p.y2) p.y2[(long)b * p.y2_bs + off] = act_apply(v, p.y2_slope);
                }
            }
        }
    }
}

// ncap: first column that no longer belongs to this tile (the k = 5 wgrad tile holds 255 valid columns: its 256th is the next
// tile's first).  The default never binds.
template <int S, int MB, int NB>
__device__ __forceinline__ void epilogue_g(const IgemmParams& p, const AccT<MB, NB>& acc, int m0, int n0, int lane, int wm, int wn,
                                           int ncap = 0x7fffffff) {
    const int Ntot = p.Q * p.k, Nlim = min(Ntot, ncap);   // row pitch of dW; first column this tile does not own
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * (NB * 32) + j * 32 + (lane & 31);
        if (n >= Nlim) continue;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (MB * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) p.y[(long)m * Ntot + n] = acc.c[i][j][r];
            }
    }
    if (!p.ad_p) return;
    // fused Adam (28 B per parameter of a separate pass -> 24 B here): eight rows at a time -- 24 loads in flight per lane, one
    // memory latency per half block, inside the GEMM loop's register budget (a scheduling barrier keeps the compiler from
    // hoisting the next chunk's loads); 32-bit element offsets (operands are < 2 GiB)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * (NB * 32) + j * 32 + (lane & 31);
        if (n >= Nlim) continue;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int mb = m0 + wm * (MB * 32) + i * 32 + 16 * q + 4 * (lane >> 5);
                float pp[8], mm[8], vv[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    const unsigned o = (unsigned)(m < p.M ? m : 0) * (unsigned)Ntot + (unsigned)n;
                    pp[r] = p.ad_p[o]; mm[r] = p.ad_m[o]; vv[r] = p.ad_v[o];
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m >= p.M) continue;
                    const unsigned o = (unsigned)m * (unsigned)Ntot + (unsigned)n;
                    pg_adam_one(pp[r], acc.c[i][j][8 * q + r], mm[r], vv[r], p.ad.omb1, p.ad.b2, p.ad.omb2, p.ad.step_size, p.ad.bc2_sqrt,
                                p.ad.eps, p.ad.gs);
                    p.ad_p[o] = pp[r]; p.ad_m[o] = mm[r]; p.ad_v[o] = vv[r];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
}

constexpr int NEVER = 0x40000000;   // a "first valid tap/position" no index ever reaches: marks rows outside the tile
// Out-of-range byte offsets that replace per-element predicates.  Descriptors span < 0x7ffffff0 bytes, so with
//   FAR (rows outside the tile) = 0x80000000 and OOB (slabs past K) = 0x7ffffff0
// every sum {valid row + OOB, FAR + valid k offset, FAR + OOB} stays >= 0x7ffffff0 as an unsigned 32-bit value and
// never wraps back into range (FAR + FAR would: the two invalid cases therefore use different constants).
constexpr int FAR = (int)0x80000000u;


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

// n / d for 0 <= n < 2^24 via the float reciprocal, exact after one correction step (branch-free selects).
__device__ __forceinline__ void divmod24(int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { r += d; --q; }
    if (r >= d) { r -= d; ++q; }
}

// raw-window tile geometry (kernels in conv_raw.hip / conv_raw_wgrad.hip; the host needs it to size grids and windows)
constexpr int RBM = 128, RBN = 256;       // raw-window workgroup tile
constexpr int RS2 = 768, RS1 = 384;       // floats reserved per channel window for column stride 2 / 1
// gap (floats) between the windows of consecutive samples inside a tile: a column's taps must never reach into the next
// sample's window, which needs >= TJ - 1 floats (TJ = taps per channel and slab); short-tap kernels take the small gap so that
// tiles spanning many short samples (U = 31 columns per sample at the U-Net's bottleneck) still fit their window slots
constexpr int raw_gap(int tj) { return tj >= 8 ? 16 : tj; }
constexpr int RTILE_A = RBM * BK;         // weight tile, same swizzled image as above (8 KB)

enum Kind { KIND_F, KIND_T, KIND_G };

}  // namespace

// Kernel launchers, one per translation unit.  `prec` = pg_conv_args.precision (0 fp32, 1 bf16, 2 bf16x3).
namespace pgconv {
hipError_t launch_im2col(int kind, const IgemmParams& p, int grid, hipStream_t st, int prec);   // conv_im2col.hip
hipError_t launch_raw_ft(int kind, const IgemmParams& p, int grid, hipStream_t st, int prec);   // conv_raw.hip (F / T, tile 128 x 256)
hipError_t launch_raw_ft_tall(int kind, const IgemmParams& p, int grid, hipStream_t st, int prec);   // conv_raw_tall.hip (256 x 128)
hipError_t launch_raw_g(const IgemmParams& p, int grid, hipStream_t st, int prec);              // conv_raw_wgrad.hip
// conv_h3.hip: 4 waves at ONE per SIMD, wave tile 256 x 64, tile 256 x 256
hipError_t launch_h3(int kind, const IgemmParams& p, int grid, hipStream_t st);
hipError_t launch_h3_fixup(int kind, const IgemmParams& p, int grid, unsigned split_tiles, bool wide, hipStream_t st);
// conv_raw3.hip: fp32 raw-window F / T kernels on 4 waves at ONE per SIMD, tile 256 x 256
bool raw3_covers(int kind, const IgemmParams& p);               // (k, s) pair and whole-slab K; the window-length bound is raw_supported's
hipError_t launch_raw3(int kind, const IgemmParams& p, int grid, hipStream_t st);
hipError_t launch_raw3_fixup(int kind, const IgemmParams& p, int grid, unsigned blocks, hipStream_t st);
bool h_supported(int kind, const IgemmParams& p);               // geometry covered by the bf16-resident kernels (conv_h3.hip)
}  // namespace pgconv
