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
// Tiling (all three): 256 threads = 4 waves in a 2x2 grid, workgroup tile 128x128, wave tile 64x64 = 2x2 MFMA
// 32x32 accumulators (64 VGPRs), BK = 16.  Operand tiles live in LDS K-contiguous ([row][BK] with an 80-B row
// stride => conflict-free ds_read_b128); lane half h = lane>>5 owns k in [8h, 8h+8) of each BK slab, so one lane
// fetches its 8 A (or B) values of a 32-row block with two ds_read_b128 instead of eight ds_read_b32 (the MFMA
// only needs A and B to agree on which k each lane half carries).  Global -> register -> LDS double buffering
// with one barrier per BK slab; activations / zero padding / im2col indexing are applied while staging, so the
// (Leaky)ReLU in front of every conv (model.py:91,96,103) and torch.cat (model.py:113) are never materialised.
// Each output element is produced by exactly one workgroup in a fixed k order: results are deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "phasegen.h"
#include "pg_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, NT = 256;
constexpr int LDT = BK + 4;               // floats per LDS row (80 B)
constexpr int TILE = BM * LDT;            // floats per operand tile (10 KB)

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
};

// Activations are applied branch-free as max(v,0) + slope*min(v,0): slope 1 = identity, 0.2 = LeakyReLU(0.2)
// (model.py:80), 0 = ReLU (model.py:82).  A runtime switch here would make hipcc branch around every gathered
// element and wait vmcnt(0) for each load in turn.
__host__ __device__ __forceinline__ float act_slope(int act) {
    return act == PG_ACT_LEAKY02 ? 0.2f : (act == PG_ACT_RELU ? 0.0f : 1.0f);
}
__device__ __forceinline__ float act_apply(float v, float slope) { return fmaxf(v, 0.f) + slope * fminf(v, 0.f); }

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

struct Acc { f32x16 c[2][2]; };

// One BK=16 slab: 8 x ds_read_b128, then 8 k-pairs x 4 MFMA.
__device__ __forceinline__ void mma_slab(const float* __restrict__ As, const float* __restrict__ Bs,
                                         int lane, int wm, int wn, Acc& acc) {
    const int r = lane & 31, h = lane >> 5;
    const float* ap = As + (wm * 64 + r) * LDT + h * 8;
    const float* bp = Bs + (wn * 64 + r) * LDT + h * 8;
    f32x4 a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i][0] = *reinterpret_cast<const f32x4*>(ap + i * 32 * LDT);
        a[i][1] = *reinterpret_cast<const f32x4*>(ap + i * 32 * LDT + 4);
        b[i][0] = *reinterpret_cast<const f32x4*>(bp + i * 32 * LDT);
        b[i][1] = *reinterpret_cast<const f32x4*>(bp + i * 32 * LDT + 4);
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const float a0 = a[0][kk >> 2][kk & 3], a1 = a[1][kk >> 2][kk & 3];
        const float b0 = b[0][kk >> 2][kk & 3], b1 = b[1][kk >> 2][kk & 3];
        acc.c[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc.c[0][0], 0, 0, 0);
        acc.c[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc.c[0][1], 0, 0, 0);
        acc.c[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc.c[1][0], 0, 0, 0);
        acc.c[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc.c[1][1], 0, 0, 0);
    }
}

// Register-staged tiles: each thread carries 2 x float4 of A and 2 x float4 of B per slab.
struct Stage { f32x4 a[2], b[2]; };

__device__ __forceinline__ void stage_store(float* As, float* Bs, const Stage& st, int tid, float slopeA, float slopeB) {
    // A: thread -> (row = tid>>2 (+64), kgroup = tid&3);  B: thread -> (row = tid&127, kgroup = tid>>7 (+2))
    // Activation happens HERE (after the MFMA block in program order), never at the load, so the gathers of the
    // next slab stay in flight underneath the matrix work.
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        f32x4 a = st.a[e], b = st.b[e];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = act_apply(a[i], slopeA); b[i] = act_apply(b[i], slopeB); }
        *reinterpret_cast<f32x4*>(As + ((tid >> 2) + 64 * e) * LDT + (tid & 3) * 4) = a;
        *reinterpret_cast<f32x4*>(Bs + (tid & 127) * LDT + ((tid >> 7) + 2 * e) * 4) = b;
    }
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

#define PG_MAINLOOP(LOAD_A, LOAD_B)                                                     \
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE];                        \
    Acc acc;                                                                            \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc.c[i][j][r] = 0.f;            \
    Stage st;                                                                           \
    const int nslab = (Ktot + BK - 1) / BK;                                             \
    { const int k0 = 0; LOAD_A; LOAD_B; }                                               \
    stage_store(lds, lds + TILE, st, tid, slopeA, slopeB);                                              \
    __syncthreads();                                                                    \
    for (int sl = 0; sl < nslab; ++sl) {                                                \
        const int cur = sl & 1;                                                         \
        const int k0 = (sl + 1) * BK;      /* past-the-end slab loads only zeros */     \
        LOAD_A; LOAD_B;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                              \
        mma_slab(lds + cur * 2 * TILE, lds + cur * 2 * TILE + TILE, lane, wm, wn, acc); \
        __builtin_amdgcn_sched_barrier(0);                                              \
        stage_store(lds + (cur ^ 1) * 2 * TILE, lds + (cur ^ 1) * 2 * TILE + TILE, st, tid, slopeA, slopeB); \
        __syncthreads();                                                                \
    }

// ------------------------------------------------------------------------------------------------------------
// F kernel
// ------------------------------------------------------------------------------------------------------------
template <int KW, int S>
__global__ __launch_bounds__(NT) void conv_f_kernel(const IgemmParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tilesN) * BM, n0 = (tile % p.tilesN) * BN;
    const int kw = KW ? KW : p.k, s = S ? S : p.s;
    const int Ktot = p.Q * kw, Ntot = p.B * p.Ly;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);

    // A operand (weights, K-contiguous rows): thread -> rows (tid>>2) + 64e, k-group tid&3
    int arow[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int m = m0 + (tid >> 2) + 64 * e;
        arow[e] = m < p.M ? m * Ktot + (tid & 3) * 4 : -1;
    }
    // B operand (im2col): one output position n per thread
    const int nB = n0 + (tid & 127);
    const bool nvalid = nB < Ntot;
    const int bB = nvalid ? nB / p.Ly : 0, tB = nvalid ? nB - bB * p.Ly : 0;
    const int xoff = bB * (int)p.x_bs + s * tB - p.p;          // element offset of tap j = 0, channel 0
    const int jlo = p.p - s * tB;                              // valid taps: 0 <= j - jlo < Lx
    const unsigned jspan = nvalid ? (unsigned)p.Lx : 0u;
    const float slopeA = 1.0f, slopeB = act_slope(p.act_x);

#define F_LOAD_A                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                \
        const int kk = k0 + (tid & 3) * 4;                                                         \
        if (p.a_vec) {                                                                             \
            const bool ok = arow[e] >= 0 && kk < Ktot;                                             \
            st.a[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? (arow[e] + k0) * 4 : OOB, 0, 0)); \
        } else {                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                          \
                st.a[e][i] = bload(rw, arow[e] + k0 + i, arow[e] >= 0 && kk + i < Ktot);           \
        }                                                                                          \
    }
#define F_LOAD_B                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                \
        const int kk0 = k0 + ((tid >> 7) + 2 * e) * 4;                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
            const int kk = kk0 + i, q = kk / kw, j = kk - q * kw;                                  \
            const bool ok = kk < Ktot && (unsigned)(j - jlo) < jspan;                              \
            st.b[e][i] = bload(rx, xoff + q * p.Lx + j, ok);                       \
        }                                                                                          \
    }
    PG_MAINLOOP(F_LOAD_A, F_LOAD_B)
#undef F_LOAD_A
#undef F_LOAD_B

    // epilogue: acc reg r of block (i,j): row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
    const Epi ep(p, (unsigned)((long)p.M * p.Ly * 4));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= Ntot) continue;
        const int b = n / p.Ly, t = n - b * p.Ly;
        float* yb = p.y + (long)b * p.y_bs;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    const int off = m * p.Ly + t;
                    float v = acc.c[i][j][r];
                    if (ep.fused) v = ep(v, b * (int)p.add_bs + off, b * (int)p.ref_bs + off);
                    yb[off] = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// T kernel.  GEMM rows m' = o*s + phi, K = (q, jj) with KJ = ceil(k/s) taps per phase, N = (b, u).
// ------------------------------------------------------------------------------------------------------------
template <int KW, int S>
__global__ __launch_bounds__(NT) void conv_t_kernel(const IgemmParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tilesN) * BM, n0 = (tile % p.tilesN) * BN;
    const int kw = KW ? KW : p.k, s = S ? S : p.s;
    const int KJ = (kw + s - 1) / s;
    const int Ktot = p.Q * KJ, Ntot = p.B * p.U, Mrows = p.M * s;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);
    const int wq = p.M * kw;                  // weight stride between input channels q

    // A operand: W[q][o][s*jj + phi]; thread -> rows m' = (tid>>2) + 64e (o = m'/s, phi = m'%s), k-group tid&3
    int arow[2], aphi[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int mr = m0 + (tid >> 2) + 64 * e, o = mr / s;
        aphi[e] = mr - o * s;
        arow[e] = mr < Mrows ? o * kw + aphi[e] : -1;
    }
    // B operand: X[b][q][u - jj]; one (b, u) per thread
    const int nB = n0 + (tid & 127);
    const bool nvalid = nB < Ntot;
    const int bB = nvalid ? nB / p.U : 0, uB = (nvalid ? nB - bB * p.U : 0) + p.u_off;
    const int xoff = bB * (int)p.x_bs + uB;
    const unsigned xspan = nvalid ? (unsigned)p.Lx : 0u;
    const float slopeA = 1.0f, slopeB = act_slope(p.act_x);

#define T_LOAD_A                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                \
        const int kk0 = k0 + (tid & 3) * 4;                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
            const int kk = kk0 + i, q = kk / KJ, jj = kk - q * KJ;                                 \
            const bool ok = arow[e] >= 0 && kk < Ktot && s * jj + aphi[e] < kw;                    \
            st.a[e][i] = bload(rw, q * wq + arow[e] + s * jj, ok);                                 \
        }                                                                                          \
    }
#define T_LOAD_B                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                \
        const int kk0 = k0 + ((tid >> 7) + 2 * e) * 4;                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
            const int kk = kk0 + i, q = kk / KJ, jj = kk - q * KJ;                                 \
            const bool ok = kk < Ktot && (unsigned)(uB - jj) < xspan;                              \
            st.b[e][i] = bload(rx, xoff + q * p.Lx - jj, ok);                      \
        }                                                                                          \
    }
    PG_MAINLOOP(T_LOAD_A, T_LOAD_B)
#undef T_LOAD_A
#undef T_LOAD_B

    const Epi ep(p, (unsigned)((long)p.M * p.Ly * 4));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= Ntot) continue;
        const int b = n / p.U, u = n - b * p.U + p.u_off;
        float* yb = p.y + (long)b * p.y_bs;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int o = mr / s, phi = mr - o * s, tau = s * u + phi - p.p;
                if (mr < Mrows && tau >= 0 && tau < p.Ly) {
                    const int off = o * p.Ly + tau;
                    float v = acc.c[i][j][r];
                    if (ep.fused) v = ep(v, b * (int)p.add_bs + off, b * (int)p.ref_bs + off);
                    yb[off] = v;
                }
            }
    }
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

template <int KW, int S>
__global__ __launch_bounds__(NT) void conv_g_kernel(const IgemmParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tilesN) * BM, n0 = (tile % p.tilesN) * BN;
    const int kw = KW ? KW : p.k, s = S ? S : p.s;
    const int Ktot = p.B * p.LP, Ntot = p.Q * kw;
    const rsrc_t rp = make_rsrc(p.pt, p.pt_bytes), rx = make_rsrc(p.x, p.x_bytes);

    // A operand: P[b][m][i]; thread -> rows (tid>>2) + 64e, k-group tid&3
    int arow[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int m = m0 + (tid >> 2) + 64 * e;
        arow[e] = m < p.M ? m * p.LP : -1;
    }
    // B operand: Q[b][q][s*i + j - p]; one (q, j) per thread
    const int nB = n0 + (tid & 127);
    const bool nvalid = nB < Ntot;
    const int qB = nvalid ? nB / kw : 0, jB = nvalid ? nB - qB * kw : 0;
    const int xoff = qB * p.Lx + jB - p.p;
    const unsigned xspan = nvalid ? (unsigned)p.Lx : 0u;
    const float slopeA = act_slope(p.act_p), slopeB = act_slope(p.act_x);
    const int pbs = (int)p.pt_bs, xbs = (int)p.x_bs;

#define G_LOAD_A                                                                                   \
    {                                                                                              \
        int bb, ii;                                                                                \
        divmod24(k0 + (tid & 3) * 4, p.LP, p.inv_LP, bb, ii);                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
            const bool okb = bb < p.B;                                                             \
            _Pragma("unroll") for (int e = 0; e < 2; ++e)                                          \
                st.a[e][i] = bload(rp, bb * pbs + arow[e] + ii, okb && arow[e] >= 0); \
            ++ii; if (ii == p.LP) { ii = 0; ++bb; }                                                \
        }                                                                                          \
    }
#define G_LOAD_B                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                \
        int bb, ii;                                                                                \
        divmod24(k0 + ((tid >> 7) + 2 * e) * 4, p.LP, p.inv_LP, bb, ii);                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
            const bool ok = bb < p.B && (unsigned)(s * ii + jB - p.p) < xspan;                     \
            st.b[e][i] = bload(rx, bb * xbs + xoff + s * ii, ok);                 \
            ++ii; if (ii == p.LP) { ii = 0; ++bb; }                                                \
        }                                                                                          \
    }
    PG_MAINLOOP(G_LOAD_A, G_LOAD_B)
#undef G_LOAD_A
#undef G_LOAD_B

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= Ntot) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) p.y[(long)m * Ntot + n] = acc.c[i][j][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
enum Kind { KIND_F, KIND_T, KIND_G };

template <int KW, int S>
hipError_t launch_kind(Kind kind, const IgemmParams& p, int grid, hipStream_t st) {
    switch (kind) {
        case KIND_F: hipLaunchKernelGGL((conv_f_kernel<KW, S>), dim3(grid), dim3(NT), 0, st, p); break;
        case KIND_T: hipLaunchKernelGGL((conv_t_kernel<KW, S>), dim3(grid), dim3(NT), 0, st, p); break;
        case KIND_G: hipLaunchKernelGGL((conv_g_kernel<KW, S>), dim3(grid), dim3(NT), 0, st, p); break;
    }
    return hipGetLastError();
}

int launch(Kind kind, IgemmParams& p, long rows, long cols, hipStream_t st) {
    p.tilesM = (int)((rows + BM - 1) / BM);
    p.tilesN = (int)((cols + BN - 1) / BN);
    const long grid = (long)p.tilesM * p.tilesN;
    if (grid <= 0 || grid > 0x7fffffffL) return pg_fail(PG_ERR_SHAPE, "conv: empty or oversize grid");
    hipError_t e;
    if (p.k == 32 && p.s == 2) e = launch_kind<32, 2>(kind, p, (int)grid, st);
    else if (p.k == 8 && p.s == 1) e = launch_kind<8, 1>(kind, p, (int)grid, st);
    else if (p.k == 8 && p.s == 2) e = launch_kind<8, 2>(kind, p, (int)grid, st);
    else if (p.k == 4 && p.s == 2) e = launch_kind<4, 2>(kind, p, (int)grid, st);
    else if (p.k == 5 && p.s == 2) e = launch_kind<5, 2>(kind, p, (int)grid, st);
    else e = launch_kind<0, 0>(kind, p, (int)grid, st);
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
    p.x = a->x; p.x_bs = a->x_bs; p.w = a->w; p.y = a->y; p.y_bs = a->y_bs;
    p.B = a->B; p.Q = a->Cin; p.M = a->Cout; p.Lx = a->Lin; p.Ly = a->Lout; p.k = a->k; p.s = a->stride; p.p = a->pad;
    p.act_x = a->x_act;
    if (int e = set_extents(p, p.Q, p.Lx, 0, 0)) return e;
    return launch(KIND_F, p, p.M, (long)p.B * p.Ly, (hipStream_t)stream);
}

// nn.ConvTranspose1d dgrad: dx[b,c,i] = sum_{o,j} w[c][o][j] dy[b,o,s*i+j-p]  -> F kernel with M = Cin, Q = Cout.
extern "C" int pg_convt1d_dgrad(const pg_conv_args* a, void* stream) {
    if (int e = check_geom(a, true)) return e;
    if (!a->dy || !a->w || !a->dx) return pg_fail(PG_ERR_NULL, "convt1d_dgrad: dy, w, dx required");
    IgemmParams p = {};
    p.x = a->dy; p.x_bs = a->dy_bs; p.w = a->w; p.y = a->dx; p.y_bs = a->dx_bs;
    p.add = a->dx_add; p.add_bs = a->dx_add_bs; p.ref = a->dx_ref; p.ref_bs = a->dx_ref_bs;
    p.mask_mode = a->dx_ref ? a->dx_mask : 0;
    p.B = a->B; p.Q = a->Cout; p.M = a->Cin; p.Lx = a->Lout; p.Ly = a->Lin; p.k = a->k; p.s = a->stride; p.p = a->pad;
    if (int e = set_extents(p, p.Q, p.Lx, 0, 0)) return e;
    return launch(KIND_F, p, p.M, (long)p.B * p.Ly, (hipStream_t)stream);
}

static int launch_t(IgemmParams& p, hipStream_t st) {
    // tau = s*u + phi - p >= 0 for some phi  <=>  u >= floor(p/s) at the latest; tau <= Ly-1 => u <= (Ly-1+p)/s
    p.u_off = p.p / p.s;
    const int u_max = (p.Ly - 1 + p.p) / p.s;
    p.U = u_max - p.u_off + 1;
    if (p.U <= 0) return pg_fail(PG_ERR_SHAPE, "convT: empty output");
    if (int e = set_extents(p, p.Q, p.Lx, 0, 0)) return e;
    return launch(KIND_T, p, (long)p.M * p.s, (long)p.B * p.U, st);
}

// nn.ConvTranspose1d forward: T kernel with M = Cout, Q = Cin.
extern "C" int pg_convt1d_fwd(const pg_conv_args* a, void* stream) {
    if (int e = check_geom(a, true)) return e;
    if (!a->x || !a->w || !a->y) return pg_fail(PG_ERR_NULL, "convt1d_fwd: x, w, y required");
    IgemmParams p = {};
    p.x = a->x; p.x_bs = a->x_bs; p.w = a->w; p.y = a->y; p.y_bs = a->y_bs;
    p.B = a->B; p.Q = a->Cin; p.M = a->Cout; p.Lx = a->Lin; p.Ly = a->Lout; p.k = a->k; p.s = a->stride; p.p = a->pad;
    p.act_x = a->x_act;
    return launch_t(p, (hipStream_t)stream);
}

// nn.Conv1d dgrad: dx[b,c,u] = sum_{o,j,t: s*t+j-p=u} w[o][c][j] dy[b,o,t]  -> T kernel with M = Cin, Q = Cout.
extern "C" int pg_conv1d_dgrad(const pg_conv_args* a, void* stream) {
    if (int e = check_geom(a, false)) return e;
    if (!a->dy || !a->w || !a->dx) return pg_fail(PG_ERR_NULL, "conv1d_dgrad: dy, w, dx required");
    IgemmParams p = {};
    p.x = a->dy; p.x_bs = a->dy_bs; p.w = a->w; p.y = a->dx; p.y_bs = a->dx_bs;
    p.add = a->dx_add; p.add_bs = a->dx_add_bs; p.ref = a->dx_ref; p.ref_bs = a->dx_ref_bs;
    p.mask_mode = a->dx_ref ? a->dx_mask : 0;
    p.B = a->B; p.Q = a->Cout; p.M = a->Cin; p.Lx = a->Lout; p.Ly = a->Lin; p.k = a->k; p.s = a->stride; p.p = a->pad;
    return launch_t(p, (hipStream_t)stream);
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
    return launch(KIND_G, p, p.M, (long)p.Q * p.k, (hipStream_t)stream);
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
    return launch(KIND_G, p, p.M, (long)p.Q * p.k, (hipStream_t)stream);
}
extern "C" int64_t pg_workspace_bytes_conv(void) { return 256; }
extern "C" int pg_conv_set_schedule(int) { return 0; }
