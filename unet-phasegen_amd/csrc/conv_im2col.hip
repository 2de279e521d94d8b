// conv_im2col.hip -- the im2col implicit-GEMM kernels (tile 256 x 128, both operands gathered element by element by
// LDS-DMA): the fallback for k = 5, generic (k, s) and windows that do not fit the raw-window kernels.  See conv_igemm.hip
// for the overview of the three GEMM forms.
#include "conv_common.h"

namespace {

// Body shared by the three GEMM kernels.  SETUP computes this thread's per-row gather constants for tile (m0, n0);
// ISSUE enqueues the 16 LDS-DMA gathers of one slab (8 per operand per thread) into the LDS buffer (As, Bs): there are
// no staging registers and no ds_write.  The only wait is the vmcnt(0) that __syncthreads() carries, and it sits AFTER
// the slab's 32 MFMAs (phase order pinned with sched_barrier: hipcc otherwise hoists the register-only MFMAs above the
// gather issue), so gather latency is covered by matrix work.  buf^1 is refilled while buf is read: its previous
// readers all passed the barrier that ended the last iteration.
#define PG_BODY(SETUP, ISSUE, ...)   /* variadic tail = the epilogue call (its template arguments contain commas) */                                                             \
    const int tid = threadIdx.x, lane = tid & 63;                                                   \
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv >> 1, wn = wv & 1;             \
    const int kw = KW ? KW : p.k, s = S ? S : p.s;                                                  \
    const int kt = dma_kt(lane, wv);                                                                \
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];                                   \
    const int g = logical_wg(blockIdx.x, gridDim.x, p.whole);                                       \
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x, p.whole);                  \
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
// (the runtime-(k, s) instantiation carries 16 more per-row registers -- phi_of -- than the specialised ones and used to spill
// 73 VGPRs at two waves per SIMD: it is built for one, the generic fallback's speed not being the point)
template <int KW, int S, int BF>
__global__ __launch_bounds__(NT, (KW == 0 ? 1 : 2)) void conv_t_kernel(const IgemmParams p) {
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

template <int KW, int S>
hipError_t launch_kind(Kind kind, const IgemmParams& p, int grid, hipStream_t st, int prec) {
#define PG_LAUNCH_KIND(PM)                                                                                      \
    switch (kind) {                                                                                             \
        case KIND_F: hipLaunchKernelGGL((conv_f_kernel<KW, S, PM>), dim3(grid), dim3(NT), 0, st, p); break;     \
        case KIND_T: hipLaunchKernelGGL((conv_t_kernel<KW, S, PM>), dim3(grid), dim3(NT), 0, st, p); break;     \
        case KIND_G: hipLaunchKernelGGL((conv_g_kernel<KW, S, PM>), dim3(grid), dim3(NT), 0, st, p); break;     \
    }
    if (prec == 1) { PG_LAUNCH_KIND(1) } else if (prec == 2) { PG_LAUNCH_KIND(2) } else { PG_LAUNCH_KIND(0) }
#undef PG_LAUNCH_KIND
    return hipGetLastError();
}

}  // namespace

// ONE instantiation per (form, operand mode): runtime (k, s).  Until 0.3 the U-Net's five (k, s) pairs had compile-time copies as well
// (54 kernels, 3.5 MB of code object); since the raw-window kernels took every layer of the network, the im2col kernels only see
// what those refuse -- generic (k, s), windows that do not fit a tile (many very short samples), k = 5 with an odd channel count,
// wgrads of samples shorter than a slab -- where a division per gather does not matter.
hipError_t pgconv::launch_im2col(int kind, const IgemmParams& p, int grid, hipStream_t st, int prec) {
    return launch_kind<0, 0>((Kind)kind, p, grid, st, prec);
}
