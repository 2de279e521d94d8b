// conv_raw_impl.h -- raw-window variants of the F and T kernels (conv fwd / convT dgrad; convT fwd / conv dgrad), as a
// template over the workgroup tile shape; instantiated by conv_raw.hip (128 x 256) and conv_raw_tall.hip (256 x 128).
#pragma once
#include "conv_common.h"

namespace {

// ================================================================================================================
// Raw-window variants of the F and T kernels: workgroup tile 128 (M) x 256 (N), wave tile 64 x 128.
//
// An im2col tile holds every activation element k/s times and needs one gather instruction per 64 elements.  Here the activation
// operand is staged RAW: for each channel of a slab one contiguous window of the input row (every element once,
// zero-filled outside [0, Lx)), and the im2col overlap is resolved when the MFMA fragments are read: column c of the
// tile reads taps at window offset vcol(c) + tap (F) or vcol(c) - tap (T), vcol(c) = s'*c + 16*seg(c), where seg(c)
// counts the sample boundaries between column 0 and c (a 16-float gap per boundary keeps windows of different samples
// apart).  The tile is made wide on the activation side, where bytes are now cheap: per slab 8 KB of weights plus ~1-2 KB
// of activations feed 128x256x16 MACs -- 60 % fewer global->LDS bytes per MFMA than the 256x128 im2col tiling.
// Supported when the taps per channel in K order (F: k, T: k/s) are 4, 8, 16 or 32 and the windows fit RS floats;
// generic (k, s) stays on the im2col kernels.
//
// k = 5, s = 2 (the innermost up-conv, model.py:94-95: k_size + 1) runs here as a VIRTUAL k = 8 (round 3): the weight tile is
// loaded as if every (row, channel) had 8 taps -- 16-byte pieces at the real 20-byte row pitch, so floats 5..7 of a chunk
// pair are the first taps of the NEXT weight row -- and the k positions of the three virtual taps are never multiplied: the
// fp32 path skips their MFMAs (v_mfma_f32_32x32x2_f32 takes one k per lane half, and both halves of such an MFMA carry a
// virtual tap: T form 6 of 16 per column block, F form 3 of 8), the bf16 modes zero the fragment elements.  MFMA work is
// exactly the algorithmic 5/8 of the virtual problem, i.e. 1.0x; only the gathers and fragment reads pay for 8 taps.
// ================================================================================================================
// ds_read_b32-based B fragments: lane (column block jb, column r, half h) needs k = 8h .. 8h+7 of the slab, i.e.
// (channel qi, tap tau) = divmod(8h + i, TJ); the element lives at  qi*RS + bbase[jb] +/- tau.
struct RawFrags { f32x4 a[2][2]; float b[4][8]; };

// K5: 0 = real taps only; 1 = T form of k = 5 on the phase-major image (taps 5, 6, 7 of each 8 are virtual);
//     2 = F form of k = 5 (k = 8 h + i: taps i = 5, 6, 7 are virtual)
template <int K5> __device__ __forceinline__ constexpr bool k5_virtual(int row_block, int kk) {
    return K5 == 1 ? (row_block == 0 ? (kk & 3) == 3 : (kk & 3) >= 2) : (K5 == 2 ? kk >= 5 : false);
}

// nbv: column blocks (of 32) of the tile that hold columns at all -- wave-uniform; < 4 only on the tall tile's fp32 pass at few columns
// (batch-1 inference: 65 / 62 / 29 / 14 columns of 128), where the fragments and MFMAs of the empty blocks are skipped: at N = 65 that
// pass was MFMA-bound on padding (135 TFLOP/s for 4 blocks), not weight-bound
template <int TJ, bool DESC, int RS, bool PM, int K5, bool ZERO_VIRTUAL>
__device__ __forceinline__ void raw_load_frags(const float* __restrict__ As, const float* __restrict__ Bw, int lane, int wm,
                                               const int (&bbase)[4], float slopeA, float slopeB, RawFrags& f, int nbv = 4) {
    const int r = lane & 31, h = lane >> 5;
    if (PM) {
        // phase-major weight image (stride-2 T kernels): row o = 32 floats = the slab's 16 k x 2 phases exactly as they lie in
        // memory (W[q][o][2 jj + phi], phases interleaved), 16-byte chunks XOR-swizzled by (o >> 1) & 7.  The lane's 16 floats
        // [16 h, 16 h + 16) de-interleave in registers: even elements are the phase-0 row's fragment, odd ones the phase-1 row's,
        // so the wave's two row blocks are (o, phi = 0) and (o, phi = 1) of the same 32 output channels.
        const int sw = (r >> 1) & 7;
        const float* ap = As + (wm * 32 + r) * 32;
        f32x4 v[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) v[cc] = *reinterpret_cast<const f32x4*>(ap + (((4 * h + cc) ^ sw) << 2));
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int phi = 0; phi < 2; ++phi) f.a[phi][kk >> 2][kk & 3] = v[(2 * kk + phi) >> 2][(2 * kk + phi) & 3];
    } else {
        const int sw = (r >> 2) & 3;
        const float* ap = As + (wm * 64 + r) * BK;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) f.a[i][c] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BK + (((2 * h + c) ^ sw) << 2));
    }
    // lane part of the index: TJ == 16 -> one channel, taps 8h + i;  TJ <= 8 -> channels (8/TJ)*h + i/TJ, taps i % TJ
    const int lanepart = (TJ == 16) ? (DESC ? -8 * h : 8 * h) : (8 / TJ) * h * RS;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        if (jb >= nbv) break;
        // descending taps are read as base - (TD - 1) + (TD - 1 - tau): every constant offset stays non-negative and small, so
        // the reads pair into ds_read2_b32 with immediate offsets off ONE address register per column block (negative offsets made
        // hipcc materialise an address per pair)
        constexpr int TD = TJ < 8 ? TJ : 8;
        typedef const __attribute__((address_space(3))) float* lds_ptr;
        lds_ptr bp = (lds_ptr)(Bw + bbase[jb] + lanepart - (DESC ? TD - 1 : 0));
        if (DESC) asm volatile("" : "+v"(bp));   // opaque base (T form: measured +1 %; F form: -1 %, left to the compiler): otherwise
                                                 // the stage / tile constants are folded into one address register per read pair
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int qoff = (TJ == 16) ? 0 : (i / TJ) * RS, tau = (TJ == 16) ? i : (i % TJ);
            f.b[jb][i] = bp[qoff + (DESC ? TD - 1 - tau : tau)];
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
    if (K5 && ZERO_VIRTUAL) {       // bf16 modes (one MFMA takes all 16 k): the virtual taps' weights -- the next row's -- become 0
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (k5_virtual<K5>(i, kk)) f.a[i][kk >> 2][kk & 3] = 0.f;
    }
    if (slopeB != 1.0f) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int i = 0; i < 8; ++i) f.b[jb][i] = act_apply(f.b[jb][i], slopeB);
    }
}

// the same with the column blocks as the OUTER loop and everything behind block nbv - 1 skipped by one branch per block
template <int K5>
__device__ __forceinline__ void raw_mfma_cols(const RawFrags& f, AccR& acc, int nbv) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j >= nbv) break;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (k5_virtual<K5>(i, kk)) continue;
                acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][kk >> 2][kk & 3], f.b[j][kk], acc.c[i][j], 0, 0, 0);
            }
    }
}

template <int BF, int K5>
__device__ __forceinline__ void raw_mfma(const RawFrags& f, AccR& acc) {
    if (BF) { mfma_low_2x4<BF>(f.a, f.b, acc); return; }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (k5_virtual<K5>(i, kk)) continue;       // both lane halves of this MFMA hold a virtual tap of k = 5: never multiplied
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][kk >> 2][kk & 3], f.b[j][kk], acc.c[i][j], 0, 0, 0);
        }
}

template <int TJ, bool DESC, int RS, int BF, bool PM, int K5>
__device__ __forceinline__ void mma_slab_raw(const float* __restrict__ As, const float* __restrict__ Bw, int lane, int wm,
                                             const int (&bbase)[4], float slopeA, float slopeB, AccR& acc) {
    RawFrags f;
    raw_load_frags<TJ, DESC, RS, PM, K5, BF != 0>(As, Bw, lane, wm, bbase, slopeA, slopeB, f);
    raw_mfma<BF, K5>(f, acc);
}
template <int TJ, bool DESC, int RS, bool PM, int K5>
__device__ __forceinline__ void mma_slab_raw_cols(int nbv, const float* __restrict__ As, const float* __restrict__ Bw, int lane, int wm,
                                                  const int (&bbase)[4], float slopeA, float slopeB, AccR& acc) {
    RawFrags f;
    raw_load_frags<TJ, DESC, RS, PM, K5, false>(As, Bw, lane, wm, bbase, slopeA, slopeB, f, nbv);
    raw_mfma_cols<K5>(f, acc, nbv);
}

// TKIND false: F (conv fwd / convT dgrad, taps ascend with stride s between columns)
// TKIND true : T (convT fwd / conv dgrad in gather form, unit column stride, taps descend)
// WN = waves along N: 2 -> workgroup tile 128 (M) x 256 (N), the training shape; 1 -> 256 x 128 ("tall"): the four waves are
// stacked in M, for problems with few columns (batch-1 inference: N = frames' <= 128), where the wide tile would be mostly
// empty and the pass is bound by MFMA time per weight byte.
template <int KW, int S, bool TKIND, int BF, int WN>
__global__ __launch_bounds__(NT, 2) void conv_raw_kernel(const IgemmParams p) {
    constexpr int TM = WN == 2 ? RBM : 2 * RBM, TN = WN == 2 ? RBN : RBN / 2;   // workgroup tile
    constexpr int TA = TM * BK;                       // floats of the weight tile (8 / 16 KB)
    constexpr int AE4 = TM / 16, AE16 = TM / 64;      // dword / 16-byte gather pieces per thread for the weight tile
    constexpr int KWV = KW == 5 ? 8 : KW;             // taps per (row, channel) of the weight tile image: k = 5 is a virtual 8
    constexpr int K5 = KW == 5 ? (TKIND ? 1 : 2) : 0;
    constexpr int KWP = TKIND ? KWV / S : KWV;        // taps per channel in K order
    constexpr int TJ = KWP < 16 ? KWP : 16, NQ = 16 / TJ;
    constexpr int SC = TKIND ? 1 : S;                 // window positions per column step
    constexpr int RG = raw_gap(TJ);                   // gap between the windows of consecutive samples
    constexpr bool PM = TKIND && S == 2;              // phase-major weight image, loaded as 16-byte pieces (raw_load_frags)
    static_assert(!TKIND || S == 1 || (S == 2 && KWP <= 16), "T kernels: stride 1, or stride 2 with at most 16 taps per phase");
    constexpr int RS = SC == 1 ? RS1 : RS2;           // floats reserved per channel window
    constexpr int NPC = (RS + NT - 1) / NT;           // gather pieces per thread and window
    constexpr int STG = TA + NQ * RS;                 // floats per LDS stage
    // slabs per barrier: two when both stages still fit 64 KB -- except the k = 32 T kernel (U0 forward), which measures 2 %
    // faster with one (146 vs 143 TFLOP/s); the F form and the k = 8 kernels gain 1-5 % from two
    constexpr int SPB = (4 * STG * 4 <= 64 * 1024 && !(TKIND && KW == 32)) ? 2 : 1;
    static_assert(KWP == 2 || KWP == 4 || KWP == 8 || KWP == 16 || KWP == 32, "raw-window kernels need 2/4/8/16/32 taps per channel");
    static_assert(!TKIND || KWV % S == 0, "T raw kernel needs s | k");
    static_assert(KW != 5 || S == 2, "k = 5 is built for stride 2 only");
    __shared__ __attribute__((aligned(16))) float lds[2 * SPB * STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = WN == 2 ? wv >> 1 : wv, wn = WN == 2 ? wv & 1 : 0;
    const int kt = dma_kt(lane, wv);
#if PG_ABL == 8   /* dev-only: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6) */
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int Lcol = TKIND ? p.U : p.Ly;              // columns (output positions) per sample
    const int Ktot = p.Q * KWP, Mrows = TKIND ? p.M * S : p.M;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);
    const float slopeA = 1.0f, slopeB = act_slope(p.act_x);
    const int wq = p.M * KW;                          // T: weight stride between input channels
    // phase-major image: the LOGICAL 16-byte chunk this lane carries into LDS (physical chunk lane & 7 of a row, XOR the row's
    // swizzle -- which depends on the wave and lane only, not on the gather instruction) = floats [f, f + 4) of the row's 32:
    // channel ql of the slab, taps `within ..` of W[q][o][:]
    const int pm_f = ((lane & 7) ^ (((wv & 1) << 2) | (lane >> 4))) << 2;
    const int pm_ql = pm_f / (2 * TJ), pm_within = pm_f - pm_ql * 2 * TJ;
    const int g = logical_wg(blockIdx.x, gridDim.x, p.whole);
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x, p.whole);
    int pos = split_lo(sp, g);
    const int pos_end = split_lo(sp, g + 1);
    int slot = 0;
    while (pos < pos_end) {
        const int tile = pos / p.nslab, sb = pos - tile * p.nslab;
        const int se = min(p.nslab, sb + (pos_end - pos));
        const int m0 = (tile / p.tilesN) * TM, n0 = p.n_lo + (tile % p.tilesN) * TN;
        const int b0 = n0 / Lcol, t0 = n0 - b0 * Lcol;            // sample / position of the tile's first column
        const int nseg = (t0 + TN - 1) / Lcol + 1;
        const int rlen = SC * (TN - 1) + TJ + RG * (nseg - 1);   // floats of a channel window that are ever read
        constexpr bool SKIPB = WN == 1 && BF == 0;               // tall tile, fp32: column blocks without columns are skipped
        const int nbv = SKIPB ? __builtin_amdgcn_readfirstlane(min(4, (p.B * Lcol - n0 + 31) / 32)) : 4;

        // --- weight-tile gather constants (BYTE offsets) --------------------------------------------------------
        int aoff[AE4], avoff[AE16];
        if (TKIND) {        // stride 1: 16-byte pieces of the tap-contiguous rows (T16 below); stride 2: phase-major image
            if (PM) {
#pragma unroll
                for (int e = 0; e < AE16; ++e) {
                    const int o = m0 / 2 + (4 * e + wv) * 8 + (lane >> 3);    // row (output channel) of this lane's 16-byte chunk
                    avoff[e] = o < p.M ? (pm_ql * wq + o * KW + pm_within) * 4 : FAR;   // W[q0 + ql][o][within ..]; q0 rides in the SGPR
                }
            } else {
#pragma unroll
                for (int e = 0; e < AE4; ++e) {
                    const int mr = m0 + dma_row(lane, wv, e);
                    aoff[e] = mr < Mrows ? mr * KW * 4 : FAR;                 // W[q][o][jj]  (general path of the stride-1 kernels)
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < AE4; ++e) {
                const int m = m0 + dma_row(lane, wv, e);
                aoff[e] = (!K5 && !p.a_vec && m < p.M) ? (m * Ktot + kt) * 4 : FAR;
            }
#pragma unroll
            for (int e = 0; e < AE16; ++e) {
                const int m = m0 + dma16_row(lane, wv, e);
                const int kc = dma16_kc(lane);      // k = 5: chunk kc of the slab = floats (kc & 7) .. + 3 of channel kc >> 3, rows 5 Q floats apart
                avoff[e] = m < p.M ? (K5 ? (m * p.Q * 5 + (kc >> 3) * 5 + (kc & 7)) * 4 : (m * Ktot + kc) * 4) : FAR;
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
        // --- fast gather path (fp32, wide tile): whole slabs inside K need no per-gather index math.  Every gather offset splits
        // into a per-lane part that is fixed for the tile and a wave-uniform part that advances with the slab and rides in the
        // SGPR offset operand of the buffer load.
        //   T weights: K index k0 + kt (k0 multiple of 16, KWP | 16) = channel k0/KWP + kt/KWP, tap kt%KWP: the kt terms are per lane
        //   F weights: row offset per lane, k0 * 4 uniform
        //   windows:   position validity is per lane; channel (and, for 32-tap channels, the 0 / 16 tap offset tau0 of the slab)
        //              is uniform.  With tau0 in {0, 16} the validity of a position can differ: two per-lane offsets (NT0 sets).
        constexpr bool FAST = !(BF == 2 && TKIND && !(KW == 32 && WN == 2));   // (the other bf16x3 T kernels are at the 256-VGPR limit: the extra offsets spill)
        constexpr int NT0 = (!TKIND && KWP == 32) ? 2 : 1;
        //   T weights at stride 1: the taps of a channel are contiguous in memory AND in K, so the tile loads as 16-byte pieces
        constexpr bool T16 = TKIND && S == 1;
        int avoffk[(FAST && T16) ? AE16 : 1], voffb[FAST ? NT0 : 1][FAST ? NPC : 1];
        if (FAST) {
            if (T16) {
                const int kc = dma16_kc(lane);
#pragma unroll
                for (int e = 0; e < AE16; ++e) {
                    const int o = m0 + dma16_row(lane, wv, e);
                    avoffk[e] = o < Mrows ? ((kc / KWP) * wq + o * KW + (kc % KWP)) * 4 : FAR;
                }
            }
#pragma unroll
            for (int h = 0; h < NT0; ++h)
#pragma unroll
                for (int e = 0; e < NPC; ++e) {
                    const int ps = posb[e] + 16 * h;              // F: + tau0 (T kernels always have tau0 = 0).  tau0 stays in the
                    voffb[h][e] = (unsigned)ps < (unsigned)p.Lx ? rowb[e] + ps * 4 : FAR;   // per-lane part: that one is range-checked
                }
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
    {   float* const As = (STAGE_PTR) + wv * 64; float* const Bw = (STAGE_PTR) + TA + wv * 64;             \
        const int k0 = (K0);                                                                              \
        const bool kok = k0 < Ktot;                                                                       \
        if (FAST && k0 + BK <= Ktot) {                                                                    \
            if (T16) {                                                                                    \
                const int sa = (k0 / KWP) * wq * 4;                                                       \
                _Pragma("unroll") for (int e = 0; e < AE16; ++e) dma16s(rw, As + wv * 192 + e * 1024, avoffk[T16 ? e : 0], sa); \
            } else if (PM) {     /* stride-2 T: the phase-major rows as 16-byte pieces */                    \
                const int sa = (k0 / KWP) * wq * 4;                                                       \
                _Pragma("unroll") for (int e = 0; e < AE16; ++e) dma16s(rw, As + wv * 192 + e * 1024, avoff[e], sa); \
            } else {     /* F: 16-byte pieces always (they only need dword alignment); no branch on a_vec here */ \
                _Pragma("unroll") for (int e = 0; e < AE16; ++e) dma16s(rw, As + wv * 192 + e * 1024, avoff[e], K5 ? (k0 >> 3) * 20 : k0 * 4); \
            }                                                                                             \
            const int fq0 = k0 / KWP, ft0 = k0 - fq0 * KWP;              /* ft0 = 16 only for the odd slabs of 32-tap channels */ \
            /* pieces that every tile needs (the shortest window is SC*(TN-1)+TJ floats) issue without the per-wave check; */ \
            /* one branch per slab picks the offset set (ft0), not one per piece */                       \
            constexpr int RMIN = SC * (TN - 1) + TJ;                                                      \
            if (NT0 == 2 && ft0) {                                                                        \
                _Pragma("unroll") for (int qi = 0; qi < NQ; ++qi) {                                       \
                    const int sq = (fq0 + qi) * p.Lx * 4;                                                 \
                    _Pragma("unroll") for (int e = 0; e < NPC; ++e)                                       \
                        if ((e + 1) * 256 <= RMIN || e * 256 + wv * 64 < rlen) dma4s(rx, Bw + qi * RS + e * 256, voffb[NT0 - 1][e], sq); \
                }                                                                                         \
            } else {                                                                                      \
                _Pragma("unroll") for (int qi = 0; qi < NQ; ++qi) {                                       \
                    const int sq = (fq0 + qi) * p.Lx * 4;                                                 \
                    _Pragma("unroll") for (int e = 0; e < NPC; ++e)                                       \
                        if ((e + 1) * 256 <= RMIN || e * 256 + wv * 64 < rlen) dma4s(rx, Bw + qi * RS + e * 256, voffb[0][e], sq); \
                }                                                                                         \
            }                                                                                             \
        } else {                                                                                          \
        if (PM) {            /* K tail: channels past Q (and slabs past K) read zeros */                         \
            const int qa = k0 / KWP;                                                                      \
            const int sa = (kok && qa + pm_ql < p.Q) ? qa * wq * 4 : OOB;                                 \
            _Pragma("unroll") for (int e = 0; e < AE16; ++e) dma16(rw, As + wv * 192 + e * 1024, avoff[e] + sa); \
        } else if (TKIND) {                                                                               \
            const int kk = k0 + kt, q = kk / KWP, jj = kk - q * KWP;                                      \
            const int wo = kok ? (q * wq + S * jj) * 4 : OOB;                                             \
            _Pragma("unroll") for (int e = 0; e < AE4; ++e) dma4(rw, As + e * 256, aoff[e] + wo);         \
        } else if (K5) {     /* F form of k = 5 (Q even: every slab inside K is whole): only slabs past K come here */ \
            const int kv = kok ? (k0 >> 3) * 20 : OOB;                                                    \
            _Pragma("unroll") for (int e = 0; e < AE16; ++e) dma16(rw, As + wv * 192 + e * 1024, avoff[e] + kv); \
        } else if (p.a_vec) {                                                                             \
            const int kv = (k0 + dma16_kc(lane) < Ktot) ? k0 * 4 : OOB;                                   \
            _Pragma("unroll") for (int e = 0; e < AE16; ++e) dma16(rw, As + wv * 192 + e * 1024, avoff[e] + kv); \
        } else {                                                                                          \
            const int ka = (k0 + kt < Ktot) ? k0 * 4 : OOB;                                               \
            _Pragma("unroll") for (int e = 0; e < AE4; ++e) dma4(rw, As + e * 256, aoff[e] + ka);         \
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
            if (SKIPB) {         // (ONE code path per kernel: a second copy of the MFMA block beside 128 accumulators spills)
                mma_slab_raw_cols<TJ, TKIND, RS, PM, K5>(nbv, lds + cur * SSTG, lds + cur * SSTG + TA, lane, wm, bbase, slopeA, slopeB, acc);
                if (SPB == 2 && sl + 1 < se)
                    mma_slab_raw_cols<TJ, TKIND, RS, PM, K5>(nbv, lds + cur * SSTG + STG, lds + cur * SSTG + STG + TA, lane, wm, bbase, slopeA, slopeB, acc);
            } else {
                mma_slab_raw<TJ, TKIND, RS, BF, PM, K5>(lds + cur * SSTG, lds + cur * SSTG + TA, lane, wm, bbase, slopeA, slopeB, acc);
                if (SPB == 2 && sl + 1 < se)
                    mma_slab_raw<TJ, TKIND, RS, BF, PM, K5>(lds + cur * SSTG + STG, lds + cur * SSTG + STG + TA, lane, wm, bbase, slopeA, slopeB, acc);
            }
            __builtin_amdgcn_sched_barrier(0);
            PG_STAMP(2)
            __syncthreads();
            PG_STAMP(3)
        }
        PG_STAMP_FLUSH
#undef RAW_ISSUE
        if (sb == 0 && se == p.nslab) {
            if (PM) epilogue_t_pm<2, 4>(p, acc, m0 / 2 + wm * 32, n0 + wn * 128, lane, 0);
            else if (TKIND) epilogue_t<S, 2, 4>(p, acc, m0, n0, lane, wm, wn);
            else epilogue_f<S, 2, 4>(p, acc, m0, n0, lane, wm, wn);
        } else store_partial(p.ws, g, slot, acc, tid, nbv);
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

template <int KW, int S, bool TK, int WN>
hipError_t launch_raw(const IgemmParams& p, int grid, hipStream_t st, int prec) {
    if (prec == 1) hipLaunchKernelGGL((conv_raw_kernel<KW, S, TK, 1, WN>), dim3(grid), dim3(NT), 0, st, p);
    else if (prec == 2) hipLaunchKernelGGL((conv_raw_kernel<KW, S, TK, 2, WN>), dim3(grid), dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((conv_raw_kernel<KW, S, TK, 0, WN>), dim3(grid), dim3(NT), 0, st, p);
    return hipGetLastError();
}


template <int WN>
hipError_t launch_raw_ft_wn(int kind, const IgemmParams& p, int grid, hipStream_t st, int prec) {
    if (kind == KIND_F) {
        if (p.k == 32) return launch_raw<32, 2, false, WN>(p, grid, st, prec);
        if (p.k == 8 && p.s == 1) return launch_raw<8, 1, false, WN>(p, grid, st, prec);
        if (p.k == 8) return launch_raw<8, 2, false, WN>(p, grid, st, prec);
        if (p.k == 5) return launch_raw<5, 2, false, WN>(p, grid, st, prec);
        return launch_raw<4, 2, false, WN>(p, grid, st, prec);
    }
    if (p.k == 32) return launch_raw<32, 2, true, WN>(p, grid, st, prec);
    if (p.k == 5) return launch_raw<5, 2, true, WN>(p, grid, st, prec);
    if (p.k == 4) return launch_raw<4, 2, true, WN>(p, grid, st, prec);
    if (p.s == 1) return launch_raw<8, 1, true, WN>(p, grid, st, prec);
    return launch_raw<8, 2, true, WN>(p, grid, st, prec);
}

}  // namespace
