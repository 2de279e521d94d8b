// conv_raw_wgrad.hip -- raw-window variant of the G (wgrad) kernel.
#include "conv_common.h"

namespace {

// ----------------------------------------------------------------------------------------------------------------
// Raw-window variant of the G (wgrad) kernel: dW[m][(q,j)] = sum_{k=(b,i)} P[b,m,i] * Q[b,q,s*i+j-p], tile 128 (m) x 256
// ((q,j) columns = 256/k whole channels; k = 5: 51 channels = 255 columns, the 256th is idle and column tiles are 255 apart).  Per slab of 16 consecutive (b,i) the columns of one channel are k shifted
// views of the SAME piece of Q's row: positions s*i0 - p + [0, 15 s + k).  That window is staged once per channel
// (LDS image [sub][channel][WLP], sub 1 only filled when the slab runs over the end of sample b into b+1) and the
// B fragment of column (q,j), slab element kl is read at  q*WLP + j + s*kl  (+ a wave-uniform shift for kl past the
// sample boundary).  For k = 32 that is 6.4x fewer bytes than the im2col tile.
// ----------------------------------------------------------------------------------------------------------------
template <int KW, int S> struct GRaw {
    static constexpr int WL = 15 * S + KW;                                   // window floats actually read
    static constexpr int WLP = (KW == 32) ? 64 : (KW == 8 ? (S == 1 ? 24 : 40) : 36);   // padded; keeps reads conflict-free
    static constexpr int NQT = RBN / KW;                                      // whole channels per tile (k = 5: 51, one idle column)
    static constexpr int TNV = NQT * KW;                                      // columns of a tile that exist; also the tile pitch in N
    static constexpr int SUB = NQT * WLP;                                     // floats per sub-window set
    static constexpr int SUBS = (SUB + 63) / 64 * 64;                         // its slot: a dword gather instruction writes 64 floats,
                                                                              //   also from lanes past SUB (k = 5: 1836 -> 1856)
    static constexpr int NE = (SUB + NT - 1) / NT;                            // gather pieces per thread and sub-window
    static constexpr int STG = RTILE_A + 2 * SUBS;                            // floats per LDS stage
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
    const int g = logical_wg(blockIdx.x, gridDim.x, p.whole);
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x, p.whole);
    int pos = split_lo(sp, g);
    const int pos_end = split_lo(sp, g + 1);
    int slot = 0;
    while (pos < pos_end) {
        const int tile = pos / p.nslab, sb = pos - tile * p.nslab;
        const int se = min(p.nslab, sb + (pos_end - pos));
        const int m0 = (tile / p.tilesN) * RBM, n0 = (tile % p.tilesN) * C::TNV;
        const int qbase = (tile % p.tilesN) * C::NQT;

        int aoff[8];                                   // P[b][m][i]: byte offset of row m (this thread's k column added per slab)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int m = m0 + dma_row(lane, wv, e);
            aoff[e] = m < p.M ? m * p.LP * 4 : FAR;
        }
        // window element idx = tid + NT e owned by this thread on the general (per-element) path: channel byte offset and v - pad,
        // held in registers -- short samples (30 frames: every other slab runs over a sample boundary) take that path often.  Only
        // the bf16-mode k = 4 / k = 5 kernels, which would spill, recompute them where used (RECOMP; the opaque copy of tid keeps the
        // compiler from hoisting the values out of the slab loop again).
        constexpr bool RECOMP = KW <= 5 && BF != 0;
        int choff[RECOMP ? 1 : C::NE], vv[RECOMP ? 1 : C::NE];
        if (!RECOMP) {
#pragma unroll
            for (int e = 0; e < C::NE; ++e) {
                const int idx = tid + NT * e, ql = idx / C::WLP, v = idx - ql * C::WLP;
                choff[e] = (idx < C::SUB && qbase + ql < p.Q) ? (qbase + ql) * p.Lx * 4 : FAR;
                vv[e] = v - p.p;
            }
        }
#define GRAW_CHOFF(e) (RECOMP ? ({ int t_ = tid; asm volatile("" : "+v"(t_)); const int idx_ = t_ + NT * (e), ql_ = idx_ / C::WLP; \
                                  (idx_ < C::SUB && qbase + ql_ < p.Q) ? (qbase + ql_) * p.Lx * 4 : FAR; }) : choff[RECOMP ? 0 : (e)])
#define GRAW_VV(e) (RECOMP ? ({ int t_ = tid; asm volatile("" : "+v"(t_)); const int idx_ = t_ + NT * (e); idx_ - (idx_ / C::WLP) * C::WLP - p.p; }) \
                           : vv[RECOMP ? 0 : (e)])
        // Slabs that lie inside one sample (all but one in LP/16) read the P tile as two 16-byte pieces per thread: 16 consecutive
        // frames of a row are contiguous (16-byte LDS-DMA only needs dword alignment, tools/probe/ldsdma16.hip), the (sample, frame)
        // of the slab is wave-uniform and rides in the SGPR offset.  Enabled where registers allow.
        constexpr bool FASTP = BF != 2 && !(KW <= 5 && BF != 0);   // (the bf16-mode k = 4 / k = 5 kernels are at the register limit)
        constexpr bool FASTW = FASTP && KW != 4;          // SGPR-offset window gathers: k = 4 has 9 pieces per thread, their offsets spill
        // window inside the row (the common case): the whole window set of the tile loads as 16-byte pieces -- piece pc =
        // floats [4 pc, 4 pc + 4) of the [channel][WLP] image (WLP is a multiple of 4, so a piece never straddles channels;
        // the source only needs dword alignment): SUB / 4 pieces = 2 wave instructions for k = 32 instead of 8
        constexpr int NE16 = (C::SUB / 4 + NT - 1) / NT;
        static_assert(C::WLP % 4 == 0, "16-byte window pieces");
        int pv[FASTP ? 2 : 1], woff[FASTW ? NE16 : 1];
        if (FASTW) {
#pragma unroll
            for (int e = 0; e < NE16; ++e) {
                const int f = 4 * (tid + NT * e), ql = f / C::WLP, v0 = f - ql * C::WLP;
                woff[e] = (f < C::SUB && qbase + ql < p.Q) ? ((qbase + ql) * p.Lx + v0) * 4 : FAR;
            }
        }
        if (FASTP) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int m = m0 + dma16_row(lane, wv, e);
                pv[e] = m < p.M ? (m * p.LP + dma16_kc(lane)) * 4 : FAR;
            }
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
        const int kc = p.LP - gi;                      /* elements of this slab left in sample gb */      \
        if (FASTP && kc >= 16 && gb < p.B) {                                                              \
            const int sa = gb * pbs4 + gi * 4;                                                            \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) dma16s(rp, As + wv * 192 + e * 1024, pv[FASTP ? e : 0], sa); \
        } else {                                                                                          \
          int bb, ii; divmod24(k0 + kt, p.LP, p.inv_LP, bb, ii);                                          \
          const int po = bb < p.B ? bb * pbs4 + ii * 4 : OOB;                                             \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) dma4(rp, As + e * 256, aoff[e] + po); }           \
        const int w0 = S * gi - p.p;                   /* memory position of window element 0 */          \
        if (FASTW && kc >= 16 && gb < p.B && w0 >= 0 && w0 + C::WLP <= p.Lx) {   /* window inside the row: no per-lane checks */ \
            const int sw_ = gb * xbs4 + w0 * 4;                                                           \
            _Pragma("unroll") for (int e = 0; e < NE16; ++e)                                              \
                if (4 * (e * NT + wv * 64) < C::SUB) dma16s(rx, (STAGE_PTR) + RTILE_A + 4 * (e * NT + wv * 64), woff[FASTW ? e : 0], sw_); \
        } else {                                                                                          \
        const int sb0 = gb < p.B ? gb * xbs4 : -NEVER, sb1 = (kc < 16 && gb + 1 < p.B) ? (gb + 1) * xbs4 : -NEVER; \
        _Pragma("unroll") for (int e = 0; e < C::NE; ++e) {                                               \
            if ((e + 1) * NT <= C::SUB || e * NT + wv * 64 < C::SUB) {                                    \
                const int ps = S * gi + GRAW_VV(e);                                                       \
                dma4(rx, Bw + e * NT, ((unsigned)ps < (unsigned)p.Lx && sb0 >= 0) ? sb0 + GRAW_CHOFF(e) + ps * 4 : FAR); \
            }                                                                                             \
        }                                                                                                 \
        if (kc < 16) {                                 /* slab runs into the next sample: second sub-window */ \
            _Pragma("unroll") for (int e = 0; e < C::NE; ++e) {                                           \
                if ((e + 1) * NT <= C::SUB || e * NT + wv * 64 < C::SUB) {                                                          \
                    const int ps = GRAW_VV(e);                                                            \
                    dma4(rx, Bw + C::SUBS + e * NT, ((unsigned)ps < (unsigned)p.Lx && sb1 >= 0) ? sb1 + GRAW_CHOFF(e) + ps * 4 : FAR); \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        }                                                                                                 \
        kc_next = kc < 16 ? kc : 16;                                                                      \
        gi += BK; if (gi >= p.LP) { gi -= p.LP; ++gb; }                                                   \
    }

        PG_STAMP_DECL
        int kc_next;
        GRAW_ISSUE(lds, sb * BK)
        kc_cur = kc_next;
        __syncthreads();
        for (int sl = sb; sl < se; ++sl) {
            const int cur = (sl - sb) & 1;
            PG_STAMP(0)
#if defined(PG_G_ABL) && PG_G_ABL == 1      /* dev ablation (wrong results): no gathers inside the loop */
            kc_next = 16;
#else
            GRAW_ISSUE(lds + (cur ^ 1) * C::STG, (sl + 1) * BK)
#endif
            __builtin_amdgcn_sched_barrier(0);
            PG_STAMP(1)
            {   // fragments + MFMA for slab sl
                const float* As = lds + cur * C::STG;
                const float* Bw = As + RTILE_A;
                const int r = lane & 31, h = lane >> 5, sw = (r >> 2) & 3;
                const float* ap = As + (wm * 64 + r) * BK;
                f32x4 a[2][2];
                float b[4][8];
                // read order = use order (the MFMAs below run kk-major): with the operands of kk = 0, 1 first the first MFMA waits for
                // 6 of the 20 fragment reads instead of 17 (hipcc keeps source order; it had sorted them operand by operand).  + 0.4 %
                // on the k = 32 layers, + 0.8-1.3 % on the k = 8 ones.  (Round 4 also tried, on this kernel: the CU's second workgroup
                // started ~2000 cycles late, or at a higher wave priority, to break the lockstep of the two; the next slab's gathers
                // issued one per 8 MFMAs instead of at the slab's start -- all neutral.  What the in-loop gathers cost as a whole:
                // without them (-DPG_G_ABL=1, wrong results) 94 % of the pipe against 88 %; a sample end inside a slab, 1 slab in 8
                // at 129 frames, costs under 2 of those points (warm: 128 frames 88.2 %, 129 frames 86.5 %), and a K order without such
                // slabs -- whole slabs per sample, then "leftover" slabs of one frame of 16 samples -- measured equal: tools/dbg/wgrad_frames.py.)
                if (kc_cur >= 16) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i][0] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BK + (((2 * h) ^ sw) << 2));
#pragma unroll
                    for (int ip = 0; ip < 4; ++ip) {
                        if (ip == 2) {
#pragma unroll
                            for (int i = 0; i < 2; ++i) a[i][1] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BK + (((2 * h + 1) ^ sw) << 2));
                        }
#pragma unroll
                        for (int jb = 0; jb < 4; ++jb) { b[jb][2 * ip] = Bw[bbase[jb] + S * 2 * ip]; b[jb][2 * ip + 1] = Bw[bbase[jb] + S * (2 * ip + 1)]; }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int i = 0; i < 2; ++i) a[i][c] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BK + (((2 * h + c) ^ sw) << 2));        // elements kl >= kc_cur live in the second sub-window, which starts at frame 0 of the next sample
                    // (volatile reads: otherwise hipcc sinks the loads of both paths into one tail with per-element selected
                    // addresses, which cost the common path 60 VALU and left its 32 reads unpaired)
                    typedef const volatile __attribute__((address_space(3))) float* lds_vptr;
                    const lds_vptr Bv = (lds_vptr)Bw;
                    const int shift = C::SUBS - S * kc_cur;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int d = (8 * h + i >= kc_cur) ? shift : 0;
#pragma unroll
                        for (int jb = 0; jb < 4; ++jb) b[jb][i] = Bv[bbase[jb] + S * i + d];
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
                    for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc.c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk >> 2][kk & 3], b[j][kk], acc.c[i][j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            PG_STAMP(2)
            kc_cur = kc_next;
#if defined(PG_G_ABL) && PG_G_ABL == 2      /* dev ablation (wrong results): the barrier does not wait for the gathers */
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
            __syncthreads();
#endif
            PG_STAMP(3)
        }
        PG_STAMP_FLUSH
#undef GRAW_ISSUE
#undef GRAW_CHOFF
#undef GRAW_VV
        if (sb == 0 && se == p.nslab) epilogue_g<S, 2, 4>(p, acc, m0, n0, lane, wm, wn, n0 + C::TNV);
        else store_partial(p.ws, g, slot, acc, tid);
        pos += se - sb;
        slot = 1;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Per-sample slabs (round 3): the variant for SHORT samples.  Above, a slab is 16 consecutive (b, i) of the flattened K axis:
// with 30 frames per sample every other slab runs over a sample boundary, and almost every remaining one has its window partly
// outside the row, so the 2 + 2 wide gathers of the fast path are the exception and 8 + 2 x 8 dword gathers (~2400 issue cycles
// beside 4096 of MFMA) the rule: 62-69 % of the fp32 pipe at the U-Net's bottleneck.  Here a slab is 16 consecutive frames of ONE
// sample, the last slab of a sample padded (K = B * ceil(LP / 16) * 16: + 6.7 % MFMA work at 30 frames, + 4.9 % at 61, + 1.6 % at
// 126; the host only takes this kernel where that is <= 7 %).  Then EVERY slab gathers with the wide pieces and no range checks:
//   P tile:   frames gi .. gi + 15 of the rows, 16-byte pieces (frames past LP are the next row's: the A fragments of k >= LP - gi
//             are zeroed in registers, one wave-uniform branch per slab);
//   windows:  positions s gi - p + [0, WLP) of every channel as 16-byte pieces; positions outside [0, Lx) hold a neighbouring
//             row's values (or zeros past the tensor): on slabs whose window leaves the row -- wave-uniform -- the B fragment
//             elements are range-checked and zeroed in registers (64 VALU beside 64 MFMAs).
// Zeroing is a select, not a product: every operand that reaches the matrix pipe is real data or 0.  The only gather whose
// offset could become negative -- channel 0's first pieces, p positions in front of the tensor's first row -- is dropped from
// the wide gathers and re-loaded element-wise by ONE extra dword instruction of wave 0 (tiles with qbase == 0 only).
// ----------------------------------------------------------------------------------------------------------------
template <int KW, int S, int BF>
__global__ __launch_bounds__(NT, 2) void conv_g_ps_kernel(const IgemmParams p) {
    using C = GRaw<KW, S>;
    constexpr int STG = RTILE_A + (C::SUB + 255) / 256 * 256;          // window image rounded up to whole 16-byte wave instructions
    constexpr int NE16 = (C::SUB / 4 + NT - 1) / NT;
    static_assert(C::WLP % 4 == 0 && C::WLP <= 64, "16-byte window pieces; channel 0 inside the first 64 floats");
    __shared__ __attribute__((aligned(16))) float lds[2 * STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv >> 1, wn = wv & 1;
    const rsrc_t rp = make_rsrc(p.pt, p.pt_bytes), rx = make_rsrc(p.x, p.x_bytes);
    const float slopeA = act_slope(p.act_p), slopeB = act_slope(p.act_x);
    const int pbs4 = (int)p.pt_bs * 4, xbs4 = (int)p.x_bs * 4;
    const int cps = (p.LP + 15) >> 4;                                  // slabs (chunks of 16 frames) per sample
    const int g = logical_wg(blockIdx.x, gridDim.x, p.whole);
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x, p.whole);
    int pos = split_lo(sp, g);
    const int pos_end = split_lo(sp, g + 1);
    int slot = 0;
    while (pos < pos_end) {
        const int tile = pos / p.nslab, sb = pos - tile * p.nslab;
        const int se = min(p.nslab, sb + (pos_end - pos));
        const int m0 = (tile / p.tilesN) * RBM, n0 = (tile % p.tilesN) * C::TNV;
        const int qbase = (tile % p.tilesN) * C::NQT;

        int pv[2], woff[NE16];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int m = m0 + dma16_row(lane, wv, e);
            pv[e] = m < p.M ? (m * p.LP + dma16_kc(lane)) * 4 : FAR;
        }
#pragma unroll
        for (int e = 0; e < NE16; ++e) {
            const int f = 4 * (tid + NT * e), ql = f / C::WLP, v0 = f - ql * C::WLP;
            const int off = ((qbase + ql) * p.Lx + v0 - p.p) * 4;         // >= 0 except for channel 0's pieces in front of the tensor
            woff[e] = (f < C::SUB && qbase + ql < p.Q && off >= 0) ? off : FAR;
        }
        // wave 0's element-wise reload of the first 64 floats of the image (channel 0 and the start of channel 1) on tiles that hold channel 0
        const bool fix0 = qbase == 0 && wv == 0;
        int f0_off, f0_v;
        { const int ql = lane / C::WLP, v = lane - ql * C::WLP; f0_off = ql < p.Q ? ql * p.Lx * 4 : FAR; f0_v = v - p.p; }
        int bbase[4], pj[4];                           // fragment base of this lane's 4 columns; their window position at slab element 0
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int c = wn * 128 + jb * 32 + (lane & 31), qc = c / KW;
            bbase[jb] = qc * C::WLP + (c - qc * KW) + S * 8 * (lane >> 5);
            pj[jb] = (c - qc * KW) + S * 8 * (lane >> 5) - p.p;
        }
        AccR acc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.c[i][j][r] = 0.f;

        int gb = sb / cps, gc = sb - gb * cps;          // (sample, chunk) of the slab being GATHERED (one ahead of the multiplied one)
#define GPS_ISSUE(STAGE_PTR)                                                                              \
    {   float* const As = (STAGE_PTR) + wv * 64;                                                          \
        if (gb < p.B) {                                                                                   \
            const int gi = gc << 4;                                                                       \
            const int sa = gb * pbs4 + gi * 4;                                                            \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) dma16s(rp, As + wv * 192 + e * 1024, pv[e], sa); \
            const int sw_ = gb * xbs4 + S * gi * 4;                                                       \
            _Pragma("unroll") for (int e = 0; e < NE16; ++e)                                              \
                if (4 * (e * NT + wv * 64) < C::SUB) dma16s(rx, (STAGE_PTR) + RTILE_A + 4 * (e * NT + wv * 64), woff[e], sw_); \
            if (fix0) {                                                                                   \
                const int ps = S * gi + f0_v;                                                             \
                dma4(rx, (STAGE_PTR) + RTILE_A, (unsigned)ps < (unsigned)p.Lx ? gb * xbs4 + f0_off + ps * 4 : FAR); \
            }                                                                                             \
        }                                                                                                 \
        if (++gc == cps) { gc = 0; ++gb; }                                                                \
    }
        int mb = gb, mc = gc;                           // (sample, chunk) of the slab being MULTIPLIED
        GPS_ISSUE(lds)
        __syncthreads();
        for (int sl = sb; sl < se; ++sl) {
            const int cur = (sl - sb) & 1;
            GPS_ISSUE(lds + (cur ^ 1) * STG)
            __builtin_amdgcn_sched_barrier(0);
            {   // fragments + MFMA for slab sl = (mb, mc)
                const float* As = lds + cur * STG;
                const float* Bw = As + RTILE_A;
                const int r = lane & 31, h = lane >> 5, sw = (r >> 2) & 3;
                const float* ap = As + (wm * 64 + r) * BK;
                const int gi = mc << 4, kc = p.LP - gi;              // valid frames of this slab (>= 16: all)
                const int w0 = S * gi;                               // window element v sits at row position w0 + v - p
                f32x4 a[2][2];
                float b[4][8];
                // read order = use order (kk-major; see conv_g_raw_kernel)
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][0] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BK + (((2 * h) ^ sw) << 2));
#pragma unroll
                for (int ip = 0; ip < 4; ++ip) {
                    if (ip == 2) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) a[i][1] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BK + (((2 * h + 1) ^ sw) << 2));
                    }
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb) { b[jb][2 * ip] = Bw[bbase[jb] + S * 2 * ip]; b[jb][2 * ip + 1] = Bw[bbase[jb] + S * (2 * ip + 1)]; }
                }
                if (kc < 16) {                          // the sample's last, padded slab: frames past LP are not this row's
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int v = 0; v < 4; ++v) a[i][c][v] = (8 * h + 4 * c + v < kc) ? a[i][c][v] : 0.f;
                }
                if (w0 < p.p || w0 - p.p + C::WLP > p.Lx) {   // the window leaves the row: zero what lies outside [0, Lx)
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                        for (int i = 0; i < 8; ++i) b[jb][i] = (unsigned)(w0 + pj[jb] + S * i) < (unsigned)p.Lx ? b[jb][i] : 0.f;
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
            if (++mc == cps) { mc = 0; ++mb; }
            __syncthreads();
        }
#undef GPS_ISSUE
        (void)mb;
        if (sb == 0 && se == p.nslab) epilogue_g<S, 2, 4>(p, acc, m0, n0, lane, wm, wn, n0 + C::TNV);
        else store_partial(p.ws, g, slot, acc, tid);
        pos += se - sb;
        slot = 1;
    }
}

template <int KW, int S>
hipError_t launch_g_raw(const IgemmParams& p, int grid, hipStream_t st, int prec) {
    if (p.g_ps) {        // per-sample slabs (host: short samples whose padded K costs <= 7 %)
        if (prec == 1) hipLaunchKernelGGL((conv_g_ps_kernel<KW, S, 1>), dim3(grid), dim3(NT), 0, st, p);
        else if (prec == 2) hipLaunchKernelGGL((conv_g_ps_kernel<KW, S, 2>), dim3(grid), dim3(NT), 0, st, p);
        else hipLaunchKernelGGL((conv_g_ps_kernel<KW, S, 0>), dim3(grid), dim3(NT), 0, st, p);
        return hipGetLastError();
    }
    if (prec == 1) hipLaunchKernelGGL((conv_g_raw_kernel<KW, S, 1>), dim3(grid), dim3(NT), 0, st, p);
    else if (prec == 2) hipLaunchKernelGGL((conv_g_raw_kernel<KW, S, 2>), dim3(grid), dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((conv_g_raw_kernel<KW, S, 0>), dim3(grid), dim3(NT), 0, st, p);
    return hipGetLastError();
}

}  // namespace

hipError_t pgconv::launch_raw_g(const IgemmParams& p, int grid, hipStream_t st, int prec) {
    if (p.k == 32) return launch_g_raw<32, 2>(p, grid, st, prec);
    if (p.k == 8 && p.s == 1) return launch_g_raw<8, 1>(p, grid, st, prec);
    if (p.k == 8) return launch_g_raw<8, 2>(p, grid, st, prec);
    if (p.k == 5) return launch_g_raw<5, 2>(p, grid, st, prec);
    return launch_g_raw<4, 2>(p, grid, st, prec);
}
