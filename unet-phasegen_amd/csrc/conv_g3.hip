// conv_g3.hip -- fp32 wgrad (G) kernels with ONE wave per SIMD: dW[m][(q, j)] = sum over (b, i) of P[b, m, i] * Q[b, q, s i + j - p]
// (operands and window image as conv_raw_wgrad.hip), 256 threads = 4 waves, one workgroup per CU, wave tile 256 x 64 (256 accumulator
// registers in AGPRs), workgroup tile 256 rows x 256 columns = 256 / k whole channels; every LDS read, gather and fragment fix-up
// pinned into its own MFMA gap (the structure of conv_raw3.hip / conv_h3.hip, whose headers say why and what it takes).
//
// K order.  The sum over (b, i) may run in any order; conv_raw_wgrad.hip's flat order (16 consecutive (b, i) per slab) makes every
// LP / 16-th slab straddle two samples, which costs a second window set and element-wise gathers.  Here a slab never straddles:
//   * main slabs: 16 consecutive frames of ONE sample (16-byte gathers, window positions s gi - p + [0, WLP) of every channel);
//   * PAD variant (short samples: what conv_g_ps_kernel does): the last slab of a sample is padded -- its frames past LP are zeroed
//     in the A fragments (they are the next row's data) -- K = B * ceil(LP / 16) slabs;
//   * leftover variant (LP = 16 cf + rem with a small rem, B a multiple of 16: the k = 32 layers at 129 frames): the rem last frames
//     of 16 consecutive samples form rem "leftover" slabs whose 16 k-elements are 16 SAMPLES at one frame: A tile and B tile (a plain
//     [16][256] matrix: every (sample, column) value once) gathered element-wise, 32 dword gathers per wave -- 1 slab in 129.
//     K = B * LP exactly: no padded work.
// Everything else (k = 5, bf16 operand modes, other batch sizes, an activation applied on load) stays on conv_raw_wgrad.hip.
#include "conv_common.h"

namespace {

constexpr int NT3 = 256;
constexpr int G3_RING = 3;
constexpr int G3_REGS = 256;
constexpr int G3_TM = 256, G3_TN = 256;
constexpr int G3_SLOTS = 26;              // gather slots per half-slab: behind MFMAs 12, 14, ... 62

typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

template <int KW, int S> struct G3Geo {
    static constexpr int WLP = (KW == 32) ? 64 : (KW == 8 ? (S == 1 ? 24 : 40) : 36);   // floats per channel window (conv_raw_wgrad.hip: GRaw)
    static constexpr int NQT = G3_TN / KW;                  // whole channels per tile
    static constexpr int SUB = NQT * WLP;                   // floats of the window image
    static constexpr int NWI = (SUB / 4 + 63) / 64;         // 16-byte wave instructions that cover it
    static constexpr int NW = (NWI + 3) / 4;                // ... per wave (slots past NWI repeat another wave's)
    static constexpr int SUBR = NWI * 256;                  // floats reserved (whole instructions)
    static constexpr int ND = 4 + NW + 1;                   // gathers per wave and main slab: weight rows, windows, the channel-0 fix
    static_assert(15 * S + KW <= WLP && WLP % 4 == 0 && WLP <= 64 && NWI >= 2, "window image");
};

template <int N> __device__ __forceinline__ void g3_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void g3_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }
__device__ __forceinline__ unsigned g3_lds_addr(const float* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
__device__ __forceinline__ float g3_acc(float v) {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(v));
    return x;
}

// position of a slab in the K order: sample smp of group g (PAD: groups of one sample; leftover variant: of 16), chunk chk of its
// whole 16-frame chunks -- or, left = 1, leftover frame t of the group's 16 samples.  Wave-uniform scalars, advanced incrementally.
struct G3Pos { int g, smp, chk, left; };
template <bool PAD> __device__ __forceinline__ void g3_advance(G3Pos& s, int cf, int rem) {
    if (!s.left) {
        if (++s.chk == cf) {
            s.chk = 0;
            if (PAD) ++s.g;
            else if (++s.smp == 16) { s.smp = 0; if (rem) s.left = 1; else ++s.g; }
        }
    } else if (++s.chk == rem) { s.chk = 0; s.left = 0; ++s.g; }
}
template <bool PAD> __device__ __forceinline__ G3Pos g3_pos_of(int slab, int cf, int rem) {
    G3Pos s;
    if (PAD) { s.g = slab / cf; s.chk = slab - s.g * cf; s.smp = 0; s.left = 0; return s; }
    const int spg = 16 * cf + rem;
    s.g = slab / spg;
    const int w = slab - s.g * spg;
    if (w < 16 * cf) { s.smp = w / cf; s.chk = w - s.smp * cf; s.left = 0; }
    else { s.smp = 0; s.chk = w - 16 * cf; s.left = 1; }
    return s;
}

// ---- fragments of one half-slab (k = 8 h + 4 c + 0..3) --------------------------------------------------------------------------
struct G3Frag { f32x4 a[8]; unsigned d[2][4]; float b[2][4]; };

__device__ __forceinline__ unsigned g3_a_addr(const float* stage, int c, int r, int h) {
    return g3_lds_addr(stage) + (r * 16 + (((2 * h + c) ^ ((r >> 2) & 3)) << 2)) * 4;
}
template <int I> __device__ __forceinline__ void g3_load_a(unsigned a0, G3Frag& f) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a[I]) : "v"(a0), "n"(I * 2048));
}
// main slab: the four window dwords of the lane's column, S dwords apart
template <int S, int JB, int P> __device__ __forceinline__ void g3_load_b_main(unsigned b_addr, G3Frag& f) {
    u32x2v t;
    if (P == 0) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:%2" : "=v"(t) : "v"(b_addr), "n"(S));
    else asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(b_addr), "n"(2 * S), "n"(3 * S));
    f.d[JB][2 * P] = t[0]; f.d[JB][2 * P + 1] = t[1];
}
// leftover slab: the B tile is a plain [16 k][256 columns] matrix: four dwords 1024 bytes apart
template <int JB, int P> __device__ __forceinline__ void g3_load_b_left(unsigned b_addr, G3Frag& f) {
    u32x2v t;
    if (P == 0) asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:4" : "=v"(t) : "v"(b_addr));
    else asm volatile("ds_read2st64_b32 %0, %1 offset0:8 offset1:12" : "=v"(t) : "v"(b_addr));
    f.d[JB][2 * P] = t[0]; f.d[JB][2 * P + 1] = t[1];
}

// what a half-slab's fragment fix-ups need to know about the slab they were read from (wave-uniform)
struct G3Fix { int left; int w0; int kc; int edge; };   // leftover slab?; main: window position of element 0 + p (= s gi); valid frames
                                                       // (>= 16: all); does the window leave the row?

// One half-slab: the 64 MFMAs of `cur` and, one piece per MFMA gap, the reads of the next half-slab `nxt` (window dwords behind MFMAs
// 0-3, weight rows behind 4-11), gathers of the slab two ahead behind the even MFMAs 12 ... 62 (`issue(E0 + n)`), the wait for the
// reads behind 24 and the fix-ups of `nxt` behind the odd MFMAs 25 ... 43: PAD, zeroing of frames past LP in the weight-row fragments
// one row block at a time; range check of the window values.
// (ONE instantiation serves both slab types: the type only selects the B read instruction -- a two-instruction wave-uniform branch
// inside the gap -- and switches the range check off.  With the MFMA stream itself duplicated under an if / else, hipcc no longer
// kept the 256 accumulators in place across the join and spilled ~650 registers.)
template <int S, bool PAD, int E0, typename Issue>
__device__ __forceinline__ void g3_half(const G3Frag& cur, G3Frag& nxt, unsigned a0, unsigned b0, unsigned b1, int c, int h,
                                        const G3Fix fx, const int (&pj)[2], int Lx, AccT<8, 2>& acc, const Issue& issue) {
#define G3_CHUNK(C, WORK)                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                           \
    acc.c[((C) >> 1) & 7][(C) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[((C) >> 1) & 7][(C) >> 4], cur.b[(C) & 1][(C) >> 4], \
                                                                         acc.c[((C) >> 1) & 7][(C) & 1], 0, 0, 0);               \
    WORK;
#if defined(PG_G3_ABL) && PG_G3_ABL == 3
#define G3_LB(JB, P) { g3_load_b_main<S, JB, P>(JB ? b1 : b0, nxt); }
#else
#define G3_LB(JB, P) { if (!PAD && fx.left) g3_load_b_left<JB, P>(JB ? b1 : b0, nxt); else g3_load_b_main<S, JB, P>(JB ? b1 : b0, nxt); }
#endif
    // Fix-ups cost MFMA issue slots even inside the gaps (measured: always-on select + multiply + two max per value -- 365 VALU per
    // slab -- took the kernel from 0.88 to 0.78 of the pipe; the same behind wave-uniform tests -- ~40 scalar branches per slab --
    // cost as much).  This kernel therefore covers the case the engine uses -- operands stored activated, loaded as they are -- and
    // carries only: PAD, the zeroing of frames past the sample's end (always on: 4 compares + 4 selects per row block); the range
    // check of window values (always on: add + compare + select per value).
#define G3_FA(I) {                                                                                                               \
        if (PAD) {                                                                                                               \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) nxt.a[I][kk] = (8 * h + 4 * c + kk < fx.kc) ? nxt.a[I][kk] : 0.f;   \
        } }
#if defined(PG_G3_ABL) && PG_G3_ABL >= 2      /* dev ablation (wrong results): no range selects; 3: no leftover-type branches either */
#define G3_FB(JB) { _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) nxt.b[JB][kk] = __builtin_bit_cast(float, nxt.d[JB][kk]); }
#else
#define G3_FB(JB) {                                                                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                                       \
            const float v = __builtin_bit_cast(float, nxt.d[JB][kk]);                                                            \
            nxt.b[JB][kk] = ((!PAD && fx.left) || (unsigned)(fx.w0 + pj[JB] + S * (4 * c + kk)) < (unsigned)Lx) ? v : 0.f;       \
        } }
#endif
    G3_CHUNK(0, G3_LB(0, 0)) G3_CHUNK(1, G3_LB(0, 1)) G3_CHUNK(2, G3_LB(1, 0)) G3_CHUNK(3, G3_LB(1, 1))
    G3_CHUNK(4, g3_load_a<0>(a0, nxt)) G3_CHUNK(5, g3_load_a<1>(a0, nxt)) G3_CHUNK(6, g3_load_a<2>(a0, nxt)) G3_CHUNK(7, g3_load_a<3>(a0, nxt))
    G3_CHUNK(8, g3_load_a<4>(a0, nxt)) G3_CHUNK(9, g3_load_a<5>(a0, nxt)) G3_CHUNK(10, g3_load_a<6>(a0, nxt)) G3_CHUNK(11, g3_load_a<7>(a0, nxt))
    G3_CHUNK(12, issue(E0 + 0)) G3_CHUNK(13, (void)0) G3_CHUNK(14, issue(E0 + 1)) G3_CHUNK(15, (void)0)
    G3_CHUNK(16, issue(E0 + 2)) G3_CHUNK(17, (void)0) G3_CHUNK(18, issue(E0 + 3)) G3_CHUNK(19, (void)0)
    G3_CHUNK(20, issue(E0 + 4)) G3_CHUNK(21, (void)0) G3_CHUNK(22, issue(E0 + 5)) G3_CHUNK(23, (void)0)
    G3_CHUNK(24, issue(E0 + 6))
    __builtin_amdgcn_sched_barrier(0);
    g3_lgkm0();                                // every read of `nxt` was issued 13 or more MFMAs (800 cycles) ago
    G3_CHUNK(25, G3_FA(0)) G3_CHUNK(26, issue(E0 + 7)) G3_CHUNK(27, G3_FA(1)) G3_CHUNK(28, issue(E0 + 8))
    G3_CHUNK(29, G3_FA(2)) G3_CHUNK(30, issue(E0 + 9)) G3_CHUNK(31, G3_FA(3)) G3_CHUNK(32, issue(E0 + 10))
    G3_CHUNK(33, G3_FA(4)) G3_CHUNK(34, issue(E0 + 11)) G3_CHUNK(35, G3_FA(5)) G3_CHUNK(36, issue(E0 + 12))
    G3_CHUNK(37, G3_FA(6)) G3_CHUNK(38, issue(E0 + 13)) G3_CHUNK(39, G3_FA(7)) G3_CHUNK(40, issue(E0 + 14))
    G3_CHUNK(41, G3_FB(0)) G3_CHUNK(42, issue(E0 + 15)) G3_CHUNK(43, G3_FB(1)) G3_CHUNK(44, issue(E0 + 16))
    G3_CHUNK(45, (void)0) G3_CHUNK(46, issue(E0 + 17)) G3_CHUNK(47, (void)0) G3_CHUNK(48, issue(E0 + 18))
    G3_CHUNK(49, (void)0) G3_CHUNK(50, issue(E0 + 19)) G3_CHUNK(51, (void)0) G3_CHUNK(52, issue(E0 + 20))
    G3_CHUNK(53, (void)0) G3_CHUNK(54, issue(E0 + 21)) G3_CHUNK(55, (void)0) G3_CHUNK(56, issue(E0 + 22))
    G3_CHUNK(57, (void)0) G3_CHUNK(58, issue(E0 + 23)) G3_CHUNK(59, (void)0) G3_CHUNK(60, issue(E0 + 24))
    G3_CHUNK(61, (void)0) G3_CHUNK(62, issue(E0 + 25)) G3_CHUNK(63, (void)0)
    __builtin_amdgcn_sched_barrier(0);
#undef G3_FB
#undef G3_FA
#undef G3_LB
#undef G3_CHUNK
}

__device__ __forceinline__ void store_partial_g3(float* ws, int g, int slot, const AccT<8, 2>& acc, int tid) {
    float* dst = ws + ((long)(g * 2 + slot) * G3_REGS) * NT3 + tid;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((i * 2 + j) * 16 + r) * NT3] = g3_acc(acc.c[i][j][r]);
}

template <int KW, int S, bool PAD>
__global__ __launch_bounds__(NT3, 1) void conv_g3_kernel(const IgemmParams p) {
    using C = G3Geo<KW, S>;
    constexpr int TA = G3_TM * BK;                                   // floats of the weight-row tile (16 KB)
    constexpr int BREG = (!PAD && C::SUBR < 16 * G3_TN) ? 16 * G3_TN : C::SUBR;   // window image, or the leftover slab's [16][256] tile
    constexpr int STG = TA + BREG;
    constexpr int NDM = C::ND;                                       // gathers per wave of a main slab
    constexpr int NDL = 32;                                          // ... of a leftover slab: 16 + 16 dword instructions
    static_assert(G3_RING * STG * 4 <= 160 * 1024 && NDM <= G3_SLOTS && NDL <= 2 * G3_SLOTS, "LDS budget / gather slots");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wv;
    const int r = lane & 31, h = lane >> 5;
    const rsrc_t rp = make_rsrc(p.pt, p.pt_bytes), rx = make_rsrc(p.x, p.x_bytes);
    // (host: this kernel only where neither operand has an activation on load)
    const int pbs4 = (int)p.pt_bs * 4, xbs4 = (int)p.x_bs * 4;
    const int cf = PAD ? (p.LP + 15) >> 4 : p.LP >> 4;               // main slabs per sample
    const int rem = PAD ? 0 : p.LP & 15;                             // leftover frames per sample
    const int g = logical_wg(blockIdx.x, gridDim.x, p.whole);
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x, p.whole);
    int pos = split_lo(sp, g);
    const int pos_end = split_lo(sp, g + 1);
    int slot = 0;
    while (pos < pos_end) {
        const int tile = pos / p.nslab, sb = pos - tile * p.nslab;
        const int se = min(p.nslab, sb + (pos_end - pos));
        const int m0 = (tile / p.tilesN) * G3_TM, n0 = (tile % p.tilesN) * G3_TN;
        const int qbase = (tile % p.tilesN) * C::NQT;

        // --- main slabs: per-lane gather offsets (bytes), fixed for the tile ---------------------------------------------------------
        int pv[4], woff[C::NW], wslot[C::NW];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + dma16_row(lane, wv, e);
            pv[e] = m < p.M ? (m * p.LP + dma16_kc(lane)) * 4 : FAR;
        }
#pragma unroll
        for (int e = 0; e < C::NW; ++e) {
            const int ws0 = wv + 4 * e;
            // a slot past the image repeats another wave's (same bytes to the same place) so that every wave issues NW window gathers --
            // never slot 0, which wave 0's channel-0 fix below must be the last to write
            const int ws = ws0 < C::NWI ? ws0 : 1 + (ws0 - C::NWI) % (C::NWI - 1);
            const int f = 4 * (ws * 64 + lane), ql = f / C::WLP, v0 = f - ql * C::WLP;
            const int off = ((qbase + ql) * p.Lx + v0 - p.p) * 4;     // >= 0 except channel 0's pieces in front of the tensor
            woff[e] = (f < C::SUB && qbase + ql < p.Q && off >= 0) ? off : FAR;
            wslot[e] = ws;
        }
        const bool fix0 = qbase == 0 && wv == 0;        // element-wise reload of the image's first 64 floats (channel 0 and the start of 1)
        int f0_off, f0_v;
        { const int ql = lane / C::WLP, v = lane - ql * C::WLP; f0_off = ql < p.Q ? ql * p.Lx * 4 : FAR; f0_v = v - p.p; }
        // --- leftover slabs: lane (row 4 wv + (lane >> 4) (+ 16 e), k = kt) of the weight-row tile, column wv 64 + lane of the B tile -----
        const int kt = dma_kt(lane, wv);
        const int la_row = m0 + 4 * wv + (lane >> 4);
        const int la_off = la_row * p.LP * 4 + kt * pbs4;
        int lb_ch, lb_pj;
        { const int cc = wv * 64 + lane, q = cc / KW; lb_ch = qbase + q < p.Q ? (qbase + q) * p.Lx * 4 : FAR; lb_pj = cc - q * KW - p.p; }
        // --- fragment bases --------------------------------------------------------------------------------------------------------
        int bbase[2], pj[2], bcol[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int cc = wn * 64 + jb * 32 + r, qc = cc / KW;
            bbase[jb] = qc * C::WLP + (cc - qc * KW) + S * 8 * h;
            pj[jb] = (cc - qc * KW) + S * 8 * h - p.p;
            bcol[jb] = 8 * h * G3_TN + cc;
        }
        AccT<8, 2> acc;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc.c[i][j][q] = 0.f;

        // gather number e of the slab at position `s` into `stage`.  Slabs past this workgroup's K range: out-of-range offsets (zeros).
        auto issue_piece = [&](float* stage, const G3Pos& s, bool live, int e) {
            if (PAD || !s.left) {
                const int smp = PAD ? s.g : 16 * s.g + s.smp, gi = s.chk << 4;
                if (e < 4) dma16s(rp, stage + (4 * e + wv) * 256, live ? pv[e < 4 ? e : 0] : FAR, smp * pbs4 + gi * 4);
                else if (e < 4 + C::NW) {
                    const int x = e < 4 ? 0 : (e - 4 < C::NW ? e - 4 : 0);
                    dma16s(rx, stage + TA + wslot[x] * 256, live ? woff[x] : FAR, smp * xbs4 + S * gi * 4);
                } else if (e == 4 + C::NW) {
                    if (fix0) {
                        const int ps = S * gi + f0_v;
                        dma4(rx, stage + TA, (live && (unsigned)ps < (unsigned)p.Lx) ? smp * xbs4 + f0_off + ps * 4 : FAR);
                    } else dma16s(rp, stage + wv * 256, live ? pv[0] : FAR, smp * pbs4 + gi * 4);     // (the other waves: their first gather again)
                }
            } else if (e < NDL) {
                const int b0 = 16 * s.g, il = 16 * cf + s.chk;          // samples b0 ... b0 + 15 at frame il
                if (e < 16) {
                    const bool ok = live && la_row + 16 * e < p.M;
                    dma4s(rp, stage + (4 * e + wv) * 64, ok ? la_off : FAR, e * 16 * p.LP * 4 + b0 * pbs4 + il * 4);
                } else {
                    const int ps = S * il + lb_pj;
                    dma4s(rx, stage + TA + (4 * (e - 16) + wv) * 64, (live && (unsigned)ps < (unsigned)p.Lx) ? lb_ch + ps * 4 : FAR,
                          (b0 + e - 16) * xbs4);
                }
            }
        };
        // what the fix-ups of fragments read from the slab at `s` need
        auto fix_of = [&](const G3Pos& s) {
            G3Fix fx;
            fx.left = PAD ? 0 : s.left;
            fx.w0 = S * (s.chk << 4);
            fx.kc = p.LP - (s.chk << 4);
            fx.edge = fx.w0 < p.p || fx.w0 - p.p + C::WLP > p.Lx;
            return fx;
        };
        G3Pos s0 = g3_pos_of<PAD>(sb, cf, rem), s1 = s0;
        g3_advance<PAD>(s1, cf, rem);
        G3Pos s2 = s1;
        g3_advance<PAD>(s2, cf, rem);
        // prefill: slabs sb and sb + 1
        // (all of slab sb's gathers before any of slab sb + 1's: the counted wait below relies on the order; a main slab ignores
        // e >= its own count -- the loop bound is the larger of the two slab types)
#pragma unroll
        for (int e = 0; e < (PAD ? NDM : NDL); ++e) issue_piece(lds, s0, true, e);
#pragma unroll
        for (int e = 0; e < (PAD ? NDM : NDL); ++e) issue_piece(lds + STG, s1, sb + 1 < se, e);
        __builtin_amdgcn_sched_barrier(0);
        if (!PAD && s1.left) g3_wait_vmcnt<NDL>(); else g3_wait_vmcnt<NDM>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        G3Frag f0, f1;
        {
            const unsigned a0 = g3_a_addr(lds, 0, r, h);
            const G3Fix fx = fix_of(s0);
            if (fx.left) {
                const unsigned b0 = g3_lds_addr(lds + TA) + bcol[0] * 4, b1 = g3_lds_addr(lds + TA) + bcol[1] * 4;
                g3_load_b_left<0, 0>(b0, f0); g3_load_b_left<0, 1>(b0, f0); g3_load_b_left<1, 0>(b1, f0); g3_load_b_left<1, 1>(b1, f0);
            } else {
                const unsigned b0 = g3_lds_addr(lds + TA) + bbase[0] * 4, b1 = g3_lds_addr(lds + TA) + bbase[1] * 4;
                g3_load_b_main<S, 0, 0>(b0, f0); g3_load_b_main<S, 0, 1>(b0, f0); g3_load_b_main<S, 1, 0>(b1, f0); g3_load_b_main<S, 1, 1>(b1, f0);
            }
            g3_load_a<0>(a0, f0); g3_load_a<1>(a0, f0); g3_load_a<2>(a0, f0); g3_load_a<3>(a0, f0);
            g3_load_a<4>(a0, f0); g3_load_a<5>(a0, f0); g3_load_a<6>(a0, f0); g3_load_a<7>(a0, f0);
            g3_lgkm0();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    float v = f0.a[i][kk];
                    if (PAD) v = (8 * h + kk < fx.kc) ? v : 0.f;
                    f0.a[i][kk] = v;
                }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    float v = __builtin_bit_cast(float, f0.d[jb][kk]);
                    if (!fx.left) v = (unsigned)(fx.w0 + pj[jb] + S * kk) < (unsigned)p.Lx ? v : 0.f;
                    f0.b[jb][kk] = v;
                }
        }
        int st = 0;
        for (int sl = sb; sl < se; ++sl) {
            const int st1 = st == 2 ? 0 : st + 1, st2 = st >= 1 ? st - 1 : 2;
            const float* const cur = lds + st * STG;
            const float* const nxs = lds + st1 * STG;
            float* const ring2 = lds + st2 * STG;
            const bool live2 = sl + 2 < se;
            const bool left2 = !PAD && s2.left;
            auto issue0 = [&](int e) { if (e < (left2 ? NDL : NDM)) issue_piece(ring2, s2, live2, e); };
            // first half: MFMAs of (slab sl, c = 0); reads of (slab sl, c = 1)
            {
                const G3Fix fx = fix_of(s0);
                const unsigned a0 = g3_a_addr(cur, 1, r, h);
                const unsigned bb = g3_lds_addr(cur + TA) + (fx.left ? 4 * G3_TN * 4 : S * 4 * 4);
                g3_half<S, PAD, 0>(f0, f1, a0, bb + (fx.left ? bcol[0] : bbase[0]) * 4, bb + (fx.left ? bcol[1] : bbase[1]) * 4, 1, h, fx, pj, p.Lx,
                                   acc, issue0);
            }
            // In front of the second half, whose gaps carry the reads of slab sl + 1's first fragments: of slab sl + 2's gathers this
            // wave has issued min(26, its count) -- they may stay in flight, everything older (slab sl + 1) is done.  The barrier also
            // orders the ring (conv_raw3.hip).
            if (left2) g3_wait_vmcnt<G3_SLOTS>(); else g3_wait_vmcnt<NDM>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // second half: MFMAs of (slab sl, c = 1); reads of (slab sl + 1, c = 0); the rest of a leftover slab's gathers
            {
                const G3Fix fx = fix_of(s1);
                const unsigned a0 = g3_a_addr(nxs, 0, r, h);
                const unsigned bb = g3_lds_addr(nxs + TA);
                g3_half<S, PAD, G3_SLOTS>(f1, f0, a0, bb + (fx.left ? bcol[0] : bbase[0]) * 4, bb + (fx.left ? bcol[1] : bbase[1]) * 4, 0, h, fx, pj,
                                          p.Lx, acc, issue0);
            }
            s0 = s1; s1 = s2;
            g3_advance<PAD>(s2, cf, rem);
            st = st1;
        }
        __syncthreads();
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");      // MFMA result -> asm accumulator reads (conv_raw3.hip)
        if (sb == 0 && se == p.nslab) {
#define G3_EPI(I, J)                                                                                              \
    {   AccT<1, 1> blk;                                                                                          \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) blk.c[0][0][q] = g3_acc(acc.c[I][J][q]);                  \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        epilogue_g<S, 1, 1>(p, blk, m0 + (I) * 32, n0 + (wn * 2 + (J)) * 32, lane, 0, 0, n0 + G3_TN);            \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    }
            G3_EPI(0, 0) G3_EPI(0, 1) G3_EPI(1, 0) G3_EPI(1, 1) G3_EPI(2, 0) G3_EPI(2, 1) G3_EPI(3, 0) G3_EPI(3, 1)
            G3_EPI(4, 0) G3_EPI(4, 1) G3_EPI(5, 0) G3_EPI(5, 1) G3_EPI(6, 0) G3_EPI(6, 1) G3_EPI(7, 0) G3_EPI(7, 1)
#undef G3_EPI
        } else store_partial_g3(p.ws, g, slot, acc, tid);
        pos += se - sb;
        slot = 1;
    }
}

// fixup of the stream-K split: one workgroup per (split tile, 32 x 32 block of the wave tile)
__global__ __launch_bounds__(NT3) void conv_g3_fixup_kernel(const IgemmParams p, int G) {
    constexpr int MB = 8, NB = 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = p.whole + blockIdx.x / (MB * NB), blk = blockIdx.x % (MB * NB), bi = blk / NB, bj = blk - bi * NB;
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, G, p.whole);
    const int first = tile * p.nslab, last = first + p.nslab - 1;
    const int g0 = split_owner(sp, first), g1 = split_owner(sp, last);
    if (g0 == g1 && split_lo(sp, g0) <= first && split_lo(sp, g0 + 1) > last) return;
    AccT<1, 1> acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.c[0][0][r] = 0.f;
    for (int g = g0; g <= g1; ++g) {
        const int slot = (split_lo(sp, g) / p.nslab == tile) ? 0 : 1;
        const float* src = p.ws + ((long)(g * 2 + slot) * G3_REGS) * NT3 + tid;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.c[0][0][r] += src[(blk * 16 + r) * NT3];
    }
    const int nt0 = (tile % p.tilesN) * p.tn_stride;
    epilogue_g<0, 1, 1>(p, acc, (tile / p.tilesN) * G3_TM + bi * 32, nt0 + (wv * NB + bj) * 32, lane, 0, 0, nt0 + p.tn_stride);
}

template <int KW, int S, bool PAD>
hipError_t launch_g3(const IgemmParams& p, int grid, hipStream_t st) {
    using C = G3Geo<KW, S>;
    constexpr int BREG = (!PAD && C::SUBR < 16 * G3_TN) ? 16 * G3_TN : C::SUBR;
    constexpr int lds_bytes = G3_RING * (G3_TM * BK + BREG) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)conv_g3_kernel<KW, S, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((conv_g3_kernel<KW, S, PAD>), dim3(grid), dim3(NT3), lds_bytes, st, p);
    return hipGetLastError();
}

}  // namespace

// mode 1: PAD (any batch size), mode 2: leftover slabs (B a multiple of 16)
hipError_t pgconv::launch_g3(const IgemmParams& p, int mode, int grid, hipStream_t st) {
    const bool pad = mode == 1;
    if (p.k == 32) return pad ? ::launch_g3<32, 2, true>(p, grid, st) : ::launch_g3<32, 2, false>(p, grid, st);
    if (p.k == 8 && p.s == 1) return pad ? ::launch_g3<8, 1, true>(p, grid, st) : ::launch_g3<8, 1, false>(p, grid, st);
    if (p.k == 8) return pad ? ::launch_g3<8, 2, true>(p, grid, st) : ::launch_g3<8, 2, false>(p, grid, st);
    return pad ? ::launch_g3<4, 2, true>(p, grid, st) : ::launch_g3<4, 2, false>(p, grid, st);
}

hipError_t pgconv::launch_g3_fixup(const IgemmParams& p, int grid, unsigned blocks, hipStream_t st) {
    hipLaunchKernelGGL(conv_g3_fixup_kernel, dim3(blocks), dim3(NT3), 0, st, p, grid);
    return hipGetLastError();
}
