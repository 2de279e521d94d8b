// conv_raw3.hip -- fp32 raw-window F and T kernels (conv fwd / convT dgrad; convT fwd / conv dgrad: operands, window layout and weight
// images exactly as conv_raw_impl.h) with ONE wave per SIMD: 256 threads = 4 waves, one workgroup per CU, each wave a 256 x 64
// sub-tile (8 x 2 blocks of 32 x 32: 256 accumulator registers, held in AGPRs), workgroup tile 256 x 256.
//
// Why: the two-waves-per-SIMD kernels sit at 0.83-0.91 of the fp32 MFMA peak because what ONE wave
// issues outside its MFMA bursts per slab -- fragment reads, gathers, the barrier with its LDS-DMA drain -- is covered by the partner
// wave only in part (utilisation = B / (B + t_n)).  A register-only loop of v_mfma_f32_32x32x2_f32 sustains 0.984 of the peak on
// random operands (tools/mfma_peak.py; unlike the bf16 pipe there is no power limit here), so the headroom is real.  Here a wave
// never leaves its burst: a v_mfma_f32_32x32x2_f32 holds the issue port for 8 of its 64 cycles, and every LDS read, every gather
// and every funnel of the next half-slab is PINNED into one of those gaps (one piece of work per MFMA, scheduling fences between:
// the structure of conv_h3.hip, whose header lists what it takes to make hipcc emit it).  Stage ring of three slabs, gathers two
// slabs ahead behind a counted vmcnt, one raw s_barrier per slab (8192 MFMA cycles) placed in front of the slab's second half, whose
// gaps carry the reads of the next slab's first fragments.
//
// Covered: fp32 operands (precision 0), F: k = 32 / 8 / 4 at stride 2 and k = 8 at stride 1; T: k = 32 / 8 / 4 at stride 2 (phase-major
// weight image), k = 8 at stride 1, and k = 5 at stride 2 in both forms (a virtual k = 8); whole 16-deep slabs only (Cin a multiple of
// 16 / taps-per-channel).  Everything else -- the bf16 operand modes, K tails, small problems -- stays on conv_raw.hip / conv_raw_tall.hip / conv_im2col.hip.
#include "conv_common.h"

namespace {

constexpr int NT3 = 256;                  // threads per workgroup
constexpr int R3_RING = 3;
constexpr int R3_REGS = 256;              // accumulator registers per thread: 16 blocks x 16
constexpr int R3_TM = 256, R3_TN = 256;
constexpr int R3_SLOTS = 12;              // gather slots per half-slab (behind MFMAs 12, 16, ... 56 of its 64)

typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ void r3_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void r3_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }      // lgkmcnt(0), the other counters untouched
__device__ __forceinline__ unsigned r3_lds_addr(const float* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
// an accumulator register read where it is USED (conv_h3.hip: h3_acc)
__device__ __forceinline__ float r3_acc(float v) {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(v));
    return x;
}

// ---- fragments of one HALF-slab: k = 8 h + 4 c + 0..3 of a 16-deep slab (lane half h, half-slab c) ---------------------------
// A (weight tile): eight ds_read_b128.  Plain image (F, T at stride 1): 16-byte group (2 h + c) of row r of each of the 8 row blocks,
// element kk of block i = a[i][kk].  Phase-major image (T at stride 2; rows = output channels, 32 floats = 16 k x 2 phases
// interleaved): groups 4 h + 2 c and + 1 of row r of each of the 4 channel blocks; row block 2 ob + phi (phase phi of channel block
// ob) takes element e = 2 kk + phi of the pair: a[2 ob + (e >> 2)][e & 3].
// B (window): four consecutive dwords of the lane's column in each of the 2 column blocks (two ds_read2_b32), taps ascending (F) or
// descending (T) -- conv_raw_impl.h: raw_load_frags.
// All reads are `asm volatile` (hipcc does not order plain LDS loads against scheduling fences) and are waited for by the loop itself.
struct R3Frag { f32x4 a[8]; unsigned d[2][4]; float b[2][4]; };

template <bool PM> __device__ __forceinline__ unsigned r3_a_addr(const float* stage, int c, int r, int h) {
    if (PM) return r3_lds_addr(stage) + (r * 32 + (((4 * h + 2 * c) ^ ((r >> 1) & 7)) << 2)) * 4;
    return r3_lds_addr(stage) + (r * 16 + (((2 * h + c) ^ ((r >> 2) & 3)) << 2)) * 4;
}
// PM: the pair's second group is the first XOR 1 -> its address differs by +-16 bytes depending on the lane: a second register
template <bool PM> __device__ __forceinline__ unsigned r3_a_addr2(const float* stage, int c, int r, int h) {
    return r3_lds_addr(stage) + (r * 32 + (((4 * h + 2 * c + 1) ^ ((r >> 1) & 7)) << 2)) * 4;
}
template <bool PM, int I> __device__ __forceinline__ void r3_load_a(unsigned a0, unsigned a1, R3Frag& f) {
    if (PM) {       // I = 2 ob + t: group t of channel block ob (32 rows x 128 bytes = 4096 bytes per block)
        if (I & 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a[I]) : "v"(a1), "n"((I >> 1) * 4096));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a[I]) : "v"(a0), "n"((I >> 1) * 4096));
    } else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a[I]) : "v"(a0), "n"(I * 2048));
}
template <bool PM> __device__ __forceinline__ float r3_a_elem(const R3Frag& f, int i, int kk) {
    if (PM) { const int e = 2 * kk + (i & 1); return f.a[(i & ~1) + (e >> 2)][e & 3]; }
    return f.a[i][kk];
}
// byte address of the first of the four dwords of column block jb in half-slab c.  TJ = taps per channel and slab (16: one channel,
// lane half h takes taps 8 h ..; 8: channel h; 4: channels 2 h + c; 2: channels 4 h + 2 c, + 1), DESC: taps descend (stored ascending from
// base - (TD - 1)).
template <int TJ, bool DESC, int RS>
__device__ __forceinline__ unsigned r3_b_addr(const float* bw, int c, int h, int bbase_jb) {
    constexpr int TD = TJ < 8 ? TJ : 8;
    const int lanepart = (TJ == 16) ? (DESC ? -8 * h : 8 * h) : (8 / TJ) * h * RS;
    // TJ = 4: the half-slab is one channel's four taps; TJ = 2: two channels' two taps each (the second RS floats further: r3_load_b)
    const int so = TJ >= 8 ? (DESC ? 4 - 4 * c : 4 * c) : (TJ == 4 ? c * RS : 2 * c * RS);
    return r3_lds_addr(bw) + (bbase_jb + lanepart - (DESC ? TD - 1 : 0) + so) * 4;
}
// W2 = byte distance of the pair's second read: 0 -> dwords 2, 3 of the same run; else (TJ = 2) dwords 0, 1 of the next channel's window
template <int JB, int P, int W2 = 0> __device__ __forceinline__ void r3_load_b(unsigned b_addr, R3Frag& f) {
    u32x2v t;
    if (P == 0) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(t) : "v"(b_addr));
    else if (W2 == 0) asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(t) : "v"(b_addr));
    else asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(t) : "v"(b_addr + W2));
    f.d[JB][2 * P] = t[0]; f.d[JB][2 * P + 1] = t[1];
}
// window dwords -> B fragment elements (tap order); ACT: with the input activation max(v, slope v).  The engine's own calls store
// activated tensors and load them as they are: that instantiation carries no VALU here (24 instructions per half-slab otherwise --
// fix-ups cost MFMA issue slots even inside the gaps, conv_g3.hip).
template <bool DESC, bool ACT, bool TJ2 = false> __device__ __forceinline__ void r3_finish_b(R3Frag& f, float slope) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int i = DESC ? (TJ2 ? kk ^ 1 : 3 - kk) : kk;        // (two taps per channel: the descending order is per channel)
            const float v = __builtin_bit_cast(float, f.d[jb][i]);
            f.b[jb][kk] = ACT ? fmaxf(v, slope * v) : v;
        }
}

// k = 5 at stride 2 runs as a VIRTUAL k = 8 (conv_raw_impl.h): the weight tile is read as if every (row, channel) had 8 taps -- taps 5, 6, 7
// are the next weight row's first floats -- and the MFMAs whose k positions hold a virtual tap in BOTH lane halves are never issued:
// K5 = 2 (F form: tap = k index 0 ... 7 of the lane half): kk = 5, 6, 7; K5 = 1 (T form, phase-major rows: taps 2 jj + phi, jj = kk & 3):
// phase 0: jj = 3, phase 1: jj = 2, 3.  MFMA work stays the algorithmic 5 / 8 of the virtual problem.
template <int K5> __device__ __forceinline__ constexpr bool r3_k5_virtual(int i, int kk) {
    return K5 == 1 ? ((i & 1) == 0 ? (kk & 3) == 3 : (kk & 3) >= 2) : (K5 == 2 ? kk >= 5 : false);
}

// One half-slab: the 64 MFMAs of `cur` (4 k x 8 row blocks x 2 column blocks) and in their gaps -- one piece per MFMA, pinned by
// scheduling fences -- the reads of the NEXT half-slab's fragments `nxt` (window dwords behind MFMAs 0-3, weight rows behind 4-11),
// gathers of the slab two ahead behind MFMAs 12, 16, ... 56 (`issue(E0 + n)`), the wait for the reads behind 60 and the activation
// of the window values behind 61.
template <int TJ, bool DESC, int RS, bool PM, bool ACT, int K5, int CC, int E0, typename Issue>
__device__ __forceinline__ void r3_half(const R3Frag& cur, R3Frag& nxt, const float* nstage, int nc, int TA_, int r, int h,
                                        const int (&bbase)[2], float slope, AccT<8, 2>& acc, const Issue& issue) {
    const unsigned a0 = r3_a_addr<PM>(nstage, nc, r, h), a1 = PM ? r3_a_addr2<PM>(nstage, nc, r, h) : 0u;
    const unsigned b0 = r3_b_addr<TJ, DESC, RS>(nstage + TA_, nc, h, bbase[0]), b1 = r3_b_addr<TJ, DESC, RS>(nstage + TA_, nc, h, bbase[1]);
#define R3_CHUNK(C, WORK)                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                           \
    if (!r3_k5_virtual<K5>(((C) >> 1) & 7, 4 * CC + ((C) >> 4)))         /* CC: which half of the slab `cur` is (k = 8 h + 4 CC + ..) */ \
        acc.c[((C) >> 1) & 7][(C) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(r3_a_elem<PM>(cur, ((C) >> 1) & 7, (C) >> 4),      \
                                                                             cur.b[(C) & 1][(C) >> 4], acc.c[((C) >> 1) & 7][(C) & 1], 0, 0, 0); \
    WORK;
#define R3_ROW(C0, W0, W1, W2, W3) R3_CHUNK(C0, W0) R3_CHUNK(C0 + 1, W1) R3_CHUNK(C0 + 2, W2) R3_CHUNK(C0 + 3, W3)
#if defined(PG_R3_ABL) && (PG_R3_ABL == 4 || PG_R3_ABL == 5 || PG_R3_ABL == 7)
    (void)a0; (void)a1; (void)b0; (void)b1;
    R3_ROW(0, (void)0, (void)0, (void)0, (void)0)
    R3_ROW(4, (void)0, (void)0, (void)0, (void)0)
    R3_ROW(8, (void)0, (void)0, (void)0, (void)0)
#else
    constexpr int W2 = TJ == 2 ? RS * 4 : 0;
    R3_ROW(0, (r3_load_b<0, 0, W2>(b0, nxt)), (r3_load_b<0, 1, W2>(b0, nxt)), (r3_load_b<1, 0, W2>(b1, nxt)), (r3_load_b<1, 1, W2>(b1, nxt)))
    R3_ROW(4, (r3_load_a<PM, 0>(a0, a1, nxt)), (r3_load_a<PM, 1>(a0, a1, nxt)), (r3_load_a<PM, 2>(a0, a1, nxt)), (r3_load_a<PM, 3>(a0, a1, nxt)))
    R3_ROW(8, (r3_load_a<PM, 4>(a0, a1, nxt)), (r3_load_a<PM, 5>(a0, a1, nxt)), (r3_load_a<PM, 6>(a0, a1, nxt)), (r3_load_a<PM, 7>(a0, a1, nxt)))
#endif
    R3_ROW(12, issue(E0 + 0), (void)0, (void)0, (void)0)
    R3_ROW(16, issue(E0 + 1), (void)0, (void)0, (void)0)
    R3_ROW(20, issue(E0 + 2), (void)0, (void)0, (void)0)
    R3_ROW(24, issue(E0 + 3), (void)0, (void)0, (void)0)
    R3_ROW(28, issue(E0 + 4), (void)0, (void)0, (void)0)
    R3_ROW(32, issue(E0 + 5), (void)0, (void)0, (void)0)
    R3_ROW(36, issue(E0 + 6), (void)0, (void)0, (void)0)
    R3_ROW(40, issue(E0 + 7), (void)0, (void)0, (void)0)
    R3_ROW(44, issue(E0 + 8), (void)0, (void)0, (void)0)
    R3_ROW(48, issue(E0 + 9), (void)0, (void)0, (void)0)
    R3_ROW(52, issue(E0 + 10), (void)0, (void)0, (void)0)
    R3_ROW(56, issue(E0 + 11), (void)0, (void)0, (void)0)
#if defined(PG_R3_ABL) && PG_R3_ABL >= 4
    R3_CHUNK(60, (void)0)
    R3_CHUNK(61, (void)0)
    nxt = cur;
#else
    R3_CHUNK(60, r3_lgkm0())
    __builtin_amdgcn_sched_barrier(0);
    R3_CHUNK(61, (r3_finish_b<DESC, ACT, TJ == 2>(nxt, slope)))
#endif
    R3_CHUNK(62, (void)0)
    R3_CHUNK(63, (void)0)
    __builtin_amdgcn_sched_barrier(0);
#undef R3_ROW
#undef R3_CHUNK
}

__device__ __forceinline__ void store_partial_r3(float* ws, int g, int slot, const AccT<8, 2>& acc, int tid) {
    float* dst = ws + ((long)(g * 2 + slot) * R3_REGS) * NT3 + tid;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((i * 2 + j) * 16 + r) * NT3] = r3_acc(acc.c[i][j][r]);
}

// TKIND false: F (conv fwd / convT dgrad);  TKIND true: T (convT fwd / conv dgrad in gather form)
template <int KW, int S, bool TKIND, bool ACT>
__global__ __launch_bounds__(NT3, 1) void conv_raw3_kernel(const IgemmParams p) {
    constexpr int TM = R3_TM, TN = R3_TN;
    constexpr int TA = TM * BK;                       // floats of the weight tile (16 KB)
    constexpr int AE16 = TM / 64;                     // 16-byte gather instructions per wave for the weight tile
    constexpr int KWV = KW == 5 ? 8 : KW;             // taps per (row, channel) of the weight image: k = 5 is a virtual 8
    constexpr int K5 = KW == 5 ? (TKIND ? 1 : 2) : 0;
    constexpr int KWP = TKIND ? KWV / S : KWV;        // taps per channel in K order
    static_assert(KW != 5 || S == 2, "k = 5 is built for stride 2 only");
    constexpr int TJ = KWP < 16 ? KWP : 16, NQ = 16 / TJ;
    constexpr int SC = TKIND ? 1 : S;
    constexpr int RG = raw_gap(TJ);
    constexpr bool PM = TKIND && S == 2;              // phase-major weight image
    constexpr bool T16 = TKIND && S == 1;             // taps of a channel contiguous in memory and in K
    constexpr int RS = SC == 1 ? RS1 : RS2;           // floats reserved per channel window
    constexpr int NPC = (RS + NT3 - 1) / NT3;         // gather pieces per thread and window
    constexpr int NT0 = (!TKIND && KWP == 32) ? 2 : 1;   // F with 32-tap channels: a channel spans two slabs (tap offset 0 / 16)
    constexpr int STG = TA + NQ * RS;                 // floats per LDS stage (one slab)
    constexpr int ND = AE16 + NQ * NPC;               // gathers per wave and slab
    constexpr int ND0 = ND < R3_SLOTS ? ND : R3_SLOTS;   // ... of them issued in the slab's first half
    static_assert(TJ == 2 || TJ == 4 || TJ == 8 || TJ == 16, "taps per channel and slab");
    static_assert(ND <= 2 * R3_SLOTS && ND < 32, "gather slots per slab / vmcnt range");
    static_assert(R3_RING * STG * 4 <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wv;
    const int r = lane & 31, h = lane >> 5;
#if PG_ABL == 8   /* dev-only: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (tools/clock_probe.py) */
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int Lcol = TKIND ? p.U : p.Ly;              // columns (output positions) per sample
    const int Ktot = p.Q * KWP, Mrows = TKIND ? p.M * S : p.M;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);
    const float slopeB = act_slope(p.act_x);
    const int wq = p.M * KW;                          // T: weight stride between input channels
    const int pm_f = ((lane & 7) ^ (((wv & 1) << 2) | (lane >> 4))) << 2;      // conv_raw_impl.h: the logical chunk this lane carries
    const int pm_ql = pm_f / (2 * TJ), pm_within = pm_f - pm_ql * 2 * TJ;
    const int g = logical_wg(blockIdx.x, gridDim.x, p.whole);
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x, p.whole);
    int pos = split_lo(sp, g);
    const int pos_end = split_lo(sp, g + 1);
    int slot = 0;
    while (pos < pos_end) {
        const int tile = pos / p.nslab, sb = pos - tile * p.nslab;
        const int se = min(p.nslab, sb + (pos_end - pos));
        int tmi, tni;
        tile_decode(p, tile, tmi, tni);
        const int m0 = tmi * TM, n0 = tni * TN;
        const int b0 = n0 / Lcol, t0 = n0 - b0 * Lcol;
        const int nseg = (t0 + TN - 1) / Lcol + 1;

        // --- weight-tile gather offsets (bytes): per lane, fixed for the tile; the slab rides in the SGPR offset ------------------
        int avoff[AE16];
#pragma unroll
        for (int e = 0; e < AE16; ++e) {
            if (PM) {
                const int o = m0 / 2 + (4 * e + wv) * 8 + (lane >> 3);
                avoff[e] = o < p.M ? (pm_ql * wq + o * KW + pm_within) * 4 : FAR;
            } else if (T16) {
                const int o = m0 + dma16_row(lane, wv, e), kc = dma16_kc(lane);
                avoff[e] = o < Mrows ? ((kc / KWP) * wq + o * KW + (kc % KWP)) * 4 : FAR;
            } else {
                const int m = m0 + dma16_row(lane, wv, e);
                const int kc = dma16_kc(lane);      // k = 5: chunk kc = floats (kc & 7) .. + 3 of channel kc >> 3, rows 5 Q floats apart
                avoff[e] = m < p.M ? (K5 ? (m * p.Q * 5 + (kc >> 3) * 5 + (kc & 7)) * 4 : (m * Ktot + kc) * 4) : FAR;
            }
        }
        // --- window gather offsets: thread owns window positions v = tid + 256 e (conv_raw_impl.h) --------------------------------
        int voff0[NPC], voff1[NPC];                  // tap offset 0 / 16 of the slab (voff1: F with 32-tap channels only)
#pragma unroll
        for (int e = 0; e < NPC; ++e) {
            const int v = tid + 256 * e;
            int k = 0;
            while (k + 1 < nseg && SC * ((k + 1) * Lcol - t0) + RG * (k + 1) <= v) ++k;
            const int cs = k ? k * Lcol - t0 : 0;
            const int vl = v - (SC * cs + RG * k);
            const int tf = k ? 0 : t0;
            const int b = b0 + k;
            const int posb = b < p.B ? (TKIND ? p.u_off + tf - (TJ - 1) + vl : S * tf - p.p + vl) : -NEVER;
            const int rowb = b * (int)p.x_bs * 4;
            voff0[e] = (unsigned)posb < (unsigned)p.Lx ? rowb + posb * 4 : FAR;
            voff1[e] = (NT0 == 2 && (unsigned)(posb + 16) < (unsigned)p.Lx) ? rowb + (posb + 16) * 4 : FAR;
        }
        int bbase[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int c = wn * 64 + jb * 32 + r;
            bbase[jb] = SC * c + RG * ((t0 + c) / Lcol) + (TKIND ? TJ - 1 : 0);
        }
        AccT<8, 2> acc;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc.c[i][j][q] = 0.f;

        // gather number e of slab `slab` into its stage: e < AE16 the weight rows, then window piece (e - AE16) % NPC of channel
        // (e - AE16) / NPC.  A slab past this workgroup's K range is gathered from out-of-range offsets: zeros, no traffic.
        auto issue_piece = [&](float* stage, int slab, int e) {
#ifdef PG_R3_ABL          // dev ablation (wrong results): 1 = no gathers inside the loop, 2 = no window gathers, 3 = no weight gathers,
                          // 4 = 1 + no LDS reads in the loop, 5 = 4 + no barrier, 6 = 1 + no activation VALU
            if (slab >= sb + 2 && (PG_R3_ABL == 1 || PG_R3_ABL >= 4 || (PG_R3_ABL == 2 && e >= AE16) || (PG_R3_ABL == 3 && e < AE16))) return;
#endif
            const bool live = slab < se;
            const int k0 = slab * BK;
            if (e < AE16) {
                const int sa = (PM || T16) ? (k0 / KWP) * wq * 4 : (K5 ? (k0 >> 3) * 20 : k0 * 4);
                dma16s(rw, stage + (4 * e + wv) * 256, live ? avoff[e < AE16 ? e : 0] : FAR, sa);
            } else if (e < ND) {
                const int x = e - AE16, qi = x / NPC, pe0 = x - qi * NPC;
                // a piece past the window's RS floats (RS = 384: the second piece of waves 2 and 3) must not be written -- it would
                // land in the next channel's window; such a wave repeats its piece 0 instead (same bytes to the same place), so that
                // every wave issues the same number of gathers and one vmcnt immediate fits all four
                const int pe = (e < AE16 || pe0 * 256 + wv * 64 < RS) ? pe0 : 0;
                const int fq0 = k0 / KWP, ft0 = k0 - fq0 * KWP;      // ft0 = 16 only for the odd slabs of 32-tap channels
                int vo = voff0[0];
#pragma unroll
                for (int q = 0; q < NPC; ++q) {
                    const int v0 = voff0[q], v1 = voff1[q];
                    if (q == pe) vo = (NT0 == 2 && ft0) ? v1 : v0;
                }
                dma4s(rx, stage + TA + qi * RS + pe * 256 + wv * 64, live ? vo : FAR, (fq0 + qi) * p.Lx * 4);
            }
        };
        // wait_first: before the first read of a segment (everything but the youngest slab has landed).  wait_next: in front of the
        // second half of slab i, whose gaps carry the reads of slab i + 1's first fragments: of slab i + 2's gathers this wave has
        // issued ND0 by then -- they may stay in flight, everything older (slab i + 1) is done.  The same barrier orders the ring:
        // slab i + 2 is gathered into slab i - 1's stage from slab i's first half on, and the last reads of that stage (slab i - 1's
        // second-half fragments) were issued and waited for in slab i - 1's FIRST half, in front of slab i - 1's barrier.
        auto wait_first = [&]() {
            r3_wait_vmcnt<ND>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto wait_next = [&]() {
#if defined(PG_R3_DBG) || defined(PG_R3_ABL)
            r3_wait_vmcnt<0>();
#else
            r3_wait_vmcnt<ND0>();
#endif
#if !(defined(PG_R3_ABL) && PG_R3_ABL == 5)
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
        };
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int e = 0; e < ND; ++e) issue_piece(lds + hf * STG, sb + hf, e);
        __builtin_amdgcn_sched_barrier(0);
        wait_first();
        R3Frag f0, f1;
        {
            const unsigned a0 = r3_a_addr<PM>(lds, 0, r, h), a1 = PM ? r3_a_addr2<PM>(lds, 0, r, h) : 0u;
            const unsigned b0a = r3_b_addr<TJ, TKIND, RS>(lds + TA, 0, h, bbase[0]), b1a = r3_b_addr<TJ, TKIND, RS>(lds + TA, 0, h, bbase[1]);
            constexpr int W2 = TJ == 2 ? RS * 4 : 0;
            r3_load_b<0, 0, W2>(b0a, f0); r3_load_b<0, 1, W2>(b0a, f0); r3_load_b<1, 0, W2>(b1a, f0); r3_load_b<1, 1, W2>(b1a, f0);
            r3_load_a<PM, 0>(a0, a1, f0); r3_load_a<PM, 1>(a0, a1, f0); r3_load_a<PM, 2>(a0, a1, f0); r3_load_a<PM, 3>(a0, a1, f0);
            r3_load_a<PM, 4>(a0, a1, f0); r3_load_a<PM, 5>(a0, a1, f0); r3_load_a<PM, 6>(a0, a1, f0); r3_load_a<PM, 7>(a0, a1, f0);
            r3_lgkm0();
            __builtin_amdgcn_sched_barrier(0);
            r3_finish_b<TKIND, ACT, TJ == 2>(f0, slopeB);
        }
        int st = 0;
        for (int sl = sb; sl < se; ++sl) {
            const int st1 = st == 2 ? 0 : st + 1, st2 = st >= 1 ? st - 1 : 2;                // (st + 1) % 3, (st + 2) % 3
            const float* const cur = lds + st * STG;
            float* const ring2 = lds + st2 * STG;
            const int s2 = sl + 2;
            auto issue0 = [&](int e) { if (e < ND0) issue_piece(ring2, s2, e); };             // first half: gathers 0 ... ND0 - 1
            auto issue1 = [&](int e) { if (e < ND) issue_piece(ring2, s2, e); };              // second half: the rest (E0 = ND0)
            r3_half<TJ, TKIND, RS, PM, ACT, K5, 0, 0>(f0, f1, cur, 1, TA, r, h, bbase, slopeB, acc, issue0);
            // (past the last slab of the segment the "next" stage holds zero-filled or older slabs: read, never multiplied; f0 is
            // carried by the loop, so its registers stay reserved until the reads have landed -- conv_h3.hip on dead asm reads)
            wait_next();
            r3_half<TJ, TKIND, RS, PM, ACT, K5, 1, ND0>(f1, f0, lds + st1 * STG, 0, TA, r, h, bbase, slopeB, acc, issue1);
            st = st1;
        }
        __syncthreads();
        // the accumulator reads below are `asm`: the wait states an MFMA result needs before a VALU may read it, spelled out
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
#if defined(PG_R3_ABL) && PG_R3_ABL == 7      /* 4 + no epilogue: one store per wave keeps the loop alive */
        if (p.nslab < 0) store_partial_r3(p.ws, g, slot, acc, tid);
        else if (lane == 0) p.y[wv] = r3_acc(acc.c[0][0][0]);
        pos += se - sb; slot = 1; continue;
#endif
        if (sb == 0 && se == p.nslab) {
#define R3_EPI(I, J)                                                                                              \
    {   AccT<1, 1> blk;                                                                                          \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) blk.c[0][0][q] = r3_acc(acc.c[I][J][q]);                  \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (PM) epilogue_t_pm<1, 1>(p, blk, m0 / 2 + ((I) >> 1) * 32, n0 + (wn * 2 + (J)) * 32, lane, (I) & 1);  \
        else if (TKIND) epilogue_t<S, 1, 1>(p, blk, m0 + (I) * 32, n0 + (wn * 2 + (J)) * 32, lane, 0, 0);        \
        else epilogue_f<S, 1, 1>(p, blk, m0 + (I) * 32, n0 + (wn * 2 + (J)) * 32, lane, 0, 0);                   \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    }
            R3_EPI(0, 0) R3_EPI(0, 1) R3_EPI(1, 0) R3_EPI(1, 1) R3_EPI(2, 0) R3_EPI(2, 1) R3_EPI(3, 0) R3_EPI(3, 1)
            R3_EPI(4, 0) R3_EPI(4, 1) R3_EPI(5, 0) R3_EPI(5, 1) R3_EPI(6, 0) R3_EPI(6, 1) R3_EPI(7, 0) R3_EPI(7, 1)
#undef R3_EPI
        } else store_partial_r3(p.ws, g, slot, acc, tid);
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

// fixup of the stream-K split: one workgroup per (split tile, 32 x 32 block of the wave tile: 16 of them).
// KIND 0: F, 1: T at stride 1, 3: T at stride 2 (phase-major rows: block row bi = phase bi & 1 of channel block bi >> 1)
template <int KIND>
__global__ __launch_bounds__(NT3) void conv_raw3_fixup_kernel(const IgemmParams p, int G) {
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
        const float* src = p.ws + ((long)(g * 2 + slot) * R3_REGS) * NT3 + tid;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.c[0][0][r] += src[(blk * 16 + r) * NT3];
    }
    int tmi, tni;
    tile_decode(p, tile, tmi, tni);
    const int mt = tmi * R3_TM, n0 = tni * p.tn_stride + (wv * NB + bj) * 32;
    if (KIND == 0) epilogue_f<0, 1, 1>(p, acc, mt + bi * 32, n0, lane, 0, 0);
    else if (KIND == 1) epilogue_t<0, 1, 1>(p, acc, mt + bi * 32, n0, lane, 0, 0);
    else epilogue_t_pm<1, 1>(p, acc, mt / 2 + (bi >> 1) * 32, n0, lane, bi & 1);
}

template <int KW, int S, bool TK, bool ACT>
hipError_t launch3a(const IgemmParams& p, int grid, hipStream_t st) {
    constexpr int KWV = KW == 5 ? 8 : KW, KWP = TK ? KWV / S : KWV, TJ = KWP < 16 ? KWP : 16, NQ = 16 / TJ, SC = TK ? 1 : S;
    constexpr int lds_bytes = R3_RING * (R3_TM * BK + NQ * (SC == 1 ? RS1 : RS2)) * 4;
    // (the attribute belongs to (function, current device): set on every call, nothing cached between calls)
    hipError_t e = hipFuncSetAttribute((const void*)conv_raw3_kernel<KW, S, TK, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((conv_raw3_kernel<KW, S, TK, ACT>), dim3(grid), dim3(NT3), lds_bytes, st, p);
    return hipGetLastError();
}
template <int KW, int S, bool TK>
hipError_t launch3(const IgemmParams& p, int grid, hipStream_t st) {
    return p.act_x == PG_ACT_NONE ? launch3a<KW, S, TK, false>(p, grid, st) : launch3a<KW, S, TK, true>(p, grid, st);
}

}  // namespace

bool pgconv::raw3_covers(int kind, const IgemmParams& p) {
    const bool k5 = p.k == 5 && p.s == 2;
    if (kind == KIND_F) {
        if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 4 && p.s == 2) || k5)) return false;
    } else if (kind == KIND_T) {
        if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 4 && p.s == 2) || k5)) return false;
    } else return false;
    const int kv = k5 ? 8 : p.k, kwp = kind == KIND_T ? kv / p.s : kv, tj = kwp < 16 ? kwp : 16;
    return ((long)p.Q * kwp) % BK == 0 && p.Q % (16 / tj > 0 ? 16 / tj : 1) == 0;     // whole slabs of whole channels only
}

hipError_t pgconv::launch_raw3(int kind, const IgemmParams& p, int grid, hipStream_t st) {
    if (kind == KIND_F) {
        if (p.k == 32) return launch3<32, 2, false>(p, grid, st);
        if (p.k == 8 && p.s == 1) return launch3<8, 1, false>(p, grid, st);
        if (p.k == 8) return launch3<8, 2, false>(p, grid, st);
        if (p.k == 5) return launch3<5, 2, false>(p, grid, st);
        return launch3<4, 2, false>(p, grid, st);
    }
    if (p.k == 32) return launch3<32, 2, true>(p, grid, st);
    if (p.k == 4) return launch3<4, 2, true>(p, grid, st);
    if (p.k == 5) return launch3<5, 2, true>(p, grid, st);
    if (p.s == 1) return launch3<8, 1, true>(p, grid, st);
    return launch3<8, 2, true>(p, grid, st);
}

hipError_t pgconv::launch_raw3_fixup(int kind, const IgemmParams& p, int grid, unsigned blocks, hipStream_t st) {
    if (kind == KIND_F) hipLaunchKernelGGL((conv_raw3_fixup_kernel<0>), dim3(blocks), dim3(NT3), 0, st, p, grid);
    else if (p.s == 2) hipLaunchKernelGGL((conv_raw3_fixup_kernel<3>), dim3(blocks), dim3(NT3), 0, st, p, grid);
    else hipLaunchKernelGGL((conv_raw3_fixup_kernel<1>), dim3(blocks), dim3(NT3), 0, st, p, grid);
    return hipGetLastError();
}
