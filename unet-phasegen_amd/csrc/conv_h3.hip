// conv_h3.hip -- the bf16-resident forward kernels (operands, layout contract: conv_h.hip; ring, gathers inside the MFMA stream:
// conv_h2.hip) with ONE wave per SIMD: 256 threads = 4 waves, one workgroup per CU, each wave a 256 x 64 sub-tile (8 x 2 blocks of
// 32 x 32, 256 accumulator registers -- the unified 512-register file of gfx950 holds them beside two fragment sets), workgroup
// tile 256 x 256.
// Why: at two waves per SIMD the pipe utilisation is B / (B + t_n) with t_n one wave's non-MFMA issue time per slab (~400 of 1024
// + 400 cycles) -- the partner covers only the other wave's.  A wave alone on its SIMD with twice the MFMAs
// per fragment set has 26 non-MFMA instructions for 16 MFMAs per k-step (8 + 6 LDS reads, 8 funnel shifts, addresses): 1.6 per
// MFMA gap, and an MFMA holds the vector issue for only 8 of its 32 cycles -- placed INSIDE the gaps they cost nothing.  The
// placement is written out: a k-step is 16 CHUNKS (one MFMA + at most one piece of other work) separated by scheduling fences.
#include "conv_common.h"

namespace {

constexpr int KB = 32;
constexpr int H_HEAD = 32;                // zero elements the caller guarantees in front of x (PG_H_HEAD)
constexpr int NT3 = 256;                  // threads per workgroup
constexpr int H3_LDS = 156 * 1024;
constexpr int H3_RING = 3;
constexpr int H3_REGS = 256;              // accumulator registers per thread: 16 blocks x 16
#ifndef PG_H3_SPBMAX
#define PG_H3_SPBMAX 4
#endif

__host__ __device__ constexpr int h3_rsd(int sc) { return sc == 1 ? 256 : 384; }
__host__ __device__ __forceinline__ int h3_round4(int v) { return (v + 3) & ~3; }
__host__ __device__ constexpr int h3_spb(int stg_floats) {
    for (int n = PG_H3_SPBMAX; n > 1; --n)
        if (H3_RING * n * stg_floats * 4 <= H3_LDS) return n;
    return 1;
}
template <int N> __device__ __forceinline__ void h3_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// ---- fragments of one MFMA k-step (k = 16 s + 8 h + 0..7 of a 32-deep slab) ------------------------------------------------
// A: 8 x ds_read_b128 (row r of each 32-row block; 16-byte groups XOR-swizzled by the row as conv_h.hip fills them).  B: per 32-column
// block the window dwords of the lane's column (5 for >= 8 taps per channel in the slab, 2 x 3 for 4 taps), finished by funnel shifts
// of the lane's parity (conv_h.hip).  The pieces are separate functions because the k-step loop below places them one per MFMA.
struct H3Frag { f32x4 a[8]; unsigned d[2][6]; u32x4v b[2]; };

// The reads are `asm volatile`: hipcc orders plain LDS loads freely against the scheduling fences of the k-step loop (it sank all eight
// A reads of a k-step behind its 11th MFMA).  What that costs: the compiler does not count them -- the loop waits for them itself
// (h3_lgkm0) before the first use of a fragment set.
__device__ __forceinline__ unsigned h3_lds_addr(const float* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
__device__ __forceinline__ void h3_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }      // lgkmcnt(0), the other counters untouched

// byte address of the lane's 16-byte group of row r in 32-row block 0 of a stage (k-step s, half h); block i is 2048 bytes further
__device__ __forceinline__ unsigned h3_a_addr(const float* stage, int s, int r, int h) {
    const int sw = (r >> 2) & 3;
    return h3_lds_addr(stage) + (r * 16 + (((2 * s + h) ^ sw) << 2)) * 4;
}
template <int I> __device__ __forceinline__ void h3_load_a(unsigned a_addr, H3Frag& f) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a[I]) : "v"(a_addr), "n"(I * 2048));
}
// byte address of the first window dword of the lane's column in block jb (k-step s, half h); TJ = 4: of the first of its two channels
template <int TJ, int RSD, int TA>
__device__ __forceinline__ unsigned h3_b_addr(const float* stage, int s, int h, int bdw_jb) {
    const int qi = TJ == 32 ? 0 : (TJ == 16 ? s : (TJ == 8 ? 2 * s + h : 4 * s + 2 * h));
    const int tap0 = TJ == 32 ? 16 * s + 8 * h : (TJ == 16 ? 8 * h : 0);
    return h3_lds_addr(stage + TA) + (qi * RSD + bdw_jb + (tap0 >> 1)) * 4;
}
template <int TJ, int RSD, int JB> __device__ __forceinline__ void h3_load_b(unsigned b_addr, H3Frag& f) {
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    u32x2v t0, t1;
    if (TJ >= 8) {          // dwords 0 ... 4 of the window
        unsigned t2;
        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(t0) : "v"(b_addr));
        asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(t1) : "v"(b_addr));
        asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(t2) : "v"(b_addr));
        f.d[JB][0] = t0[0]; f.d[JB][1] = t0[1]; f.d[JB][2] = t1[0]; f.d[JB][3] = t1[1]; f.d[JB][4] = t2;
    } else {                // two channels, dwords 0 ... 2 of each (RSD dwords apart: offsets in dwords, RSD + 2 <= 255 does not hold -> bytes)
        unsigned t2, t3;
        static_assert(RSD * 4 + 8 < 65536, "16-bit byte offset");
        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(t0) : "v"(b_addr));
        asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(t2) : "v"(b_addr));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t3) : "v"(b_addr), "n"(RSD * 4));
        asm volatile("ds_read2_b32 %0, %1 offset0:1 offset1:2" : "=v"(t1) : "v"(b_addr + RSD * 4));
        f.d[JB][0] = t0[0]; f.d[JB][1] = t0[1]; f.d[JB][2] = t2; f.d[JB][3] = t3; f.d[JB][4] = t1[0]; f.d[JB][5] = t1[1];
    }
}
template <int TJ>
__device__ __forceinline__ void h3_finish_b(int jb, const int (&bsh)[2], H3Frag& f) {
    if (TJ >= 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f.b[jb][i] = __builtin_amdgcn_alignbit(f.d[jb][i + 1], f.d[jb][i], bsh[jb]);
    } else {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int i = 0; i < 2; ++i) f.b[jb][2 * cc + i] = __builtin_amdgcn_alignbit(f.d[jb][3 * cc + i + 1], f.d[jb][3 * cc + i], bsh[jb]);
    }
}

// One k-step: the 16 MFMAs of `cur`, and in their gaps -- one piece per MFMA, pinned by scheduling fences -- the reads of the NEXT
// k-step's fragments `nxt` (B windows behind MFMAs 0-1, A rows behind 2-9, the wait for them and the funnel shifts behind 15),
// and gathers of a later stage group behind MFMAs 10, 12, 14 (and 15): `issue(e)` for e = E0 ... E0 + 3.
// (An LDS-DMA instruction costs its wave 60-185 cycles of issue, MI355X_MICROARCH.md; back to back behind one MFMA -- hipcc's own
// placement, and sched_group_barrier's with the M0 writes between them -- all but the first 24 of those cycles idle the pipe.)
template <int TJ, int RSD, int TA, int E0, bool NEXT, typename Issue>
__device__ __forceinline__ void h3_kstep(const H3Frag& cur, H3Frag& nxt, const float* nstage, int ns, int r, int h, const int (&bdw)[2],
                                         const int (&bsh)[2], AccT<8, 2>& acc, const Issue& issue) {
    const unsigned aa = h3_a_addr(nstage, ns, r, h);
    const unsigned ba0 = h3_b_addr<TJ, RSD, TA>(nstage, ns, h, bdw[0]), ba1 = h3_b_addr<TJ, RSD, TA>(nstage, ns, h, bdw[1]);
#define H3_CHUNK(C, WORK)                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                           \
    acc.c[(C) >> 1][(C) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a[(C) >> 1]),              \
                                                                      __builtin_bit_cast(bf16x8, cur.b[(C) & 1]), acc.c[(C) >> 1][(C) & 1], 0, 0, 0); \
    WORK;
    // NEXT = false (last k-step of a stage group): no reads at all.  A read whose result is never used is not harmless here: hipcc
    // hands the "dead" destination registers out again at once (one became the address of the next gather), and the LDS answer,
    // which it does not know about, lands in them 64+ cycles later.
    H3_CHUNK(0, if (NEXT) (h3_load_b<TJ, RSD, 0>(ba0, nxt)))
    H3_CHUNK(1, if (NEXT) (h3_load_b<TJ, RSD, 1>(ba1, nxt)))
    H3_CHUNK(2, if (NEXT) h3_load_a<0>(aa, nxt))
    H3_CHUNK(3, if (NEXT) h3_load_a<1>(aa, nxt))
    H3_CHUNK(4, if (NEXT) h3_load_a<2>(aa, nxt))
    H3_CHUNK(5, if (NEXT) h3_load_a<3>(aa, nxt))
    H3_CHUNK(6, if (NEXT) h3_load_a<4>(aa, nxt))
    H3_CHUNK(7, if (NEXT) h3_load_a<5>(aa, nxt))
    H3_CHUNK(8, if (NEXT) h3_load_a<6>(aa, nxt))
    H3_CHUNK(9, if (NEXT) h3_load_a<7>(aa, nxt))
    H3_CHUNK(10, issue(E0 + 0))
    H3_CHUNK(11, (void)0)
    H3_CHUNK(12, issue(E0 + 1))
    H3_CHUNK(13, (void)0)
    H3_CHUNK(14, issue(E0 + 2))
    H3_CHUNK(15, issue(E0 + 3))                // (the 4th slot: only the second k-step of a slab with 7 gathers uses it)
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT) {                                // every read of `nxt` was issued six or more MFMAs (190 cycles) ago
        h3_lgkm0();
        __builtin_amdgcn_sched_barrier(0);
        h3_finish_b<TJ>(0, bsh, nxt);
        h3_finish_b<TJ>(1, bsh, nxt);
    }
    __builtin_amdgcn_sched_barrier(0);
#undef H3_CHUNK
}

// An accumulator register read where it is USED: the "a" constraint keeps the value in its AGPR up to this instruction.  With plain
// uses hipcc split all 256 live ranges to VGPRs at the loop exit in one go (256 v_accvgpr_read in a row) and spilled what did not fit.
__device__ __forceinline__ float h3_acc(float v) {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(v));
    return x;
}

// cols_left: columns of the problem from this wave's first column on (wave-uniform).  Column blocks without columns are neither
// written here nor read by the fixup: at batch 1 (65 ... 14 columns in a 256-wide tile) the partial tiles were 40 % of the bytes moved
__device__ __forceinline__ void store_partial3(float* ws, int g, int slot, const AccT<8, 2>& acc, int tid, int cols_left) {
    float* dst = ws + ((long)(g * 2 + slot) * H3_REGS) * NT3 + tid;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (j * 32 < cols_left) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((i * 2 + j) * 16 + r) * NT3] = h3_acc(acc.c[i][j][r]);
            }
}

template <int KW, int S, bool TKIND>
__global__ __launch_bounds__(NT3, 1) void conv_h3_kernel(const IgemmParams p) {
    constexpr int TM = 256, TN = 256;
    constexpr int KWP = TKIND ? KW / S : KW;
    constexpr int TJ = KWP < 32 ? KWP : 32, NQ = 32 / TJ;
    constexpr int SC = TKIND ? 1 : S;
    constexpr int RSD = h3_rsd(SC);
    constexpr int NP = NQ * RSD / 4;                  // 16-byte window pieces per slab
    constexpr int NI = (NP + 63) / 64;                // ... = NI wave instructions, dealt round-robin to the 4 waves
    constexpr int NPW = (NI + 3) / 4;
    constexpr int NAW = TM / 16 / 4;                  // weight tile: 16 rows x 64 B per wave instruction, 4 per wave
    constexpr int TA = TM * 16;
    constexpr int STG = TA + 4 * NPW * 256;           // every wave owns NPW window slots (slots past NI take zero writes: no branch in the issue)
    constexpr int SPB = h3_spb(STG);
    constexpr int SSTG = SPB * STG;
    constexpr int NGRP = SPB * (NAW + NPW);           // LDS-DMA instructions of a stage group, the same for every wave
    constexpr int MBW = 8, NBW = 2;
    static_assert(KWP == 4 || KWP == 8 || KWP == 16 || KWP == 32, "taps per channel in K order");
    static_assert(H3_RING * SSTG * 4 <= 160 * 1024 && NGRP < 64, "LDS budget / vmcnt range");
    static_assert(NAW + NPW <= 7, "seven gather slots per slab in the k-step loop");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wv;
    const int r = lane & 31, h = lane >> 5;
    const int Lcol = TKIND ? p.U : p.Ly;
    const int Ktot = p.Q * KWP, Mrows = TKIND ? p.M * S : p.M;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes);
    const rsrc_t rx = make_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p.x) - H_HEAD), p.x_bytes + 2 * H_HEAD);
    const int g = logical_wg(blockIdx.x, gridDim.x, p.whole);
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, gridDim.x, p.whole);
    int pos = split_lo(sp, g);
    const int pos_end = split_lo(sp, g + 1);
    int slot = 0;
    while (pos < pos_end) {
        const int tile = pos / p.nslab, sb = pos - tile * p.nslab;
        const int se = min(p.nslab, sb + (pos_end - pos));
        const int m0 = (tile / p.tilesN) * TM, n0 = (tile % p.tilesN) * TN;
        const int b0 = n0 / Lcol, t0 = n0 - b0 * Lcol;
        const int nc0 = min(Lcol - t0, TN);
        const int pos_first = TKIND ? p.u_off + t0 - (TJ - 1) : S * t0 - p.p;
        const int pos_mid = TKIND ? p.u_off - (TJ - 1) : -p.p;
        const int sh0 = pos_first & 1, shm = pos_mid & 1;
        const int nd0 = h3_round4((SC * (nc0 - 1) + TJ + sh0 + 1) >> 1);
        const int ndm = h3_round4((SC * (Lcol - 1) + TJ + shm + 1) >> 1);

        int avoff[NAW];
#pragma unroll
        for (int e = 0; e < NAW; ++e) {
            const int m = m0 + 16 * (wv + 4 * e) + (lane >> 2);
            avoff[e] = m < Mrows ? m * Ktot * 2 + dma16_kc(lane) * 4 : FAR;
        }
        int voff[NPW];
#pragma unroll
        for (int e = 0; e < NPW; ++e) {
            const int i = 64 * (wv + 4 * e) + lane, qi = i / (RSD / 4), d = 4 * (i - qi * (RSD / 4));
            int k, dl;
            if (d < nd0) { k = 0; dl = d; } else { k = 1 + (d - nd0) / ndm; dl = (d - nd0) - (k - 1) * ndm; }
            const int e0 = (k ? pos_mid - shm : pos_first - sh0) + 2 * dl;
            const int b = b0 + k;
            const bool ok = i < NP && b < p.B && k * Lcol < t0 + TN;      // (i >= NP: the wave's padding slot)
            voff[e] = ok ? (b * (int)p.x_bs + qi * p.x_pitch + e0 + H_HEAD) * 2 : FAR;
        }
        int bdw[NBW], bsh[NBW];
#pragma unroll
        for (int jb = 0; jb < NBW; ++jb) {
            const int c = wn * (NBW * 32) + jb * 32 + r, seg = (t0 + c) / Lcol;
            const int cin = seg ? (t0 + c) - seg * Lcol : c;
            const int el = SC * cin + (seg ? shm : sh0);
            bdw[jb] = (seg ? nd0 + (seg - 1) * ndm : 0) + (el >> 1);
            bsh[jb] = (el & 1) << 4;
        }
        AccT<MBW, NBW> acc;
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
            for (int j = 0; j < NBW; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc.c[i][j][q] = 0.f;

        // Stage groups are always WHOLE (the fragment phase has no per-slab branch: conv_h_frag.h FULL): a slab past the end of this
        // workgroup's K range is gathered from out-of-range offsets -- the buffer unit writes zeros, no memory traffic -- for the
        // weights AND the window (0 x stale LDS could be 0 x NaN).  Every wave therefore has at least NGRP gathers in flight per group.
        // gather number e of slab `slab` into its stage: e < NAW the weight rows, then the wave's window pieces
        auto issue_piece = [&](float* stage, int slab, int e) {
#ifdef PG_H3_ABL          // dev ablation (wrong results): no gathers inside the loop
            if (PG_H3_ABL == 1 && slab >= sb + 2 * SPB) return;
#endif
            const bool live = slab < se;
            const int k0 = slab * KB;
            if (e < NAW) dma16s(rw, stage + 256 * (wv + 4 * e), live ? avoff[e < NAW ? e : 0] : FAR, k0 * 2);
            else if (e < NAW + NPW) dma16s(rx, stage + TA + 256 * (wv + 4 * (e - NAW)), live ? voff[e < NAW ? 0 : e - NAW] : FAR, (k0 / KWP) * p.x_pitch * 2);
        };
        // Waits.  wait_first: before the first read of a segment -- everything but the youngest group (group 1) has landed.  wait_next:
        // in front of the LAST k-step of stage group i, whose MFMA gaps carry the reads of group i + 1's first fragments -- so the
        // next group starts on loaded registers instead of 16 exposed LDS reads behind its barrier (~250 of ~2000 + cycles per group).
        // By then this wave has issued, of group i + 2, the gathers of slabs 0 ... SPB - 2 and the first k-step's three of the last
        // slab: NMID younger instructions may stay in flight, everything older (group i + 1) is done.  The barrier also closes the
        // reads of group i for every wave (its last k-step multiplies registers): group i + 3's gathers may overwrite its slot.
        constexpr int NMID = (SPB - 1) * (NAW + NPW) + 3;
        auto wait_first = [&]() {
            h3_wait_vmcnt<NGRP>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto wait_next = [&]() {
#ifdef PG_H3_ABL
            h3_wait_vmcnt<0>();
#else
            h3_wait_vmcnt<NMID>();
#endif
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
#pragma unroll
        for (int hf = 0; hf < 2 * SPB; ++hf)
#pragma unroll
            for (int e = 0; e < NAW + NPW; ++e) issue_piece(lds + hf * STG, sb + hf, e);
        __builtin_amdgcn_sched_barrier(0);
        wait_first();
        H3Frag f0, f1;
        {
            const unsigned aa = h3_a_addr(lds, 0, r, h);
            h3_load_b<TJ, RSD, 0>(h3_b_addr<TJ, RSD, TA>(lds, 0, h, bdw[0]), f0);
            h3_load_b<TJ, RSD, 1>(h3_b_addr<TJ, RSD, TA>(lds, 0, h, bdw[1]), f0);
            h3_load_a<0>(aa, f0); h3_load_a<1>(aa, f0); h3_load_a<2>(aa, f0); h3_load_a<3>(aa, f0);
            h3_load_a<4>(aa, f0); h3_load_a<5>(aa, f0); h3_load_a<6>(aa, f0); h3_load_a<7>(aa, f0);
            h3_lgkm0();
            __builtin_amdgcn_sched_barrier(0);
            h3_finish_b<TJ>(0, bsh, f0);
            h3_finish_b<TJ>(1, bsh, f0);
        }
        int st = 0;
        for (int sl = sb; sl < se; sl += SPB) {
            const int st1 = st == 2 ? 0 : st + 1, st2 = st >= 1 ? st - 1 : 2;                // (st + 1) % 3, (st + 2) % 3
            const float* const cur = lds + st * SSTG;
            float* const ring2 = lds + st2 * SSTG;
            const int s2 = sl + 2 * SPB;
#pragma unroll
            for (int hf = 0; hf < SPB; ++hf) {
                const float* const stg = cur + hf * STG;
                auto issue = [&](int e) { issue_piece(ring2 + hf * STG, s2 + hf, e - 1); };                 // second k-step: gathers 3 ... 6
                auto issue0 = [&](int e) { if (e < 3) issue_piece(ring2 + hf * STG, s2 + hf, e); };       // first: 0 ... 2
                h3_kstep<TJ, RSD, TA, 0, true>(f0, f1, stg, 1, r, h, bdw, bsh, acc, issue0);
                if (hf + 1 < SPB) h3_kstep<TJ, RSD, TA, 4, true>(f1, f0, stg + STG, 0, r, h, bdw, bsh, acc, issue);
                else {
                    // (past the last group of the segment the "next" stage holds zero-filled or older slabs: read, never multiplied;
                    // f0 is carried by the loop, so its registers stay reserved until the reads have landed)
                    wait_next();
                    h3_kstep<TJ, RSD, TA, 4, true>(f1, f0, lds + st1 * SSTG, 0, r, h, bdw, bsh, acc, issue);
                }
            }
            st = st1;
        }
        __syncthreads();
        // the accumulator reads below are `asm` (h3_acc): hipcc's hazard recogniser does not see them, so the wait states an MFMA
        // result needs before a VALU may read it (its 8 passes = 32 cycles) are spelled out -- 48 cycles, once per tile segment
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
        if (sb == 0 && se == p.nslab) {
            // one 32 x 32 block at a time behind scheduling fences: with all 256 accumulator registers of the wave tile in one
            // epilogue hipcc moved them to VGPRs wholesale and spilled ~200 of them -- through the MAIN loop as well
            // (written out: a `#pragma unroll` nest over the 16 blocks exceeds hipcc's unroll budget, stays rolled, and indexes
            // the accumulators dynamically -- i.e. keeps them in scratch)
#define H3_EPI(I, J)                                                                                              \
    {   AccT<1, 1> blk;                                                                                          \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) blk.c[0][0][q] = h3_acc(acc.c[I][J][q]);                          \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (TKIND) epilogue_t<S, 1, 1>(p, blk, m0 + (I) * 32, n0 + (wn * NBW + (J)) * 32, lane, 0, 0);           \
        else epilogue_f<S, 1, 1>(p, blk, m0 + (I) * 32, n0 + (wn * NBW + (J)) * 32, lane, 0, 0);                 \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    }
            H3_EPI(0, 0) H3_EPI(0, 1) H3_EPI(1, 0) H3_EPI(1, 1) H3_EPI(2, 0) H3_EPI(2, 1) H3_EPI(3, 0) H3_EPI(3, 1)
            H3_EPI(4, 0) H3_EPI(4, 1) H3_EPI(5, 0) H3_EPI(5, 1) H3_EPI(6, 0) H3_EPI(6, 1) H3_EPI(7, 0) H3_EPI(7, 1)
#undef H3_EPI
        } else store_partial3(p.ws, g, slot, acc, tid, __builtin_amdgcn_readfirstlane(p.B * (TKIND ? p.U : p.Ly) - n0 - (tid >> 6) * 64));
        pos += se - sb;
        slot = 1;
    }
}

// fixup of the stream-K split: one workgroup per (split tile, 32 x 32 block index of the wave tile: 16 of them); WIDE (many
// segments per tile, small-batch inference): four workgroups per block, one per GEMM wave, whose four waves each sum every fourth
// segment and wave 0 adds the four sums in order (conv_igemm.hip's conv_fixup_kernel has the same two forms)
template <int KIND, bool WIDE>
__global__ __launch_bounds__(NT3) void conv_h3_fixup_kernel(const IgemmParams p, int G) {
    constexpr int MB = 8, NB = 2;
    __shared__ float red[WIDE ? 3 * 16 * 64 : 1];
    const int unit = WIDE ? blockIdx.x >> 2 : blockIdx.x, q = WIDE ? threadIdx.x >> 6 : 0;
    const int tid = WIDE ? (blockIdx.x & 3) * 64 + (threadIdx.x & 63) : threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = 0, wn = wv;
    const int tile = p.whole + unit / (MB * NB), blk = unit % (MB * NB), bi = blk / NB, bj = blk - bi * NB;
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, G, p.whole);
    const int first = tile * p.nslab, last = first + p.nslab - 1;
    const int g0 = split_owner(sp, first), g1 = split_owner(sp, last);
    if (g0 == g1 && split_lo(sp, g0) <= first && split_lo(sp, g0 + 1) > last) return;
    if ((tile % p.tilesN) * p.tn_stride + wn * NB * 32 + bj * 32 >= p.B * (KIND == 0 ? p.Ly : p.U)) return;     // a block without columns
    AccT<1, 1> acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.c[0][0][r] = 0.f;
#pragma unroll 2
    for (int g = g0 + q; g <= g1; g += WIDE ? 4 : 1) {
        const int slot = (split_lo(sp, g) / p.nslab == tile) ? 0 : 1;
        const float* src = p.ws + ((long)(g * 2 + slot) * H3_REGS) * NT3 + tid;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.c[0][0][r] += src[(blk * 16 + r) * NT3];
    }
    if (WIDE) {         // (the returns above are uniform over the workgroup here)
        if (q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((q - 1) * 16 + r) * 64 + lane] = acc.c[0][0][r];
        }
        __syncthreads();
        if (q) return;
#pragma unroll
        for (int qq = 0; qq < 3; ++qq)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.c[0][0][r] += red[(qq * 16 + r) * 64 + lane];
    }
    const int m0 = (tile / p.tilesN) * 256 + wm * (MB - 1) * 32 + bi * 32;
    const int n0 = (tile % p.tilesN) * p.tn_stride + wn * (NB - 1) * 32 + bj * 32;
    if (KIND == 0) epilogue_f<0, 1, 1>(p, acc, m0, n0, lane, wm, wn);
    else epilogue_t<0, 1, 1>(p, acc, m0, n0, lane, wm, wn);
}

template <int KW, int S, bool TK>
hipError_t launch3(const IgemmParams& p, int grid, hipStream_t st) {
    constexpr int KWP = TK ? KW / S : KW, TJ = KWP < 32 ? KWP : 32, NQ = 32 / TJ, SC = TK ? 1 : S;
    constexpr int NI = (NQ * h3_rsd(SC) / 4 + 63) / 64, STG = 256 * 16 + 4 * ((NI + 3) / 4) * 256;
    constexpr int lds_bytes = H3_RING * h3_spb(STG) * STG * 4;
    hipError_t e = hipFuncSetAttribute((const void*)conv_h3_kernel<KW, S, TK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((conv_h3_kernel<KW, S, TK>), dim3(grid), dim3(NT3), lds_bytes, st, p);
    return hipGetLastError();
}

}  // namespace

// do the windows of every possible 256-column tile fit the reserved dwords, and is the rows' zero tail long enough?
// (host-side mirror of the kernel's geometry: pg_conv_fwd_h_supported)
bool pgconv::h_supported(int kind, const IgemmParams& p) {
    constexpr int tn = 256;
    const bool t = kind == KIND_T;
    if (kind == KIND_G) return false;
    if (t) { if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 5 && p.s == 2))) return false; }
    else if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 4 && p.s == 2))) return false;
    const int kwp = t ? pg_shadow_taps(p.k, p.s) : p.k, tj = kwp < 32 ? kwp : 32, nq = 32 / tj, sc = t ? 1 : p.s;
    if (p.Q % nq || (p.x_pitch & 1) || (p.x_bs & 1) || p.x_pitch <= p.Lx) return false;
    const int lcol = t ? p.U : p.Ly, rsd = h3_rsd(sc);
    const int pos_mid = t ? p.u_off - (tj - 1) : -p.p, shm = pos_mid & 1;
    const int ndm = h3_round4((sc * (lcol - 1) + tj + shm + 1) >> 1);
    // elements of a row's neighbourhood a window piece can touch: [pos_mid - shm, pos_mid - shm + 2 ndm) for a sample's first
    // column at frame 0; a first segment that starts at column t0 ends no later (its window is the same one cut at t0, rounded
    // up to a piece: + 6 elements at most)
    const int left = pos_mid - shm < 0 ? -(pos_mid - shm) : 0;
    const int right = pos_mid - shm + 2 * ndm + 6 - p.Lx;
    const int tail = p.x_pitch - p.Lx;
    if (left > H_HEAD || tail < left || tail < right) return false;
    // the kernel lays the samples' windows out back to back (segment 0, full middle segments, last partial one): the worst
    // first-column position t0 must fit the reserved dwords
    int need = 0;
    for (int t0 = 0; t0 < lcol; ++t0) {
        const int nc0 = lcol - t0 < tn ? lcol - t0 : tn, rem = tn - nc0;
        const int sh0 = (pos_mid + sc * t0) & 1;
        int nmid = rem / lcol, nlast = rem - nmid * lcol;                        // full middle samples, columns of the last one
        if (1 + nmid + (nlast ? 1 : 0) > p.B) { nlast = 0; if (1 + nmid > p.B) nmid = p.B - 1; }   // only B samples exist
        const int n = h3_round4((sc * (nc0 - 1) + tj + sh0 + 1) >> 1) + nmid * ndm + (nlast ? (sc * (nlast - 1) + tj + shm + 2) >> 1 : 0);
        if (n > need) need = n;
    }
    return need <= rsd;
}

hipError_t pgconv::launch_h3(int kind, const IgemmParams& p, int grid, hipStream_t st) {
    if (kind == KIND_F) {
        if (p.k == 32) return launch3<32, 2, false>(p, grid, st);
        if (p.k == 8 && p.s == 1) return launch3<8, 1, false>(p, grid, st);
        if (p.k == 8) return launch3<8, 2, false>(p, grid, st);
        return launch3<4, 2, false>(p, grid, st);
    }
    if (p.k == 32) return launch3<32, 2, true>(p, grid, st);
    if (p.s == 1) return launch3<8, 1, true>(p, grid, st);
    return launch3<8, 2, true>(p, grid, st);
}

hipError_t pgconv::launch_h3_fixup(int kind, const IgemmParams& p, int grid, unsigned split_tiles, bool wide, hipStream_t st) {
    const dim3 fg(split_tiles * (wide ? 64 : 16));
    if (kind == KIND_F) { if (wide) hipLaunchKernelGGL((conv_h3_fixup_kernel<0, true>), fg, dim3(NT3), 0, st, p, grid);
                          else hipLaunchKernelGGL((conv_h3_fixup_kernel<0, false>), fg, dim3(NT3), 0, st, p, grid); }
    else { if (wide) hipLaunchKernelGGL((conv_h3_fixup_kernel<1, true>), fg, dim3(NT3), 0, st, p, grid);
           else hipLaunchKernelGGL((conv_h3_fixup_kernel<1, false>), fg, dim3(NT3), 0, st, p, grid); }
    return hipGetLastError();
}
