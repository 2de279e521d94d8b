// conv_h2.hip -- the bf16-resident forward kernels (see conv_h.hip for operands, layout contract and fragment scheme) on a LARGER
// workgroup tile: 512 threads = 8 waves, ONE workgroup per CU, each wave still a 128 x 64 sub-tile (4 x 2 blocks of 32 x 32, 128
// accumulator registers), arranged
//     WM = 1:  1 (M) x 8 (N) waves -> tile 128 x 512    WM = 2:  2 (M) x 4 (N) waves -> tile 256 x 256.
// Why: conv_h.hip's 128 x 256 tile is bound by the bytes a CU pulls from L2 per MFMA cycle (19 KB per CU
// and 1000 cycles at the full rate; weights from L1 -> 1.6x faster).  Per 32-deep slab a tile needs TM x 64 B of weights and a
// window set that grows with TN, for TM x TN x 64 FLOP: the weight bytes per FLOP fall with TN, the window bytes with TM.
//     U0 (16 taps per channel):  128 x 256: 8 + 2 KB   128 x 512: 8 + 4 KB per 2x the FLOP (-40 %)   256 x 256: 16 + 2 KB per 2x (-10 %)
//     D3 (4 taps, stride 2):     128 x 256: 8 + 12 KB  128 x 512: 8 + 24 KB per 2x (-20 %)          256 x 256: 16 + 12 KB per 2x (-30 %)
// so long-tap layers take the wide tile (host: conv_fwd_h_impl).  With one workgroup per CU the stages may use most of the 160 KB
// of LDS: a ring of three groups of up to four slabs, gathered two groups ahead behind counted waits (kernel body).
#include "conv_common.h"
#include "conv_h_frag.h"

namespace {

constexpr int KB = 32;                    // k per slab (two MFMA k-steps of 16)
constexpr int H_HEAD = 32;                // zero elements the caller guarantees in front of x (PG_H_HEAD)
constexpr int NT2 = 512;                  // threads per workgroup
constexpr int H2_LDS = 156 * 1024;        // LDS budget of the ring of three stage groups
#ifndef PG_H2_STAGGER
#define PG_H2_STAGGER 0
#endif
#ifndef PG_H2_SPREAD
#define PG_H2_SPREAD 1
#endif
#ifndef PG_H2_RING
#define PG_H2_RING 3
#endif
#ifndef PG_H2_SPBMAX
#define PG_H2_SPBMAX 4
#endif
constexpr int H2_RING = PG_H2_RING;       // stage groups: one being multiplied, one landed / landing, one just issued
__host__ __device__ constexpr int h2_spb(int stg_floats) {      // slabs per stage group (= per barrier): as many as the ring affords, <= 4
    for (int n = PG_H2_SPBMAX; n > 1; --n)
        if (H2_RING * n * stg_floats * 4 <= H2_LDS) return n;
    return 1;
}
// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4, expcnt [6:4] and lgkmcnt [11:8] at their maxima)
template <int N> __device__ __forceinline__ void h2_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

__host__ __device__ constexpr int h2_rsd(int sc, int tn) { return (sc == 1 ? 256 : 384) * (tn / 256); }
__host__ __device__ __forceinline__ int h2_round4(int v) { return (v + 3) & ~3; }

// partial tiles of the stream-K split: [workgroup][slot][register][thread]
__device__ __forceinline__ void store_partial2(float* ws, int g, int slot, const AccT<4, 2>& acc, int tid) {
    float* dst = ws + ((long)(g * 2 + slot) * ACC_REGS) * NT2 + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((i * 2 + j) * 16 + r) * NT2] = acc.c[i][j][r];
}

template <int KW, int S, bool TKIND, int WM>
__global__ __launch_bounds__(NT2, 2) void conv_h2_kernel(const IgemmParams p) {
    constexpr int WNW = 8 / WM;                       // waves along N
    constexpr int TM = 128 * WM, TN = 64 * WNW;       // workgroup tile
    constexpr int KWP = TKIND ? KW / S : KW;          // taps per channel in K order
    constexpr int TJ = KWP < 32 ? KWP : 32, NQ = 32 / TJ;
    constexpr int SC = TKIND ? 1 : S;                 // window elements per column step
    constexpr int RSD = h2_rsd(SC, TN);               // dwords reserved per channel window
    constexpr int NP = NQ * RSD / 4;                  // 16-byte window pieces per slab
    constexpr int NI = (NP + 63) / 64;                // ... = NI wave instructions, dealt round-robin to the 8 waves
    constexpr int NPW = (NI + 7) / 8;
    constexpr int NAI = TM / 16, NAW = NAI / 8;       // weight tile: 16 rows x 64 B per wave instruction; 1 or 2 per wave
    constexpr int TA = TM * 16;                       // dwords of the weight tile
    constexpr int STG = TA + NI * 256;                // (window region rounded up to whole wave instructions)
    constexpr int SPB = h2_spb(STG);                  // slabs per stage group = per barrier
    constexpr int SSTG = SPB * STG;
    // LDS-DMA instructions of one FULL stage group that every wave issues at least (waves with a window instruction more over-wait by it)
    constexpr int NGRP = SPB * (NAW + NI / 8);
    static_assert(KWP == 4 || KWP == 8 || KWP == 16 || KWP == 32, "taps per channel in K order");
    static_assert(H2_RING * SSTG * 4 <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int MBW = 4, NBW = 2;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WM == 1 ? 0 : wv / WNW, wn = WM == 1 ? wv : wv % WNW;
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
        // segments (samples) of the tile, as in conv_h.hip: whole 16-byte pieces laid back to back
        const int nc0 = min(Lcol - t0, TN);
        const int pos_first = TKIND ? p.u_off + t0 - (TJ - 1) : S * t0 - p.p;
        const int pos_mid = TKIND ? p.u_off - (TJ - 1) : -p.p;
        const int sh0 = pos_first & 1, shm = pos_mid & 1;
        const int nd0 = h2_round4((SC * (nc0 - 1) + TJ + sh0 + 1) >> 1);
        const int ndm = h2_round4((SC * (Lcol - 1) + TJ + shm + 1) >> 1);

        int avoff[NAW];
#pragma unroll
        for (int e = 0; e < NAW; ++e) {
            const int m = m0 + 16 * (wv + 8 * e) + (lane >> 2);
            avoff[e] = m < Mrows ? m * Ktot * 2 + dma16_kc(lane) * 4 : FAR;
        }
        int voff[NPW];
#pragma unroll
        for (int e = 0; e < NPW; ++e) {
            const int i = 64 * (wv + 8 * e) + lane, qi = i / (RSD / 4), d = 4 * (i - qi * (RSD / 4));
            int k, dl;
            if (d < nd0) { k = 0; dl = d; } else { k = 1 + (d - nd0) / ndm; dl = (d - nd0) - (k - 1) * ndm; }
            const int e0 = (k ? pos_mid - shm : pos_first - sh0) + 2 * dl;
            const int b = b0 + k;
            const bool ok = i < NP && b < p.B && k * Lcol < t0 + TN;
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

#define H2_ISSUE(STAGE_PTR, SLAB)                                                                             \
    {   float* const As = (STAGE_PTR); float* const Bw = (STAGE_PTR) + TA;                                   \
        const int k0 = (SLAB) * KB;                                                                          \
        if (k0 < Ktot) {                                                                                     \
            _Pragma("unroll") for (int e = 0; e < NAW; ++e) dma16s(rw, As + 256 * (wv + 8 * e), avoff[e], k0 * 2); \
            const int sq = (k0 / KWP) * p.x_pitch * 2;                                                       \
            _Pragma("unroll") for (int e = 0; e < NPW; ++e)                                                  \
                if (wv + 8 * e < NI) dma16s(rx, Bw + 256 * (wv + 8 * e), voff[e], sq);                       \
        }                                                                                                    \
    }

        // Ring of three stage groups with the gathers TWO groups ahead: the weights stream from HBM (a layer's shadow is larger than the
        // Infinity Cache) and every workgroup that shares a weight panel waits for the same fill, so a gather needs ~2.5 us to land --
        // one group (4 slabs ~ 3 us of MFMAs) was not enough: stamps showed 280 cycles per slab in the closing barrier's vmcnt(0).
        // The wait is COUNTED (the youngest group stays in flight) and the barrier is the raw s_barrier: __syncthreads() would drain
        // every pending LDS-DMA (vmcnt(0)).  Order per iteration: [issue group i+2] [fragments + MFMAs of group i] [wait: group i+1
        // landed for THIS wave] [barrier: ... for every wave; all reads of group i done, its stage is free for group i+3].
        PG_STAMP_DECL
        auto wait_landed = [&](int youngest_first_slab) {       // all gathers done except (at most) the youngest group's
            if (youngest_first_slab + SPB <= p.nslab) h2_wait_vmcnt<NGRP>();     // that group was issued in full
            else h2_wait_vmcnt<0>();                             // partial or empty youngest group: drain
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
#pragma unroll
        for (int hf = 0; hf < SPB; ++hf) H2_ISSUE(lds + hf * STG, sb + hf)
#pragma unroll
        for (int hf = 0; hf < SPB; ++hf) H2_ISSUE(lds + SSTG + hf * STG, sb + SPB + hf)
        __builtin_amdgcn_sched_barrier(0);
        wait_landed(sb + SPB);
        int st = 0;
        for (int sl = sb; sl < se; sl += SPB) {
            const int st2 = st >= 1 ? st - 1 : 2;                // (st + 2) % 3
#if PG_H2_STAGGER        /* dev knob: waves 4-7 (the SIMD partners of 0-3) start each group PG_H2_STAGGER x 64 cycles late */
            if (wv >= 4) __builtin_amdgcn_s_sleep(PG_H2_STAGGER);
#endif
            PG_STAMP(0)
            const int nsl = min(SPB, se - sl);
#if PG_H2_SPREAD
            // the gathers of group i+2 ride inside this group's MFMA stream, one slab's worth between the two k-steps of each slab
            // (a partial last group issues what is left in a burst: nothing of it is ever multiplied by this workgroup's range)
            float* const ring2 = lds + st2 * SSTG;
            const int s2 = sl + 2 * SPB;
            auto issue_one = [&](int hf) { H2_ISSUE(ring2 + hf * STG, s2 + hf) };
            PG_STAMP(1)
            h_mma_group<TJ, RSD, TA, STG, SPB>(lds + st * SSTG, nsl, r, h, wm, bdw, bsh, acc, issue_one);
            for (int hf = nsl; hf < SPB; ++hf) issue_one(hf);
#else
#pragma unroll
            for (int hf = 0; hf < SPB; ++hf) H2_ISSUE(lds + st2 * SSTG + hf * STG, sl + 2 * SPB + hf)
            __builtin_amdgcn_sched_barrier(0);
            PG_STAMP(1)
            h_mma_group<TJ, RSD, TA, STG, SPB>(lds + st * SSTG, nsl, r, h, wm, bdw, bsh, acc);
#endif
            __builtin_amdgcn_sched_barrier(0);
            PG_STAMP(2)
            wait_landed(sl + 2 * SPB);
            PG_STAMP(3)
            st = st == 2 ? 0 : st + 1;
        }
        __syncthreads();                                         // (drains the gathers issued past the range: the next tile restarts the ring)
        PG_STAMP_FLUSH
#undef H2_ISSUE
        if (sb == 0 && se == p.nslab) {
            if (TKIND) epilogue_t<S, MBW, NBW>(p, acc, m0, n0, lane, wm, wn);
            else epilogue_f<S, MBW, NBW>(p, acc, m0, n0, lane, wm, wn);
        } else store_partial2(p.ws, g, slot, acc, tid);
        pos += se - sb;
        slot = 1;
    }
}

// fixup of the stream-K split: one workgroup per (split tile, 32 x 32 block index of the wave tile), as conv_fixup_kernel
template <int KIND, int WM>
__global__ __launch_bounds__(NT2) void conv_h2_fixup_kernel(const IgemmParams p, int G) {
    constexpr int WNW = 8 / WM, MB = 4, NB = 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = WM == 1 ? 0 : wv / WNW, wn = WM == 1 ? wv : wv % WNW;
    const int tile = p.whole + blockIdx.x / (MB * NB), blk = blockIdx.x % (MB * NB), bi = blk / NB, bj = blk - bi * NB;
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, G, p.whole);
    const int first = tile * p.nslab, last = first + p.nslab - 1;
    const int g0 = split_owner(sp, first), g1 = split_owner(sp, last);
    if (g0 == g1 && split_lo(sp, g0) <= first && split_lo(sp, g0 + 1) > last) return;   // computed whole by one workgroup
    AccT<1, 1> acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.c[0][0][r] = 0.f;
    for (int g = g0; g <= g1; ++g) {
        const int slot = (split_lo(sp, g) / p.nslab == tile) ? 0 : 1;
        const float* src = p.ws + ((long)(g * 2 + slot) * ACC_REGS) * NT2 + tid;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.c[0][0][r] += src[(blk * 16 + r) * NT2];
    }
    // the epilogues place block (0, 0) of wave (wm, wn) at m0 + wm * 32, n0 + wn * 32 (MB = NB = 1): shift the origin to block (bi, bj)
    const int m0 = (tile / p.tilesN) * (128 * WM) + wm * (MB - 1) * 32 + bi * 32;
    const int n0 = (tile % p.tilesN) * p.tn_stride + wn * (NB - 1) * 32 + bj * 32;
    if (KIND == 0) epilogue_f<0, 1, 1>(p, acc, m0, n0, lane, wm, wn);
    else epilogue_t<0, 1, 1>(p, acc, m0, n0, lane, wm, wn);
}

template <int KW, int S, bool TK, int WM>
hipError_t launch2(const IgemmParams& p, int grid, hipStream_t st) {
    constexpr int KWP = TK ? KW / S : KW, TJ = KWP < 32 ? KWP : 32, NQ = 32 / TJ, SC = TK ? 1 : S, TN = 64 * (8 / WM);
    constexpr int NI = (NQ * h2_rsd(SC, TN) / 4 + 63) / 64, STG = 128 * WM * 16 + NI * 256;
    constexpr int lds_bytes = H2_RING * h2_spb(STG) * STG * 4;
    // (the attribute belongs to (function, current device): set on every call, nothing cached between calls)
    hipError_t e = hipFuncSetAttribute((const void*)conv_h2_kernel<KW, S, TK, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((conv_h2_kernel<KW, S, TK, WM>), dim3(grid), dim3(NT2), lds_bytes, st, p);
    return hipGetLastError();
}

template <int WM>
hipError_t launch2_wm(int kind, const IgemmParams& p, int grid, hipStream_t st) {
    if (kind == KIND_F) {
        if (p.k == 32) return launch2<32, 2, false, WM>(p, grid, st);
        if (p.k == 8 && p.s == 1) return launch2<8, 1, false, WM>(p, grid, st);
        if (p.k == 8) return launch2<8, 2, false, WM>(p, grid, st);
        return launch2<4, 2, false, WM>(p, grid, st);
    }
    if (p.k == 32) return launch2<32, 2, true, WM>(p, grid, st);
    if (p.s == 1) return launch2<8, 1, true, WM>(p, grid, st);
    return launch2<8, 2, true, WM>(p, grid, st);      // k = 8 and k = 5 (shadow padded to 4 taps per phase with zero weights)
}

}  // namespace

// do the windows of every possible tile of width `tn` fit the reserved dwords, and is the rows' zero tail long enough?
// (host-side mirror of the kernels' geometry; tn = 256: conv_h.hip and the 256 x 256 tile, tn = 512: the 128 x 512 tile)
bool pgconv::h_supported_tn(int kind, const IgemmParams& p, int tn) {
    const bool t = kind == KIND_T;
    if (kind == KIND_G) return false;
    if (t) { if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 5 && p.s == 2))) return false; }
    else if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 4 && p.s == 2))) return false;
    const int kwp = t ? pg_shadow_taps(p.k, p.s) : p.k, tj = kwp < 32 ? kwp : 32, nq = 32 / tj, sc = t ? 1 : p.s;
    if (p.Q % nq || (p.x_pitch & 1) || (p.x_bs & 1) || p.x_pitch <= p.Lx) return false;
    const int lcol = t ? p.U : p.Ly, rsd = h2_rsd(sc, tn);
    const int pos_mid = t ? p.u_off - (tj - 1) : -p.p, shm = pos_mid & 1;
    const int ndm = h2_round4((sc * (lcol - 1) + tj + shm + 1) >> 1);
    // elements of a row's neighbourhood a window piece can touch: [pos_mid - shm, pos_mid - shm + 2 ndm) for a sample's first
    // column at frame 0; a first segment that starts at column t0 ends no later (its window is the same one cut at t0, rounded
    // up to a piece: + 6 elements at most)
    const int left = pos_mid - shm < 0 ? -(pos_mid - shm) : 0;
    const int right = pos_mid - shm + 2 * ndm + 6 - p.Lx;
    const int tail = p.x_pitch - p.Lx;
    if (left > H_HEAD || tail < left || tail < right) return false;
    // the kernels lay the samples' windows out back to back (segment 0, full middle segments, last partial one): the worst
    // first-column position t0 must fit the reserved dwords
    int need = 0;
    for (int t0 = 0; t0 < lcol; ++t0) {
        const int nc0 = lcol - t0 < tn ? lcol - t0 : tn, rem = tn - nc0;
        const int sh0 = (pos_mid + sc * t0) & 1;
        int nmid = rem / lcol, nlast = rem - nmid * lcol;                        // full middle samples, columns of the last one
        if (1 + nmid + (nlast ? 1 : 0) > p.B) { nlast = 0; if (1 + nmid > p.B) nmid = p.B - 1; }   // only B samples exist
        const int n = h2_round4((sc * (nc0 - 1) + tj + sh0 + 1) >> 1) + nmid * ndm + (nlast ? (sc * (nlast - 1) + tj + shm + 2) >> 1 : 0);
        if (n > need) need = n;
    }
    return need <= rsd;
}

hipError_t pgconv::launch_h2(int kind, int wm, const IgemmParams& p, int grid, hipStream_t st) {
    return wm == 1 ? launch2_wm<1>(kind, p, grid, st) : launch2_wm<2>(kind, p, grid, st);
}

hipError_t pgconv::launch_h2_fixup(int kind, int wm, const IgemmParams& p, int grid, unsigned blocks, hipStream_t st) {
    if (kind == KIND_F) {
        if (wm == 1) hipLaunchKernelGGL((conv_h2_fixup_kernel<0, 1>), dim3(blocks), dim3(NT2), 0, st, p, grid);
        else hipLaunchKernelGGL((conv_h2_fixup_kernel<0, 2>), dim3(blocks), dim3(NT2), 0, st, p, grid);
    } else {
        if (wm == 1) hipLaunchKernelGGL((conv_h2_fixup_kernel<1, 1>), dim3(blocks), dim3(NT2), 0, st, p, grid);
        else hipLaunchKernelGGL((conv_h2_fixup_kernel<1, 2>), dim3(blocks), dim3(NT2), 0, st, p, grid);
    }
    return hipGetLastError();
}
