// conv_h_frag.h -- fragment phase shared by the bf16-resident forward kernels (conv_h.hip: 4 waves, conv_h2.hip: 8 waves): a wave
// owns 128 rows x 64 columns of its workgroup's tile (4 x 2 blocks of 32 x 32); `wm` = index of its 128-row band in the weight tile.
#pragma once
#include "conv_common.h"

namespace {

typedef short s16x8 __attribute__((ext_vector_type(8)));

// ---- fragments of one MFMA k-step (k = 16 s + 8 h + 0..7 of a 32-deep slab), software-pipelined ---------------------------
// hipcc's own schedule of the fragment phase re-used one register quad for all four A reads of a k-step and waited lgkmcnt(0)
// in front of every MFMA pair: each pair (64 pipe cycles) exposed one LDS round trip, which the partner wave's MFMAs only cover
// while the LDS is idle -- not while LDS-DMA writes of the next stage stream in.  Here a k-step's ten LDS reads are issued one
// k-step AHEAD, in front of the previous k-step's eight MFMAs, and the funnel shifts that finish the B fragments run behind
// them: no MFMA waits for an LDS read issued less than eight MFMAs (256 pipe cycles) earlier.
struct HRaw { f32x4 a[4]; unsigned d[2][6]; };      // a k-step's fragments as they come out of LDS
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

template <int TJ, int RSD, int TA>
__device__ __forceinline__ void h_load_raw(const float* stage, int s, int r, int h, int wm, const int (&bdw)[2], HRaw& f) {
    typedef const __attribute__((address_space(3))) unsigned* lds_u32;
    const lds_u32 Bd = (lds_u32)(stage + TA);
    const int sw = (r >> 2) & 3;
    const float* ap = stage + (wm * 128 + r) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * 16 + (((2 * s + h) ^ sw) << 2));
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        if (TJ >= 8) {
            const int qi = TJ == 32 ? 0 : (TJ == 16 ? s : 2 * s + h);
            const int tap0 = TJ == 32 ? 16 * s + 8 * h : (TJ == 16 ? 8 * h : 0);
            lds_u32 bp = Bd + qi * RSD + bdw[jb] + (tap0 >> 1);
            asm volatile("" : "+v"(bp));        // ONE address per fragment: the five dwords then pair into ds_read2_b32 with 8-bit offsets
#pragma unroll                                  // (folded into the stage constants, hipcc built an address per read pair)
            for (int i = 0; i < 5; ++i) f.d[jb][i] = bp[i];
        } else {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                lds_u32 bp = Bd + (4 * s + 2 * h + cc) * RSD + bdw[jb];
                asm volatile("" : "+v"(bp));
#pragma unroll
                for (int i = 0; i < 3; ++i) f.d[jb][3 * cc + i] = bp[i];
            }
        }
    }
}

// the B fragments of a k-step: funnel shift of the window dwords by the lane's parity
template <int TJ>
__device__ __forceinline__ void h_finish(const HRaw& f, const int (&bsh)[2], u32x4v (&b)[2]) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        if (TJ >= 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[jb][i] = __builtin_amdgcn_alignbit(f.d[jb][i + 1], f.d[jb][i], bsh[jb]);
        } else {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int i = 0; i < 2; ++i) b[jb][2 * cc + i] = __builtin_amdgcn_alignbit(f.d[jb][3 * cc + i + 1], f.d[jb][3 * cc + i], bsh[jb]);
        }
    }
}

__device__ __forceinline__ void h_mfma(const HRaw& f, const u32x4v (&b)[2], AccT<4, 2>& acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
            acc.c[i][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i]), __builtin_bit_cast(bf16x8, b[jb]), acc.c[i][jb], 0, 0, 0);
}

// the 2 * nsl k-steps of nsl consecutive slabs (LDS stages STG floats apart), pipelined one k-step deep on two register sets
// (even / odd k-steps, no copies): per k-step the ten LDS reads of the NEXT one are issued, then -- behind a scheduling fence --
// the eight MFMAs of the current one, then the funnel shifts of the next one's B fragments, whose wait for the reads therefore
// sits behind the MFMAs.  (The read past the last slab re-reads that slab: a valid address, never multiplied.)
struct HNoIssue { __device__ __forceinline__ void operator()(int) const {} };

// The 2 * nsl k-steps of nsl <= NSL consecutive slabs (LDS stages STG floats apart), fully unrolled over the slabs so that every
// LDS read address is a per-lane base (set up once per group by the caller-independent prologue here) plus an IMMEDIATE -- with a
// rolled slab loop hipcc spent one v_add per read (~32 VALU per slab, a third of a wave's non-MFMA issue time) -- and pipelined
// one k-step deep on two register sets (even / odd k-steps, no copies): per k-step the ten LDS reads of the NEXT one are issued
// between the first MFMAs of the current one and the funnel shifts of the next one's B fragments between its last MFMAs.
// Instructions a wave issues BETWEEN its own MFMAs are free (an MFMA holds the vector issue for 8 of its 32 cycles); what a wave
// issues outside its bursts is exposed once per iteration even with a partner wave on the SIMD (utilisation =
// B / (B + t_n)).  `issue(hf)` is called once per slab, between its two k-steps: conv_h2.hip issues the gathers of slab hf of the
// stage group two ahead there, so that its LDS-DMA instructions (60-185 cycles of issue each) sit inside the MFMA stream as well.
// (The read past the last slab of the group re-reads that slab: a valid address, never multiplied.)
template <int TJ, int RSD, int TA, int STG, int NSL, typename Issue = HNoIssue>
__device__ __forceinline__ void h_mma_group(const float* stage0, int nsl, int r, int h, int wm, const int (&bdw)[2], const int (&bsh)[2],
                                            AccT<4, 2>& acc, const Issue& issue = Issue()) {
    constexpr int NDS = 4 + 2 * (TJ >= 8 ? 3 : 4);          // LDS read instructions of a k-step
#define H_INTERLEAVE                                                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, NDS - 8, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    HRaw f0, f1;
    u32x4v b0[2], b1[2];
    h_load_raw<TJ, RSD, TA>(stage0, 0, r, h, wm, bdw, f0);
    h_finish<TJ>(f0, bsh, b0);
#pragma unroll
    for (int hf = 0; hf < NSL; ++hf) {
        if (hf < nsl) {
            const float* st = stage0 + hf * STG;
            const float* nx = hf + 1 < NSL ? st + STG : st;
            __builtin_amdgcn_sched_barrier(0);
            h_load_raw<TJ, RSD, TA>(st, 1, r, h, wm, bdw, f1);
            h_mfma(f0, b0, acc);
            h_finish<TJ>(f1, bsh, b1);
            H_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
            issue(hf);
            __builtin_amdgcn_sched_barrier(0);
            h_load_raw<TJ, RSD, TA>(nx, 0, r, h, wm, bdw, f0);
            h_mfma(f1, b1, acc);
            h_finish<TJ>(f0, bsh, b0);
            H_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef H_INTERLEAVE
}

}  // namespace
