// conv_h.hip -- bf16-RESIDENT forward convolutions (BASELINE configs[4]: "bf16 MFMA convs"): conv forward (F) and transposed-
// conv forward (T, gather form) whose operands already live in HBM as bf16 -- activations (B, C, pitch) written by the
// producing layer's epilogue / BatchNorm, weights as a bf16 "shadow" of the fp32 master weights in the kernel's own
// GEMM layout [row][K] (pg_shadow_weights) -- multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// Why a separate kernel: in the fp32-tensor kernels' bf16 mode (conv_raw_impl.h, BF = 1) the matrix pipe is 16x faster but
// every operand still travels as fp32 -- per 16-deep slab a wave spends 256 cycles in MFMAs and ~580 in 36 LDS reads, 24
// conversions and the gather issue.  Here a slab is 32 deep, and per slab a wave issues
//   A: 8 x ds_read_b128 (weights: 8 consecutive k of a row are one aligned 16-byte read, no conversion),
//   B: 12 x ds_read2_b32 + 16 x v_alignbit (activations: raw bf16 row windows; a lane's 8 consecutive taps start at an
//      arbitrary ELEMENT, i.e. at a dword + 0 or 2 bytes: five dwords are read and funnel-shifted by the lane's parity),
// for 16 MFMAs.
//
// Layout contract for the activation operand (round 3: windows are gathered as 16-BYTE pieces, 8 elements each, with no
// per-element range check -- the zeros of the convolution's padding are READ, not synthesised):
//   * rows (b, c, :) are `x_pitch` elements apart, x_pitch and the batch stride even (pieces start at even elements);
//   * elements [L, x_pitch) of every row are ZERO, and that tail is long enough for this layer (pg_conv_fwd_h_supported says
//     so: the window of a row reaches at most `left` elements in front of it -- into the previous row's tail -- and `right`
//     behind element L - 1; tail >= max(left, right); 40 elements cover every layer of the U-Net, ops.h_pitch);
//   * the PG_H_HEAD (32) elements in front of x are readable and zero (row (0, 0) has no previous row's tail: ops.h_alloc
//     puts a zero head pad in front; a channel slice of a larger tensor has the previous channel's tail there).
// Producers (epilogues here, pg_bn_fwd, pg_cast_rows_bf16) only ever write elements [0, L) and keep the tails zero.
// K must be whole channels per slab (Q % (32 / min(taps, 32)) == 0); otherwise the caller uses the fp32-tensor path.
// Work decomposition, stream-K split, fixup kernels and epilogues are the shared ones (conv_common.h).
#include "conv_common.h"
#include "conv_h_frag.h"

namespace {

constexpr int KB = 32;                    // k per slab (two MFMA k-steps of 16)
constexpr int H_HEAD = 32;                // zero elements the caller guarantees in front of x (bytes: 64)

// window geometry shared by the kernel and the host-side check: dwords reserved per channel window
__host__ __device__ constexpr int h_rsd(int sc) { return sc == 1 ? 256 : 384; }
__host__ __device__ __forceinline__ int h_round4(int v) { return (v + 3) & ~3; }

template <int KW, int S, bool TKIND>
__global__ __launch_bounds__(NT, 2) void conv_h_kernel(const IgemmParams p) {
    constexpr int TM = RBM, TN = RBN;                 // workgroup tile 128 x 256
    constexpr int KWP = TKIND ? KW / S : KW;          // taps per channel in K order
    constexpr int TJ = KWP < 32 ? KWP : 32, NQ = 32 / TJ;
    constexpr int SC = TKIND ? 1 : S;                 // window elements per column step
    constexpr int RSD = h_rsd(SC);                    // dwords reserved per channel window (a multiple of 4: 16-byte pieces)
    constexpr int NP = NQ * RSD / 4;                  // 16-byte window pieces per slab; piece i fills dwords [4 i, 4 i + 4) of the
    constexpr int NI = (NP + 63) / 64;                //   [channel][RSD] image -> NI wave instructions, dealt round-robin to the waves
    constexpr int NPW = (NI + 3) / 4;                 // ... at most NPW per wave (D3: 3, D2 / U2: 2, the others 1)
    constexpr int TA = TM * 16;                       // dwords of the weight tile: 128 rows x 64 B (32 bf16)
    constexpr int STG = TA + NI * 256;                // window region rounded up to whole wave instructions: lanes past NP write zeros
    // slabs per barrier: two 32-deep slabs are gathered together and multiplied one after the other where both stage pairs fit
    // 64 KB (two workgroups per CU) -- halves the barrier / gather-burst rate; a bf16 slab is only 16 MFMAs (512 cycles) per wave
    constexpr int SPB = 4 * STG * 4 <= 64 * 1024 ? 2 : 1;
    constexpr int SSTG = SPB * STG;
    static_assert(KWP == 4 || KWP == 8 || KWP == 16 || KWP == 32, "taps per channel in K order");
    static_assert(2 * SSTG * 4 <= 64 * 1024, "LDS budget");
    static_assert(RSD % 4 == 0 && TA % 4 == 0, "16-byte aligned window images");
    __shared__ __attribute__((aligned(16))) float lds[2 * SSTG];
    const int tid = threadIdx.x, lane = tid & 63;
    // waves 1 (M) x 4 (N): a wave owns ALL 128 rows of the tile and 64 of its columns (4 x 2 blocks).  The B fragments -- scalar
    // window reads + funnel shifts, the expensive operand here -- are then shared by four row blocks instead of two: per slab
    // 12 ds_read2_b32 + 16 v_alignbit + 8 ds_read_b128 instead of 24 + 32 + 4 for the same 16 MFMAs
    constexpr int MBW = 4, NBW = 2;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = 0, wn = wv;
    const int r = lane & 31, h = lane >> 5;
    const int Lcol = TKIND ? p.U : p.Ly;
    const int Ktot = p.Q * KWP, Mrows = TKIND ? p.M * S : p.M;
    // the activation descriptor starts H_HEAD elements in front of x (zero by contract): every window offset below is >= 0
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
        // ---- segments (samples) of the tile and their window geometry, in DWORDS of bf16 pairs --------------------------
        // segment k holds the columns of sample b0 + k.  Its window starts at memory element pos_k (first tap of its first
        // column), loaded from the even element pe_k = pos_k - sh_k below it.  Every middle segment has the same pos (frame 0).
        // Segment windows are whole 16-byte pieces (dword counts rounded up to 4) laid back to back: [seg 0 | full middle ... | last].
        const int nc0 = min(Lcol - t0, TN);                                       // columns of segment 0
        const int pos_first = TKIND ? p.u_off + t0 - (TJ - 1) : S * t0 - p.p;     // memory element of window element 0, segment 0
        const int pos_mid = TKIND ? p.u_off - (TJ - 1) : -p.p;                    //   ... of the later segments (frame 0)
        const int sh0 = pos_first & 1, shm = pos_mid & 1;                         // (two's complement: also right for negatives)
        const int nd0 = h_round4((SC * (nc0 - 1) + TJ + sh0 + 1) >> 1);           // dwords of segment 0's window
        const int ndm = h_round4((SC * (Lcol - 1) + TJ + shm + 1) >> 1);          //   ... of a full middle segment

        // ---- weight-tile gather: 16-byte pieces of the K-contiguous bf16 rows, swizzled image as in the fp32 kernels ------
        int avoff[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int m = m0 + dma16_row(lane, wv, e);
            avoff[e] = m < Mrows ? m * Ktot * 2 + dma16_kc(lane) * 4 : FAR;
        }
        // ---- window gather: wave instruction j = wv + 4 e carries pieces 64 j .. 64 j + 63; this lane's piece i = 64 j + lane
        // is dwords [4 pc, 4 pc + 4) of channel qi = i / (RSD / 4).  Its source is fixed for the tile up to the slab's first
        // channel, which rides in the SGPR offset.  No range check: zeros come from the rows' tails (contract above).
        int voff[NPW];
#pragma unroll
        for (int e = 0; e < NPW; ++e) {
            const int i = 64 * (wv + 4 * e) + lane, qi = i / (RSD / 4), d = 4 * (i - qi * (RSD / 4));
            int k, dl;
            if (d < nd0) { k = 0; dl = d; } else { k = 1 + (d - nd0) / ndm; dl = (d - nd0) - (k - 1) * ndm; }
            const int e0 = (k ? pos_mid - shm : pos_first - sh0) + 2 * dl;       // even memory element of the piece's first dword
            const int b = b0 + k;
            const bool ok = i < NP && b < p.B && k * Lcol < t0 + TN;             // the sample exists and has columns in this tile
            voff[e] = ok ? (b * (int)p.x_bs + qi * p.x_pitch + e0 + H_HEAD) * 2 : FAR;
        }
        // ---- fragment bases: dword and parity of window element 0 of each of this lane's 2 column blocks -------------------
        int bdw[NBW], bsh[NBW];
#pragma unroll
        for (int jb = 0; jb < NBW; ++jb) {
            const int c = wn * (NBW * 32) + jb * 32 + r, seg = (t0 + c) / Lcol;
            const int cin = seg ? (t0 + c) - seg * Lcol : c;                      // column inside its segment
            const int el = SC * cin + (seg ? shm : sh0);                          // window element of tap 0
            bdw[jb] = (seg ? nd0 + (seg - 1) * ndm : 0) + (el >> 1);
            bsh[jb] = (el & 1) << 4;                                              // funnel shift in bits
        }
        AccT<MBW, NBW> acc;
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
            for (int j = 0; j < NBW; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc.c[i][j][q] = 0.f;

#define H_ISSUE(STAGE_PTR, SLAB)                                                                              \
    {   float* const As = (STAGE_PTR) + wv * 64; float* const Bw = (STAGE_PTR) + TA;                         \
        const int k0 = (SLAB) * KB;                                                                          \
        if (k0 < Ktot) {                                                                                     \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) dma16s(rw, As + wv * 192 + e * 1024, avoff[e], k0 * 2); \
            const int sq = (k0 / KWP) * p.x_pitch * 2;                                                       \
            _Pragma("unroll") for (int e = 0; e < NPW; ++e)                                                  \
                if (wv + 4 * e < NI) dma16s(rx, Bw + 256 * (wv + 4 * e), voff[e], sq);                       \
        }                                                                                                    \
    }

#pragma unroll
        for (int hf = 0; hf < SPB; ++hf) H_ISSUE(lds + hf * STG, sb + hf)
        __syncthreads();
        for (int sl = sb; sl < se; sl += SPB) {
            const int cur = ((sl - sb) / SPB) & 1;
#pragma unroll
            for (int hf = 0; hf < SPB; ++hf) H_ISSUE(lds + (cur ^ 1) * SSTG + hf * STG, sl + SPB + hf)
            __builtin_amdgcn_sched_barrier(0);
            h_mma_group<TJ, RSD, TA, STG, SPB>(lds + cur * SSTG, min(SPB, se - sl), r, h, wm, bdw, bsh, acc);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
#undef H_ISSUE
        if (sb == 0 && se == p.nslab) {
            if (TKIND) epilogue_t<S, MBW, NBW>(p, acc, m0, n0, lane, wm, wn);
            else epilogue_f<S, MBW, NBW>(p, acc, m0, n0, lane, wm, wn);
        } else store_partial(p.ws, g, slot, acc, tid);
        pos += se - sb;
        slot = 1;
    }
}

template <int KW, int S, bool TK>
hipError_t launch1(const IgemmParams& p, int grid, hipStream_t st) {
    hipLaunchKernelGGL((conv_h_kernel<KW, S, TK>), dim3(grid), dim3(NT), 0, st, p);
    return hipGetLastError();
}

}  // namespace

// (the host-side mirror of the window geometry -- pgconv::h_supported_tn -- lives in conv_h2.hip, shared by both tile families)

hipError_t pgconv::launch_h(int kind, const IgemmParams& p, int grid, hipStream_t st) {
    if (kind == KIND_F) {
        if (p.k == 32) return launch1<32, 2, false>(p, grid, st);
        if (p.k == 8 && p.s == 1) return launch1<8, 1, false>(p, grid, st);
        if (p.k == 8) return launch1<8, 2, false>(p, grid, st);
        return launch1<4, 2, false>(p, grid, st);
    }
    if (p.k == 32) return launch1<32, 2, true>(p, grid, st);
    if (p.s == 1) return launch1<8, 1, true>(p, grid, st);
    return launch1<8, 2, true>(p, grid, st);      // k = 8 and k = 5 (shadow padded to 4 taps per phase with zero weights)
}
