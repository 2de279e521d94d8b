// conv_igemm.hip -- the six 1-D convolution passes of the U-Net as three implicit-GEMM kernels on the
// gfx950 matrix cores (default v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain; optional bf16-pipe operand
// modes, pg_conv_args.precision).
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
// Two generations of kernels, selected per launch by the host (launch() below); sources:
//   conv_common.h        problem descriptor, LDS-DMA helpers, MFMA operand modes, stream-K split, epilogues
//   conv_raw_impl.h      RAW-WINDOW F/T kernel template; conv_raw.hip instantiates the 128 x 256 tile (training), conv_raw_tall.hip the
//                        256 x 128 tile (few columns: small-batch inference)
//   conv_raw*.hip        RAW-WINDOW F/T kernels  } the fast path for every layer geometry of the U-Net except k = 5:
//   conv_raw_wgrad.hip   RAW-WINDOW G kernel     } workgroup tile 128 (M) x 256 (N), 4 waves of 64 x 128 (128 accumulator
//                        registers, 2 waves/SIMD).  The weight / P tile is gathered by LDS-DMA into a swizzled K-contiguous
//                        image; the ACTIVATION operand is staged as raw row windows (every element once) and the im2col
//                        overlap is resolved when fragments are read: ~60 % fewer global->LDS bytes and far fewer gather
//                        instructions per MFMA than an im2col tile.
//   conv_im2col.hip      IM2COL kernels (conv_f/t/g_kernel): 256 x 128 tile, both operands gathered element by element by
//                        LDS-DMA with per-lane source addresses (im2col, phase split, zero padding and the XOR swizzle all
//                        live in the address).  They serve k = 5, generic (k, s) and shapes whose windows do not fit, and
//                        stay covered by the tests (schedule bit 2).
//   conv_igemm.hip       (this file) fixup kernel of the stream-K split, grid policy, geometry checks, the C ABI entry points.
//
// Common to both: operands reach LDS through buffer_load ... lds (no staging registers, no ds_write; out-of-range lanes
// write 0.0, which implements conv padding, tile edges and K tails); (Leaky)ReLU in front of every conv is applied
// branch-free on the MFMA fragments, so the in-place activations (model.py:80,82) and torch.cat (model.py:113) are
// never materialised; phase order per slab is pinned with sched_barrier(0): gathers for slab s+1, then fragment reads
// + MFMAs of slab s, then one __syncthreads() whose vmcnt(0) therefore sits behind the matrix work; double-buffered
// LDS.  Work decomposition is a persistent stream-K split over (tile, slab) with a deterministic fixup kernel (below).
// Each output element is accumulated in a fixed order: results are bit-reproducible.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cmath>
#include "conv_common.h"

namespace {

// ---- fixup: add the partial segments of every split tile in ascending workgroup order, then the epilogue ---------
// One workgroup per (tile, 32 x 32 block of the wave tile): with few tiles and many segments (small-batch inference: 8 tiles
// split over 512 workgroups) one workgroup per tile would read 8 MB on its own; per block the reduction is MB*NB times wider.
// WN = waves of the GEMM kernel along N (2: tile 64 MB x 64 NB; 1: the raw "tall" tile 128 MB x 32 NB).
// WIDE (small-batch inference: tens of segments per tile): four workgroups per block, one per wave of the GEMM kernel; the four
// waves of a fixup workgroup each sum every fourth segment (two segments' loads in flight per wave) and wave 0 adds the four
// sums in order -- a fixed order, chosen by the host from (grid, tiles) alone.  8 tiles x 64 segments: 36 us -> see DESIGN 4.3.
template <int KIND, int MB, int NB, int WN = 2, bool WIDE = false>
__global__ __launch_bounds__(NT) void conv_fixup_kernel(const IgemmParams p, int G) {
    // WN = waves of the GEMM kernel along N: 2 -> 2 x 2 waves, 1 -> 4 x 1 (tall raw tile), 4 -> 1 x 4 (bf16-resident kernels)
    __shared__ float red[WIDE ? 3 * 16 * 64 : 1];
    const int unit = WIDE ? blockIdx.x >> 2 : blockIdx.x, q = WIDE ? threadIdx.x >> 6 : 0;
    const int tid = WIDE ? (blockIdx.x & 3) * 64 + (threadIdx.x & 63) : threadIdx.x;     // the GEMM thread whose accumulators this lane sums
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = WN == 2 ? wv >> 1 : (WN == 4 ? 0 : wv), wn = WN == 2 ? wv & 1 : (WN == 4 ? wv : 0);
    static_assert(MB * NB * 16 == ACC_REGS, "8 blocks per wave tile: the host launches 8 workgroups per tile");
    // hybrid split: tiles below p.whole were computed whole by one workgroup each -- the grid covers the split tiles only
    const int tile = p.whole + unit / (MB * NB), blk = unit % (MB * NB), bi = blk / NB, bj = blk - bi * NB;
    const Split sp = make_split(p.tilesM * p.tilesN, p.nslab, G, p.whole);
    const int first = tile * p.nslab, last = first + p.nslab - 1;
    const int g0 = split_owner(sp, first), g1 = split_owner(sp, last);
    if (g0 == g1 && split_lo(sp, g0) <= first && split_lo(sp, g0 + 1) > last) return;   // computed whole by one workgroup
    // a 32-column block that starts past the problem's last column holds nothing (few-column problems on the tall tile: the GEMM
    // kernel did not write it either)
    if (KIND != 2 && p.n_lo + (tile % p.tilesN) * p.tn_stride + (WN == 2 ? wn * (NB - 1) * 32 : (WN == 4 ? wn * (NB * 32) : 0)) + bj * 32 >= p.B * (KIND == 0 ? p.Ly : p.U)) return;
    AccT<1, 1> acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.c[0][0][r] = 0.f;
#pragma unroll 2
    for (int g = g0 + q; g <= g1; g += WIDE ? 4 : 1) {
        const int slot = (split_lo(sp, g) / p.nslab == tile) ? 0 : 1;     // the range's first segment, or its last
        const float* src = p.ws + ((long)(g * 2 + slot) * ACC_REGS) * NT + tid;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.c[0][0][r] += src[(blk * 16 + r) * NT];
    }
    if (WIDE) {         // (the returns above are uniform over the workgroup here: all four waves stand for the same GEMM wave)
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
    // the epilogues place block (0, 0) of wave (wm, wn) at m0 + wm * 32, n0 + wn * 32: shift the origin to block (bi, bj)
    const int m0 = (tile / p.tilesN) * ((4 / WN) * 32 * MB) + wm * (MB - 1) * 32 + bi * 32;
    // column tiles are p.tn_stride columns apart: the tile width, except the k = 5 wgrad's 255 (51 whole channels) of 256
    const int nt0 = p.n_lo + (tile % p.tilesN) * p.tn_stride;
    const int n0 = nt0 + wn * (NB - 1) * 32 + bj * 32;
    if (KIND == 0) epilogue_f<0, 1, 1>(p, acc, m0, n0, lane, wm, wn);
    else if (KIND == 3)       // stride-2 raw T kernels (phase-major rows): block row bi is phase bi of the wave's 32 output channels
        epilogue_t_pm<1, 1>(p, acc, (tile / p.tilesN) * ((4 / WN) * 16 * MB) + wm * 32, nt0 + wn * (NB * 32) + bj * 32, lane, bi);
    else if (KIND == 1) epilogue_t<0, 1, 1>(p, acc, m0, n0, lane, wm, wn);
    else epilogue_g<0, 1, 1>(p, acc, m0, n0, lane, wm, wn, nt0 + p.tn_stride);
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
constexpr int WG_PER_CU = 2;                    // <= 256 VGPR+AGPR per lane -> 2 waves per SIMD; 48 KB LDS per workgroup
constexpr int MAX_STREAMK_WG = 2048;            // bound on the persistent grid (sizes the caller's workspace)
constexpr long WS_PER_WG = 2L * ACC_REGS * NT * 4;   // two partial tiles of 256x128 fp32 per workgroup

// Per-call knobs decoded from pg_conv_args.precision / .schedule (no process-wide state: two streams or threads can run
// different precisions and schedules concurrently).
struct Knobs {
    int prec;         // PG_PREC_*: 0 fp32 MFMA, 1 bf16 operands, 2 bf16x3 split (all fp32 accumulate)
    int force_mode;   // work split: 0 automatic, 1 one tile per workgroup, 2 force stream-K
    int no_raw;       // 1 = never use the raw-window kernels (exercise the im2col kernels)
    int no_tall;      // 1 = never use the tall 256 x 128 raw tile
    int oversub;      // stream-K grid = up to oversub x resident workgroup slots
    int contended;    // other kernels (RCCL collectives) are expected to hold part of the chip: always take the finer split
    int no_ps;        // wgrad: 1 = never the per-sample-slab kernel (A/B, tests)
    int no_raw3;      // 1 = never the one-wave-per-SIMD fp32 kernels (conv_raw3.hip; schedule bit 13: A/B, tests of the older kernels)
    int all_raw3;     // 1 = the one-wave-per-SIMD kernels wherever they cover the problem (bit 14), also where auto prefers the older ones
    int sr;           // conv_raw3 tile order: super-row height forced by schedule bits 15-16 (0 = default)
    int force_colsplit;   // 1 = split wherever the geometry allows, whatever the cost model says (bit 18: tests reach the tail launch on small problems)
    int no_colsplit;  // 1 = never split the columns past the last full 256-wide tile off into a tail launch (bit 17: A/B, tests)
    char* desc; int desc_len;   // pg_conv_describe: write the launch plan here INSTEAD of launching
};
int decode_knobs(const pg_conv_args* a, Knobs& k) {
    if (a->precision < 0 || a->precision > 2) return pg_fail(PG_ERR_UNSUPPORTED, "conv: precision must be PG_PREC_FP32, PG_PREC_BF16 or PG_PREC_BF16X3");
    const int sc = a->schedule;
    if (sc < 0 || (sc & ~0x7ffff) || ((sc >> 17) & 3) == 3 || (sc & 3) == 3 || ((sc >> 8) & 15) > 8) return pg_fail(PG_ERR_SHAPE, "conv: bad schedule bits");
    k.prec = a->precision;
    k.force_mode = sc & 3; k.no_raw = (sc >> 2) & 1; k.no_tall = (sc >> 3) & 1;
    k.oversub = (sc >> 8) & 15; if (!k.oversub) k.oversub = 4;
    k.contended = (sc >> 4) & 1;
    // bits 5-6 selected the two-waves-per-SIMD / eight-wave tile families of pg_conv_fwd_h until 0.3 (conv_h.hip, conv_h2.hip): no
    // automatic choice reached them once conv_h3 was the default for every layer, and they are gone (0.4); bit 12 (conv_h3) is a no-op
    if ((sc >> 5) & 3) return pg_fail(PG_ERR_UNSUPPORTED, "conv: schedule bits 5-6 (tile families removed in 0.4)");
    k.no_ps = (sc >> 7) & 1;
    k.no_raw3 = (sc >> 13) & 1;
    k.all_raw3 = (sc >> 14) & 1;
    k.sr = ((sc >> 15) & 3) ? 1 << (((sc >> 15) & 3) - 1) : 0;      // bits 15-16: 1 -> R = 1 (row-major), 2 -> 2, 3 -> 4; 0 = default
    k.no_colsplit = (sc >> 17) & 1;
    k.force_colsplit = (sc >> 18) & 1;
    if (k.no_raw3 && k.all_raw3) return pg_fail(PG_ERR_SHAPE, "conv: schedule bits 13 and 14 exclude each other");
    k.desc = nullptr; k.desc_len = 0;
    return PG_OK;
}

int cu_count() { return pg_cu_count(); }
// the wide fixup (four workgroups per 32 x 32 block, segments summed four abreast) from 8 segments per split tile on
bool fixup_wide(int grid, long split_tiles) { return split_tiles > 0 && grid >= 8 * split_tiles; }

// Grid policy.  Default: a persistent stream-K grid of up to oversub (4) x the resident workgroup slots, each
// workgroup owning an equal contiguous range of the (tile, slab) space, plus the fixup launch.  Measured on MI355X
// (tools/contention.py): the oversubscribed split costs nothing on a free chip, removes tile-count quantisation, and --
// what matters for data-parallel training, where RCCL's collective kernels hold part of the chip during backward --
// degrades gracefully when slots are taken (16 of 512 slots held: 1x split 33 -> 58 ms, one-tile-per-workgroup 32 -> 43 ms,
// 4x split 33 -> 37 ms).  Small problems (less than 8 slabs per resident slot, or no workspace) run one tile per
// workgroup.  mode: 0 auto, 1 force one tile per workgroup, 2 force stream-K (tests).
int pick_grid(long tiles, int nslab, IgemmParams& p, long ws_bytes, int mode, int oversub, int contended, int wg_per_cu = WG_PER_CU,
              long ws_per_wg = WS_PER_WG) {
    const long total = tiles * (long)nslab;
    const long slots = (long)cu_count() * wg_per_cu;
    p.whole = 0;
    // A tile count that is a whole multiple of the resident slots quantises perfectly: whole tiles per workgroup, no partial
    // tiles through the workspace and no fixup launch (measured: the fixups of the five such layers of the U-Net cost 0.5 ms
    // per step).  Not when the chip is shared (data-parallel backward beside RCCL): there the finer split bounds the tail.
    if (mode == 0 && !contended && tiles % slots == 0) {
        if (tiles <= slots * oversub) return (int)tiles;
        for (long mult = oversub; mult >= 1; --mult)
            if (tiles % (slots * mult) == 0) return (int)(slots * mult);       // several whole tiles per workgroup
    }
    long mult = total / (256 * slots);               // whole multiples of the slot count only (a ragged second wave is
    if (mult > oversub) mult = oversub;              // worse than none), and >= 256 slabs per workgroup so that partial-
    if (mult < 1) mult = 1;                          // tile traffic stays negligible
    long G = slots * mult;
    if (G > MAX_STREAMK_WG) G = (MAX_STREAMK_WG / slots) * slots;
    if (G > total) G = total;
    const bool can = p.ws && ws_bytes >= G * ws_per_wg && total < 0x7fffffffL;
    if (mode == 1 || !can) return (int)tiles;
    if (mode == 2) return (int)G;
    // Hybrid: a tile count slightly above a multiple of the slots (1056, 528) runs its full waves as whole tiles and splits only
    // the remainder over one more wave of workgroups: the same balance as the even split with (almost) no partial tiles -- the
    // fixup then touches 32 tiles instead of 1056.  Not when the chip is shared (see above).
    if (!contended && tiles > slots) {
        const long whole = tiles / slots * slots, rem = tiles - whole;
        if (rem * nslab >= slots * 8 && whole + slots <= MAX_STREAMK_WG && ws_bytes >= (whole + slots) * ws_per_wg) {
            p.whole = (int)whole;
            return (int)(whole + slots);
        }
    }
    return total >= slots * 8 ? (int)G : (int)tiles;
}

// raw-window kernels: supported (k, s) pairs and the window-length bound
// k = 5, s = 2 runs the raw-window kernels as a virtual k = 8 (conv_raw_impl.h): taps of the weight image per (row, channel)
inline int virtual_k(int k, int s) { return (k == 5 && s == 2) ? 8 : k; }

bool raw_supported(Kind kind, const IgemmParams& p, const Knobs& kn, int tn = RBN) {
    if (kn.no_raw == 1) return false;
    int kwp, sc, lcol;
    const bool k5 = p.k == 5 && p.s == 2;
    if (kind == KIND_F) {
        if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 4 && p.s == 2) || (k5 && p.Q % 2 == 0))) return false;
        kwp = virtual_k(p.k, p.s); sc = p.s; lcol = p.Ly;       // (k = 5: a slab is two whole channels; an odd channel count stays on im2col)
    } else if (kind == KIND_T) {
        if (!((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 4 && p.s == 2) || k5)) return false;
        kwp = virtual_k(p.k, p.s) / p.s; sc = 1; lcol = p.U;
    } else {
        // a 16-element slab of (b, i) may run over at most ONE sample boundary in the raw-window wgrad kernel
        return p.LP >= 16 && ((p.k == 32 && p.s == 2) || (p.k == 8 && p.s == 1) || (p.k == 8 && p.s == 2) || (p.k == 4 && p.s == 2) || k5);
    }
    const int tj = kwp < 16 ? kwp : 16;
    const int nseg_max = (lcol - 1 + tn - 1) / lcol + 1;
    return sc * (tn - 1) + tj + raw_gap(tj) * (nseg_max - 1) + (kind == KIND_T ? tj : 0) <= (sc == 1 ? RS1 : RS2);
}

int launch(Kind kind, IgemmParams& p, const Knobs& kn, long rows, long cols, long Ktot, long ws_bytes, hipStream_t st, bool may_split = true) {
    const long Ktot_in = Ktot;
    bool raw = raw_supported(kind, p, kn);
    // F / T problems whose columns the tall 256 x 128 tile covers with at least 3 % fewer computed ones take it (and those whose
    // windows only fit the narrower tile: many short samples per tile): small-batch
    // inference above all (a 128 x 256 tile over 65 columns is 3/4 idle MFMA work per weight byte), and training shapes such as
    // N = 16 x 65 (5 wide tiles = 1280 columns vs 9 tall = 1152: +15 % measured) or 64 x 30.  On ties the wide tile wins (it
    // runs two slabs per barrier; measured 1-7 % faster at equal column counts).
    const long cols_wide = (cols + RBN - 1) / RBN * RBN, cols_tall = (cols + RBN / 2 - 1) / (RBN / 2) * (RBN / 2);
    const bool tall = kind != KIND_G && kn.no_raw == 0 && kn.no_tall == 0 && (cols_tall * 100 <= cols_wide * 97 || !raw) &&
                      raw_supported(kind, p, kn, RBN / 2);
    if (tall) raw = true;
    // fp32 F / T problems the one-wave-per-SIMD kernels cover (conv_raw3.hip: 256 x 256 tile) take them, unless 256-row tiles
    // would compute over 3 % more rows than 128-row ones
    // Over the tall tile too where 256-wide tiles compute at most 8 % more columns (D2 forward / U2 dgrad / D3 forward at batch 64:
    // -6 / -2 / -6 %, and 2.2 instead of 6.95 GB of L2 fills; batch-1 inference and N = 16 x 65 keep the tall tile).
    const bool r3_over_tall = tall && raw_supported(kind, p, kn) && (kn.all_raw3 || cols_wide * 100 <= cols_tall * 108);
    const bool r3_ok = raw && raw_supported(kind, p, kn) && kind != KIND_G && kn.prec == 0 && !kn.no_raw3 && pgconv::raw3_covers(kind, p) &&
                       (rows + 255) / 256 * 256 * 100 <= (rows + RBM - 1) / RBM * RBM * 103;
    const bool r3 = r3_ok && (!tall || r3_over_tall);
    const int bm = (tall || r3) ? 2 * RBM : (raw ? RBM : BM);
    int bn = (tall && !r3) ? RBN / 2 : (raw ? RBN : BN);
    const bool k5 = raw && p.k == 5 && p.s == 2;
    if (k5 && kind == KIND_G) bn = (RBN / 5) * 5;               // wgrad: a column tile is 51 whole channels x 5 taps = 255 columns (+ 1 idle)
    p.g_ps = 0;
    if (raw && kind == KIND_G && p.k != 32) {
        // short samples: slabs of 16 frames of ONE sample (conv_g_ps_kernel) where padding every sample to whole slabs costs <= 7 %
        // of MFMA work (30 frames: 6.7 %, 61: 4.9 %, 126: 1.6 %, 256: none; 129 would cost 11.6 % and keeps the flat K axis)
        const long cps = (p.LP + 15) / 16;
        if ((cps * 16 - p.LP) * 100 <= 7L * p.LP && !kn.no_ps) { p.g_ps = 1; Ktot = (long)p.B * cps * 16; }
    }
    if (k5 && kind != KIND_G) Ktot = (long)p.Q * (kind == KIND_T ? 4 : 8);   // F / T: K runs over the virtual taps
    // Column split (round 4).  conv_raw3's tiles are 256 columns wide: 64 x 129 frames are 32.25 of them, and the 33rd tile column
    // costs what the other 32 cost each (2.3 % of D0 forward / U0 dgrad; at the reference's own batch of 16 x 65 = 1040 columns a
    // FIFTH of five).  Where the columns past the last full tile are few (<= 128), the launch covers full tiles only and a second
    // launch of the tall two-waves-per-SIMD kernel (256 x 128, column blocks without columns skipped) takes the tail from column
    // n_lo on: a weight-streaming pass like demo.py's single clip (the weights once at ~3 TB/s, or its own MFMA work), taken when
    // the model below says it costs under 1 / 1.3 of the tile column it replaces.  A pure function of the geometry.
    if (may_split && r3_ok && kn.force_mode != 1 && !kn.no_colsplit && !kn.no_tall) {      // (r3_ok: also where the whole problem would take the tall tile)
        const long full = cols / RBN * RBN, rem = cols - full;
        if (full > 0 && rem > 0 && rem <= RBN / 2 && raw_supported(kind, p, kn, RBN / 2)) {
            const double rows_p = (double)((rows + 255) / 256 * 256), rem32 = (double)((rem + 31) / 32 * 32), Kd = (double)Ktot;
            const double t_col = 2.0 * rows_p * 256.0 * Kd / 140e6;                                              // us at 140 TFLOP/s
            const double t_tail = fmax(4.0 * (double)rows * Kd / 3e6, 2.0 * rows_p * rem32 * Kd / 100e6) + 25.0;  // us: 3 TB/s | 100 TFLOP/s, + launches
            // ... and only where the full tiles then split into ALIGNED ranges (whole tiles per workgroup, or a whole number of
            // workgroups per tile): those walk K in lockstep and share weight / activation panels in L2.  120 or 240 full tiles
            // (64 x 61 frames) over 256 CUs split unaligned: measured 3 % faster than with the tail kept, but 3.9 instead of
            // 1.4 GB of L2 fills per launch -- not taken.
            IgemmParams pa = p;
            const long tiles_a = (rows + 255) / 256 * (full / RBN), nslab_a = (Ktot + BK - 1) / BK;
            const long grid_a = pick_grid(tiles_a, (int)nslab_a, pa, ws_bytes, kn.force_mode, kn.oversub, kn.contended, 1, 2 * WS_PER_WG);
            const bool aligned = pa.whole == 0 && (grid_a % tiles_a == 0 || tiles_a % grid_a == 0);
            if ((t_col > 1.3 * t_tail && aligned) || kn.force_colsplit) {
                Knobs kb = kn;
                kb.no_raw3 = 1; kb.all_raw3 = 0;
                char tail[160] = "";
                if (kn.desc) { kb.desc = tail; kb.desc_len = (int)sizeof tail; }
                int rc = launch(kind, p, kn, rows, full, Ktot_in, ws_bytes, st, false);
                if (rc != PG_OK) return rc;
                IgemmParams pb = p;
                pb.n_lo = (int)full;
                rc = launch(kind, pb, kb, rows, rem, Ktot_in, ws_bytes, st, false);
                if (rc == PG_OK && kn.desc) {          // "...|tail=conv_raw_kernel<...>,grid=G"
                    char* bar = strchr(tail, '|');
                    if (bar) { *bar = ','; bar = strchr(bar, '|'); if (bar) *bar = 0; }
                    const size_t n = strlen(kn.desc);
                    if (n + 7 < (size_t)kn.desc_len) snprintf(kn.desc + n, (size_t)kn.desc_len - n, "|tail=%s", tail);
                }
                return rc;
            }
        }
    }
    if (p.n_lo && !(tall && !r3)) return pg_fail(PG_ERR_UNSUPPORTED, "conv: internal -- a column tail off the tall tile");
    p.tn_stride = bn;
    // conv_raw3's tile order: the 32 workgroups of an XCD (256 CUs / 8) run consecutive tiles.  Row-major (R = 1) they are one tile
    // row: they share the weight panel but each reads its own activation panel.  In super-rows of R tile rows an XCD covers R x 32/R
    // tiles and an activation panel serves R rows at once.  Measured (tools/dbg/sr_ab.py, FETCH_SIZE per launch at the bench shape,
    // R = 1 / 2 / 4): k = 8 layers 1.85 / 1.57 / 1.77, 1.49 / 1.22 / 1.51, 0.68 / 0.54 / 0.68 GB -- R = 2 saves 15-20 % of the L2
    // fills; k = 32: 5.41 / 5.47 / 7.41 and 3.24 / 3.23 / 4.84 GB -- nothing at R = 2, + 40 % at R = 4: those fills are weight-panel
    // re-reads of workgroups that drift apart over 4096 slabs, and fewer sharers per panel make it worse.  Time: equal to 0.3 %.
    // Hence R = 2 for k = 8, row-major otherwise (schedule bits 15-16 force R = 1 / 2 / 4 for the A/B).
    p.sr = r3 ? (kn.sr ? kn.sr : (p.k == 8 ? 2 : 1)) : 0;
    p.tilesM = (int)((rows + bm - 1) / bm);
    p.tilesN = (int)((cols + bn - 1) / bn);
    p.nslab = (int)((Ktot + BK - 1) / BK);
    const long tiles = (long)p.tilesM * p.tilesN;
    if (tiles <= 0 || tiles > 0x0fffffffL || p.nslab <= 0 || cols + bn >= 0x7fffffffL) return pg_fail(PG_ERR_SHAPE, "conv: empty or oversize grid");   // the fixup launches 8 workgroups per tile
    const int grid = r3 ? pick_grid(tiles, p.nslab, p, ws_bytes, kn.force_mode, kn.oversub, kn.contended, 1, 2 * WS_PER_WG)
                        : pick_grid(tiles, p.nslab, p, ws_bytes, kn.force_mode, kn.oversub, kn.contended);
    // ranges made of whole tiles (grid == tiles, or a grid that divides the tile count) leave nothing for the fixup
    const bool split = grid != tiles && !(tiles % grid == 0);
    if (kn.desc) {      // the kernel this call would launch, named as rocprofv3 names it (profiles/*_kernel_stats.csv)
        char name[96];
        if (raw && kind == KIND_G) snprintf(name, sizeof name, "conv_g_%s_kernel<%d, %d, %d>", p.g_ps ? "ps" : "raw", p.k, p.s, kn.prec);
        else if (r3) snprintf(name, sizeof name, "conv_raw3_kernel<%d, %d, %s, %s>", p.k, p.s, kind == KIND_T ? "true" : "false", p.act_x == PG_ACT_NONE ? "false" : "true");
        else if (raw) snprintf(name, sizeof name, "conv_raw_kernel<%d, %d, %s, %d, %d>", p.k, p.s, kind == KIND_T ? "true" : "false", kn.prec, tall ? 1 : 2);
        else snprintf(name, sizeof name, "conv_%c_kernel<0, 0, %d>", kind == KIND_F ? 'f' : (kind == KIND_T ? 't' : 'g'), kn.prec);
        snprintf(kn.desc, (size_t)kn.desc_len, "%s|grid=%d|tiles=%ld|slabs=%d|split=%d|whole=%d|fixup=%s", name, grid, tiles, p.nslab, (int)split, p.whole,
                 !split ? "none" : (!r3 && fixup_wide(grid, tiles - p.whole) ? "wide" : "plain"));
        return PG_OK;
    }
    hipError_t e;
    if (r3) e = pgconv::launch_raw3(kind, p, grid, st);
    else if (raw && kind == KIND_G) e = pgconv::launch_raw_g(p, grid, st, kn.prec);
    else if (tall) e = pgconv::launch_raw_ft_tall(kind, p, grid, st, kn.prec);
    else if (raw) e = pgconv::launch_raw_ft(kind, p, grid, st, kn.prec);
    else e = pgconv::launch_im2col(kind, p, grid, st, kn.prec);
    if (e == hipSuccess && split && r3) e = pgconv::launch_raw3_fixup(kind, p, grid, (unsigned)((tiles - p.whole) * 16), st);
    else if (e == hipSuccess && split) {
        // many segments per split tile (small-batch inference) -> the wide fixup: the order in which a tile's segments are added is a
        // function of (grid, tiles) only, so a geometry always takes the same one
        const bool wide = fixup_wide(grid, tiles - p.whole);
        const dim3 fg((unsigned)((tiles - p.whole) * (wide ? 32 : 8)));
#define PG_FIXUP(...) { if (wide) hipLaunchKernelGGL((conv_fixup_kernel<__VA_ARGS__, true>), fg, dim3(NT), 0, st, p, grid); \
                        else hipLaunchKernelGGL((conv_fixup_kernel<__VA_ARGS__, false>), fg, dim3(NT), 0, st, p, grid); }
        if (tall) {
            if (kind == KIND_F) PG_FIXUP(0, 2, 4, 1)
            else if (p.s == 2) PG_FIXUP(3, 2, 4, 1)
            else PG_FIXUP(1, 2, 4, 1)
        } else if (raw) {
            if (kind == KIND_F) PG_FIXUP(0, 2, 4, 2)
            else if (kind == KIND_T && p.s == 2) PG_FIXUP(3, 2, 4, 2)
            else if (kind == KIND_T) PG_FIXUP(1, 2, 4, 2)
            else PG_FIXUP(2, 2, 4, 2)
        } else switch (kind) {
            case KIND_F: PG_FIXUP(0, WMB, 2, 2) break;
            case KIND_T: PG_FIXUP(1, WMB, 2, 2) break;
            case KIND_G: PG_FIXUP(2, WMB, 2, 2) break;
        }
#undef PG_FIXUP
        e = hipGetLastError();
    }
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
static int run_conv1d_fwd(const pg_conv_args* a, void* stream, char* desc, int desc_len) {
    if (int e = check_geom(a, false)) return e;
    Knobs kn; if (int e = decode_knobs(a, kn)) return e;
    kn.desc = desc; kn.desc_len = desc_len;
    if (!a->x || !a->w || !a->y) return pg_fail(PG_ERR_NULL, "conv1d_fwd: x, w, y required");
    IgemmParams p = {};
    p.y_slope = act_slope(a->y_act); p.y2 = a->y2; p.y2_bs = a->y2_bs; p.y2_slope = act_slope(a->y2_act);
    p.x = a->x; p.x_bs = a->x_bs; p.w = a->w; p.y = a->y; p.y_bs = a->y_bs;
    p.B = a->B; p.Q = a->Cin; p.M = a->Cout; p.Lx = a->Lin; p.Ly = a->Lout; p.k = a->k; p.s = a->stride; p.p = a->pad;
    p.act_x = a->x_act;
    if (int e = set_extents(p, p.Q, p.Lx, 0, 0)) return e;
    p.ws = (float*)a->workspace;
    return launch(KIND_F, p, kn, p.M, (long)p.B * p.Ly, (long)p.Q * p.k, a->workspace_bytes, (hipStream_t)stream);
}

// nn.ConvTranspose1d dgrad: dx[b,c,i] = sum_{o,j} w[c][o][j] dy[b,o,s*i+j-p]  -> F kernel with M = Cin, Q = Cout.
static int run_convt1d_dgrad(const pg_conv_args* a, void* stream, char* desc, int desc_len) {
    if (int e = check_geom(a, true)) return e;
    Knobs kn; if (int e = decode_knobs(a, kn)) return e;
    kn.desc = desc; kn.desc_len = desc_len;
    if (!a->dy || !a->w || !a->dx) return pg_fail(PG_ERR_NULL, "convt1d_dgrad: dy, w, dx required");
    IgemmParams p = {};
    p.y_slope = 1.0f; p.y2_slope = 1.0f;
    p.x = a->dy; p.x_bs = a->dy_bs; p.w = a->w; p.y = a->dx; p.y_bs = a->dx_bs;
    p.add = a->dx_add; p.add_bs = a->dx_add_bs; p.ref = a->dx_ref; p.ref_bs = a->dx_ref_bs;
    p.mask_mode = a->dx_ref ? a->dx_mask : 0;
    p.B = a->B; p.Q = a->Cout; p.M = a->Cin; p.Lx = a->Lout; p.Ly = a->Lin; p.k = a->k; p.s = a->stride; p.p = a->pad;
    if (int e = set_extents(p, p.Q, p.Lx, 0, 0)) return e;
    p.ws = (float*)a->workspace;
    return launch(KIND_F, p, kn, p.M, (long)p.B * p.Ly, (long)p.Q * p.k, a->workspace_bytes, (hipStream_t)stream);
}

static int launch_t(IgemmParams& p, const Knobs& kn, long ws_bytes, hipStream_t st) {
    // tau = s*u + phi - p >= 0 for some phi  <=>  u >= floor(p/s) at the latest; tau <= Ly-1 => u <= (Ly-1+p)/s
    p.u_off = p.p / p.s;
    const int u_max = (p.Ly - 1 + p.p) / p.s;
    p.U = u_max - p.u_off + 1;
    if (p.U <= 0) return pg_fail(PG_ERR_SHAPE, "convT: empty output");
    if (int e = set_extents(p, p.Q, p.Lx, 0, 0)) return e;
    return launch(KIND_T, p, kn, (long)p.M * p.s, (long)p.B * p.U, (long)p.Q * ((p.k + p.s - 1) / p.s), ws_bytes, st);
}

// nn.ConvTranspose1d forward: T kernel with M = Cout, Q = Cin.
static int run_convt1d_fwd(const pg_conv_args* a, void* stream, char* desc, int desc_len) {
    if (int e = check_geom(a, true)) return e;
    Knobs kn; if (int e = decode_knobs(a, kn)) return e;
    kn.desc = desc; kn.desc_len = desc_len;
    if (!a->x || !a->w || !a->y) return pg_fail(PG_ERR_NULL, "convt1d_fwd: x, w, y required");
    IgemmParams p = {};
    p.y_slope = act_slope(a->y_act); p.y2 = a->y2; p.y2_bs = a->y2_bs; p.y2_slope = act_slope(a->y2_act);
    p.x = a->x; p.x_bs = a->x_bs; p.w = a->w; p.y = a->y; p.y_bs = a->y_bs;
    p.B = a->B; p.Q = a->Cin; p.M = a->Cout; p.Lx = a->Lin; p.Ly = a->Lout; p.k = a->k; p.s = a->stride; p.p = a->pad;
    p.act_x = a->x_act;
    p.ws = (float*)a->workspace;
    return launch_t(p, kn, a->workspace_bytes, (hipStream_t)stream);
}

// nn.Conv1d dgrad: dx[b,c,u] = sum_{o,j,t: s*t+j-p=u} w[o][c][j] dy[b,o,t]  -> T kernel with M = Cin, Q = Cout.
static int run_conv1d_dgrad(const pg_conv_args* a, void* stream, char* desc, int desc_len) {
    if (int e = check_geom(a, false)) return e;
    Knobs kn; if (int e = decode_knobs(a, kn)) return e;
    kn.desc = desc; kn.desc_len = desc_len;
    if (!a->dy || !a->w || !a->dx) return pg_fail(PG_ERR_NULL, "conv1d_dgrad: dy, w, dx required");
    IgemmParams p = {};
    p.y_slope = 1.0f; p.y2_slope = 1.0f;
    p.x = a->dy; p.x_bs = a->dy_bs; p.w = a->w; p.y = a->dx; p.y_bs = a->dx_bs;
    p.add = a->dx_add; p.add_bs = a->dx_add_bs; p.ref = a->dx_ref; p.ref_bs = a->dx_ref_bs;
    p.mask_mode = a->dx_ref ? a->dx_mask : 0;
    p.B = a->B; p.Q = a->Cout; p.M = a->Cin; p.Lx = a->Lout; p.Ly = a->Lin; p.k = a->k; p.s = a->stride; p.p = a->pad;
    p.ws = (float*)a->workspace;
    return launch_t(p, kn, a->workspace_bytes, (hipStream_t)stream);
}

// pg_conv_args.adam: the optimiser step of this weight in the wgrad epilogue (whole tiles: the GEMM kernel's; split tiles: the
// fixup kernel's -- every element of dW passes through exactly one epilogue_g)
static int set_fused_adam(IgemmParams& p, const pg_conv_args* a) {
    const pg_adam_args* ad = a->adam;
    if (!ad) return PG_OK;
    if (!ad->p || !ad->m || !ad->v) return pg_fail(PG_ERR_NULL, "wgrad: fused adam needs p, m and v");
    if (ad->step < 1) return pg_fail(PG_ERR_SHAPE, "wgrad: fused adam: step is 1-based");
    if (ad->n != (int64_t)a->Cin * a->Cout * a->k) return pg_fail(PG_ERR_SHAPE, "wgrad: fused adam: n must equal Cin * Cout * k (p / m / v shaped like dw)");
    if (((uintptr_t)ad->p | (uintptr_t)ad->m | (uintptr_t)ad->v) & 3) return pg_fail(PG_ERR_ALIGN, "wgrad: fused adam: misaligned pointer");
    if ((const float*)ad->p == a->dw || ad->m == a->dw || ad->v == a->dw) return pg_fail(PG_ERR_SHAPE, "wgrad: fused adam: p / m / v alias dw");
    p.ad_p = ad->p; p.ad_m = ad->m; p.ad_v = ad->v;
    p.ad = pg_adam_scalars(ad);
    return PG_OK;
}

// nn.Conv1d wgrad: dw[o][c][j] = sum_{b,t} dy[b,o,t] act(x)[b,c,s*t+j-p]  -> G with P = dy (M = Cout), Q = x.
static int run_conv1d_wgrad(const pg_conv_args* a, void* stream, char* desc, int desc_len) {
    if (int e = check_geom(a, false)) return e;
    Knobs kn; if (int e = decode_knobs(a, kn)) return e;
    kn.desc = desc; kn.desc_len = desc_len;
    if (!a->dy || !a->x || !a->dw) return pg_fail(PG_ERR_NULL, "conv1d_wgrad: dy, x, dw required");
    IgemmParams p = {};
    p.pt = a->dy; p.pt_bs = a->dy_bs; p.LP = a->Lout; p.act_p = PG_ACT_NONE;
    p.x = a->x; p.x_bs = a->x_bs; p.act_x = a->x_act; p.y = a->dw;
    p.B = a->B; p.Q = a->Cin; p.M = a->Cout; p.Lx = a->Lin; p.k = a->k; p.s = a->stride; p.p = a->pad;
    if (int e = set_extents(p, p.Q, p.Lx, p.M, p.LP)) return e;
    if (int e = set_fused_adam(p, a)) return e;
    p.ws = (float*)a->workspace;
    return launch(KIND_G, p, kn, p.M, (long)p.Q * p.k, (long)p.B * p.LP, a->workspace_bytes, (hipStream_t)stream);
}

// nn.ConvTranspose1d wgrad: dw[c][o][j] = sum_{b,i} act(x)[b,c,i] dy[b,o,s*i+j-p]  -> G with P = x (M = Cin), Q = dy.
static int run_convt1d_wgrad(const pg_conv_args* a, void* stream, char* desc, int desc_len) {
    if (int e = check_geom(a, true)) return e;
    Knobs kn; if (int e = decode_knobs(a, kn)) return e;
    kn.desc = desc; kn.desc_len = desc_len;
    if (!a->dy || !a->x || !a->dw) return pg_fail(PG_ERR_NULL, "convt1d_wgrad: dy, x, dw required");
    IgemmParams p = {};
    p.pt = a->x; p.pt_bs = a->x_bs; p.LP = a->Lin; p.act_p = a->x_act;
    p.x = a->dy; p.x_bs = a->dy_bs; p.act_x = PG_ACT_NONE; p.y = a->dw;
    p.B = a->B; p.Q = a->Cout; p.M = a->Cin; p.Lx = a->Lout; p.k = a->k; p.s = a->stride; p.p = a->pad;
    if (int e = set_extents(p, p.Q, p.Lx, p.M, p.LP)) return e;
    if (int e = set_fused_adam(p, a)) return e;
    p.ws = (float*)a->workspace;
    return launch(KIND_G, p, kn, p.M, (long)p.Q * p.k, (long)p.B * p.LP, a->workspace_bytes, (hipStream_t)stream);
}

extern "C" int pg_conv1d_fwd(const pg_conv_args* a, void* stream) { return run_conv1d_fwd(a, stream, nullptr, 0); }
extern "C" int pg_conv1d_dgrad(const pg_conv_args* a, void* stream) { return run_conv1d_dgrad(a, stream, nullptr, 0); }
extern "C" int pg_conv1d_wgrad(const pg_conv_args* a, void* stream) { return run_conv1d_wgrad(a, stream, nullptr, 0); }
extern "C" int pg_convt1d_fwd(const pg_conv_args* a, void* stream) { return run_convt1d_fwd(a, stream, nullptr, 0); }
extern "C" int pg_convt1d_dgrad(const pg_conv_args* a, void* stream) { return run_convt1d_dgrad(a, stream, nullptr, 0); }
extern "C" int pg_convt1d_wgrad(const pg_conv_args* a, void* stream) { return run_convt1d_wgrad(a, stream, nullptr, 0); }

// The launch plan of a conv call without launching it: "kernel<template args>|grid=..|tiles=..|slabs=..|split=0/1", the kernel
// named as rocprofv3 reports it.  Pure function of the arguments (same checks as the real call); bench.py uses it to group
// its per-launch timings by kernel so that its roofline numbers can be recomputed from profiles/*_kernel_stats.csv.
extern "C" int pg_conv_describe(const pg_conv_args* a, int32_t op, char* buf, int32_t buflen) {
    if (!buf || buflen < 128) return pg_fail(PG_ERR_NULL, "conv_describe: buf of >= 128 bytes required");
    buf[0] = 0;
    switch (op) {
        case PG_OP_CONV1D_FWD: return run_conv1d_fwd(a, nullptr, buf, buflen);
        case PG_OP_CONV1D_DGRAD: return run_conv1d_dgrad(a, nullptr, buf, buflen);
        case PG_OP_CONV1D_WGRAD: return run_conv1d_wgrad(a, nullptr, buf, buflen);
        case PG_OP_CONVT1D_FWD: return run_convt1d_fwd(a, nullptr, buf, buflen);
        case PG_OP_CONVT1D_DGRAD: return run_convt1d_dgrad(a, nullptr, buf, buflen);
        case PG_OP_CONVT1D_WGRAD: return run_convt1d_wgrad(a, nullptr, buf, buflen);
    }
    return pg_fail(PG_ERR_UNSUPPORTED, "conv_describe: op must be a PG_OP_* value");
}

// ---- bf16-resident forward (conv_h.hip) ----------------------------------------------------------------------------------
static int conv_fwd_h_impl(const pg_convh_args* a, void* stream, bool query, char* desc = nullptr, int desc_len = 0) {
    if (!a) return pg_fail(PG_ERR_NULL, "conv_fwd_h: null args");
    if (a->B <= 0 || a->Cin <= 0 || a->Cout <= 0 || a->Lin <= 0 || a->Lout <= 0 || a->k <= 0 || a->stride <= 0 || a->pad < 0)
        return pg_fail(PG_ERR_SHAPE, "conv_fwd_h: non-positive dimension");
    const bool tr = a->transposed != 0;
    const long lo = tr ? (long)(a->Lin - 1) * a->stride - 2L * a->pad + a->k : ((long)a->Lin + 2L * a->pad - a->k) / a->stride + 1;
    if (lo != a->Lout) return pg_fail(PG_ERR_SHAPE, "conv_fwd_h: Lout inconsistent with Lin/k/stride/pad");
    if (!query && (!a->x || !a->w || (!a->y && !a->yh && !a->yh2))) return pg_fail(PG_ERR_NULL, "conv_fwd_h: x, w and at least one output required");
    if (((uintptr_t)a->x & 3) || ((uintptr_t)a->w & 15) || (a->x_bs & 1) || (a->x_pitch & 1))
        return pg_fail(PG_ERR_ALIGN, "conv_fwd_h: x must be 4-byte aligned with even pitch / batch stride, w 16-byte aligned");
    if (a->x_pitch <= a->Lin) return pg_fail(PG_ERR_SHAPE, "conv_fwd_h: x_pitch must exceed Lin (zero tail of at least one element)");
    if ((a->yh && a->yh_pitch < a->Lout) || (a->yh2 && a->yh2_pitch < a->Lout)) return pg_fail(PG_ERR_SHAPE, "conv_fwd_h: output pitch below Lout");
    pg_conv_args kb = {};
    kb.schedule = a->schedule;
    Knobs kn; if (int e = decode_knobs(&kb, kn)) return e;
    IgemmParams p = {};
    p.x = reinterpret_cast<const float*>(a->x); p.x_bs = a->x_bs; p.x_pitch = a->x_pitch;
    p.w = reinterpret_cast<const float*>(a->w);
    p.y = a->y; p.y_bs = a->y_bs; p.y_slope = 1.0f; p.y2_slope = 1.0f;
    p.yh = a->yh; p.yh_bs = a->yh_bs; p.yh_pitch = a->yh_pitch; p.yh_slope = act_slope(a->yh_act);
    p.yh2 = a->yh2; p.yh2_bs = a->yh2_bs; p.yh2_pitch = a->yh2_pitch; p.yh2_slope = act_slope(a->yh2_act);
    p.B = a->B; p.Q = a->Cin; p.M = a->Cout; p.Lx = a->Lin; p.Ly = a->Lout; p.k = a->k; p.s = a->stride; p.p = a->pad;
    const long xe = ((long)(p.B - 1) * p.x_bs + (long)p.Q * p.x_pitch) * 2;
    if (xe <= 0 || xe >= 0x7ffffff0L) return pg_fail(PG_ERR_SHAPE, "conv_fwd_h: activation tensor exceeds 2 GiB (31-bit buffer offsets)");
    p.x_bytes = (unsigned)xe;
    const long we = (long)p.M * p.Q * p.k * 2 * (tr ? 1 : 1);
    const int kwp = tr ? pg_shadow_taps(p.k, p.s) : p.k;
    const long wbytes = tr ? (long)p.M * p.s * p.Q * kwp * 2 : we;
    if (wbytes >= 0x7ffffff0L) return pg_fail(PG_ERR_SHAPE, "conv_fwd_h: weight shadow exceeds 2 GiB");
    p.w_bytes = (unsigned)wbytes;
    const Kind kind = tr ? KIND_T : KIND_F;
    if (tr) {
        p.u_off = p.p / p.s;
        p.U = (p.Ly - 1 + p.p) / p.s - p.u_off + 1;
        if (p.U <= 0) return pg_fail(PG_ERR_SHAPE, "conv_fwd_h: empty output");
    }
    // ONE tile family since 0.4: 256 x 256 on 4 waves, one workgroup per CU and one wave per SIMD (conv_h3.hip).  Round 3 measured it
    // level with or ahead of the 128 x 256 (conv_h.hip) and the eight-wave 128 x 512 / 256 x 256 tiles (conv_h2.hip) on all eight layers
    // inside the forward (6.44 ms against 6.55 / 6.73 / 6.88), so those were never reached automatically and have been removed.
    if (!pgconv::h_supported(kind, p)) return pg_fail(PG_ERR_UNSUPPORTED, "conv_fwd_h: geometry not covered by the bf16-resident kernels (use the fp32-tensor entry points)");
    if (query) return PG_OK;
    const long rows = tr ? (long)p.M * p.s : p.M, cols = (long)p.B * (tr ? p.U : p.Ly), Ktot = (long)p.Q * kwp;
    const int tm = 2 * RBM, tn = RBN;
    p.tilesM = (int)((rows + tm - 1) / tm);
    p.tilesN = (int)((cols + tn - 1) / tn);
    p.tn_stride = tn;
    p.nslab = (int)(Ktot / 32);
    p.ws = (float*)a->workspace;
    const long tiles = (long)p.tilesM * p.tilesN;
    if (tiles <= 0 || tiles > 0x0fffffffL || p.nslab <= 0) return pg_fail(PG_ERR_SHAPE, "conv_fwd_h: empty or oversize grid");
    const int grid = pick_grid(tiles, p.nslab, p, a->workspace_bytes, kn.force_mode, kn.oversub, kn.contended, 1, 2 * WS_PER_WG);
    const bool split = grid != tiles && !(tiles % grid == 0);
    if (desc) {
        snprintf(desc, (size_t)desc_len, "conv_h3_kernel<%d, %d, %s>|grid=%d|tiles=%ld|slabs=%d|split=%d|whole=%d|fixup=%s",
                 (tr && p.k == 5) ? 8 : p.k, p.s, tr ? "true" : "false", grid, tiles, p.nslab, (int)split, p.whole,
                 !split ? "none" : (fixup_wide(grid, tiles - p.whole) ? "wide" : "plain"));
        return PG_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = pgconv::launch_h3(kind, p, grid, st);
    if (e == hipSuccess && split) e = pgconv::launch_h3_fixup(kind, p, grid, (unsigned)(tiles - p.whole), fixup_wide(grid, tiles - p.whole), st);
    if (e != hipSuccess) return pg_fail((int)e, hipGetErrorString(e));
    return PG_OK;
}

extern "C" int pg_conv_fwd_h(const pg_convh_args* a, void* stream) { return conv_fwd_h_impl(a, stream, false); }
// 1 if pg_conv_fwd_h covers this geometry (sizes, strides and pitches of `a`; pointers may be NULL), else 0: pure host check
extern "C" int pg_conv_fwd_h_supported(const pg_convh_args* a) { return conv_fwd_h_impl(a, nullptr, true) == PG_OK ? 1 : 0; }
// launch plan of a pg_conv_fwd_h call without launching it (as pg_conv_describe; pointers must be non-NULL, they are not read)
extern "C" int pg_conv_fwd_h_describe(const pg_convh_args* a, char* buf, int32_t buflen) {
    if (!buf || buflen < 128) return pg_fail(PG_ERR_NULL, "conv_fwd_h_describe: buf of >= 128 bytes required");
    buf[0] = 0;
    return conv_fwd_h_impl(a, nullptr, false, buf, buflen);
}

// Workspace a caller should hand to the conv entry points (pg_conv_args.workspace) so that badly quantised tile counts
// can be balanced over all CUs (stream-K).  Without it every call falls back to one-tile-per-workgroup scheduling.
extern "C" int64_t pg_workspace_bytes_conv(void) { return (int64_t)MAX_STREAMK_WG * WS_PER_WG; }
