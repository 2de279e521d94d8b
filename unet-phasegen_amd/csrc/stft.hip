// stft.hip -- librosa-convention STFT / ISTFT framing on device (preproc_mdb.py:84-97, utils.py:34-42, demo.py:39).
//
// One workgroup per (signal, frame): the frame is gathered (reflect padding resolved by integer index math, the
// bit-exact part of the contract), windowed with a periodic Hann, transformed by a radix-2 Stockham FFT that
// lives entirely in LDS (two ping-pong complex buffers + a twiddle table), and written as [re; im] or, fused
// with data.py:39-47, as [log1p|z|; angle z].  The inverse runs the conjugate transform on the Hermitian
// extension (zero DC row prepended, utils.py:38-39), windows, and leaves frames in a workspace; the
// overlap-add is a GATHER over the <= n_fft/hop frames covering each sample (deterministic, no atomics),
// fused with the window-sum-square division, the n_fft/2 trim and the peak search.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "phasegen.h"
#include "pg_common.h"
#include "pg_fastmath.h"

#ifndef PG_STFT_ABL
#define PG_STFT_ABL 0
#endif
#ifndef PG_W_ABL         /* dev-only timing ablations of stft_w_kernel / istft_frames_w_kernel (wrong results): 1 no sample / row loads, */
#define PG_W_ABL 0       /* 2 no transform, 4 no global stores, 8 no workgroup-wide phase at all                                          */
#endif


namespace {

constexpr int FFT_THREADS = 256;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL load and store
// (s_waitcnt vmcnt(0)), which serialises a prefetch issued in front of the FFT passes with the passes it was meant to hide under;
// this one leaves vector-memory operations in flight (the compiler still waits for a load's registers at their first use).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// numpy 'reflect' padding (edge sample not repeated): index into y[0..n) of position pos (may be <0 or >=n)
__device__ __host__ __forceinline__ int reflect_index(int pos, int n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    int m = pos % period;
    if (m < 0) m += period;
    return m < n ? m : period - m;
}

__device__ __forceinline__ float hann(int k, int n) { return 0.5f - 0.5f * cospif(2.0f * (float)k / (float)n); }

// In-LDS Stockham radix-2 FFT of length N (power of two).  buf0 holds the input; returns the buffer holding the
// natural-order output.  tw[i] = exp(-2 pi i / N * i), i < N/2; sign = +1 forward, -1 inverse (conjugate twiddles).
__device__ float2* fft_lds(float2* buf0, float2* buf1, const float2* tw, int N, float sign) {
    float2* x = buf0;
    float2* y = buf1;
    const int half = N >> 1;
    for (int Ns = 1; Ns < N; Ns <<= 1) {
        const int tstride = half / Ns;
        for (int j = threadIdx.x; j < half; j += blockDim.x) {
            const int k = j & (Ns - 1);
            const float2 w = tw[k * tstride];
            const float wi = sign * w.y;
            const float2 a = x[j], b = x[j + half];
            const float2 v = make_float2(b.x * w.x - b.y * wi, b.x * wi + b.y * w.x);
            const int j0 = 2 * j - k;
            y[j0] = make_float2(a.x + v.x, a.y + v.y);
            y[j0 + Ns] = make_float2(a.x - v.x, a.y - v.y);
        }
        __syncthreads();
        float2* t = x; x = y; y = t;
    }
    return x;
}

__device__ __forceinline__ void fill_twiddles(float2* tw, int N) {
    for (int i = threadIdx.x; i < (N >> 1); i += blockDim.x) {
        float s, c;
        sincospif(-2.0f * (float)i / (float)N, &s, &c);
        tw[i] = make_float2(c, s);
    }
}

// where signal `sig` lives: first sample and how many of its n_samples exist in the source (the rest read as zero)
struct Src { const float* p; int lim; };
__device__ __forceinline__ Src signal_src(const pg_stft_args& a, int sig) {
    if (!a.chunk_start) return { a.y + (long)sig * a.n_samples, a.n_samples };
    const long st = a.chunk_start[sig];
    const long row = a.chunk_row ? a.chunk_row[sig] : 0;
    long lim = a.src_len - st;
    lim = lim < 0 ? 0 : (lim > a.n_samples ? a.n_samples : lim);
    return { a.y + row * a.src_stride + st, (int)lim };
}

__global__ __launch_bounds__(FFT_THREADS) void stft_kernel(const pg_stft_args a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int N = a.n_fft, bins = N >> 1;
    float2* buf0 = smem; float2* buf1 = smem + N; float2* tw = smem + 2 * N;
    const int t = blockIdx.x % a.n_frames, sig = blockIdx.x / a.n_frames;
    const Src src = signal_src(a, sig);
    fill_twiddles(tw, N);
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const int idx = reflect_index(t * a.hop + k - (N >> 1), a.n_samples);
        buf0[k] = make_float2((idx < src.lim ? src.p[idx] : 0.f) * hann(k, N), 0.f);
    }
    __syncthreads();
    const float2* X = fft_lds(buf0, buf1, tw, N, 1.f);
    float* o_re = a.out + ((long)sig * 2 * bins) * a.n_frames + t;
    float* o_im = o_re + (long)bins * a.n_frames;
    for (int k = 1 + threadIdx.x; k <= bins; k += blockDim.x) {      // bin 0 (DC) dropped, preproc_mdb.py:93
        float2 v = X[k];
        if (a.polar) {
            float mg, an;
            pg_polar_one(v.x, v.y, 1, mg, an);
            o_re[(long)(k - 1) * a.n_frames] = mg;
            o_im[(long)(k - 1) * a.n_frames] = an;
        } else {
            o_re[(long)(k - 1) * a.n_frames] = v.x;
            o_im[(long)(k - 1) * a.n_frames] = v.y;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Batched transforms: SF consecutive frames of one signal per group, real FFT through an (n_fft/2)-point complex
// transform (even samples -> re, odd -> im, then the split X[k] = E[k] + w^k O[k]).  Workgroups are persistent (two
// per CU) and walk the groups of their XCD, so the twiddle tables, window values and split factors are built once per
// workgroup; inside a pass the index math and twiddles are shared by the SF frames.  Radix-4 Stockham passes give one
// butterfly per thread per pass at n_fft = 2048.  LDS indices go through swz(), a GF(2)-linear swizzle under which
// every read (32-lane groups, 64 banks) and write (16-lane groups, 32 banks) of every pass is conflict-free; twiddles
// are stored per pass in butterfly order, so their reads are contiguous too.
// The frame-minor output (bins, frames) is written SF frames (16 B) per row; consecutive groups -- the ones sharing
// a 128 B line -- run on the same XCD, i.e. behind the same L2.
constexpr int SF = 4;
constexpr int MAX_HALF = 1024;                       // n_fft <= 2048
constexpr int BT = 256;                              // threads per workgroup (512 measured slower: the store phase dominates)
constexpr int M_ITERS = MAX_HALF / BT;               // points per thread and frame in the load / store phases
constexpr int K_ITERS = (MAX_HALF / 2 + BT - 1) / BT; // bin pairs (k, M-k) per thread
constexpr int FG = BT / 256, PF = SF / FG;           // passes: thread = (butterfly j, frame subset fg), PF frames each

__device__ __forceinline__ int swz(int e) { return e ^ (((e >> 4) & 1) * 5) ^ (((e >> 5) & 1) * 10); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(__fmaf_rn(a.x, w.x, -(a.y * w.y)), __fmaf_rn(a.x, w.y, a.y * w.x));
}

// Twiddles of the radix-4 passes: the pass with sub-transform length Ns uses w^(r k), w = exp(-+ 2 pi i / (4 Ns)),
// k < Ns, r = 1..3, stored at T_r[(Ns - 1) / 3 + k] with T_r = tw + (r - 1) * TL, TL = tw_len(M).
__device__ __host__ __forceinline__ int tw_len(int M) { return M / 3 + 1; }
__device__ void fill_tw4(float2* tw, int M, float sign) {
    const int TL = tw_len(M);
    for (int Ns = 1; Ns * 4 <= M; Ns <<= 2) {
        const int off = (Ns - 1) / 3;
        for (int k = threadIdx.x; k < Ns; k += blockDim.x) {
            const float ang = sign * 2.0f * (float)k / (float)(4 * Ns);
#pragma unroll
            for (int r = 1; r <= 3; ++r) {
                float sn, cs;
                sincospif(ang * (float)r, &sn, &cs);
                tw[(r - 1) * TL + off + k] = make_float2(cs, sn);
            }
        }
    }
}

// M-point complex transforms of SF frames stored as x[f*M + swz(e)]; DIR = +1 forward, -1 inverse (unnormalised).
// Returns the buffer holding the natural-order result.  Ends with a barrier.
template <int DIR, int PF, int UNR = PF>   // PF frames per thread: a workgroup of 256 FG threads transforms FG x PF frames; UNR of them
__device__ float2* fft_frames_t(float2* x, float2* y, const float2* tw, int M) {      // in flight at once (register budget of the caller)
    const int q = M >> 2, TL = tw_len(M);
    const int jt = threadIdx.x & 255, f0 = (int)(threadIdx.x >> 8) * PF;
    x += f0 * M; y += f0 * M;                      // this thread's frames (the swap below keeps the offset)
    int Ns = 1;
    for (; Ns * 4 <= M; Ns <<= 2) {
        const int off = (Ns - 1) / 3;
        for (int j = jt; j < q; j += 256) {
            const int k = j & (Ns - 1);
            const float2 w1 = tw[off + k], w2 = tw[TL + off + k], w3 = tw[2 * TL + off + k];
            const int i0 = swz(j), i1 = swz(j + q), i2 = swz(j + 2 * q), i3 = swz(j + 3 * q);
            const int j0 = ((j - k) << 2) + k;
            const int o0 = swz(j0), o1 = swz(j0 + Ns), o2 = swz(j0 + 2 * Ns), o3 = swz(j0 + 3 * Ns);
#pragma unroll(UNR)
            for (int f = 0; f < PF; ++f) {
                const float2* xf = x + f * M;
                float2* yf = y + f * M;
                float2 a = xf[i0], b = xf[i1], c = xf[i2], d = xf[i3];
                if (Ns > 1) { b = cmul(b, w1); c = cmul(c, w2); d = cmul(d, w3); }          // first pass: all twiddles are 1
                const float2 s0 = make_float2(a.x + c.x, a.y + c.y), s1 = make_float2(a.x - c.x, a.y - c.y);
                const float2 s2 = make_float2(b.x + d.x, b.y + d.y), s3 = make_float2(b.x - d.x, b.y - d.y);
                const float2 r3 = DIR > 0 ? make_float2(s3.y, -s3.x) : make_float2(-s3.y, s3.x);   // -/+ i * s3
                yf[o0] = make_float2(s0.x + s2.x, s0.y + s2.y);
                yf[o1] = make_float2(s1.x + r3.x, s1.y + r3.y);
                yf[o2] = make_float2(s0.x - s2.x, s0.y - s2.y);
                yf[o3] = make_float2(s1.x - r3.x, s1.y - r3.y);
            }
        }
        lds_barrier();
        float2* t = x; x = y; y = t;
    }
    if (Ns < M) {                                  // log2 M odd: one closing radix-2 pass, twiddle computed in place
        const int half = M >> 1;
        for (int j = jt; j < half; j += 256) {
            float sn, cs;
            sincospif((DIR > 0 ? -1.0f : 1.0f) * (float)j / (float)half, &sn, &cs);       // k = j here: Ns == half
            const float2 w = make_float2(cs, sn);
            const int i0 = swz(j), i1 = swz(j + half), o0 = swz(j), o1 = swz(j + half);
#pragma unroll(UNR)
            for (int f = 0; f < PF; ++f) {
                const float2 a = x[f * M + i0], v = cmul(x[f * M + i1], w);
                y[f * M + o0] = make_float2(a.x + v.x, a.y + v.y);
                y[f * M + o1] = make_float2(a.x - v.x, a.y - v.y);
            }
        }
        lds_barrier();
        float2* t = x; x = y; y = t;
    }
    return x - f0 * M;
}
template <int DIR> __device__ __forceinline__ float2* fft_frames(float2* x, float2* y, const float2* tw, int M) { return fft_frames_t<DIR, PF>(x, y, tw, M); }

// one output row, SF consecutive frames: a 16 B store when the row segment is aligned, scalar stores otherwise
__device__ __forceinline__ void stft_store_row(const pg_stft_args& a, float* o_re, float* o_im, long row, int nfr, bool vec,
                                               float (&re)[SF], float (&im)[SF]) {
    if (a.polar) {
#pragma unroll
        for (int f = 0; f < SF; ++f) {
            const float r = re[f], i = im[f];
            pg_polar_one(r, i, 1, re[f], im[f]);
        }
    }
    float* pr = o_re + row * a.n_frames;
    float* pi = o_im + row * a.n_frames;
#if PG_STFT_ABL == 1 || (PG_W_ABL & 4)   /* dev-only: everything but the global stores (values kept alive) */
    asm volatile("" :: "v"(re[0]), "v"(re[1]), "v"(re[2]), "v"(re[3]), "v"(im[0]), "v"(im[1]), "v"(im[2]), "v"(im[3]), "v"(pr), "v"(pi));
    return;
#endif
    if (vec) {
        *(float4*)pr = make_float4(re[0], re[1], re[2], re[3]);
        *(float4*)pi = make_float4(im[0], im[1], im[2], im[3]);
    } else {
#pragma unroll
        for (int f = 0; f < SF; ++f)
            if (f < nfr) { pr[f] = re[f]; pi[f] = im[f]; }
    }
}

// The groups of this workgroup: XCD x (workgroups are dealt round-robin over the 8 XCDs) owns the contiguous range
// [x * chunk, (x + 1) * chunk), walked by its gridDim.x / 8 workgroups with that stride.
struct GroupWalk { int g, end, step; };
__device__ __forceinline__ GroupWalk group_walk(int total) {
    const int chunk = (total + 7) >> 3, xcd = blockIdx.x & 7;
    GroupWalk w;
    w.g = xcd * chunk + (int)(blockIdx.x >> 3);
    w.end = min(total, (xcd + 1) * chunk);
    w.step = (int)(gridDim.x >> 3);
    return w;
}

template <bool CHUNKED>   // CHUNKED: signals are chunks of longer source rows (pg_stft_args.chunk_start); its own instantiation so
                          // that the plain path keeps its vector loads and register budget
__global__ __launch_bounds__(BT) void stft_frames_kernel(const pg_stft_args a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int N = a.n_fft, M = N >> 1;
    float2* x = smem; float2* y = smem + SF * M; float2* tw = smem + 2 * SF * M;
    const int groups = (a.n_frames + SF - 1) / SF, total = a.n_signals * groups;
    fill_tw4(tw, M, -1.f);
    // per-thread constants: window values of its sample pairs, split factors w^k of its bin pairs
    float w0[M_ITERS], w1[M_ITERS], sc[K_ITERS], ss[K_ITERS];
#pragma unroll
    for (int i = 0; i < M_ITERS; ++i) {
        const int m = threadIdx.x + i * BT;
        w0[i] = hann(2 * m, N); w1[i] = hann(2 * m + 1, N);
    }
#pragma unroll
    for (int i = 0; i < K_ITERS; ++i)
        sincospif(-(float)(1 + threadIdx.x + i * BT) / (float)M, &ss[i], &sc[i]);   // w = exp(-2 pi i / n_fft)
    const bool vec2 = !CHUNKED && ((a.hop | a.n_samples) & 1) == 0 && (((uintptr_t)a.y) & 7) == 0;   // sample pairs are 8 B aligned
    const bool vec4 = (a.n_frames & 3) == 0 && (((uintptr_t)a.out) & 15) == 0;               // row segments are 16 B aligned
    __syncthreads();
    // sample pairs of one group -> registers (windowing and the LDS write happen one iteration later, so the loads of
    // group i+1 are in flight during the passes of group i)
    float2 pre[M_ITERS][SF];
    auto load_group = [&](int g) {
        const int sig = g / groups, t0 = (g - sig * groups) * SF;
        const int nfr = min(SF, a.n_frames - t0);
        const float* sgn; int lim;
        if (CHUNKED) { const Src src = signal_src(a, sig); sgn = src.p; lim = src.lim; }
        else { sgn = a.y + (long)sig * a.n_samples; lim = a.n_samples; }
        auto at = [&](int q) { return (!CHUNKED || q < lim) ? sgn[q] : 0.f; };
#pragma unroll
        for (int i = 0; i < M_ITERS; ++i) {
            const int m = threadIdx.x + i * BT;
#pragma unroll
            for (int f = 0; f < SF; ++f) {
                float v0 = 0.f, v1 = 0.f;
                if (m < M && f < nfr) {
                    const int start = (t0 + f) * a.hop - M, p = start + 2 * m;              // frame tap k sits at start + k
                    if (start >= 0 && start + N <= a.n_samples) {
                        if (vec2) { const float2 v = *(const float2*)(sgn + p); v0 = v.x; v1 = v.y; }
                        else { v0 = at(p); v1 = at(p + 1); }
                    } else { v0 = at(reflect_index(p, a.n_samples)); v1 = at(reflect_index(p + 1, a.n_samples)); }
                }
                pre[i][f] = make_float2(v0, v1);
            }
        }
    };
    GroupWalk gw = group_walk(total);
    if (gw.g < gw.end) load_group(gw.g);
    for (; gw.g < gw.end; gw.g += gw.step) {
        const int sig = gw.g / groups, t0 = (gw.g - sig * groups) * SF;
        const int nfr = min(SF, a.n_frames - t0);
#pragma unroll
        for (int i = 0; i < M_ITERS; ++i) {
            const int m = threadIdx.x + i * BT;
            if (m < M) {
                const int e = swz(m);
#pragma unroll
                for (int f = 0; f < SF; ++f) x[f * M + e] = make_float2(pre[i][f].x * w0[i], pre[i][f].y * w1[i]);
            }
        }
        lds_barrier();
#if PG_STFT_ABL != 3
        if (gw.g + gw.step < gw.end) load_group(gw.g + gw.step);
#endif
#if PG_STFT_ABL == 2
        const float2* Z = x;
#else
        const float2* Z = fft_frames<1>(x, y, tw, M);
#endif
        float* o_re = a.out + ((long)sig * 2 * M) * a.n_frames + t0;
        float* o_im = o_re + (long)M * a.n_frames;
        const bool vec = vec4 && nfr == SF;
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {                               // bins k and M-k from Z[k], Z[M-k]; DC dropped
            const int k = 1 + threadIdx.x + i * BT;
            if (k <= (M >> 1)) {
                const int ea = swz(k), eb = swz(M - k);
                float rk[SF], ik[SF], rm[SF], im[SF];
#pragma unroll
                for (int f = 0; f < SF; ++f) {                                // frames past the end were loaded as zeros
                    const float2 A = Z[f * M + ea], B = Z[f * M + eb];
                    const float2 E = make_float2(0.5f * (A.x + B.x), 0.5f * (A.y - B.y));
                    const float2 O = make_float2(0.5f * (A.y + B.y), -0.5f * (A.x - B.x));
                    const float2 T = cmul(O, make_float2(sc[i], ss[i]));
                    rk[f] = E.x + T.x; ik[f] = E.y + T.y;
                    rm[f] = E.x - T.x; im[f] = T.y - E.y;
                }
                stft_store_row(a, o_re, o_im, k - 1, nfr, vec, rk, ik);
                if (k != M - k) stft_store_row(a, o_re, o_im, M - k - 1, nfr, vec, rm, im);
            }
        }
        if (threadIdx.x == 0) {                                               // Nyquist bin: X[M] = Re Z0 - Im Z0
            float rn[SF], in[SF];
#pragma unroll
            for (int f = 0; f < SF; ++f) { const float2 Z0 = Z[f * M]; rn[f] = Z0.x - Z0.y; in[f] = 0.f; }
            stft_store_row(a, o_re, o_im, M - 1, nfr, vec, rn, in);
        }
        lds_barrier();                                                      // Z may live in the buffer the next group loads into
    }
}

// SF consecutive frames of spectrum row `bin` (1-based FFT bin): the raw values of both tensors (frames past the end read as zero) ...
struct RowRaw { float va[SF], vb[SF]; };
__device__ __forceinline__ void istft_row_load(const pg_istft_args& a, const float* pa, const float* pb, int bin, int nfr, bool vec, RowRaw& r) {
    const float* ra = pa + (long)(bin - 1) * a.n_frames;
    const float* rb = pb + (long)(bin - 1) * a.n_frames;
    if (vec) {
        const float4 qa = *(const float4*)ra, qb = *(const float4*)rb;
        r.va[0] = qa.x; r.va[1] = qa.y; r.va[2] = qa.z; r.va[3] = qa.w;
        r.vb[0] = qb.x; r.vb[1] = qb.y; r.vb[2] = qb.z; r.vb[3] = qb.w;
    } else {
#pragma unroll
        for (int f = 0; f < SF; ++f) { r.va[f] = f < nfr ? ra[f] : 0.f; r.vb[f] = f < nfr ? rb[f] : 0.f; }
    }
}
// ... and their complex values
__device__ __forceinline__ void istft_row_cvt(int mode, const RowRaw& r, float2 (&X)[SF]) {
#pragma unroll
    for (int f = 0; f < SF; ++f) {
        if (mode == 0) {                                                  // demo.py:39: (exp(m) - 1) e^{j phi}
            const float mag = pg_expm1_ref(r.va[f]);
            float s, c;
            pg_sincos(r.vb[f], s, c);
            X[f] = make_float2(mag * c, mag * s);
        } else X[f] = make_float2(r.va[f], r.vb[f]);
    }
}
__device__ __forceinline__ void istft_row(const pg_istft_args& a, const float* pa, const float* pb, int bin, int nfr, bool vec,
                                          float2 (&X)[SF]) {
    RowRaw r;
    istft_row_load(a, pa, pb, bin, nfr, vec, r);
    istft_row_cvt(a.mode, r, X);
}

__global__ __launch_bounds__(BT) void istft_frames4_kernel(const pg_istft_args a, float* frames) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int M = a.bins, N = 2 * M;
    float2* x = smem; float2* y = smem + SF * M; float2* tw = smem + 2 * SF * M;
    const int groups = (a.n_frames + SF - 1) / SF, total = a.n_signals * groups;
    fill_tw4(tw, M, 1.f);
    float w0[M_ITERS], w1[M_ITERS], sc[K_ITERS], ss[K_ITERS];
    const float inv = 1.0f / (float)M;
#pragma unroll
    for (int i = 0; i < M_ITERS; ++i) {
        const int m = threadIdx.x + i * BT;
        w0[i] = inv * hann(2 * m, N); w1[i] = inv * hann(2 * m + 1, N);
    }
#pragma unroll
    for (int i = 0; i < K_ITERS; ++i)
        sincospif((float)(1 + threadIdx.x + i * BT) / (float)M, &ss[i], &sc[i]);    // exp(+2 pi i k / n_fft)
    const bool vec4 = (a.n_frames & 3) == 0 && ((a.a_bs | a.b_bs) & 3) == 0 && ((((uintptr_t)a.a) | ((uintptr_t)a.b)) & 15) == 0;
    __syncthreads();
    for (GroupWalk gw = group_walk(total); gw.g < gw.end; gw.g += gw.step) {
        const int sig = gw.g / groups, t0 = (gw.g - sig * groups) * SF;
        const int nfr = min(SF, a.n_frames - t0);
        const float* pa = a.a + (long)sig * a.a_bs + t0;
        const float* pb = a.b + (long)sig * a.b_bs + t0;
        const bool vec = vec4 && nfr == SF;
        // Z[k] = E[k] + i O[k] with E = (X[k] + conj X[M-k]) / 2, O = (X[k] - conj X[M-k]) / 2 * exp(+2 pi i k / n_fft);
        // X[0] = 0 (the zero DC row of utils.py:38-39), X[M] real (irfft ignores the Nyquist imaginary part)
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            const int k = 1 + threadIdx.x + i * BT;
            if (k <= (M >> 1)) {
                const int ea = swz(k), eb = swz(M - k);
                float2 Xk[SF], Xm[SF];
                istft_row(a, pa, pb, k, nfr, vec, Xk);
                istft_row(a, pa, pb, M - k, nfr, vec, Xm);
#pragma unroll
                for (int f = 0; f < SF; ++f) {
                    const float2 E = make_float2(0.5f * (Xk[f].x + Xm[f].x), 0.5f * (Xk[f].y - Xm[f].y));
                    const float2 D = make_float2(0.5f * (Xk[f].x - Xm[f].x), 0.5f * (Xk[f].y + Xm[f].y));
                    const float2 O = cmul(D, make_float2(sc[i], ss[i]));
                    x[f * M + ea] = make_float2(E.x - O.y, E.y + O.x);
                    if (k != M - k) x[f * M + eb] = make_float2(E.x + O.y, O.x - E.y);
                }
            }
        }
        if (threadIdx.x == 0) {
            float2 Xn[SF];
            istft_row(a, pa, pb, M, nfr, vec, Xn);
#pragma unroll
            for (int f = 0; f < SF; ++f) x[f * M] = make_float2(0.5f * Xn[f].x, -0.5f * Xn[f].x);
        }
        lds_barrier();
        const float2* z = fft_frames<-1>(x, y, tw, M);
#pragma unroll
        for (int i = 0; i < M_ITERS; ++i) {
            const int m = threadIdx.x + i * BT;
            if (m < M) {
                const int e = swz(m);
#pragma unroll
                for (int f = 0; f < SF; ++f) {
                    if (f >= nfr) break;
                    const float2 v = z[f * M + e];
                    float2* dst = (float2*)(frames + ((long)sig * a.n_frames + t0 + f) * N) + m;
                    *dst = make_float2(v.x * w0[i], v.y * w1[i]);
                }
            }
        }
        lds_barrier();
    }
}

// peak of signal blockIdx.y = max over its `nparts` partial peaks (max is exact in any order), then audio /= peak, 16 B per lane
// where the row allows it (librosa.util.normalize(norm=inf): tiny peaks leave the clip as it is, utils.py:41-42)
__global__ __launch_bounds__(256) void istft_peak_normalize_kernel(float* audio, int len, const float* partial, int nparts) {
    __shared__ float red[4];
    const int sig = blockIdx.y;
    float mx = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) mx = fmaxf(mx, partial[(long)sig * nparts + i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    const float pk = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (!(pk > 1.17549435e-38f)) return;
    float* row = audio + (long)sig * len;
    if ((len & 3) == 0 && (((uintptr_t)audio) & 15) == 0) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (len >> 2); i += gridDim.x * blockDim.x) {
            float4 v = ((float4*)row)[i];
            v.x /= pk; v.y /= pk; v.z /= pk; v.w /= pk;
            ((float4*)row)[i] = v;
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) row[i] /= pk;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-per-frame transforms (round 4; n_fft = 2048, the reference's default: preproc_mdb.py:202-204, and n_fft = 1024).
// The radix-4 Stockham passes above spread one frame over a whole workgroup: 5 passes = 5 LDS round trips + 5 workgroup barriers per
// group of frames, 2 waves per SIMD (72 KB of LDS per workgroup) -- measured latency-bound, 3x off its VALU time.  Here ONE WAVE owns
// one frame: the 1024-point complex transform behind the real 2048-point one is 16 points per lane, factored 16 x 16 x 4:
//     n = l + 64 r            radix-16 over the lane's registers r, twiddle W1024^(l k1)                      (no LDS)
//     exchange 1              lane (k1, j) gets l = j + 4 r' of sub-transform k1            (LDS: 16 writes + 16 reads per lane)
//     radix-16 over r', twiddle W64^(j k2); exchange 2: lane gets the four j of four (k1, k2) pairs           (16 + 16)
//     radix-4 over j          lane l'' holds X[k1 + 16 k2 + 256 k3], k1 = l'' & 15, k2 = 4 q + (l'' >> 4), register 4 q + k3
// Exchanges stay inside the wave's own 8.5 KB region: LDS operations of one wave complete in order, so no barrier and no wait sits
// between a stage's writes and the next stage's reads; the eight waves of a workgroup (eight consecutive frames: rows of the
// (bins, frames) layout are touched 32 B at a time -- at 16 B the row stores alone were 28 of 74 us, one L2 request per 16 B piece)
// meet twice per group.  78.3 KB of LDS per workgroup: 2 workgroups = 16 waves per CU.
// Layouts (pads 68 / 264) are bank-conflict-free for every access: tools/fit/lds_banks.py.
// P points per lane (16: the 1024-point transform of n_fft = 2048; 8: the 512-point one of n_fft = 1024, factored 8 x 8 x 8 with
// pads 72 / 68 and the output already in natural order: X[lane + 64 r] in register r)
template <int P> struct WaveFft;
template <> struct WaveFft<16> {
    static constexpr int M = 1024, REG = 1088, T1N = 1024, T2N = 64;       // REG: float2 per wave region, 16 sub-transforms x (64 + 4 pad)
    __device__ static __forceinline__ int out(int lane, int reg) { return (lane & 15) + 16 * (4 * (reg >> 2) + (lane >> 4)) + 256 * (reg & 3); }
    __device__ static __forceinline__ constexpr int rot_out(int reg) { return (reg >> 2) + 4 * (reg & 3); }   // theta(out) - theta(lane), units of pi / 8
    static constexpr float CD = 0.999995293809576172f, SD = 0.00306795676296597627f;     // cos, sin (pi / 1024)
};
template <> struct WaveFft<8> {
    static constexpr int M = 512, REG = 576, T1N = 512, T2N = 64;          // 8 x (64 + 8 pad)
    __device__ static __forceinline__ int out(int lane, int reg) { return lane + 64 * reg; }
    __device__ static __forceinline__ constexpr int rot_out(int reg) { return 2 * reg; }
    static constexpr float CD = 0.999981175282601109f, SD = 0.00613588464915447536f;     // cos, sin (pi / 512)
};

__device__ __forceinline__ void wave_order() {              // compiler-level ordering of a wave's own LDS traffic (no instruction)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int DIR> __device__ __forceinline__ void radix4(float2& a, float2& b, float2& c, float2& d) {
    const float2 s0 = make_float2(a.x + c.x, a.y + c.y), s1 = make_float2(a.x - c.x, a.y - c.y);
    const float2 s2 = make_float2(b.x + d.x, b.y + d.y), s3 = make_float2(b.x - d.x, b.y - d.y);
    const float2 r3 = DIR > 0 ? make_float2(s3.y, -s3.x) : make_float2(-s3.y, s3.x);      // -/+ i s3
    a = make_float2(s0.x + s2.x, s0.y + s2.y); b = make_float2(s1.x + r3.x, s1.y + r3.y);
    c = make_float2(s0.x - s2.x, s0.y - s2.y); d = make_float2(s1.x - r3.x, s1.y - r3.y);
}
// v * (c -+ i s): the forward (DIR > 0) twiddle exp(-i theta) with cos = c, sin = s
template <int DIR> __device__ __forceinline__ float2 tmul(float2 v, float c, float s) {
    return DIR > 0 ? make_float2(__fmaf_rn(v.x, c, v.y * s), __fmaf_rn(v.y, c, -(v.x * s)))
                   : make_float2(__fmaf_rn(v.x, c, -(v.y * s)), __fmaf_rn(v.y, c, v.x * s));
}
// 16-point transform of v[0..15] in place; the result X[k] sits at v[(k & 3) * 4 + (k >> 2)], i.e. v[i] = X[(i >> 2) + 4 (i & 3)]
template <int DIR> __device__ __forceinline__ void radix16(float2 (&v)[16]) {
    constexpr float C1 = 0.923879532511286756f, S1 = 0.382683432365089772f, R2 = 0.707106781186547524f;
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0) radix4<DIR>(v[r0], v[r0 + 4], v[r0 + 8], v[r0 + 12]);     // over r1: v[r0 + 4 q] = b[r0][q]
    // twiddles w16^(r0 q), w16 = exp(-+ 2 pi i / 16)
    v[5] = tmul<DIR>(v[5], C1, S1);   v[9] = tmul<DIR>(v[9], R2, R2);    v[13] = tmul<DIR>(v[13], S1, C1);     // r0 = 1: w, w^2, w^3
    v[6] = tmul<DIR>(v[6], R2, R2);   v[10] = tmul<DIR>(v[10], 0.f, 1.f); v[14] = tmul<DIR>(v[14], -R2, R2);   // r0 = 2: w^2, w^4, w^6
    v[7] = tmul<DIR>(v[7], S1, C1);   v[11] = tmul<DIR>(v[11], -R2, R2);  v[15] = tmul<DIR>(v[15], -C1, -S1);  // r0 = 3: w^3, w^6, w^9
#pragma unroll
    for (int q = 0; q < 4; ++q) radix4<DIR>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);   // over r0: v[4 q + s] = X[q + 4 s]
}
// 8-point transform in place: v[m] = X[2 m], v[4 + m] = X[2 m + 1] (m = 0..3)
template <int DIR> __device__ __forceinline__ void radix8(float2 (&v)[8]) {
    constexpr float R2 = 0.707106781186547524f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float2 a = v[r], b = v[r + 4];
        v[r] = make_float2(a.x + b.x, a.y + b.y);
        v[r + 4] = make_float2(a.x - b.x, a.y - b.y);
    }
    v[5] = tmul<DIR>(v[5], R2, R2); v[6] = tmul<DIR>(v[6], 0.f, 1.f); v[7] = tmul<DIR>(v[7], -R2, R2);      // (a - b) w8^r
    radix4<DIR>(v[0], v[1], v[2], v[3]);
    radix4<DIR>(v[4], v[5], v[6], v[7]);
}
template <int P> __device__ __forceinline__ constexpr int rP_out(int i) { return P == 16 ? (i >> 2) + 4 * (i & 3) : (i < 4 ? 2 * i : 2 * (i - 4) + 1); }
template <int DIR> __device__ __forceinline__ void radixP(float2 (&v)[16]) { radix16<DIR>(v); }
template <int DIR> __device__ __forceinline__ void radixP(float2 (&v)[8]) { radix8<DIR>(v); }

// T1[k1][l] = exp(-2 pi i l k1 / M), T2[k2][j] = exp(-2 pi i j k2 / 64) (P = 16: j < 4; P = 8: j < 8): built once per workgroup
template <int P> __device__ void wave_fft_tables(float2* T1, float2* T2) {
    constexpr int M = WaveFft<P>::M, J = P == 16 ? 4 : 8;
    for (int e = threadIdx.x; e < M; e += blockDim.x) {
        float sn, cs;
        sincospif(-2.0f * (float)((e & 63) * (e >> 6)) / (float)M, &sn, &cs);
        T1[e] = make_float2(cs, sn);
    }
    for (int e = threadIdx.x; e < 64; e += blockDim.x) {
        float sn, cs;
        sincospif(-2.0f * (float)((e % J) * (e / J)) / 64.0f, &sn, &cs);
        T2[e] = make_float2(cs, sn);
    }
}

// in: v[r] = z[lane + 64 r].  out: v[reg] = Z[WaveFft<P>::out(lane, reg)]   (unnormalised)
template <int DIR>
__device__ __forceinline__ void wave_fft(float2 (&v)[16], float2* reg, const float2* T1, const float2* T2, int lane) {
    radix16<DIR>(v);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k1 = rP_out<16>(i);
        if (k1) { const float2 w = T1[64 * k1 + lane]; v[i] = tmul<DIR>(v[i], w.x, -w.y); }   // (the table holds exp(-i theta): sin = -w.y)
        reg[68 * k1 + lane] = v[i];
    }
    wave_order();
    const int kb = lane >> 2, j = lane & 3;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = reg[68 * kb + j + 4 * r];
    wave_order();
    radix16<DIR>(v);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k2 = rP_out<16>(i);
        if (k2) { const float2 w = T2[4 * k2 + j]; v[i] = tmul<DIR>(v[i], w.x, -w.y); }
        reg[264 * j + kb + 16 * k2] = v[i];
    }
    wave_order();
    const int k1 = lane & 15, kh = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) v[4 * q + jj] = reg[264 * jj + k1 + 16 * (4 * q + kh)];
    wave_order();
#pragma unroll
    for (int q = 0; q < 4; ++q) radix4<DIR>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
template <int DIR>
__device__ __forceinline__ void wave_fft(float2 (&v)[8], float2* reg, const float2* T1, const float2* T2, int lane) {
    radix8<DIR>(v);                                           // over r: v[i] = y[k1], k1 = rP_out<8>(i)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k1 = rP_out<8>(i);
        if (k1) { const float2 w = T1[64 * k1 + lane]; v[i] = tmul<DIR>(v[i], w.x, -w.y); }
        reg[72 * k1 + lane] = v[i];
    }
    wave_order();
    const int kb = lane >> 3, j = lane & 7;                   // lane (k1, j) gets l = j + 8 r' of sub-transform k1
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = reg[72 * kb + j + 8 * r];
    wave_order();
    radix8<DIR>(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k2 = rP_out<8>(i);
        if (k2) { const float2 w = T2[8 * k2 + j]; v[i] = tmul<DIR>(v[i], w.x, -w.y); }
        reg[68 * j + kb + 8 * k2] = v[i];
    }
    wave_order();
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) v[jj] = reg[68 * jj + lane];          // lane = k1 + 8 k2 holds the eight j of its pair
    wave_order();
    radix8<DIR>(v);                                           // over j: v[i] = X[lane + 64 k3], k3 = rP_out<8>(i)
    float2 t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[rP_out<8>(i)] = v[i];       // (compile-time permutation: register k3 holds X[lane + 64 k3])
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = t[i];
}

// Periodic Hann window of length n_fft at the sample pair (2 m, 2 m + 1) from (cos, sin) of theta_m = 2 pi m / M, which the lanes
// compose from a per-lane base angle and compile-time rotations (multiples of pi / 8): the window values of a frame cost ~8 flops
// each instead of 2 P registers per lane held for the whole kernel.
constexpr float W8C[16] = {1.f, 0.923879532511286756f, 0.707106781186547524f, 0.382683432365089772f, 0.f, -0.382683432365089772f,
                           -0.707106781186547524f, -0.923879532511286756f, -1.f, -0.923879532511286756f, -0.707106781186547524f,
                           -0.382683432365089772f, 0.f, 0.382683432365089772f, 0.707106781186547524f, 0.923879532511286756f};
constexpr float W8S[16] = {0.f, 0.382683432365089772f, 0.707106781186547524f, 0.923879532511286756f, 1.f, 0.923879532511286756f,
                           0.707106781186547524f, 0.382683432365089772f, 0.f, -0.382683432365089772f, -0.707106781186547524f,
                           -0.923879532511286756f, -1.f, -0.923879532511286756f, -0.707106781186547524f, -0.382683432365089772f};
// window pair at theta = base + rot pi / 8, base given as (cb, sb)
template <int P> __device__ __forceinline__ void hann_pair(float cb, float sb, int rot, float& w0, float& w1) {
    const float c = __fmaf_rn(cb, W8C[rot & 15], -(sb * W8S[rot & 15])), s = __fmaf_rn(sb, W8C[rot & 15], cb * W8S[rot & 15]);
    w0 = 0.5f - 0.5f * c;
    w1 = 0.5f - 0.5f * __fmaf_rn(c, WaveFft<P>::CD, -(s * WaveFft<P>::SD));       // half a pair further: + pi / M
}

constexpr int NW = 8;                                       // waves = frames per workgroup: row segments of 8 frames (32 B)
constexpr int WT = NW * 64;                                 // threads; a thread pair owns the bin pairs (k, M - k), k = 1 + (tid >> 1) + 256 i

// dynamic LDS of the wave-per-frame kernels: NW wave regions + T1 + T2 (P = 16: 78.3 KB, two workgroups = 16 waves per CU; P = 8: 41.5 KB)
template <int P> constexpr size_t wave_lds() { return (size_t)(NW * WaveFft<P>::REG + WaveFft<P>::T1N + WaveFft<P>::T2N) * sizeof(float2); }

// one output row, 4 of the group's 8 frames (half = 0 / 1): as stft_store_row
__device__ __forceinline__ void stft_store_row4(const pg_stft_args& a, float* o_re, float* o_im, long row, int half, int nfr, bool vec,
                                                float (&re)[4], float (&im)[4]) {
    if (a.polar) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {                         // (fenced: measured 87 us against 91-93 with the four chains interleaved)
            const float r = re[f], i = im[f];
            pg_polar_one(r, i, 1, re[f], im[f]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* pr = o_re + row * a.n_frames + 4 * half;
    float* pi = o_im + row * a.n_frames + 4 * half;
#if (PG_W_ABL & 4)
    asm volatile("" :: "v"(re[0]), "v"(re[1]), "v"(re[2]), "v"(re[3]), "v"(im[0]), "v"(im[1]), "v"(im[2]), "v"(im[3]), "v"(pr), "v"(pi));
    return;
#endif
    if (vec) {
        *(float4*)pr = make_float4(re[0], re[1], re[2], re[3]);
        *(float4*)pi = make_float4(im[0], im[1], im[2], im[3]);
    } else {
#pragma unroll
        for (int f = 0; f < 4; ++f)
            if (4 * half + f < nfr) { pr[f] = re[f]; pi[f] = im[f]; }
    }
}

// STFT, n_fft = 128 P: wave w of a workgroup transforms frame t0 + w of its group of NW frames; the split X[k] = E[k] + w^k O[k], the
// optional polar epilogue and the row stores (two 16 B pieces = 32 B per row and workgroup) are the workgroup-wide phase, reading the
// waves' spectra from LDS.
template <bool CHUNKED, int P>
__global__ __launch_bounds__(WT, CHUNKED ? 2 : 4) void stft_w_kernel(const pg_stft_args a) {
    using W = WaveFft<P>;
    extern __shared__ __attribute__((aligned(16))) float2 wsm[];
    float2 (*regs)[W::REG] = (float2 (*)[W::REG])wsm;
    float2* T1 = wsm + NW * W::REG; float2* T2 = T1 + W::T1N;
    constexpr int M = W::M, N = 2 * M, PAIRS = M / 512;       // bin pairs per thread in the split phase
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = (a.n_frames + NW - 1) / NW, total = a.n_signals * groups;
    wave_fft_tables<P>(T1, T2);
    float cb, sb, sc, ss;                                     // (cos, sin)(2 pi lane / M): base of this lane's window angles; split factor w^k
    sincospif(2.0f * (float)lane / (float)M, &sb, &cb);
    float sc2 = 0.f, ss2 = 0.f;                               // split factors of this thread's bin pairs k = 1 + (tid >> 1) + 256 i
    sincospif(-(float)(1 + (int)(threadIdx.x >> 1)) / (float)M, &ss, &sc);
    if (PAIRS > 1) sincospif(-(float)(257 + (int)(threadIdx.x >> 1)) / (float)M, &ss2, &sc2);
    const bool vec2 = !CHUNKED && ((a.hop | a.n_samples) & 1) == 0 && (((uintptr_t)a.y) & 7) == 0;   // sample pairs are 8 B aligned
    const bool vec4 = (a.n_frames & 3) == 0 && (((uintptr_t)a.out) & 15) == 0;               // row segments are 16 B aligned
    // This wave's frame of group g as RAW sample pairs v[r] = (y[2 m], y[2 m + 1]), m = lane + 64 r (reflect padding by index math: the
    // bit-exact part of the contract).  Issued one group AHEAD, into the registers the finished transform has just vacated, so the
    // samples travel while the workgroup splits, converts and stores the current group.
    float2 v[P];
    auto load_frame = [&](int g) {
        const int sig = g / groups, t0 = (g - sig * groups) * NW;
        if (wave >= min(NW, a.n_frames - t0)) return;
        const float* sgn; int lim;
        if (CHUNKED) { const Src src = signal_src(a, sig); sgn = src.p; lim = src.lim; }
        else { sgn = a.y + (long)sig * a.n_samples; lim = a.n_samples; }
        auto at = [&](int q) { return (!CHUNKED || q < lim) ? sgn[q] : 0.f; };
        const int start = (t0 + wave) * a.hop - M;                              // frame tap k sits at sample start + k
        const bool inside = start >= 0 && start + N <= a.n_samples;
        int lo = lane;
        asm volatile("" : "+v"(lo));                          // (opaque: keeps the per-lane sample offsets out of the loop-invariant registers)
#pragma unroll
        for (int r = 0; r < P; ++r) {
            const int p = start + 2 * (lo + 64 * r);
            float v0, v1;
            if (PG_W_ABL & 1) { v0 = (float)p; v1 = 1.f; }
            else if (inside) {
                if (vec2) { const float2 t = *(const float2*)(sgn + p); v0 = t.x; v1 = t.y; }
                else { v0 = at(p); v1 = at(p + 1); }
            } else { v0 = at(reflect_index(p, a.n_samples)); v1 = at(reflect_index(p + 1, a.n_samples)); }
            v[r] = make_float2(v0, v1);
        }
    };
    __syncthreads();
    GroupWalk gw = group_walk(total);
    if (gw.g < gw.end) load_frame(gw.g);
    for (; gw.g < gw.end; gw.g += gw.step) {
        const int sig = gw.g / groups, t0 = (gw.g - sig * groups) * NW;
        const int nfr = min(NW, a.n_frames - t0);
        if (wave < nfr) {
            float cbo = cb, sbo = sb;
            asm volatile("" : "+v"(cbo), "+v"(sbo));          // (opaque: or the window values are hoisted out of the loop and spilled)
#pragma unroll
            for (int r = 0; r < P; ++r) {
                float w0, w1;                                                   // m = lane + 64 r: theta = base + r 2 pi / P
                hann_pair<P>(cbo, sbo, r * (16 / P), w0, w1);
                v[r] = make_float2(v[r].x * w0, v[r].y * w1);
            }
            if (!(PG_W_ABL & 2)) wave_fft<1>(v, regs[wave], T1, T2, lane);
#pragma unroll
            for (int r = 0; r < P; ++r) regs[wave][W::out(lane, r)] = v[r];     // natural order
        }
        lds_barrier();
        if (gw.g + gw.step < gw.end) load_frame(gw.g + gw.step);              // (v is free: the spectra are in LDS)
        float* o_re = a.out + ((long)sig * 2 * M) * a.n_frames + t0;
        float* o_im = o_re + (long)M * a.n_frames;
        if (PG_W_ABL & 8) { lds_barrier(); continue; }
        // Lanes 2 i and 2 i + 1 take the two 16 B halves of the SAME rows, so a wave's store instruction covers 32 rows x 32 contiguous
        // bytes (one L2 request per row instead of two); a thread does PAIRS bin pairs: k = 1 + (tid >> 1) + 256 i.
        int tid_o = threadIdx.x;
        asm volatile("" : "+v"(tid_o));                       // (opaque: the row addresses below are computed here, not carried across the barrier)
        const int half = tid_o & 1;
        if (4 * half < nfr) {
            const bool vec = vec4 && 4 * half + 4 <= nfr;
#pragma unroll 1
            for (int i = 0; i < PAIRS; ++i) {                                 // bins k and M-k from Z[k], Z[M-k]; DC dropped
                const int k = 1 + (tid_o >> 1) + 256 * i;                     // 1 .. M / 2
                const float sn = i ? ss2 : ss, cs = i ? sc2 : sc;
                float rk[4], ik[4], rm[4], im[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) {                                 // (frames past the end: never stored)
                    const float2 A = regs[4 * half + f][k], B = regs[4 * half + f][M - k];
                    const float2 E = make_float2(0.5f * (A.x + B.x), 0.5f * (A.y - B.y));
                    const float2 O = make_float2(0.5f * (A.y + B.y), -0.5f * (A.x - B.x));
                    const float2 T = cmul(O, make_float2(cs, sn));
                    rk[f] = E.x + T.x; ik[f] = E.y + T.y;
                    rm[f] = E.x - T.x; im[f] = T.y - E.y;
                }
                stft_store_row4(a, o_re, o_im, k - 1, half, nfr, vec, rk, ik);
                if (k != M - k) stft_store_row4(a, o_re, o_im, M - k - 1, half, nfr, vec, rm, im);
                else {                                                        // the self-paired threads also own the Nyquist bin:
                    float rn[4], in[4];                                       // X[M] = Re Z0 - Im Z0
#pragma unroll
                    for (int f = 0; f < 4; ++f) { const float2 Z0 = regs[4 * half + f][0]; rn[f] = Z0.x - Z0.y; in[f] = 0.f; }
                    stft_store_row4(a, o_re, o_im, M - 1, half, nfr, vec, rn, in);
                }
            }
        }
        lds_barrier();                                                        // the regions are the next group's work space
    }
}

// the workgroup-wide build of the NW half-length spectra Z_f[k] (natural order) from the spectrum rows of one group: lanes 2 i and
// 2 i + 1 read the two 16 B halves of the same rows (a wave's load instruction covers 32 rows x 32 contiguous bytes)
template <int P>
__device__ __forceinline__ void istft_build_w(const pg_istft_args& a, float2 (*regs)[WaveFft<P>::REG], int sig, int t0, int nfr, bool vec4,
                                              float sc, float ss, float sc2, float ss2) {
    constexpr int M = WaveFft<P>::M, PAIRS = M / 512;
    const int half = threadIdx.x & 1, n4 = nfr - 4 * half;
    if (n4 <= 0) return;
    const float* pa = a.a + (long)sig * a.a_bs + t0 + 4 * half;
    const float* pb = a.b + (long)sig * a.b_bs + t0 + 4 * half;
    const bool vec = vec4 && n4 >= 4;
#pragma unroll 1
    for (int i = 0; i < PAIRS; ++i) {                         // Z[k] = E[k] + i O[k] (see istft_frames4_kernel)
        const int k = 1 + (int)(threadIdx.x >> 1) + 256 * i;
        const float sn = i ? ss2 : ss, cs = i ? sc2 : sc;
        float2 Xk[SF], Xm[SF];
        istft_row(a, pa, pb, k, n4, vec, Xk);
        istft_row(a, pa, pb, k == M - k ? M : M - k, n4, vec, Xm);           // (the self-paired threads read the Nyquist row here)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float2 Xo = k == M - k ? Xk[f] : Xm[f];
            const float2 E = make_float2(0.5f * (Xk[f].x + Xo.x), 0.5f * (Xk[f].y - Xo.y));
            const float2 D = make_float2(0.5f * (Xk[f].x - Xo.x), 0.5f * (Xk[f].y + Xo.y));
            const float2 O = cmul(D, make_float2(cs, sn));
            regs[4 * half + f][k] = make_float2(E.x - O.y, E.y + O.x);
            if (k != M - k) regs[4 * half + f][M - k] = make_float2(E.x + O.y, O.x - E.y);
            else regs[4 * half + f][0] = make_float2(0.5f * Xm[f].x, -0.5f * Xm[f].x);      // Z[0]: X[0] = 0, X[M] real
        }
    }
}

// ISTFT frames, n_fft = 128 P: the workgroup builds the spectra together, then every wave inverts its own frame and stores it
// windowed, 8 B per lane.  (A prefetch of the next group's rows during the transforms needs 32 more registers than two workgroups per
// CU leave: measured 151 us at one workgroup per CU with it against 125 us without.)
template <int P>
__global__ __launch_bounds__(WT, 4) void istft_frames_w_kernel(const pg_istft_args a, float* frames) {
    using W = WaveFft<P>;
    extern __shared__ __attribute__((aligned(16))) float2 wsm[];
    float2 (*regs)[W::REG] = (float2 (*)[W::REG])wsm;
    float2* T1 = wsm + NW * W::REG; float2* T2 = T1 + W::T1N;
    constexpr int M = W::M, N = 2 * M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = (a.n_frames + NW - 1) / NW, total = a.n_signals * groups;
    wave_fft_tables<P>(T1, T2);
    float cb, sb, sc, ss, sc2 = 0.f, ss2 = 0.f;               // (cos, sin)(2 pi lane / M): base of the window angles of this lane's outputs
    sincospif(2.0f * (float)lane / (float)M, &sb, &cb);
    const float inv = 1.0f / (float)M;
    sincospif((float)(1 + (int)(threadIdx.x >> 1)) / (float)M, &ss, &sc);          // exp(+2 pi i k / n_fft) of this thread's bin pairs
    if (M > 512) sincospif((float)(257 + (int)(threadIdx.x >> 1)) / (float)M, &ss2, &sc2);
    const bool vec4 = (a.n_frames & 3) == 0 && ((a.a_bs | a.b_bs) & 3) == 0 && ((((uintptr_t)a.a) | ((uintptr_t)a.b)) & 15) == 0;
    __syncthreads();
    for (GroupWalk gw = group_walk(total); gw.g < gw.end; gw.g += gw.step) {
        const int sig = gw.g / groups, t0 = (gw.g - sig * groups) * NW;
        const int nfr = min(NW, a.n_frames - t0);
        istft_build_w<P>(a, regs, sig, t0, nfr, vec4, sc, ss, sc2, ss2);
        lds_barrier();
        if (wave < nfr) {
            float2 v[P];
#pragma unroll
            for (int r = 0; r < P; ++r) v[r] = regs[wave][lane + 64 * r];
            wave_order();
            if (!(PG_W_ABL & 2)) wave_fft<-1>(v, regs[wave], T1, T2, lane);
            int lo = lane;
            asm volatile("" : "+v"(lo));                      // (opaque: keeps the store addresses out of the loop-invariant registers)
            float2* dst = (float2*)(frames + ((long)sig * a.n_frames + t0 + wave) * N);
            float cbo = cb, sbo = sb;
            asm volatile("" : "+v"(cbo), "+v"(sbo));          // (opaque: or the window values are hoisted out of the loop)
#pragma unroll
            for (int r = 0; r < P; ++r) {
                float w0, w1;
                hann_pair<P>(cbo, sbo, W::rot_out(r), w0, w1);
                dst[W::out(lo, r)] = make_float2(v[r].x * (inv * w0), v[r].y * (inv * w1));
            }
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ISTFT at hop = n_fft / 4 (2048 / 512, the reference's defaults; 1024 / 256) with the overlap-add INSIDE the transform kernel
// (round 4): a workgroup's eight windowed frames never leave LDS.  Of the 8 hop + (n_fft - hop) output positions they touch, blocks
// 3..7 (of hop samples) are complete and are finalised on the spot (window-sum-square division, n_fft / 2 trim, peak, one 16 B store
// per lane); blocks 0..2 ("head") still miss the previous group's last three frames and blocks 8..10 ("tail") are this group's share
// of the next group's head: the head is stored to the audio buffer as a partial sum, the tail to a small workspace (3 hop floats per
// group), and a seam kernel adds the two -- exactly two addends per sample, in a fixed order -- and finalises those 3 / 8 of the
// samples.  The frame workspace of the three-kernel path (134 MB at 64 x 256 frames of 2048: written, then read again by the
// overlap-add) is gone: 134 + 33 + 12.5 MB in the main kernel, 37 MB at the seams.
constexpr int OW_COVER = 4, OW_HB = OW_COVER - 1;            // frames covering a sample; head / tail blocks per group

struct WssCtx { float sd, cd, sh, ch; float iw[4]; };
// a thread's constants: window rotations by one sample / one hop, and the interior window-sum-square of its four positions
// (positions 4 e + j with e = tid + T i: the residue mod hop does not depend on i because hop divides 4 T)
__device__ __forceinline__ WssCtx wss_ctx(int first_pos, int N, int hop) {
    WssCtx c;
    sincospif(2.0f / (float)N, &c.sd, &c.cd);
    sincospif(2.0f * (float)hop / (float)N, &c.sh, &c.ch);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
        for (int t = 0; t < N / hop; ++t) { const float w = hann((first_pos + j) % hop + t * hop, N); acc += w * w; }
        c.iw[j] = acc;
    }
    return c;
}
// four consecutive output samples at padded position ip (multiple of 4): divide the overlap-added sums by librosa's
// window_sumsquare over the frames that exist there (utils.py:40, librosa.istft); returns max |y|
__device__ __forceinline__ float ola_finalize(float4& acc, int ip, int N, int hop, int n_frames, const WssCtx& c) {
    int t_hi = ip / hop; if (t_hi > n_frames - 1) t_hi = n_frames - 1;
    int t_lo = (ip - N + hop) / hop; if (ip - N + 1 <= 0) t_lo = 0;
    float wss[4];
    if (t_hi - t_lo + 1 == N / hop) {                                       // interior: the thread's constants
        wss[0] = c.iw[0]; wss[1] = c.iw[1]; wss[2] = c.iw[2]; wss[3] = c.iw[3];
    } else {                                                                // ends of the signal: by rotation
        float cs[4], sn[4];
        wss[0] = wss[1] = wss[2] = wss[3] = 0.f;
        sincospif(2.0f * (float)(ip - t_lo * hop) / (float)N, &sn[0], &cs[0]);
#pragma unroll
        for (int j = 1; j < 4; ++j) { cs[j] = cs[j - 1] * c.cd - sn[j - 1] * c.sd; sn[j] = sn[j - 1] * c.cd + cs[j - 1] * c.sd; }
        for (int t = t_lo; t <= t_hi; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = 0.5f - 0.5f * cs[j];
                wss[j] += w * w;
                const float cn = cs[j] * c.ch + sn[j] * c.sh;               // the next frame sees this sample hop taps earlier
                sn[j] = sn[j] * c.ch - cs[j] * c.sh;
                cs[j] = cn;
            }
        }
    }
    acc.x = wss[0] > 1.17549435e-38f ? acc.x / wss[0] : acc.x;
    acc.y = wss[1] > 1.17549435e-38f ? acc.y / wss[1] : acc.y;
    acc.z = wss[2] > 1.17549435e-38f ? acc.z / wss[2] : acc.z;
    acc.w = wss[3] > 1.17549435e-38f ? acc.w / wss[3] : acc.w;
    return fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w)));
}

// partial[] layout of this path, per signal: [groups] peaks of the main kernel's finalised samples, then [groups] peaks of the seams
template <int P>
__global__ __launch_bounds__(WT, 4) void istft_ola_w_kernel(const pg_istft_args a, float* tails, float* partial) {
    using W = WaveFft<P>;
    extern __shared__ __attribute__((aligned(16))) float2 wsm[];
    __shared__ float red[NW];
    float2 (*regs)[W::REG] = (float2 (*)[W::REG])wsm;
    float2* T1 = wsm + NW * W::REG; float2* T2 = T1 + W::T1N;
    constexpr int M = W::M, N = 2 * M, hop = N / OW_COVER;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = (a.n_frames + NW - 1) / NW, total = a.n_signals * groups;
    const int len = hop * (a.n_frames - 1);
    wave_fft_tables<P>(T1, T2);
    float cb, sb, sc, ss, sc2 = 0.f, ss2 = 0.f;
    sincospif(2.0f * (float)lane / (float)M, &sb, &cb);
    const float inv = 1.0f / (float)M;
    sincospif((float)(1 + (int)(threadIdx.x >> 1)) / (float)M, &ss, &sc);
    if (M > 512) sincospif((float)(257 + (int)(threadIdx.x >> 1)) / (float)M, &ss2, &sc2);
    const bool vec4 = (a.n_frames & 3) == 0 && ((a.a_bs | a.b_bs) & 3) == 0 && ((((uintptr_t)a.a) | ((uintptr_t)a.b)) & 15) == 0;
    const WssCtx wc = wss_ctx(4 * (int)threadIdx.x, N, hop);
    __syncthreads();
    for (GroupWalk gw = group_walk(total); gw.g < gw.end; gw.g += gw.step) {
        const int sig = gw.g / groups, grp = gw.g - sig * groups, t0 = grp * NW;
        const int nfr = min(NW, a.n_frames - t0);
        istft_build_w<P>(a, regs, sig, t0, nfr, vec4, sc, ss, sc2, ss2);
        lds_barrier();
        if (wave < nfr) {                                     // this wave's frame: inverse transform, window, back to its region as n_fft reals
            float2 v[P];
#pragma unroll
            for (int r = 0; r < P; ++r) v[r] = regs[wave][lane + 64 * r];
            wave_order();
            wave_fft<-1>(v, regs[wave], T1, T2, lane);
            int lo = lane;
            asm volatile("" : "+v"(lo));
            float cbo = cb, sbo = sb;
            asm volatile("" : "+v"(cbo), "+v"(sbo));
#pragma unroll
            for (int r = 0; r < P; ++r) {
                float w0, w1;
                hann_pair<P>(cbo, sbo, W::rot_out(r), w0, w1);
                regs[wave][W::out(lo, r)] = make_float2(v[r].x * (inv * w0), v[r].y * (inv * w1));
            }
        }
        lds_barrier();
        // overlap-add over the group's frames: position pr (relative to t0 hop) gets frame f's sample pr - f hop
        const bool first = grp == 0, last = grp == groups - 1;
        float mx = 0.f;
        float* out = a.audio + (long)sig * len;
        float* tl = tails + ((long)sig * groups + grp) * (OW_HB * hop);
        for (int pr = 4 * (int)threadIdx.x; pr < (NW + OW_HB) * hop; pr += 4 * WT) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int f = 0; f < NW; ++f) {
                const int n = pr - f * hop;
                if (f < nfr && n >= 0 && n < N) {
                    const float4 q = *(const float4*)((const float*)regs[f] + n);
                    acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
                }
            }
            const int blk = pr / hop, ip = t0 * hop + pr, i0 = ip - (N >> 1);
            if (blk >= NW && !last) { *(float4*)(tl + (pr - NW * hop)) = acc; continue; }        // tail: the next group's seam adds it
            if (i0 < 0 || i0 >= len) continue;                                                  // trimmed away
            if (blk < OW_HB && !first) { *(float4*)(out + i0) = acc; continue; }                 // head: partial sum, finalised at the seam
            mx = fmaxf(mx, ola_finalize(acc, ip, N, hop, a.n_frames, wc));
            *(float4*)(out + i0) = acc;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        if (lane == 0) red[wave] = mx;
        lds_barrier();                                        // (also: the regions are the next group's work space)
        if (threadIdx.x == 0) {
            for (int i = 1; i < NW; ++i) mx = fmaxf(mx, red[i]);
            partial[(long)sig * 2 * groups + grp] = mx;
            if (first) partial[(long)sig * 2 * groups + groups] = 0.f;       // group 0 has no seam in front of it
        }
    }
}

// the seams: head blocks of every group but the first = their own partial sums (in the audio buffer) + the previous group's tail
__global__ __launch_bounds__(256) void istft_seam_kernel(const pg_istft_args a, const float* tails, float* partial) {
    __shared__ float red[4];
    const int N = 2 * a.bins, hop = a.hop;                                   // (hop = n_fft / 4 on this path)
    const int groups = (a.n_frames + NW - 1) / NW;
    const int sig = blockIdx.y, grp = 1 + blockIdx.x;                        // grid (groups - 1, signals)
    const int len = hop * (a.n_frames - 1), t0 = grp * NW;
    float* out = a.audio + (long)sig * len;
    const float* tl = tails + ((long)sig * groups + grp - 1) * (OW_HB * hop);
    const WssCtx wc = wss_ctx(4 * (int)threadIdx.x, N, hop);                 // (256 threads: 1024 positions per pass, hop divides it)
    float mx = 0.f;
    for (int pr = 4 * (int)threadIdx.x; pr < OW_HB * hop; pr += 4 * 256) {
        const int ip = t0 * hop + pr, i0 = ip - (N >> 1);
        if (i0 < 0 || i0 >= len) continue;
        float4 acc = *(const float4*)(out + i0);
        const float4 t = *(const float4*)(tl + pr);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        mx = fmaxf(mx, ola_finalize(acc, ip, N, hop, a.n_frames, wc));
        *(float4*)(out + i0) = acc;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long)sig * 2 * groups + groups + grp] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ void frame_index_kernel(int n_samples, int n_fft, int hop, int n_frames, int* idx) {
    const long total = (long)n_frames * n_fft;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int t = (int)(e / n_fft), k = (int)(e - (long)t * n_fft);
        idx[e] = reflect_index(t * hop + k - (n_fft >> 1), n_samples);
    }
}

// inverse transform of one frame -> windowed real frame in the workspace
__global__ __launch_bounds__(FFT_THREADS) void istft_frames_kernel(const pg_istft_args a, float* frames) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int bins = a.bins, N = 2 * bins;
    float2* buf0 = smem; float2* buf1 = smem + N; float2* tw = smem + 2 * N;
    const int t = blockIdx.x % a.n_frames, sig = blockIdx.x / a.n_frames;
    const float* pa = a.a + (long)sig * a.a_bs + t;
    const float* pb = a.b + (long)sig * a.b_bs + t;
    fill_twiddles(tw, N);
    for (int k = threadIdx.x; k < bins; k += blockDim.x) {           // spectrum row k is FFT bin k+1
        const float va = pa[(long)k * a.n_frames], vb = pb[(long)k * a.n_frames];
        float re, im;
        if (a.mode == 0) {                                           // demo.py:39: (exp(m) - 1) e^{j phi}
            const float mag = pg_expm1_ref(va);
            float s, c;
            pg_sincos(vb, s, c);
            re = mag * c; im = mag * s;
        } else { re = va; im = vb; }
        const int bin = k + 1;
        if (bin == bins) buf0[bin] = make_float2(re, 0.f);           // Nyquist: imaginary part ignored by irfft
        else { buf0[bin] = make_float2(re, im); buf0[N - bin] = make_float2(re, -im); }
    }
    if (threadIdx.x == 0) buf0[0] = make_float2(0.f, 0.f);           // zero DC row, utils.py:38-39
    __syncthreads();
    const float2* x = fft_lds(buf0, buf1, tw, N, -1.f);
    float* f = frames + ((long)sig * a.n_frames + t) * N;
    const float inv = 1.0f / (float)N;
    for (int n = threadIdx.x; n < N; n += blockDim.x) f[n] = x[n].x * inv * hann(n, N);
}

// Overlap-add, 4 consecutive samples per thread: the <= n_fft/hop frames covering them are read as 16 B pieces, the
// squared-window sum follows the frames by rotating (cos, sin) instead of re-evaluating the window, and each workgroup
// leaves its peak in partial[] (no atomics: thousands of same-line device-scope atomics were most of the old kernel's time).
constexpr int OLA_SPT = 4;

__global__ __launch_bounds__(256) void istft_ola4_kernel(const pg_istft_args a, const float* frames, float* partial) {
    __shared__ float scratch[4];
    const int N = 2 * a.bins, len = a.hop * (a.n_frames - 1);
    const int sig = blockIdx.y;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * OLA_SPT;
    const float* fs = frames + (long)sig * a.n_frames * N;
    float mx = 0.f;
    if (i0 < len) {
        float* out = a.audio + (long)sig * len + i0;
        const bool vec = (a.hop & 3) == 0 && N <= 16 * a.hop && i0 + OLA_SPT <= len && (((uintptr_t)a.audio) & 15) == 0;
        if (vec) {                                   // ip, hop, N/2 are multiples of 4: the 4 samples share their frame range
            const int ip = i0 + (N >> 1);
            int t_hi = ip / a.hop; if (t_hi > a.n_frames - 1) t_hi = a.n_frames - 1;
            int t_lo = (ip - N + a.hop) / a.hop; if (ip - N + 1 <= 0) t_lo = 0;
            float sd, cd, sh, ch, c[OLA_SPT], sn[OLA_SPT];
            sincospif(2.0f / (float)N, &sd, &cd);
            sincospif(2.0f * (float)a.hop / (float)N, &sh, &ch);
            sincospif(2.0f * (float)(ip - t_lo * a.hop) / (float)N, &sn[0], &c[0]);
#pragma unroll
            for (int j = 1; j < OLA_SPT; ++j) { c[j] = c[j - 1] * cd - sn[j - 1] * sd; sn[j] = sn[j - 1] * cd + c[j - 1] * sd; }
            float acc[OLA_SPT] = {0.f, 0.f, 0.f, 0.f}, wss[OLA_SPT] = {0.f, 0.f, 0.f, 0.f};
            for (int t = t_lo; t <= t_hi; ++t) {
                const float4 v = *(const float4*)(fs + (long)t * N + (ip - t * a.hop));
                acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
#pragma unroll
                for (int j = 0; j < OLA_SPT; ++j) {
                    const float w = 0.5f - 0.5f * c[j];
                    wss[j] += w * w;
                    const float cn = c[j] * ch + sn[j] * sh;             // the next frame sees this sample hop taps earlier
                    sn[j] = sn[j] * ch - c[j] * sh;
                    c[j] = cn;
                }
            }
            float y[OLA_SPT];
#pragma unroll
            for (int j = 0; j < OLA_SPT; ++j) {
                y[j] = wss[j] > 1.17549435e-38f ? acc[j] / wss[j] : acc[j];
                mx = fmaxf(mx, fabsf(y[j]));
            }
            *(float4*)out = make_float4(y[0], y[1], y[2], y[3]);
        } else {
            for (int j = 0; j < OLA_SPT && i0 + j < len; ++j) {
                const int ip = i0 + j + (N >> 1);
                int t_hi = ip / a.hop; if (t_hi > a.n_frames - 1) t_hi = a.n_frames - 1;
                int t_lo = (ip - N + a.hop) / a.hop; if (ip - N + 1 <= 0) t_lo = 0;          // ceil((ip-N+1)/hop), clamped
                float sum = 0.f, wss = 0.f;
                for (int t = t_lo; t <= t_hi; ++t) {
                    const int n = ip - t * a.hop;
                    const float w = hann(n, N);
                    sum += fs[(long)t * N + n];
                    wss += w * w;
                }
                const float yv = wss > 1.17549435e-38f ? sum / wss : sum;
                out[j] = yv;
                mx = fmaxf(mx, fabsf(yv));
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long)sig * gridDim.x + blockIdx.x] = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}

// Griffin-Lim projection onto the target magnitudes: keep the phase of S, impose mag (utils.py:122-124)
// (blockIdx.y = clip: S / spec_out (n, 2, bins, frames), mag (n, bins, frames), x (n, 2 bins - 2, frames))
__global__ __launch_bounds__(256) void gl_project_kernel(const pg_gl_args a) {
    const long total = (long)a.bins * a.frames;
    const float* S = a.S + 2 * total * blockIdx.y;
    const float* mag = a.mag + total * blockIdx.y;
    float* x = a.x + (long)(2 * a.bins - 2) * a.frames * blockIdx.y;
    float* so = a.spec_out ? a.spec_out + 2 * total * blockIdx.y : nullptr;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / a.frames), t = (int)(e - (long)r * a.frames);
        float re = S[e], im = S[total + e];
        pg_complex_from_parts(re, im);                      // np.angle on re + 1j*im
        const float mod = hypotf(re, im), m = mag[e];
        const float c = mod > 0.f ? re / mod : 1.f, s = mod > 0.f ? im / mod : 0.f;   // angle(0) = 0
        const float nr = m * c, ni = m * s;
        if (so) { so[e] = nr; so[total + e] = ni; }
        x[e] = nr;
        if (r >= 1 && r <= a.bins - 2) x[(long)(a.bins + r - 1) * a.frames + t] = ni;
    }
}

// overlap-add of (n_fft, frames)-major windowed frames, any even n_fft
// (blockIdx.y = clip: fr (n, n_fft, frames), audio (n, len), one peak word per clip)
__global__ __launch_bounds__(256) void ola_nt_kernel(pg_ola_args a, unsigned* peak) {
    __shared__ float scratch[16];
    const int N = a.n_fft, len = a.hop * (a.frames - 1);
    a.fr += (long)N * a.frames * blockIdx.y;
    a.audio += (long)len * blockIdx.y;
    peak += blockIdx.y;
    float mx = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
        const int ip = i + (N >> 1);
        int t_hi = ip / a.hop; if (t_hi > a.frames - 1) t_hi = a.frames - 1;
        int t_lo = (ip - N + a.hop) / a.hop; if (ip - N + 1 <= 0) t_lo = 0;
        float s = 0.f, wss = 0.f;
        for (int t = t_lo; t <= t_hi; ++t) {
            const int n = ip - t * a.hop;
            const float w = hann(n, N);
            s += a.fr[(long)n * a.frames + t];
            wss += w * w;
        }
        const float yv = wss > 1.17549435e-38f ? s / wss : s;
        a.audio[i] = yv;
        mx = fmaxf(mx, fabsf(yv));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) mx = fmaxf(mx, scratch[i]);
        atomicMax(peak, __float_as_uint(mx));
    }
}

bool pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// the batched kernels hold 2 x SF frames of n_fft/2 complex points plus the twiddle table: 72 KB at n_fft = 2048
constexpr int BATCHED_MAX_NFFT = 2048;
size_t batched_lds(int n_fft) { return (size_t)(2 * SF * (n_fft / 2) + 3 * tw_len(n_fft / 2)) * sizeof(float2); }
int wave_grid(int total, int per_cu = 2) { const int g = 8 * ((total + 7) / 8), cap = (per_cu * pg_cu_count()) / 8 * 8; return g < cap ? g : (cap < 8 ? 8 : cap); }   // 78.3 KB of LDS (n_fft 2048): 2 per CU; 41.5 KB (1024): 3
int batched_grid(int total) { const int g = 8 * ((total + 7) / 8), cap = (2 * pg_cu_count()) / 8 * 8; return g < cap ? g : (cap < 8 ? 8 : cap); }
// the attribute belongs to (function, CURRENT device): set on every call (a host-side table write), so a process that drives
// several devices is served too and nothing is cached between calls
hipError_t batched_lds_ready() {
    const int lds = (int)batched_lds(BATCHED_MAX_NFFT);
    hipError_t e = hipFuncSetAttribute((const void*)stft_frames_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stft_frames_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)istft_frames4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int w16 = (int)wave_lds<16>();                     // (the 512-point kernels' 41.5 KB are below the default limit)
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stft_w_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, w16);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stft_w_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, w16);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)istft_frames_w_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, w16);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)istft_ola_w_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, w16);

    return e;
}

}  // namespace

extern "C" int pg_stft(const pg_stft_args* a, void* stream) {
    if (!a || !a->y || !a->out) return pg_fail(PG_ERR_NULL, "stft: y, out required");
    if (!pow2(a->n_fft) || a->n_fft < 32 || a->n_fft > 4096) return pg_fail(PG_ERR_UNSUPPORTED, "stft: n_fft must be a power of two in [32, 4096]");
    if (a->n_signals <= 0 || a->hop <= 0 || a->n_samples <= a->n_fft / 2) return pg_fail(PG_ERR_SHAPE, "stft: bad sizes (reflect padding needs n_samples > n_fft/2)");
    if (a->n_frames != 1 + a->n_samples / a->hop) return pg_fail(PG_ERR_SHAPE, "stft: n_frames must equal 1 + n_samples / hop");
    if (a->chunk_start && (a->src_len <= 0 || a->src_stride < a->src_len)) return pg_fail(PG_ERR_SHAPE, "stft: chunked source needs 0 < src_len <= src_stride");
    if (!a->chunk_start && a->chunk_row) return pg_fail(PG_ERR_NULL, "stft: chunk_row without chunk_start");
    hipError_t e = batched_lds_ready();
    if (e != hipSuccess) return pg_fail((int)e, hipGetErrorString(e));
    if (a->n_fft <= BATCHED_MAX_NFFT && !a->single_frame) {
        const int total = a->n_signals * ((a->n_frames + SF - 1) / SF);
        if (a->n_fft == 2048 || a->n_fft == 1024) {        // one wave per frame (wave_fft), NW frames per workgroup
            const int totw = a->n_signals * ((a->n_frames + NW - 1) / NW);
            hipStream_t st = (hipStream_t)stream;
            if (a->n_fft == 2048) {
                if (a->chunk_start) hipLaunchKernelGGL((stft_w_kernel<true, 16>), dim3((unsigned)wave_grid(totw)), dim3(WT), wave_lds<16>(), st, *a);
                else hipLaunchKernelGGL((stft_w_kernel<false, 16>), dim3((unsigned)wave_grid(totw)), dim3(WT), wave_lds<16>(), st, *a);
            } else {
                if (a->chunk_start) hipLaunchKernelGGL((stft_w_kernel<true, 8>), dim3((unsigned)wave_grid(totw, 3)), dim3(WT), wave_lds<8>(), st, *a);
                else hipLaunchKernelGGL((stft_w_kernel<false, 8>), dim3((unsigned)wave_grid(totw, 3)), dim3(WT), wave_lds<8>(), st, *a);
            }
        }
        else if (a->chunk_start) hipLaunchKernelGGL(stft_frames_kernel<true>, dim3((unsigned)batched_grid(total)), dim3(BT), batched_lds(a->n_fft), (hipStream_t)stream, *a);
        else hipLaunchKernelGGL(stft_frames_kernel<false>, dim3((unsigned)batched_grid(total)), dim3(BT), batched_lds(a->n_fft), (hipStream_t)stream, *a);
    } else {
        const size_t lds = (size_t)(2 * a->n_fft + a->n_fft / 2) * sizeof(float2);
        hipLaunchKernelGGL(stft_kernel, dim3((unsigned)(a->n_signals * a->n_frames)), dim3(FFT_THREADS), lds, (hipStream_t)stream, *a);
    }
    e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
}

extern "C" int pg_stft_frame_index(int32_t n_samples, int32_t n_fft, int32_t hop, int32_t n_frames, int32_t* idx, void* stream) {
    if (!idx) return pg_fail(PG_ERR_NULL, "stft_frame_index: idx required");
    if (n_samples <= 0 || n_fft <= 0 || hop <= 0 || n_frames <= 0) return pg_fail(PG_ERR_SHAPE, "stft_frame_index: bad sizes");
    long blocks = ((long)n_frames * n_fft + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(frame_index_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_samples, n_fft, hop, n_frames, idx);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
}

static int ola_blocks(const pg_istft_args* a) { return (a->hop * (a->n_frames - 1) + 256 * OLA_SPT - 1) / (256 * OLA_SPT); }
static int64_t ola_partial_bytes(const pg_istft_args* a) {      // per-workgroup peaks: the overlap-add kernel's blocks, or (n_fft = 2048 at
    const int64_t per = ola_blocks(a) > 2 * ((a->n_frames + 7) / 8) ? ola_blocks(a) : 2 * ((a->n_frames + 7) / 8);   // hop 512) 2 x groups
    const int64_t b = (int64_t)a->n_signals * per * (int64_t)sizeof(float);
    return (b + 255) / 256 * 256;
}

extern "C" int64_t pg_workspace_bytes_istft(const pg_istft_args* a) {
    if (!a) return 0;
    // [256 B reserved][per-workgroup peaks of the overlap-add, padded to 256 B][frames]
    return 256 + ola_partial_bytes(a) + (int64_t)a->n_signals * a->n_frames * 2 * a->bins * (int64_t)sizeof(float);
}

extern "C" int pg_istft(const pg_istft_args* a, void* stream) {
    if (!a || !a->a || !a->b || !a->audio || !a->workspace) return pg_fail(PG_ERR_NULL, "istft: a, b, audio, workspace required");
    const int N = 2 * a->bins;
    if (!pow2(N) || N < 32 || N > 4096) return pg_fail(PG_ERR_UNSUPPORTED, "istft: 2*bins must be a power of two in [32, 4096]");
    if (a->n_signals <= 0 || a->n_signals > 64 || a->n_frames < 2 || a->hop <= 0 || a->hop > N) return pg_fail(PG_ERR_SHAPE, "istft: bad sizes (1..64 signals per call)");
    if (a->workspace_bytes < pg_workspace_bytes_istft(a)) return pg_fail(PG_ERR_WORKSPACE, "istft: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)((char*)a->workspace + 256);
    float* frames = (float*)((char*)a->workspace + 256 + ola_partial_bytes(a));
    hipError_t e;
    if ((e = batched_lds_ready()) != hipSuccess) return pg_fail((int)e, hipGetErrorString(e));
    const int len = a->hop * (a->n_frames - 1);
    int bx = (len / 4 + 255) / 256; if (bx > 256) bx = 256; if (bx < 1) bx = 1;
    // n_fft = 2048 at hop 512 (the reference's defaults, preproc_mdb.py:202-204) or 1024 at 256, 16 B-aligned audio: overlap-add inside the transform
    // kernel, seams fixed by a second one; the workspace holds [256 B][2 x groups peaks per signal][tails]
    if ((N == 2048 || N == 1024) && a->hop * OW_COVER == N && !a->single_frame && (((uintptr_t)a->audio) & 15) == 0) {
        const int groups = (a->n_frames + NW - 1) / NW;
        float* tails = (float*)((char*)a->workspace + 256 + ola_partial_bytes(a));
        if (N == 2048) hipLaunchKernelGGL(istft_ola_w_kernel<16>, dim3((unsigned)wave_grid(a->n_signals * groups)), dim3(WT), wave_lds<16>(), st, *a, tails, partial);
        else hipLaunchKernelGGL(istft_ola_w_kernel<8>, dim3((unsigned)wave_grid(a->n_signals * groups, 3)), dim3(WT), wave_lds<8>(), st, *a, tails, partial);
        if (groups > 1) hipLaunchKernelGGL(istft_seam_kernel, dim3(groups - 1, a->n_signals), dim3(256), 0, st, *a, (const float*)tails, partial);
        if (a->normalize) hipLaunchKernelGGL(istft_peak_normalize_kernel, dim3(bx, a->n_signals), dim3(256), 0, st, a->audio, len, (const float*)partial, 2 * groups);
        e = hipGetLastError();
        return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
    }
    if (N <= BATCHED_MAX_NFFT && !a->single_frame) {
        const int total = a->n_signals * ((a->n_frames + SF - 1) / SF);
        if (N == 2048) hipLaunchKernelGGL(istft_frames_w_kernel<16>, dim3((unsigned)wave_grid(a->n_signals * ((a->n_frames + NW - 1) / NW))), dim3(WT), wave_lds<16>(), st, *a, frames);
        else if (N == 1024) hipLaunchKernelGGL(istft_frames_w_kernel<8>, dim3((unsigned)wave_grid(a->n_signals * ((a->n_frames + NW - 1) / NW), 3)), dim3(WT), wave_lds<8>(), st, *a, frames);
        else hipLaunchKernelGGL(istft_frames4_kernel, dim3((unsigned)batched_grid(total)), dim3(BT), batched_lds(N), st, *a, frames);
    } else {
        const size_t lds = (size_t)(2 * N + N / 2) * sizeof(float2);
        hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)(a->n_signals * a->n_frames)), dim3(FFT_THREADS), lds, st, *a, frames);
    }
    const int nblk = ola_blocks(a);
    hipLaunchKernelGGL(istft_ola4_kernel, dim3(nblk, a->n_signals), dim3(256), 0, st, *a, (const float*)frames, partial);
    // the peak over the overlap-add's per-workgroup peaks and the division by it are ONE launch (round 3: two)
    if (a->normalize) hipLaunchKernelGGL(istft_peak_normalize_kernel, dim3(bx, a->n_signals), dim3(256), 0, st, a->audio, len, (const float*)partial, nblk);
    e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
}

extern "C" int pg_gl_project(const pg_gl_args* a, void* stream) {
    if (!a || !a->S || !a->mag || !a->x) return pg_fail(PG_ERR_NULL, "gl_project: S, mag, x required");
    if (a->bins < 3 || a->frames <= 0 || a->n < 0 || a->n > 65535) return pg_fail(PG_ERR_SHAPE, "gl_project: bad sizes");
    const unsigned n = a->n ? (unsigned)a->n : 1u;          // n = 0: one clip (the v0.2 layout of the struct)
    long blocks = ((long)a->bins * a->frames + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gl_project_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, *a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
}

extern "C" int pg_ola_nt(const pg_ola_args* a, void* stream) {
    if (!a || !a->fr || !a->audio || !a->workspace) return pg_fail(PG_ERR_NULL, "ola_nt: fr, audio, workspace required");
    if (a->n_fft < 4 || (a->n_fft & 1) || a->frames < 2 || a->hop <= 0 || a->hop > a->n_fft) return pg_fail(PG_ERR_SHAPE, "ola_nt: bad sizes");
    if (a->n < 0 || a->n > 64) return pg_fail(PG_ERR_SHAPE, "ola_nt: 1..64 clips per call");
    if (a->workspace_bytes < 256) return pg_fail(PG_ERR_WORKSPACE, "ola_nt: workspace too small");
    const unsigned n = a->n ? (unsigned)a->n : 1u;          // n = 0: one clip (the v0.2 layout of the struct)
    hipStream_t st = (hipStream_t)stream;
    unsigned* peak = (unsigned*)a->workspace;
    hipError_t e = hipMemsetAsync(peak, 0, 256, st);
    if (e != hipSuccess) return pg_fail((int)e, hipGetErrorString(e));
    const int len = a->hop * (a->frames - 1);
    int bx = (len + 255) / 256; if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(ola_nt_kernel, dim3(bx, n), dim3(256), 0, st, *a, peak);
    if (a->normalize) hipLaunchKernelGGL(istft_peak_normalize_kernel, dim3(bx > 256 ? 256 : bx, n), dim3(256), 0, st, a->audio, len, (const float*)peak, 1);
    e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
}
