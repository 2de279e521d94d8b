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

namespace {

constexpr int FFT_THREADS = 256;

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

__global__ __launch_bounds__(FFT_THREADS) void stft_kernel(const pg_stft_args a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int N = a.n_fft, bins = N >> 1;
    float2* buf0 = smem; float2* buf1 = smem + N; float2* tw = smem + 2 * N;
    const int t = blockIdx.x % a.n_frames, sig = blockIdx.x / a.n_frames;
    const float* y = a.y + (long)sig * a.n_samples;
    fill_twiddles(tw, N);
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const int idx = reflect_index(t * a.hop + k - (N >> 1), a.n_samples);
        buf0[k] = make_float2(y[idx] * hann(k, N), 0.f);
    }
    __syncthreads();
    const float2* X = fft_lds(buf0, buf1, tw, N, 1.f);
    float* o_re = a.out + ((long)sig * 2 * bins) * a.n_frames + t;
    float* o_im = o_re + (long)bins * a.n_frames;
    for (int k = 1 + threadIdx.x; k <= bins; k += blockDim.x) {      // bin 0 (DC) dropped, preproc_mdb.py:93
        float2 v = X[k];
        if (a.polar) {
            pg_complex_from_parts(v.x, v.y);
            o_re[(long)(k - 1) * a.n_frames] = log1pf(hypotf(v.x, v.y));
            o_im[(long)(k - 1) * a.n_frames] = atan2f(v.y, v.x);
        } else {
            o_re[(long)(k - 1) * a.n_frames] = v.x;
            o_im[(long)(k - 1) * a.n_frames] = v.y;
        }
    }
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
            const float mag = expf(va) - 1.0f;
            float s, c;
            sincosf(vb, &s, &c);
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

// overlap-add as a gather + / window-sum-square + trim + peak |y|
__global__ __launch_bounds__(256) void istft_ola_kernel(const pg_istft_args a, const float* frames, unsigned* peak) {
    __shared__ float scratch[16];
    const int N = 2 * a.bins, len = a.hop * (a.n_frames - 1);
    const int sig = blockIdx.y;
    float mx = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
        const int ip = i + (N >> 1);
        int t_hi = ip / a.hop; if (t_hi > a.n_frames - 1) t_hi = a.n_frames - 1;
        int t_lo = (ip - N + a.hop) / a.hop; if (ip - N + 1 <= 0) t_lo = 0;          // ceil((ip-N+1)/hop), clamped
        float s = 0.f, wss = 0.f;
        for (int t = t_lo; t <= t_hi; ++t) {
            const int n = ip - t * a.hop;
            const float w = hann(n, N);
            s += frames[((long)sig * a.n_frames + t) * N + n];
            wss += w * w;
        }
        const float yv = wss > 1.17549435e-38f ? s / wss : s;
        a.audio[(long)sig * len + i] = yv;
        mx = fmaxf(mx, fabsf(yv));
    }
    // block max, then one atomicMax on the float bits (non-negative floats order like unsigned ints)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) mx = fmaxf(mx, scratch[i]);
        atomicMax(peak + sig, __float_as_uint(mx));
    }
}

__global__ __launch_bounds__(256) void istft_normalize_kernel(float* audio, int len, const unsigned* peak) {
    const int sig = blockIdx.y;
    const float pk = __uint_as_float(peak[sig]);
    if (!(pk > 1.17549435e-38f)) return;                              // librosa.util.normalize: tiny norms -> leave as is
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x)
        audio[(long)sig * len + i] /= pk;
}

// Griffin-Lim projection onto the target magnitudes: keep the phase of S, impose mag (utils.py:122-124)
__global__ __launch_bounds__(256) void gl_project_kernel(const pg_gl_args a) {
    const long total = (long)a.bins * a.frames;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / a.frames), t = (int)(e - (long)r * a.frames);
        float re = a.S[e], im = a.S[total + e];
        pg_complex_from_parts(re, im);                      // np.angle on re + 1j*im
        const float mod = hypotf(re, im), m = a.mag[e];
        const float c = mod > 0.f ? re / mod : 1.f, s = mod > 0.f ? im / mod : 0.f;   // angle(0) = 0
        const float nr = m * c, ni = m * s;
        if (a.spec_out) { a.spec_out[e] = nr; a.spec_out[total + e] = ni; }
        a.x[e] = nr;
        if (r >= 1 && r <= a.bins - 2) a.x[(long)(a.bins + r - 1) * a.frames + t] = ni;
    }
}

// overlap-add of (n_fft, frames)-major windowed frames, any even n_fft
__global__ __launch_bounds__(256) void ola_nt_kernel(const pg_ola_args a, unsigned* peak) {
    __shared__ float scratch[16];
    const int N = a.n_fft, len = a.hop * (a.frames - 1);
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

}  // namespace

extern "C" int pg_stft(const pg_stft_args* a, void* stream) {
    if (!a || !a->y || !a->out) return pg_fail(PG_ERR_NULL, "stft: y, out required");
    if (!pow2(a->n_fft) || a->n_fft < 32 || a->n_fft > 4096) return pg_fail(PG_ERR_UNSUPPORTED, "stft: n_fft must be a power of two in [32, 4096]");
    if (a->n_signals <= 0 || a->hop <= 0 || a->n_samples <= a->n_fft / 2) return pg_fail(PG_ERR_SHAPE, "stft: bad sizes (reflect padding needs n_samples > n_fft/2)");
    if (a->n_frames != 1 + a->n_samples / a->hop) return pg_fail(PG_ERR_SHAPE, "stft: n_frames must equal 1 + n_samples / hop");
    const size_t lds = (size_t)(2 * a->n_fft + a->n_fft / 2) * sizeof(float2);
    hipLaunchKernelGGL(stft_kernel, dim3((unsigned)(a->n_signals * a->n_frames)), dim3(FFT_THREADS), lds, (hipStream_t)stream, *a);
    hipError_t e = hipGetLastError();
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

extern "C" int64_t pg_workspace_bytes_istft(const pg_istft_args* a) {
    if (!a) return 0;
    // [peak words, padded to 256 B][frames]
    return 256 + (int64_t)a->n_signals * a->n_frames * 2 * a->bins * (int64_t)sizeof(float);
}

extern "C" int pg_istft(const pg_istft_args* a, void* stream) {
    if (!a || !a->a || !a->b || !a->audio || !a->workspace) return pg_fail(PG_ERR_NULL, "istft: a, b, audio, workspace required");
    const int N = 2 * a->bins;
    if (!pow2(N) || N < 32 || N > 4096) return pg_fail(PG_ERR_UNSUPPORTED, "istft: 2*bins must be a power of two in [32, 4096]");
    if (a->n_signals <= 0 || a->n_signals > 64 || a->n_frames < 2 || a->hop <= 0 || a->hop > N) return pg_fail(PG_ERR_SHAPE, "istft: bad sizes (1..64 signals per call)");
    if (a->workspace_bytes < pg_workspace_bytes_istft(a)) return pg_fail(PG_ERR_WORKSPACE, "istft: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned* peak = (unsigned*)a->workspace;
    float* frames = (float*)((char*)a->workspace + 256);
    hipError_t e = hipMemsetAsync(peak, 0, 256, st);
    if (e != hipSuccess) return pg_fail((int)e, hipGetErrorString(e));
    const size_t lds = (size_t)(2 * N + N / 2) * sizeof(float2);
    hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)(a->n_signals * a->n_frames)), dim3(FFT_THREADS), lds, st, *a, frames);
    const int len = a->hop * (a->n_frames - 1);
    int bx = (len + 255) / 256; if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(istft_ola_kernel, dim3(bx, a->n_signals), dim3(256), 0, st, *a, (const float*)frames, peak);
    if (a->normalize) hipLaunchKernelGGL(istft_normalize_kernel, dim3(bx, a->n_signals), dim3(256), 0, st, a->audio, len, (const unsigned*)peak);
    e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
}

extern "C" int pg_gl_project(const pg_gl_args* a, void* stream) {
    if (!a || !a->S || !a->mag || !a->x) return pg_fail(PG_ERR_NULL, "gl_project: S, mag, x required");
    if (a->bins < 3 || a->frames <= 0) return pg_fail(PG_ERR_SHAPE, "gl_project: bad sizes");
    long blocks = ((long)a->bins * a->frames + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gl_project_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
}

extern "C" int pg_ola_nt(const pg_ola_args* a, void* stream) {
    if (!a || !a->fr || !a->audio || !a->workspace) return pg_fail(PG_ERR_NULL, "ola_nt: fr, audio, workspace required");
    if (a->n_fft < 4 || (a->n_fft & 1) || a->frames < 2 || a->hop <= 0 || a->hop > a->n_fft) return pg_fail(PG_ERR_SHAPE, "ola_nt: bad sizes");
    if (a->workspace_bytes < 256) return pg_fail(PG_ERR_WORKSPACE, "ola_nt: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned* peak = (unsigned*)a->workspace;
    hipError_t e = hipMemsetAsync(peak, 0, 256, st);
    if (e != hipSuccess) return pg_fail((int)e, hipGetErrorString(e));
    const int len = a->hop * (a->frames - 1);
    int bx = (len + 255) / 256; if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(ola_nt_kernel, dim3(bx), dim3(256), 0, st, *a, peak);
    if (a->normalize) hipLaunchKernelGGL(istft_normalize_kernel, dim3(bx, 1), dim3(256), 0, st, a->audio, len, (const unsigned*)peak);
    e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
}
