// pointwise.hip -- the HBM-bound side of the path: train-mode BatchNorm (forward / backward), the fused
// cos/sin/magnitude loss with its gradient, Adam over the flat parameter arena, and the polar transform.
// All are streaming kernels: coalesced frame-contiguous reads, wavefront shuffle reductions, no atomics
// (every reduction has a fixed order => bit-reproducible run to run).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "phasegen.h"
#include "pg_common.h"
#include "pg_fastmath.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// BatchNorm, one workgroup per channel.  Tensor element (b, c, l) = base[b*bs + c*L + l].
// model.py:81,83 (nn.BatchNorm on (B, C, L)): mean / biased var over (B, L), eps inside the sqrt,
// running_var gets the unbiased variance, momentum 0.1.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bn_slope(int act) { return act == PG_ACT_LEAKY02 ? 0.2f : (act == PG_ACT_RELU ? 0.0f : 1.0f); }
__device__ __forceinline__ unsigned short to_bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
// all outputs of a BatchNorm forward for element (b, c, l): fp32 y / y2 and the bf16 copies of the bf16-resident path
__device__ __forceinline__ void bn_store(const pg_bn_args& a, int b, int c, int l, float o) {
    if (a.y) { const float s1 = bn_slope(a.y_act); a.y[(long)b * a.y_bs + (long)c * a.L + l] = fmaxf(o, s1 * o); }   // slope 1 = identity, 0.2 = LeakyReLU, 0 = ReLU
    if (a.y2) { const float s2 = bn_slope(a.y2_act); a.y2[(long)b * a.y2_bs + (long)c * a.L + l] = fmaxf(o, s2 * o); }
    if (a.yh) { const float s3 = bn_slope(a.yh_act); a.yh[(long)b * a.yh_bs + (long)c * a.yh_pitch + l] = to_bf16_bits(fmaxf(o, s3 * o)); }
    if (a.yh2) { const float s4 = bn_slope(a.yh2_act); a.yh2[(long)b * a.yh2_bs + (long)c * a.yh2_pitch + l] = to_bf16_bits(fmaxf(o, s4 * o)); }
}

__global__ __launch_bounds__(256) void bn_fwd_kernel(const pg_bn_args a) {
    __shared__ float scratch[16];
    const int c = blockIdx.x, n = a.B * a.L;
    const float* xc = a.x + (long)c * a.L;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int b = e / a.L, l = e - b * a.L;
        s += xc[(long)b * a.x_bs + l];
    }
    const float mean = pg_block_sum(s, scratch) / (float)n;
    float q = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int b = e / a.L, l = e - b * a.L;
        const float d = xc[(long)b * a.x_bs + l] - mean;
        q += d * d;
    }
    const float var = pg_block_sum(q, scratch) / (float)n;
    const float invstd = 1.0f / sqrtf(var + a.eps);
    const float g = a.gamma[c], be = a.beta[c];
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int b = e / a.L, l = e - b * a.L;
        bn_store(a, b, c, l, (xc[(long)b * a.x_bs + l] - mean) * invstd * g + be);
    }
    if (threadIdx.x == 0) {
        a.save_mean[c] = mean;
        a.save_invstd[c] = invstd;
        if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
        if (a.running_var) {
            const float unbiased = var * ((float)n / (float)(n > 1 ? n - 1 : 1));
            a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * unbiased;
        }
        if (c == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;       // nn.BatchNorm's counter: no separate launch for it
    }
}

// Register-resident variants (every shape of the U-Net: B * L <= 64 * 256): the channel's B x L values are read from HBM ONCE
// into registers and mean, variance and the normalised / activated outputs are all computed from there: 1 read + 1-2 writes
// instead of 3 reads.  The channel is walked FLAT: unit e = tid + 256 i of the B * L / VEC units (VEC = 4: float4 units when
// frames and strides allow 16-byte accesses -- one wave instruction moves 1 KB --, else single floats), (sample, position)
// advanced incrementally (no division in the loop), so every lane works whatever L is (129 frames used to idle 127 of 256
// lanes of a power-of-two row map).  Same two-pass arithmetic (mean first, then the centred squares), block sums in a fixed
// order: bit-reproducible.
typedef float bnf4 __attribute__((ext_vector_type(4)));
template <int VEC> struct BnVec;
template <> struct BnVec<1> { typedef float T; };
template <> struct BnVec<4> { typedef bnf4 T; };
typedef float bnf2 __attribute__((ext_vector_type(2)));
template <> struct BnVec<2> { typedef bnf2 T; };
template <int VEC> __device__ __forceinline__ float bn_lane(const typename BnVec<VEC>::T& v, int k);
template <> __device__ __forceinline__ float bn_lane<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float bn_lane<4>(const bnf4& v, int k) { return v[k]; }
template <> __device__ __forceinline__ float bn_lane<2>(const bnf2& v, int k) { return v[k]; }

struct BnWalk { int b, u, db, du, Lu; };
__device__ __forceinline__ BnWalk bn_walk(int L, int vec) {
    BnWalk w; w.Lu = L / vec;
    w.b = threadIdx.x / w.Lu; w.u = threadIdx.x - w.b * w.Lu;
    w.db = 256 / w.Lu; w.du = 256 - w.db * w.Lu;
    return w;
}
__device__ __forceinline__ void bn_next(BnWalk& w) {
    w.u += w.du; w.b += w.db;
    if (w.u >= w.Lu) { w.u -= w.Lu; w.b += 1; }
}

// all outputs of a BatchNorm forward for VEC consecutive elements starting at (b, c, l)
template <int VEC>
__device__ __forceinline__ void bn_store_v(const pg_bn_args& a, int b, int c, int l, const float* o) {
    if (VEC != 4) { for (int k = 0; k < VEC; ++k) bn_store(a, b, c, l + k, o[k]); return; }
    if (a.y) { const float s1 = bn_slope(a.y_act); bnf4 t; for (int k = 0; k < 4; ++k) t[k] = fmaxf(o[k], s1 * o[k]); *(bnf4*)(a.y + (long)b * a.y_bs + (long)c * a.L + l) = t; }
    if (a.y2) { const float s2 = bn_slope(a.y2_act); bnf4 t; for (int k = 0; k < 4; ++k) t[k] = fmaxf(o[k], s2 * o[k]); *(bnf4*)(a.y2 + (long)b * a.y2_bs + (long)c * a.L + l) = t; }
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    if (a.yh) { const float s3 = bn_slope(a.yh_act); us4 t; for (int k = 0; k < 4; ++k) t[k] = to_bf16_bits(fmaxf(o[k], s3 * o[k])); *(us4*)(a.yh + (long)b * a.yh_bs + (long)c * a.yh_pitch + l) = t; }
    if (a.yh2) { const float s4 = bn_slope(a.yh2_act); us4 t; for (int k = 0; k < 4; ++k) t[k] = to_bf16_bits(fmaxf(o[k], s4 * o[k])); *(us4*)(a.yh2 + (long)b * a.yh2_bs + (long)c * a.yh2_pitch + l) = t; }
}

// The walk runs ONCE: unit i's (sample, position) is kept packed in one register (pk = b << 16 | u, -1 = past the end) and the
// later passes decode it -- re-walking made the compiler keep every intermediate of three identical walks alive.
template <int UPT, int VEC>          // UPT units of VEC floats per thread
__global__ __launch_bounds__(256) void bn_fwd_reg_kernel(const pg_bn_args a) {
    typedef typename BnVec<VEC>::T V;
    __shared__ float scratch[16];
    const int c = blockIdx.x, n = a.B * a.L;
    const float* xc = a.x + (long)c * a.L;
    V v[UPT];
    int pk[UPT];
    float s = 0.f;
    BnWalk w = bn_walk(a.L, VEC);
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        const bool ok = w.b < a.B;
        pk[i] = ok ? (w.b << 16) | w.u : -1;
        v[i] = *(const V*)(xc + (ok ? (long)w.b * a.x_bs + VEC * w.u : 0L));      // branch-free: all loads issue back to back
        if (!ok) v[i] = V(0.f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) s += bn_lane<VEC>(v[i], k);
        bn_next(w);
    }
    const float mean = pg_block_sum(s, scratch) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < UPT; ++i)
        if (pk[i] >= 0) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) { const float d = bn_lane<VEC>(v[i], k) - mean; q += d * d; }
        }
    const float var = pg_block_sum(q, scratch) / (float)n;
    const float invstd = 1.0f / sqrtf(var + a.eps);
    const float ga = a.gamma[c], be = a.beta[c];
#pragma unroll
    for (int i = 0; i < UPT; ++i)
        if (pk[i] >= 0) {
            float o[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = (bn_lane<VEC>(v[i], k) - mean) * invstd * ga + be;
            bn_store_v<VEC>(a, pk[i] >> 16, c, VEC * (pk[i] & 0xffff), o);
        }
    if (threadIdx.x == 0) {
        a.save_mean[c] = mean;
        a.save_invstd[c] = invstd;
        if (a.running_mean) a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * mean;
        if (a.running_var) {
            const float unbiased = var * ((float)n / (float)(n > 1 ? n - 1 : 1));
            a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * unbiased;
        }
        if (c == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;       // nn.BatchNorm's counter: no separate launch for it
    }
}

template <int UPT, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reg_kernel(const pg_bn_args a) {
    typedef typename BnVec<VEC>::T V;
    __shared__ float scratch[16];
    const int c = blockIdx.x, n = a.B * a.L;
    const float* xc = a.x + (long)c * a.L;
    const float* dyc = a.dy + (long)c * a.L;
    const float mean = a.save_mean[c], invstd = a.save_invstd[c];
    V xh[UPT], dy[UPT];
    int pk[UPT];
    float s1 = 0.f, s2 = 0.f;
    BnWalk w = bn_walk(a.L, VEC);
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        const bool ok = w.b < a.B;
        pk[i] = ok ? (w.b << 16) | w.u : -1;
        dy[i] = *(const V*)(dyc + (ok ? (long)w.b * a.dy_bs + VEC * w.u : 0L));   // branch-free: all loads issue back to back
        const V xv = *(const V*)(xc + (ok ? (long)w.b * a.x_bs + VEC * w.u : 0L));
        xh[i] = (xv - mean) * invstd;
        if (!ok) { dy[i] = V(0.f); xh[i] = V(0.f); }
#pragma unroll
        for (int k = 0; k < VEC; ++k) { s1 += bn_lane<VEC>(dy[i], k); s2 += bn_lane<VEC>(dy[i], k) * bn_lane<VEC>(xh[i], k); }
        bn_next(w);
    }
    const float sum_dy = pg_block_sum(s1, scratch);
    const float sum_dy_xhat = pg_block_sum(s2, scratch);
    const float kk = a.gamma[c] * invstd, m1 = sum_dy / (float)n, m2 = sum_dy_xhat / (float)n;
    float* dxc = a.dx + (long)c * a.L;
#pragma unroll
    for (int i = 0; i < UPT; ++i)
        if (pk[i] >= 0) *(V*)(dxc + (long)(pk[i] >> 16) * a.dx_bs + VEC * (pk[i] & 0xffff)) = kk * (dy[i] - m1 - xh[i] * m2);
    if (threadIdx.x == 0) {
        a.dgamma[c] = sum_dy_xhat;
        a.dbeta[c] = sum_dy;
    }
}

// dx = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat));  dgamma = sum(dy * xhat);  dbeta = sum(dy)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const pg_bn_args a) {
    __shared__ float scratch[16];
    const int c = blockIdx.x, n = a.B * a.L;
    const float* xc = a.x + (long)c * a.L;
    const float* dyc = a.dy + (long)c * a.L;
    const float mean = a.save_mean[c], invstd = a.save_invstd[c];
    float s1 = 0.f, s2 = 0.f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int b = e / a.L, l = e - b * a.L;
        const float dy = dyc[(long)b * a.dy_bs + l];
        s1 += dy;
        s2 += dy * (xc[(long)b * a.x_bs + l] - mean) * invstd;
    }
    const float sum_dy = pg_block_sum(s1, scratch);
    const float sum_dy_xhat = pg_block_sum(s2, scratch);
    const float k = a.gamma[c] * invstd, m1 = sum_dy / (float)n, m2 = sum_dy_xhat / (float)n;
    float* dxc = a.dx + (long)c * a.L;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int b = e / a.L, l = e - b * a.L;
        const float xhat = (xc[(long)b * a.x_bs + l] - mean) * invstd;
        dxc[(long)b * a.dx_bs + l] = k * (dyc[(long)b * a.dy_bs + l] - m1 - xhat * m2);
    }
    if (threadIdx.x == 0) {
        a.dgamma[c] = sum_dy_xhat;
        a.dbeta[c] = sum_dy;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Loss (train.py:45-60) fused with d loss / d pred.  One thread per (b, c, l); stage 1 leaves 3 partial sums
// per workgroup in the workspace, stage 2 (one workgroup) adds them in double in a fixed order.
// ---------------------------------------------------------------------------------------------------------
constexpr int LOSS_BLOCKS = 1024;

__global__ __launch_bounds__(256) void loss_partial_kernel(const pg_loss_args a, float* partial) {
    __shared__ float scratch[16];
    const long CL = (long)a.C * a.L, N = (long)a.B * CL;
    const float scale = 2.0f / (float)N;
    float s_cos = 0.f, s_sin = 0.f, s_mag = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < N; e += (long)gridDim.x * blockDim.x) {
        const long b = e / CL, r = e - b * CL;
        const float ph = a.pred[b * 2 * CL + r], mh = a.pred[b * 2 * CL + CL + r];
        const float m = a.batch[b * 2 * CL + r], th = a.batch[b * 2 * CL + CL + r];
        float sp, cp, st, ct;
        sincosf(ph, &sp, &cp);
        sincosf(th, &st, &ct);
        const float dc = cp - ct, ds = sp - st, dm = mh - m;
        s_cos += dc * dc; s_sin += ds * ds; s_mag += dm * dm;
        if (a.dpred) {
            // autograd of MSE(cos p, cos th) + MSE(sin p, sin th):  2/N * (dc * -sin p + ds * cos p)
            a.dpred[b * 2 * CL + r] = scale * (ds * cp - dc * sp);
            a.dpred[b * 2 * CL + CL + r] = a.mag_weight * scale * dm;
        }
    }
    s_cos = pg_block_sum(s_cos, scratch);
    s_sin = pg_block_sum(s_sin, scratch);
    s_mag = pg_block_sum(s_mag, scratch);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 3 + 0] = s_cos; partial[blockIdx.x * 3 + 1] = s_sin; partial[blockIdx.x * 3 + 2] = s_mag;
    }
}

__global__ __launch_bounds__(64) void loss_final_kernel(const float* partial, int nblocks, double n, float mag_weight, float* losses) {
    // lane i adds partials i, i + 64, ... in double, then a fixed shuffle tree: deterministic, 64x shorter than one lane alone
    double c = 0, s = 0, m = 0;
    for (int i = threadIdx.x; i < nblocks; i += 64) { c += partial[i * 3]; s += partial[i * 3 + 1]; m += partial[i * 3 + 2]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off, 64); s += __shfl_xor(s, off, 64); m += __shfl_xor(m, off, 64); }
    if (threadIdx.x == 0) {
        const float cos_l = (float)(c / n), sin_l = (float)(s / n), mag_l = (float)(m / n);
        const float ang = cos_l + sin_l;
        losses[0] = ang + mag_l * mag_weight; losses[1] = ang; losses[2] = mag_l;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Whole-array standardisation of the feature set (preproc_mdb.py:182): moments in double (fixed reduction order:
// per-thread grid-stride sums, wave shuffle tree, per-workgroup partials summed by one wave), then one in-place pass.
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MOM_BLOCKS = 1024;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// pass 0: sum of x; pass 1: sum of (x - mean)^2 with the mean read from stats[0] (two-pass, like numpy's std)
__global__ __launch_bounds__(256) void moments_partial_kernel(const float* __restrict__ x, long n, const double* stats, int pass, double* partial) {
    __shared__ double red[4];
    const double mean = pass ? stats[0] : 0.0;
    double acc = 0;
    const long stride = (long)gridDim.x * 256;
    const bool vec = (((uintptr_t)x) & 15) == 0;
    const long n4 = vec ? (n >> 2) : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = ((const f32x4*)x)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double d = (double)v[j] - mean; acc += pass ? d * d : d; }
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double d = (double)x[i] - mean; acc += pass ? d * d : d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void moments_final_kernel(const double* partial, int nblocks, double n, int pass, double* stats) {
    double acc = 0;
    for (int i = threadIdx.x; i < nblocks; i += 64) acc += partial[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) stats[pass] = pass ? sqrt(acc / n) : acc / n;
}

__global__ __launch_bounds__(256) void standardize_kernel(float* __restrict__ x, long n, const double* stats) {
    const float mean = (float)stats[0], std = (float)stats[1];             // float32 arithmetic, as numpy on a float32 array
    const long stride = (long)gridDim.x * 256;
    const bool vec = (((uintptr_t)x) & 15) == 0;
    const long n4 = vec ? (n >> 2) : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f32x4 v = ((f32x4*)x)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (v[j] - mean) / std;
        ((f32x4*)x)[i] = v;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] = (x[i] - mean) / std;
}

// ---------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults, single-tensor path of torch 2.x): m.lerp_(g, 1-b1); v = b2 v + (1-b2) g g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).  28 B of HBM traffic per parameter.
// ---------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float omb1, float b2, float omb2,
                                                   float step_size, float bc2_sqrt, float eps, float gs) {
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = pp[k], b = mm[k], c = vv[k];
            pg_adam_one(a, gg[k], b, c, omb1, b2, omb2, step_size, bc2_sqrt, eps, gs);
            pp[k] = a; mm[k] = b; vv[k] = c;
        }
        reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        pg_adam_one(p[i], g[i], m[i], v[i], omb1, b2, omb2, step_size, bc2_sqrt, eps, gs);
    }
}

// "Thin" variant for running BESIDE the MFMA-bound convolution kernels of backward (pg_adam_args.thin; phasegen.trainer):
// one 256-thread workgroup per CU at most, <= 32 VGPRs so that its four waves fit into the registers the conv kernels
// leave free (2 waves x <= 240 of 512 per SIMD), non-temporal loads and stores so that the 28 B per parameter it streams do
// not evict the conv kernels' operands from L2 / Infinity Cache.  Slower than adam_kernel on an idle chip (fewer bytes in
// flight), which does not matter in the shadow of ~100 ms of convolutions.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256)
void adam_thin_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                      float omb1, float b2, float omb2, float step_size, float bc2_sqrt, float eps, float gs) {
    const long n2 = n >> 1;          // two parameters per thread and iteration: 23 VGPRs (four need 41)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        f32x2 pp = __builtin_nontemporal_load(reinterpret_cast<f32x2*>(p) + i), gg = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(g) + i);
        f32x2 mm = __builtin_nontemporal_load(reinterpret_cast<f32x2*>(m) + i), vv = __builtin_nontemporal_load(reinterpret_cast<f32x2*>(v) + i);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float a = pp[k], b = mm[k], c = vv[k];
            pg_adam_one(a, gg[k], b, c, omb1, b2, omb2, step_size, bc2_sqrt, eps, gs);
            pp[k] = a; mm[k] = b; vv[k] = c;
        }
        __builtin_nontemporal_store(pp, reinterpret_cast<f32x2*>(p) + i);
        __builtin_nontemporal_store(mm, reinterpret_cast<f32x2*>(m) + i);
        __builtin_nontemporal_store(vv, reinterpret_cast<f32x2*>(v) + i);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) pg_adam_one(p[n - 1], g[n - 1], m[n - 1], v[n - 1], omb1, b2, omb2, step_size, bc2_sqrt, eps, gs);
}

// ---------------------------------------------------------------------------------------------------------
// data.py:39-47: [re; im] -> [log1p(|z|); angle(z)], 16 B of traffic per bin-frame.
// ---------------------------------------------------------------------------------------------------------
// (the arithmetic is pg_polar_one, pg_fastmath.h: shared with the STFT kernel's fused epilogue, so the two stay bit-identical)
__device__ __forceinline__ void polar_one(float re, float im, int use_exp, float& mag, float& ang) { pg_polar_one(re, im, use_exp, mag, ang); }

// VEC = 4: 16-B loads/stores (inner % 4 == 0 and 16-B aligned planes); VEC = 1: any shape
template <int VEC>
__global__ __launch_bounds__(256) void polar_kernel(const float* __restrict__ in, float* __restrict__ out, long n_items, long inner, int use_exp) {
    const long per = inner / VEC, total = n_items * per;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long it = e / per, r = (e - it * per) * VEC;
        const float* pre = in + it * 2 * inner + r;
        float* pm = out + it * 2 * inner + r;
        if (VEC == 4) {
            const f32x4 re = *reinterpret_cast<const f32x4*>(pre), im = *reinterpret_cast<const f32x4*>(pre + inner);
            f32x4 mg, an;
#pragma unroll
            for (int k = 0; k < 4; ++k) { float a, b; polar_one(re[k], im[k], use_exp, a, b); mg[k] = a; an[k] = b; }
            *reinterpret_cast<f32x4*>(pm) = mg;
            *reinterpret_cast<f32x4*>(pm + inner) = an;
        } else {
            float a, b;
            polar_one(pre[0], pre[inner], use_exp, a, b);
            pm[0] = a; pm[inner] = b;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16-resident path helpers (conv_h.hip): the weight shadow and the fp32 -> bf16 row cast.
// ---------------------------------------------------------------------------------------------------------
// Conv1d weights (Cout, Cin, k) are already [row = o][K = (q, j)]: a plain cast.  ConvTranspose1d weights (Cin, Cout, k) become
// A[(o * s + phi)][q * KJ + jj'] = W[q][o][s * (KJ - 1 - jj') + phi]  (KJ = taps per phase, ceil(k / s) rounded up to a power
// of two with zero weights for the taps that do not exist -- k = 5, s = 2 is stored as 4 taps per phase; taps are stored so
// that window positions ASCEND with jj', the order conv_h_kernel's T form reads them in).
__global__ __launch_bounds__(256) void shadow_kernel(const float* __restrict__ w, unsigned short* __restrict__ wh, int Cin, int Cout, int k, int s, int transposed, int KJ) {
    const long n = transposed ? (long)Cin * Cout * s * KJ : (long)Cin * Cout * k;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        if (!transposed) { wh[e] = to_bf16_bits(w[e]); continue; }
        const long Ktot = (long)Cin * KJ;
        const long row = e / Ktot; const int kk = (int)(e - row * Ktot);
        const int o = (int)(row / s), phi = (int)(row - (long)o * s), q = kk / KJ, jj = kk - q * KJ;
        const int j = s * (KJ - 1 - jj) + phi;            // taps past k (KJ rounded up to a power of two; k = 5: 3 -> 4) are zero
        wh[e] = j < k ? to_bf16_bits(w[((long)q * Cout + o) * k + j]) : (unsigned short)0;
    }
}

// (B, C, L) fp32 rows -> (B, C, pitch) bf16 rows, activation applied first; the tail [L, pitch) of every row is zeroed.
__global__ __launch_bounds__(256) void cast_rows_kernel(const pg_cast_args a) {
    const long rows = (long)a.B * a.C, total = rows * a.pitch;
    const float slope = bn_slope(a.act);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / a.pitch; const int l = (int)(e - row * a.pitch);
        const long b = row / a.C, c = row - b * a.C;
        float v = 0.f;
        if (l < a.L) { v = a.x[b * a.x_bs + c * a.L + l]; v = fmaxf(v, slope * v); }
        a.y[b * a.y_bs + c * a.pitch + l] = to_bf16_bits(v);
    }
}

int launch_ok(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, what);
}

int bn_check(const pg_bn_args* a) {
    if (!a) return pg_fail(PG_ERR_NULL, "bn: null args");
    if (a->B <= 0 || a->C <= 0 || a->L <= 0) return pg_fail(PG_ERR_SHAPE, "bn: non-positive dimension");
    if ((long)a->B * a->L > 0x7fffffffL) return pg_fail(PG_ERR_SHAPE, "bn: B*L too large");
    return PG_OK;
}

// register-resident BN: the channel's B * L values fit 64 registers per thread; 16-byte units where every tensor allows them
bool bn_reg_plan(const pg_bn_args* a, bool bwd, int& vec, int& upt) {
    const long n = (long)a->B * a->L;
    if (n > 64L * 256) return false;
    auto ok16 = [](const void* p, long bs) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (bs & 3) == 0); };
    auto ok8 = [](const void* p, long bs, int pitch) { return p == nullptr || ((((uintptr_t)p) & 7) == 0 && (bs & 3) == 0 && (pitch & 3) == 0); };
    bool v4 = (a->L & 3) == 0 && ok16(a->x, a->x_bs);
    if (bwd) v4 = v4 && ok16(a->dy, a->dy_bs) && ok16(a->dx, a->dx_bs);
    else v4 = v4 && ok16(a->y, a->y_bs) && ok16(a->y2, a->y2_bs) && ok8(a->yh, a->yh_bs, a->yh_pitch) && ok8(a->yh2, a->yh2_bs, a->yh2_pitch);
    auto ok8f = [](const void* p, long bs) { return p == nullptr || ((((uintptr_t)p) & 7) == 0 && (bs & 1) == 0); };
    bool v2 = (a->L & 1) == 0 && ok8f(a->x, a->x_bs);
    if (bwd) v2 = v2 && ok8f(a->dy, a->dy_bs) && ok8f(a->dx, a->dx_bs);      // (forward stores of a float2 unit are scalar)
    vec = v4 ? 4 : (v2 ? 2 : 1);
    const long units = n / vec;
    upt = (int)((units + 255) / 256);
    return true;
}

template <bool BWD>
void bn_launch_reg(const pg_bn_args* a, int vec, int upt, hipStream_t st) {
#define PG_BN_LAUNCH(U, V) { if (BWD) hipLaunchKernelGGL((bn_bwd_reg_kernel<U, V>), dim3(a->C), dim3(256), 0, st, *a); \
                             else hipLaunchKernelGGL((bn_fwd_reg_kernel<U, V>), dim3(a->C), dim3(256), 0, st, *a); }
    // (units-per-thread values are the ones hipcc allocates sanely: <8, 4> and <16, 2> take 180-245 VGPRs and spill)
    // (the small ones are for single clips -- demo.py's batch of one: 64 values per channel -- where walking 16 or 32 empty units
    // per thread made a 2048-channel layer take 12-26 us)
    if (vec == 4) { if (upt <= 1) PG_BN_LAUNCH(1, 4) else if (upt <= 4) PG_BN_LAUNCH(4, 4) else PG_BN_LAUNCH(16, 4) }
    else if (vec == 2) { if (upt <= 2) PG_BN_LAUNCH(2, 2) else PG_BN_LAUNCH(32, 2) }
    else { if (upt <= 2) PG_BN_LAUNCH(2, 1) else if (upt <= 16) PG_BN_LAUNCH(16, 1) else if (upt <= 33) PG_BN_LAUNCH(33, 1) else PG_BN_LAUNCH(64, 1) }
#undef PG_BN_LAUNCH
}

}  // namespace

// taps per output phase of a transposed conv in the shadow layout: ceil(k / stride) rounded up to a power of two
int pg_shadow_taps(int k, int stride) { int kj = (k + stride - 1) / stride, p2 = 1; while (p2 < kj) p2 <<= 1; return p2; }
extern "C" int64_t pg_shadow_elems(int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t transposed) {
    return transposed ? (int64_t)Cin * Cout * stride * pg_shadow_taps(k, stride) : (int64_t)Cin * Cout * k;
}

extern "C" int pg_shadow_weights(const float* w, uint16_t* wh, int32_t Cin, int32_t Cout, int32_t k, int32_t stride, int32_t transposed, void* stream) {
    if (!w || !wh) return pg_fail(PG_ERR_NULL, "shadow_weights: w, wh required");
    if (Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0) return pg_fail(PG_ERR_SHAPE, "shadow_weights: non-positive dimension");
    const int KJ = pg_shadow_taps(k, stride);
    const long n = transposed ? (long)Cin * Cout * stride * KJ : (long)Cin * Cout * k;
    long blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(shadow_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, wh, Cin, Cout, k, stride, transposed, KJ);
    return launch_ok("shadow_weights launch failed");
}

extern "C" int pg_cast_rows_bf16(const pg_cast_args* a, void* stream) {
    if (!a || !a->x || !a->y) return pg_fail(PG_ERR_NULL, "cast_rows_bf16: x, y required");
    if (a->B <= 0 || a->C <= 0 || a->L <= 0 || a->pitch < a->L) return pg_fail(PG_ERR_SHAPE, "cast_rows_bf16: bad sizes");
    const long total = (long)a->B * a->C * a->pitch;
    long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    return launch_ok("cast_rows_bf16 launch failed");
}

extern "C" int pg_bn_fwd(const pg_bn_args* a, void* stream) {
    if (int e = bn_check(a)) return e;
    if (!a->x || (!a->y && !a->yh) || !a->gamma || !a->beta || !a->save_mean || !a->save_invstd)
        return pg_fail(PG_ERR_NULL, "bn_fwd: x, y (or yh), gamma, beta, save_mean, save_invstd required");
    if ((a->yh && a->yh_pitch < a->L) || (a->yh2 && a->yh2_pitch < a->L)) return pg_fail(PG_ERR_SHAPE, "bn_fwd: bf16 output pitch below L");
    int vec, upt;
    if (bn_reg_plan(a, false, vec, upt)) bn_launch_reg<false>(a, vec, upt, (hipStream_t)stream);
    else hipLaunchKernelGGL(bn_fwd_kernel, dim3(a->C), dim3(256), 0, (hipStream_t)stream, *a);
    return launch_ok("bn_fwd launch failed");
}

extern "C" int pg_bn_bwd(const pg_bn_args* a, void* stream) {
    if (int e = bn_check(a)) return e;
    if (!a->x || !a->dy || !a->dx || !a->gamma || !a->save_mean || !a->save_invstd || !a->dgamma || !a->dbeta)
        return pg_fail(PG_ERR_NULL, "bn_bwd: x, dy, dx, gamma, save_mean, save_invstd, dgamma, dbeta required");
    int vec, upt;
    if (bn_reg_plan(a, true, vec, upt)) bn_launch_reg<true>(a, vec, upt, (hipStream_t)stream);
    else hipLaunchKernelGGL(bn_bwd_kernel, dim3(a->C), dim3(256), 0, (hipStream_t)stream, *a);
    return launch_ok("bn_bwd launch failed");
}

extern "C" int64_t pg_workspace_bytes_loss(const pg_loss_args*) { return (int64_t)LOSS_BLOCKS * 3 * sizeof(float); }

extern "C" int pg_loss_fwd_bwd(const pg_loss_args* a, void* stream) {
    if (!a || !a->pred || !a->batch || !a->losses || !a->workspace) return pg_fail(PG_ERR_NULL, "loss: pred, batch, losses, workspace required");
    if (a->B <= 0 || a->C <= 0 || a->L <= 0) return pg_fail(PG_ERR_SHAPE, "loss: non-positive dimension");
    if (a->workspace_bytes < pg_workspace_bytes_loss(a)) return pg_fail(PG_ERR_WORKSPACE, "loss: workspace too small");
    const long N = (long)a->B * a->C * a->L;
    int blocks = (int)((N + 255) / 256); if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
    float* partial = (float*)a->workspace;
    hipLaunchKernelGGL(loss_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a, partial);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, blocks, (double)N, a->mag_weight, a->losses);
    return launch_ok("loss launch failed");
}

extern "C" int64_t pg_workspace_bytes_moments(void) { return (int64_t)MOM_BLOCKS * (int64_t)sizeof(double); }

extern "C" int pg_moments(const pg_moments_args* a, void* stream) {
    if (!a || !a->x || !a->stats || !a->workspace) return pg_fail(PG_ERR_NULL, "moments: x, stats, workspace required");
    if (a->n <= 0) return pg_fail(PG_ERR_SHAPE, "moments: non-positive size");
    if (a->workspace_bytes < pg_workspace_bytes_moments()) return pg_fail(PG_ERR_WORKSPACE, "moments: workspace too small");
    if (((uintptr_t)a->x & 3) || ((uintptr_t)a->stats & 7) || ((uintptr_t)a->workspace & 7)) return pg_fail(PG_ERR_ALIGN, "moments: misaligned pointer");
    long blocks = ((a->n >> 2) + 255) / 256; if (blocks > MOM_BLOCKS) blocks = MOM_BLOCKS; if (blocks < 1) blocks = 1;
    double* partial = (double*)a->workspace;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(moments_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a->x, (long)a->n, a->stats, pass, partial);
        hipLaunchKernelGGL(moments_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, (int)blocks, (double)a->n, pass, a->stats);
    }
    return launch_ok("moments launch failed");
}

extern "C" int pg_standardize(float* x, int64_t n, const double* stats, void* stream) {
    if (!x || !stats) return pg_fail(PG_ERR_NULL, "standardize: x, stats required");
    if (n <= 0) return pg_fail(PG_ERR_SHAPE, "standardize: non-positive size");
    if (((uintptr_t)x & 3) || ((uintptr_t)stats & 7)) return pg_fail(PG_ERR_ALIGN, "standardize: misaligned pointer");
    long blocks = ((n >> 2) + 255) / 256; if (blocks > 256 * 16) blocks = 256 * 16; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(standardize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, stats);
    return launch_ok("standardize launch failed");
}

// Python floats -> the fp32 scalars of the update, rounded exactly where torch rounds them (one definition for the streaming
// kernel and the wgrad epilogue)
PgAdamScalars pg_adam_scalars(const pg_adam_args* a) {
    const double bc1 = 1.0 - pow(a->beta1, a->step), bc2 = 1.0 - pow(a->beta2, a->step);
    return { (float)(1.0 - a->beta1), (float)a->beta2, (float)(1.0 - a->beta2), (float)(a->lr / bc1), (float)sqrt(bc2), (float)a->eps,
             (float)a->grad_scale };
}

extern "C" int pg_adam_step(const pg_adam_args* a, void* stream) {
    if (!a || !a->p || !a->g || !a->m || !a->v) return pg_fail(PG_ERR_NULL, "adam: p, g, m, v required");
    if (a->n <= 0) return PG_OK;
    if (a->step < 1) return pg_fail(PG_ERR_SHAPE, "adam: step is 1-based");
    if (((uintptr_t)a->p | (uintptr_t)a->g | (uintptr_t)a->m | (uintptr_t)a->v) & 15) return pg_fail(PG_ERR_ALIGN, "adam: pointers must be 16-byte aligned");
    const PgAdamScalars h = pg_adam_scalars(a);
    long blocks = ((a->n >> 2) + 255) / 256; if (blocks > 256 * 16) blocks = 256 * 16; if (blocks < 1) blocks = 1;
    if (a->thin) {
        const long cus = pg_cu_count();
        if (blocks > cus) blocks = cus;
        hipLaunchKernelGGL(adam_thin_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a->p, a->g, a->m, a->v, (long)a->n,
                           h.omb1, h.b2, h.omb2, h.step_size, h.bc2_sqrt, h.eps, h.gs);
        return launch_ok("adam launch failed");
    }
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a->p, a->g, a->m, a->v, (long)a->n,
                       h.omb1, h.b2, h.omb2, h.step_size, h.bc2_sqrt, h.eps, h.gs);
    return launch_ok("adam launch failed");
}

extern "C" int pg_polar(const pg_polar_args* a, void* stream) {
    if (!a || !a->in || !a->out) return pg_fail(PG_ERR_NULL, "polar: in, out required");
    if (a->n_items <= 0 || a->inner <= 0) return pg_fail(PG_ERR_SHAPE, "polar: non-positive size");
    const bool vec = (a->inner & 3) == 0 && (((uintptr_t)a->in | (uintptr_t)a->out) & 15) == 0;
    long blocks = (a->n_items * a->inner / (vec ? 4 : 1) + 255) / 256; if (blocks > 256 * 16) blocks = 256 * 16; if (blocks < 1) blocks = 1;
    if (vec) hipLaunchKernelGGL(polar_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a->in, a->out, (long)a->n_items, (long)a->inner, a->use_exp);
    else hipLaunchKernelGGL(polar_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a->in, a->out, (long)a->n_items, (long)a->inner, a->use_exp);
    return launch_ok("polar launch failed");
}
