// pg_common.h -- internal helpers shared by the libphasegen translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

// Records `msg` as the calling thread's last error and returns `code` (see pg_last_error_string()).
int pg_fail(int code, const char* msg);
// Number of CUs of the current device (immutable per device; cached).
int pg_cu_count();
// taps per output phase in a transposed conv's bf16 weight shadow (pointwise.hip)
int pg_shadow_taps(int k, int stride);

// Wave(64)-level and block-level sum reductions (wavefront shuffles, then 4..16 partials through LDS).
__device__ __forceinline__ float pg_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// All threads get the block-wide sum.  `scratch` = at least 16 floats of LDS; safe to reuse after return.
__device__ __forceinline__ float pg_block_sum(float v, float* scratch) {
    v = pg_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

// data.py:40 builds z = d[:,0] + d[:,1]*1j from two real arrays.  In IEEE arithmetic that is
//   imag = 0 + (im*1 + 0*0) = im + 0   (so -0.0 becomes +0.0: a point on the negative real axis with im = -0.0
//                                        gets angle +pi, not -pi)
//   real = re + (im*0 - 0*1)           (only the sign of a zero real part can change)
// Reproduced literally so np.angle parity holds on the branch cut.  (No fast-math: x + 0.0f is not folded.)
__device__ __forceinline__ void pg_complex_from_parts(float& re, float& im) {
    const float t = im * 0.0f - 0.0f;
    re = re + t;
    im = 0.0f + (im + 0.0f);
}

// Adam (torch.optim.Adam defaults, single-tensor path of torch 2.x): m.lerp_(g, 1-b1); v = b2 v + (1-b2) g g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).  ONE definition, used by the streaming kernel (pointwise.hip) and by the
// wgrad epilogue (conv_common.h), so that the fused update is bit-identical to the separate one.
__device__ __forceinline__ void pg_adam_one(float& p, float g, float& m, float& v, float omb1, float b2, float omb2,
                                            float step_size, float bc2_sqrt, float eps, float gs) {
    g *= gs;
    m = m + omb1 * (g - m);
    v = v * b2 + omb2 * (g * g);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}
// Host-side folding of the hyper-parameters into what the kernels take.
struct PgAdamScalars { float omb1, b2, omb2, step_size, bc2_sqrt, eps, gs; };
struct pg_adam_args;
PgAdamScalars pg_adam_scalars(const pg_adam_args* a);   // pointwise.hip
