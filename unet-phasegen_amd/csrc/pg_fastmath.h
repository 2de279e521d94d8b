// pg_fastmath.h -- the transcendental side of the spectrogram framing kernels (data.py:39-47: log1p|z|, angle z; demo.py:39:
// (exp(m) - 1) e^{j phi}), written for the VALU budget of an HBM-bound kernel.  libm's hypotf + log1pf + atan2f cost ~115 VALU
// instructions per bin-frame (63 us of the 164 us fused STFT at 64 x 256 frames of 2048 points, measured round 3); sincosf with its
// Payne-Hanek tail ~100.  These take ~40 and ~22.  Accuracy (float64 reference, tests/test_signal_gpu.py and the G4 golden of the
// imported data.py at 2e-6 absolute): atan2 <= 3e-7 rad, log1p <= 1 ulp, sincos <= 2.5e-7 absolute for |phi| <= 100.
// Branch cuts are atan2f's own (signed zeros, both axes exact).  Coefficients: tools/fit/fit_math.py (Remez on the absolute error,
// checked in float32 Horner arithmetic).  One definition each, used by every kernel that needs it, so that the fused STFT+polar
// kernel and the standalone polar kernel stay bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include "pg_common.h"

// atan(a) for a in [0, 1]: a * Q(a^2), 9 terms, |error| <= 1.8e-7
__device__ __forceinline__ float pg_atan01(float a) {
    const float s = a * a;
    float q = 4.240624806e-03f;
    q = __fmaf_rn(q, s, -2.142428631e-02f);
    q = __fmaf_rn(q, s, 5.095542386e-02f);
    q = __fmaf_rn(q, s, -8.156170715e-02f);
    q = __fmaf_rn(q, s, 1.091886227e-01f);
    q = __fmaf_rn(q, s, -1.426591545e-01f);
    q = __fmaf_rn(q, s, 1.999918899e-01f);
    q = __fmaf_rn(q, s, -3.333332497e-01f);
    q = __fmaf_rn(q, s, 9.999999999e-01f);
    return q * a;
}

// np.angle / atan2f semantics for finite arguments: signed zeros select the branch (atan2(+0, -x) = +pi, atan2(-0, -x) = -pi,
// atan2(+-0, -0) = +-pi), the axes are exact (0, +-pi/2, +-pi as float32).
__device__ __forceinline__ float pg_atan2(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    // v_rcp_f32 flushes denormal inputs and results: bring the pair into its comfortable range first (exact power-of-two scaling)
    const float sc = mx < 0x1p-60f ? 0x1p64f : (mx > 0x1p60f ? 0x1p-64f : 1.0f);
    mx *= sc; mn *= sc;
    const float a = mx > 0.0f ? mn * __builtin_amdgcn_rcpf(mx) : 0.0f;
    float r = pg_atan01(a);
    if (ay > ax) r = 1.57079632679489662f - r;
    if (__float_as_uint(x) >> 31) r = 3.14159265358979324f - r;
    return copysignf(r, y);
}

// log1p(x) for x >= 0: log(u) + ((1 + x) - u) / u with u = fl(1 + x).  The rounding error of the addition, (1 + x) - u = x - (u - 1),
// is exact in float32 and enters to first order only, so the result is as good as logf(u): <= 1 ulp (the G4 golden of the imported
// data.py holds at 1e-7 absolute).  x tiny: logf(1) + x = x.
__device__ __forceinline__ float pg_log1p_pos(float x) {
    const float u = 1.0f + x, d = u - 1.0f;
    return logf(u) + (x - d) * __builtin_amdgcn_rcpf(u);
}

// |re + j im| without hypotf's scaling ladder: exact power-of-two pre-scaling only where the squares would leave the normal range
__device__ __forceinline__ float pg_abs2(float re, float im) {
    const float m = fmaxf(fabsf(re), fabsf(im));
    const float sc = m < 0x1p-60f ? 0x1p64f : (m > 0x1p60f ? 0x1p-64f : 1.0f);
    const float isc = m < 0x1p-60f ? 0x1p-64f : (m > 0x1p60f ? 0x1p64f : 1.0f);
    re *= sc; im *= sc;
    return __builtin_sqrtf(__fmaf_rn(re, re, im * im)) * isc;
}

// sin and cos of phi (radians): phi / pi = k + f, f in [-1/2, 1/2]; (sin, cos)(phi) = (-1)^k (sin, cos)(pi f), polynomials in
// r = f / 2 revolutions on [-1/4, 1/4].  The reduction is exact in float32 except for the rounding of phi / pi (6e-8 |phi| rad).
__device__ __forceinline__ void pg_sincos(float phi, float& sn, float& cs) {
    const float t = phi * 0.318309886183790672f;
    const float k = rintf(t);
    const float r = 0.5f * (t - k), s = r * r;
    float ps = 3.987323178e+01f;
    ps = __fmaf_rn(ps, s, -7.659820792e+01f);
    ps = __fmaf_rn(ps, s, 8.160326573e+01f);
    ps = __fmaf_rn(ps, s, -4.134169186e+01f);
    ps = __fmaf_rn(ps, s, 6.283185302e+00f);
    ps *= r;
    float pc = -2.498223781e+01f;
    pc = __fmaf_rn(pc, s, 6.014401509e+01f);
    pc = __fmaf_rn(pc, s, -8.545357163e+01f);
    pc = __fmaf_rn(pc, s, 6.493934663e+01f);
    pc = __fmaf_rn(pc, s, -1.973920855e+01f);
    pc = __fmaf_rn(pc, s, 9.999999998e-01f);
    const unsigned flip = ((unsigned)(int)k & 1u) << 31;          // k odd: both change sign (|k| < 2^31 for every phase a network emits)
    sn = __uint_as_float(__float_as_uint(ps) ^ flip);
    cs = __uint_as_float(__float_as_uint(pc) ^ flip);
}

// exp(m) - 1 as the reference computes it (np.exp then - 1 in float32: demo.py:39), on v_exp_f32
__device__ __forceinline__ float pg_expm1_ref(float m) { return __builtin_amdgcn_exp2f(m * 1.44269504088896341f) - 1.0f; }

// data.py:39-47 for one bin-frame: [re; im] -> [log1p|z| (use_exp) or |z| ; angle z], with data.py:40's signed-zero arithmetic
__device__ __forceinline__ void pg_polar_one(float re, float im, int use_exp, float& mag, float& ang) {
    pg_complex_from_parts(re, im);
    const float m = pg_abs2(re, im);
    mag = use_exp ? pg_log1p_pos(m) : m;
    ang = pg_atan2(im, re);
}
