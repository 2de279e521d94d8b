// Dev tool: what the chip SUSTAINS on v_mfma_f32_32x32x16_bf16 with no memory traffic at all -- the power/clock-limited MFMA rate the
// bf16-resident conv kernels can be priced against next to the 2516.6 TFLOP/s data-sheet peak (tools/mfma_peak.py).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, long long* clk, int random) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // operands: constants (few bits toggle between MFMAs: the least power) or NSET sets of uniform random values in [-1, 1) used in
    // rotation (what a real GEMM on random data feeds the pipe)
    constexpr int NSET = 4;
    bf16x8 av[NSET], bv[NSET];
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int s = 0; s < NSET; ++s)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; const float ra = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
            h = h * 1664525u + 1013904223u; const float rb = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
            av[s][i] = random ? (__bf16)ra : (__bf16)(float)(threadIdx.x & 7);
            bv[s][i] = random ? (__bf16)rb : (__bf16)(float)(i + 1);
        }
    const long long t0 = __builtin_readcyclecounter();      // s_memtime: shader clock
    const long long r0 = wall_clock64();                    // s_memrealtime: 100 MHz
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(u + i) % NSET], bv[(u * 3 + i) % NSET], acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// the same loop on v_mfma_f32_32x32x2_f32 (the fp32 training path's instruction: 64 cycles for 4096 FLOP)
template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop_f32(float* out, int iters, long long* clk, int random) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int NSET = 8;
    float av[NSET], bv[NSET];
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int s = 0; s < NSET; ++s) {
        h = h * 1664525u + 1013904223u; const float ra = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
        h = h * 1664525u + 1013904223u; const float rb = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
        av[s] = random ? ra : (float)(threadIdx.x & 7);
        bv[s] = random ? rb : (float)(s + 1);
    }
    const long long t0 = __builtin_readcyclecounter();
    const long long r0 = wall_clock64();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u + i) % NSET], bv[(u * 3 + i) % NSET], acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// 16 accumulators (256 registers: the compiler keeps them in AGPRs, as the one-wave-per-SIMD conv kernels do), 256 threads
template <int NACC> __global__ __launch_bounds__(256, 1) void mfma_loop_f32_wide(float* out, int iters, long long* clk, int random);
extern "C" int mfma_peak_f32_16(int wgs, int iters, int reps, int random, double* tflops, double* mhz);

extern "C" int mfma_peak_f32(int wgs, int threads, int iters, int reps, int random, double* tflops, double* mhz) {
    float* out; long long* clk;
    if (hipMalloc(&out, (size_t)wgs * threads * 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    hipLaunchKernelGGL((mfma_loop_f32<4>), dim3(wgs), dim3(threads), 0, 0, out, iters, clk, random);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_loop_f32<4>), dim3(wgs), dim3(threads), 0, 0, out, iters, clk, random);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 2;
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2] = {0, 1};
    if (hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) return 4;
    const double flops = (double)wgs * (threads / 64) * (double)iters * 4 * (2.0 * 32 * 32 * 2) * reps;
    *tflops = flops / (ms * 1e-3) / 1e12;
    *mhz = (double)h[0] / ((double)h[1] / 100.0);
    (void)hipFree(out); (void)hipFree(clk);
    return 0;
}

extern "C" int mfma_peak(int wgs, int threads, int iters, int reps, int random, double* tflops, double* mhz) {
    float* out; long long* clk;
    if (hipMalloc(&out, (size_t)wgs * threads * 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    hipLaunchKernelGGL((mfma_loop<4>), dim3(wgs), dim3(threads), 0, 0, out, iters, clk, random);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_loop<4>), dim3(wgs), dim3(threads), 0, 0, out, iters, clk, random);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 2;
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2] = {0, 1};
    if (hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) return 4;
    const double flops = (double)wgs * (threads / 64) * (double)iters * 4 * (2.0 * 32 * 32 * 16) * reps;
    *tflops = flops / (ms * 1e-3) / 1e12;
    *mhz = (double)h[0] / ((double)h[1] / 100.0);           // shader cycles per microsecond of the 100 MHz clock
    (void)hipFree(out); (void)hipFree(clk);
    return 0;
}

template <int NACC>
__global__ __launch_bounds__(256, 1) void mfma_loop_f32_wide(float* out, int iters, long long* clk, int random) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int NSET = 8;
    float av[NSET], bv[NSET];
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int s = 0; s < NSET; ++s) {
        h = h * 1664525u + 1013904223u; const float ra = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
        h = h * 1664525u + 1013904223u; const float rb = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
        av[s] = random ? ra : (float)(threadIdx.x & 7);
        bv[s] = random ? rb : (float)(s + 1);
    }
    const long long t0 = __builtin_readcyclecounter();
    const long long r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u + (i >> 1)) % NSET], bv[(u * 2 + (i & 1)) % NSET], acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

extern "C" int mfma_peak_f32_16(int wgs, int iters, int reps, int random, double* tflops, double* mhz) {
    float* out; long long* clk;
    if (hipMalloc(&out, (size_t)wgs * 256 * 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    hipLaunchKernelGGL((mfma_loop_f32_wide<16>), dim3(wgs), dim3(256), 0, 0, out, iters, clk, random);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_loop_f32_wide<16>), dim3(wgs), dim3(256), 0, 0, out, iters, clk, random);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 2;
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2] = {0, 1};
    if (hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost) != hipSuccess) return 4;
    const double flops = (double)wgs * 4 * (double)iters * 4 * 16 * (2.0 * 32 * 32 * 2) * reps;
    *tflops = flops / (ms * 1e-3) / 1e12;
    *mhz = (double)h[0] / ((double)h[1] / 100.0);
    (void)hipFree(out); (void)hipFree(clk);
    return 0;
}
