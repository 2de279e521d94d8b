// dev tool: a register-heavy kernel that holds `blocks` workgroup slots busy for ~`cycles` clock ticks, emulating a
// collective kernel that occupies part of the chip while the conv kernels run.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void spin_kernel(long long cycles, float* sink) {
    float r[192];
#pragma unroll
    for (int i = 0; i < 192; ++i) r[i] = threadIdx.x * 0.001f + i;
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < 192; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 192; ++i) s += r[i];
    if (s == 12345.678f) sink[0] = s;
}
extern "C" int pg_dev_spin(int blocks, long long cycles, float* sink, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cycles, sink);
    return (int)hipGetLastError();
}
