// probe: semantics of __builtin_amdgcn_raw_ptr_buffer_load_lds on gfx950 (lane -> LDS mapping, OOB -> 0?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(const float* src, int n, float* out) {
    __shared__ float lds[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) lds[i] = -7.f;
    __syncthreads();
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * 4, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-lane gather: lane reads src[(63 - lane) * 2]; lanes 60..63 out of range
    int off = (lane < 60) ? (63 - lane) * 2 * 4 : 0x7ffffff0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, &lds[wave * 128], 4, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, &lds[wave * 128 + 64], 4, off, 0, 4, 0);   // inst offset 4 bytes
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = lds[i];
}
int main() {
    const int n = 256;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, n, o);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
        printf("wave %d first: ", w);
        for (int i = 0; i < 8; ++i) printf("%g ", r[w * 128 + i]);
        printf("... last: ");
        for (int i = 56; i < 64; ++i) printf("%g ", r[w * 128 + i]);
        printf("| second: ");
        for (int i = 0; i < 4; ++i) printf("%g ", r[w * 128 + 64 + i]);
        printf("\n");
    }
    return 0;
}
