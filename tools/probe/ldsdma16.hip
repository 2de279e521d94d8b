// probe: 16-byte LDS-DMA (buffer_load ... lds, size 16) from global addresses that are only dword-aligned, with an SGPR
// offset, and out-of-range lanes.  Prints the number of mismatches per alignment (0 = the piece lands exactly).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(const float* src, int n, int shift, int soff, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lds[i] = -7.f;
    __syncthreads();
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    const int off = lane < 60 ? (lane * 5 + shift) * 4 : 0x7ffffff0;        // lane reads 4 floats starting at lane*5 + shift
    float* dst = lds + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, off, __builtin_amdgcn_readfirstlane(soff * 4), 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = lds[i];
}
int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift)
        for (int soff : {0, 3, 64}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, shift, soff, o);
            std::vector<float> r(256);
            hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int lane = 0; lane < 64; ++lane)
                for (int c = 0; c < 4; ++c) {
                    const float want = lane < 60 ? (float)(lane * 5 + shift + soff + c) : 0.f;
                    if (r[lane * 4 + c] != want) ++bad;
                }
            printf("shift %d soffset %d: %d mismatches (lane 1 got %g %g %g %g)\n", shift, soff, bad, r[4], r[5], r[6], r[7]);
        }
    return 0;
}
