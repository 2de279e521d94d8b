// Probe (dev tool): which blockIdx.x values share a CU when a 256-thread, 2-workgroups-per-CU kernel fills the chip.
// Every block records XCC_ID and HW_ID (SE / SH / CU) and spins long enough for the first 512 blocks to be co-resident.
// Speed-only information (HIP promises nothing about placement): used to decide whether co-resident workgroups can be given
// tiles that share an operand panel.   hipcc --offload-arch=gfx950 -O3 tools/probe/cu_map.hip -o tools/probe/cu_map && ./cu_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    __shared__ float pad[9 * 1024];                    // 36 KB like the kernels under study
    pad[threadIdx.x] = threadIdx.x;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) { pad[(threadIdx.x * 7) & 1023] += 1.f; }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    if (pad[5] == -1.f) out[0] = 0;
}
int main() {
    const int G = 1024;
    unsigned* d; hipMalloc(&d, G * 8);
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * G); hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by;
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        by[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    printf("%zu distinct (xcc, se, sh, cu)\n", by.size());
    int shown = 0; std::map<int, int> deltas;
    for (auto& kv : by) {
        if (shown++ < 12) { printf("xcc %u se %u sh %u cu %u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15); for (int b : kv.second) printf(" %d", b); printf("\n"); }
        if (kv.second.size() >= 2) deltas[kv.second[1] - kv.second[0]]++;
    }
    printf("delta between the first two blocks of a CU:"); for (auto& d2 : deltas) printf(" %d x%d", d2.first, d2.second); printf("\n");
    return 0;
}
