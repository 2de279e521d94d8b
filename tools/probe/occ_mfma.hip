// Dev probe: how much of the fp32 MFMA pipe do N co-resident workgroups keep busy when every workgroup alternates a
// barrier + a non-MFMA phase of t_n cycles with a burst of MFMAs (the slab loop of the conv kernels)?
//   A: 8 accumulators (64 x 128 wave tile), 64 MFMAs per burst, 2 workgroups per CU   (the shipped structure)
//   B: 4 accumulators (64 x 64 wave tile),  32 MFMAs per burst, 3 / 4 workgroups per CU
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/occ_mfma.hip -o tools/probe/occ_mfma ; run: tools/probe/occ_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: barrier + dependent VALU filler; 1: no barrier; 2: barrier + s_sleep filler (no VALU issue); 3: barrier, filler
// split in two halves around the burst; 4: MODE 0 with s_setprio 3 during the filler
template <int NACC, int OCC, int MODE>
__global__ __launch_bounds__(256, OCC) void burst(float* out, int iters, int filler) {
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    if (MODE >= 5) {      // stagger: the workgroup that arrives SECOND on its CU starts half a burst late
        __shared__ int rank;
        if (threadIdx.x == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;   // gfx9 HW_ID: [11:8] cu, [12] sh, [15:13] se
            unsigned* ctr = reinterpret_cast<unsigned*>(out) + 1048576;
            rank = (int)atomicAdd(ctr + (((xcc & 7) * 8 + se) * 2 + sh) * 16 + cu, 1u);
        }
        __syncthreads();
        const int r = rank;
        if (threadIdx.x == 0) out[1048576 + 2048 + blockIdx.x] = (float)r;
        if (r & 1) for (int f = 0; f < (MODE == 5 ? 32 : 16); ++f) __builtin_amdgcn_s_sleep(1);   // ~2000 / ~1000 cycles
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1 && MODE < 5) __syncthreads();
        if (MODE >= 5) __syncthreads();
        if (MODE == 4) __builtin_amdgcn_s_setprio(3);
        if (MODE == 2) { for (int f = 0; f < filler; ++f) __builtin_amdgcn_s_sleep(1); }
        else for (int f = 0; f < filler; ++f) { a = a * 1.0001f + b; asm volatile("" : "+v"(a)); }   // dependent chain: the non-MFMA phase
        if (MODE == 4) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                if (MODE == 7 && kk == 0 && j == 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(3); __builtin_amdgcn_sched_barrier(0); }
                if (MODE == 8 && kk == 4 && j == 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(3); __builtin_amdgcn_sched_barrier(0); }
            }
        if (MODE == 7 || MODE == 8) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); }
        if (MODE == 9 && it == iters - 1 && threadIdx.x == 0) out[2097152 / 4 + blockIdx.x] = (float)(__builtin_amdgcn_s_memtime() & 0xffffff);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 2) void stamp_kernel(float* out, int iters, int filler, unsigned long long* st) {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    if (threadIdx.x == 0) st[3 * blockIdx.x] = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (it == iters / 2 && threadIdx.x == 0) st[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
        for (int f = 0; f < filler; ++f) { a = a * 1.0001f + b; asm volatile("" : "+v"(a)); }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    if (threadIdx.x == 0) st[3 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int OCC, int MODE = 0>
void run(const char* name, int wg_per_cu, int filler, float* out) {
    const int cus = 256, iters = 4000 * 8 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((burst<NACC, OCC, MODE>), dim3(cus * wg_per_cu), dim3(256), 0, 0, out, 200, filler);
    hipEventRecord(e0);
    hipLaunchKernelGGL((burst<NACC, OCC, MODE>), dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters, filler);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)cus * wg_per_cu * 4 * iters * 8.0 * NACC;           // wave-level MFMAs
    const double tflops = mfma * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
    printf("%-28s wg/cu %d filler %4d: %8.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", name, wg_per_cu, filler, ms, tflops, tflops / 157.3 * 100);
}

int main() {
    float* out; hipMalloc(&out, (1048576 + 4096) * sizeof(float));
    for (int filler : {20}) {
        run<8, 2, 0>("barrier + VALU filler", 1, filler, out);
        run<8, 2, 0>("barrier + VALU filler", 2, filler, out);
        run<8, 2, 1>("no barrier", 1, filler, out);
        run<8, 2, 1>("no barrier", 2, filler, out);
        run<8, 2, 2>("barrier + s_sleep filler", 1, filler, out);
        run<8, 2, 2>("barrier + s_sleep filler", 2, filler, out);
        run<8, 2, 4>("barrier + prio-3 filler", 1, filler, out);
        run<8, 2, 4>("barrier + prio-3 filler", 2, filler, out);
        run<8, 2, 7>("prio 3 after the 1st MFMA", 2, filler, out);
        run<8, 2, 8>("prio 3 in the 2nd half", 2, filler, out);
        run<4, 4, 7>("B 4acc: prio 3 after 1st MFMA", 4, filler, out);
        hipMemset(out + 1048576, 0, 4096 * 4);
        run<8, 2, 5>("stagger 2000 (2nd WG on CU)", 2, filler, out);
        {   std::vector<float> rk(512); hipMemcpy(rk.data(), out + 1048576 + 2048, 512 * 4, hipMemcpyDeviceToHost);
            int hist[8] = {0}; for (float v : rk) hist[(int)v < 7 ? (int)v : 7]++;
            printf("  ranks on their CU of the 512 workgroups: 0:%d 1:%d 2:%d 3:%d >=4:%d  (first 16 by blockIdx:", hist[0], hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
            for (int i = 0; i < 16; ++i) printf(" %d", (int)rk[i]); printf(")\n"); }
        hipMemset(out + 1048576, 0, 4096 * 4);
        run<8, 2, 6>("stagger 1000 (2nd WG on CU)", 2, filler, out);
    }
    {   // per-workgroup start / finish stamps for 2 WG/CU: are the two co-resident workgroups served fairly?
        const int n = 512, iters = 4000;
        unsigned long long* st; hipMalloc(&st, n * 3 * sizeof(unsigned long long));
        hipLaunchKernelGGL(stamp_kernel, dim3(n), dim3(256), 0, 0, out, iters, 20, st);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(n * 3); hipMemcpy(h.data(), st, n * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull; for (int i = 0; i < n; ++i) t0 = h[3 * i] < t0 ? h[3 * i] : t0;
        double dmin = 1e30, dmax = 0, dsum = 0; int late = 0;
        std::vector<double> dur(n), mid(n);
        for (int i = 0; i < n; ++i) { dur[i] = (double)(h[3 * i + 2] - h[3 * i]); mid[i] = (double)(h[3 * i + 1] - h[3 * i]); dmin = dur[i] < dmin ? dur[i] : dmin; dmax = dur[i] > dmax ? dur[i] : dmax; dsum += dur[i]; }
        for (int i = 0; i < n; ++i) if (dur[i] > 0.95 * dmax) ++late;
        double mmin = 1e30, mmax = 0; for (int i = 0; i < n; ++i) { mmin = mid[i] < mmin ? mid[i] : mmin; mmax = mid[i] > mmax ? mid[i] : mmax; }
        printf("2 WG/CU, %d iterations: workgroup duration (cycles) min %.0f max %.0f mean %.0f; time to HALF the iterations min %.0f max %.0f; ideal per WG alone %.0f\n",
               iters, dmin, dmax, dsum / n, mmin, mmax, (double)iters * 4096);
        for (int i = 0; i < 16; ++i) printf("  wg %3d: start %8llu half %10.0f end %10.0f\n", i, h[3 * i] - t0, mid[i], dur[i]);
    }
    return 0;
}
