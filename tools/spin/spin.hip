// dev tool (not product): kernels that HOLD part of the chip while the conv kernels run, emulating what a collective's kernels
// do to a data-parallel backward.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/spin/spin.hip -o tools/spin/libspin.so
//
//   pg_dev_spin   round-1 form: `blocks` workgroups of a register-heavy kernel (192 live floats) for a fixed number of cycles.
//   pg_dev_hold   `blocks` workgroups of 256 threads, resident until the host raises a flag or `max_us` microseconds have passed
//                 (s_memrealtime, 100 MHz: a bound every wave reaches whatever the shader clock does).
//                   shape 0  "collective-like": 116 VGPRs (<= 128: four such waves fit a SIMD), `lds_bytes` of LDS, ALU spin
//                   shape 1  the same, and every workgroup streams its slice of `buf` (read + write, 16 B per lane) the whole time --
//                            the HBM / fabric side of a collective
//                   shape 2  "fat": 200 VGPRs -- a one-wave-per-SIMD conv workgroup (372 registers) cannot share the CU with it
//                 The control block lives in COHERENT HOST memory (pg_dev_hold_ctl): the host raises the flag with a plain store and
//                 reads the records without touching any stream -- a flag written by a device kernel on another stream never arrived
//                 (HIP multiplexes streams onto a few hardware queues; the writer sat behind the hold kernel in the same queue).
//                 Per block: HW_ID, XCC_ID | bit 31 once resident, ticks (100 MHz) it stayed, loop iterations it made.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

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

struct HoldCtl { int flag; int pad[15]; unsigned rec[1024][4]; };      // rec: hw_id, xcc | started, ticks held, iterations

__device__ __forceinline__ unsigned long long realtime() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

template <int MODE, int NR>
__global__ __launch_bounds__(256) void hold_kernel(HoldCtl* ctl, unsigned long long max_ticks, float4* buf, long n4_per_block, float* sink) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_store(&ctl->rec[blockIdx.x][0], hw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ctl->rec[blockIdx.x][1], xcc | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = threadIdx.x * 0.001f + i;
    const unsigned long long t0 = realtime();
    float4* mine = buf + (long)blockIdx.x * n4_per_block;
    long pos = threadIdx.x;
    unsigned iters = 0;
    unsigned long long now = t0;
    while (true) {
        if (MODE == 1) {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                float4 v = mine[pos];
                v.x += 1.f;
                mine[pos] = v;
                pos += 256; if (pos >= n4_per_block) pos = threadIdx.x;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) r[i] = r[i] * 1.0001f + 0.5f;
        }
        ++iters;
        now = realtime();
        if (__hip_atomic_load(&ctl->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        if (now - t0 > max_ticks) break;
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&ctl->rec[blockIdx.x][2], (unsigned)(now - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&ctl->rec[blockIdx.x][3], iters, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float s = lds[(threadIdx.x * 7) & 255];
#pragma unroll
    for (int i = 0; i < NR; ++i) s += r[i];
    if (s == 12345.678f) sink[0] = s;
}

// control block in coherent (fine-grained) host memory, device-accessible at the same address
extern "C" void* pg_dev_hold_ctl(void) {
    void* p = nullptr;
    if (hipHostMalloc(&p, sizeof(HoldCtl), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return nullptr;
    memset(p, 0, sizeof(HoldCtl));
    return p;
}
extern "C" void pg_dev_hold_reset(void* ctl) { memset(ctl, 0, sizeof(HoldCtl)); }
extern "C" void pg_dev_hold_release(void* ctl) { __atomic_store_n(&((HoldCtl*)ctl)->flag, 1, __ATOMIC_SEQ_CST); }
extern "C" int pg_dev_hold_started(void* ctl, int blocks) {
    int n = 0;
    for (int b = 0; b < blocks; ++b) n += (__atomic_load_n(&((HoldCtl*)ctl)->rec[b][1], __ATOMIC_ACQUIRE) >> 31) & 1;
    return n;
}
extern "C" void pg_dev_hold_records(void* ctl, int blocks, unsigned* out) { memcpy(out, ((HoldCtl*)ctl)->rec, (size_t)blocks * 16); }

extern "C" int pg_dev_hold(int blocks, int shape, int lds_bytes, void* ctl, long long max_us, void* buf, long long buf_bytes, float* sink, void* stream) {
    if (blocks <= 0 || blocks > 1024 || lds_bytes < 1024 || lds_bytes > 64 * 1024 || max_us <= 0 || max_us > 20000000LL || !ctl) return -1;
    const unsigned long long ticks = (unsigned long long)max_us * 100ULL;
    const long n4 = buf ? (long)(buf_bytes / 16 / blocks) : 0;
    hipStream_t st = (hipStream_t)stream;
    if (shape == 1) {
        if (n4 < 256) return -1;
        hipLaunchKernelGGL((hold_kernel<1, 96>), dim3(blocks), dim3(256), lds_bytes, st, (HoldCtl*)ctl, ticks, (float4*)buf, n4, sink);
    } else if (shape == 2) {
        hipLaunchKernelGGL((hold_kernel<0, 190>), dim3(blocks), dim3(256), lds_bytes, st, (HoldCtl*)ctl, ticks, (float4*)buf, n4, sink);
    } else {
        hipLaunchKernelGGL((hold_kernel<0, 108>), dim3(blocks), dim3(256), lds_bytes, st, (HoldCtl*)ctl, ticks, (float4*)buf, n4, sink);
    }
    return (int)hipGetLastError();
}
