// common.hip -- error string, version, fill.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "phasegen.h"
#include "pg_common.h"

static thread_local char g_err[256] = "";

int pg_cu_count() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cached[dev]) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

int pg_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s (code %d)", msg ? msg : "error", code);
    return code;
}

extern "C" const char* pg_last_error_string(void) { return g_err; }
extern "C" int pg_version(void) { return PG_VERSION; }

namespace {
__global__ void fill_kernel(float* p, long n, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
}  // namespace

extern "C" int pg_fill(float* p, int64_t n, float value, void* stream) {
    if (!p && n > 0) return pg_fail(PG_ERR_NULL, "fill: null pointer");
    if (n <= 0) return PG_OK;
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, (long)n, value);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PG_OK : pg_fail((int)e, hipGetErrorString(e));
}
