// Library-level entry points and the thread-local error string.
#include <stdarg.h>

#include <algorithm>

#include "s2d_common.h"

namespace s2d {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ __launch_bounds__(256) void zero_fill_kernel(uint4 *p16, size_t n16, uint32_t *tail, int ntail) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p16[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0u;
}
__global__ __launch_bounds__(256) void zero_fill_bytes_kernel(uint8_t *p, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0;
}

int zero_async(void *p, size_t bytes, hipStream_t st) {
    if (!bytes) return S2D_OK;
    if (((uintptr_t)p | bytes) & 3) {   // not dword granular: byte stores (no caller on the hot path)
        hipLaunchKernelGGL(zero_fill_bytes_kernel, dim3((unsigned)std::min<size_t>(2048, (bytes + 255) / 256)), dim3(256), 0, st, (uint8_t *)p, bytes);
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
    // head dwords up to the first 16-byte boundary, 16-byte stores, tail dwords
    uint8_t *b = (uint8_t *)p;
    size_t head = (16 - ((uintptr_t)b & 15)) & 15;
    if (head > bytes) head = bytes;
    if (head) {
        hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, st, (uint4 *)nullptr, (size_t)0, (uint32_t *)b, (int)(head / 4));
        b += head;
        bytes -= head;
    }
    const size_t n16 = bytes / 16;
    const int ntail = (int)((bytes % 16) / 4);
    if (n16 || ntail)
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)std::max<size_t>(1, std::min<size_t>(4096, (n16 + 255) / 256))), dim3(256), 0, st, (uint4 *)b, n16,
                           (uint32_t *)(b + n16 * 16), ntail);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// debugging aid: every resident workgroup fills 64 KB of LDS with `value` (two workgroups per CU cover 128 of the 160 KB) and spins for
// `spin` rounds - a kernel that reads LDS words it did not write picks the value up when it lands on that CU next
__global__ __launch_bounds__(256) void lds_fill_kernel(float value, int spin, float *sink) {
    __shared__ float buf[16384];
    for (int r = 0; r <= spin; ++r) {
        for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = value;
        __syncthreads();
    }
    if (sink && buf[(threadIdx.x * 61) & 16383] == 12345.678f) sink[0] = 1.f;   // (keeps the stores alive)
}
}  // namespace s2d

extern "C" int s2d_debug_lds_fill(float value, int blocks, int spin, float *sink, s2d_stream_t stream) {
    S2D_CHECK_ARG(blocks > 0 && spin >= 0, "debug_lds_fill: bad argument");
    hipLaunchKernelGGL(s2d::lds_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, value, spin, sink);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_version(void) { return 100; }

/* compiler / runtime the library was built with (the MFMA builtin path of csrc/spconv_rg.hip ties its code quality to the hipcc release:
 * bench.py prints this string next to its numbers) */
extern "C" int s2d_build_info(char *buf, size_t buf_len) {
    char text[256];
    snprintf(text, sizeof(text), "hipcc clang %s; HIP %d.%d.%d; gfx950; built %s", __clang_version__, HIP_VERSION_MAJOR, HIP_VERSION_MINOR,
             HIP_VERSION_PATCH, __DATE__);
    size_t n = strlen(text);
    if (buf && buf_len) {
        size_t m = n < buf_len - 1 ? n : buf_len - 1;
        memcpy(buf, text, m);
        buf[m] = 0;
    }
    return (int)n;
}

extern "C" int s2d_last_error(char *buf, size_t buf_len) {
    size_t n = strlen(s2d::g_err);
    if (buf && buf_len) {
        size_t m = n < buf_len - 1 ? n : buf_len - 1;
        memcpy(buf, s2d::g_err, m);
        buf[m] = 0;
    }
    return (int)n;
}
