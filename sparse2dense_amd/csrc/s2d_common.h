// Shared helpers for the gfx950 kernels of libs2d_hip.so (error state, launch checks, small
// device utilities).  Wave size is 64 on CDNA4; block sizes are multiples of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/s2d.h"

namespace s2d {

void set_error(const char *fmt, ...);

#define S2D_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            s2d::set_error(__VA_ARGS__);         \
            return S2D_ERR_INVALID_ARG;          \
        }                                        \
    } while (0)

#define S2D_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            s2d::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                           __LINE__);                                                      \
            return (int)e__;                                                               \
        }                                                                                  \
    } while (0)

#define S2D_LAUNCH_CHECK() S2D_HIP(hipGetLastError())

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// carve aligned sub-buffers out of a caller workspace
struct Carver {
    char *base;
    size_t off;
    explicit Carver(void *p) : base((char *)p), off(0) {}
    template <typename T>
    T *take(size_t count) {
        off = align_up(off, 256);
        T *p = (T *)(base ? base + off : nullptr);
        off += count * sizeof(T);
        return p;
    }
    size_t total() const { return align_up(off, 256); }
};

}  // namespace s2d
