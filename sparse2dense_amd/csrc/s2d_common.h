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

// MI355X dispatches consecutive workgroup ids round-robin over its 8 XCDs, each with a private 4 MiB L2.
// Launch xcd_grid(tiles) workgroups and map id -> xcd_tile(id, grid): every XCD then works on one contiguous
// eighth of the tiles, so neighbouring tiles (which share gathered rows / halo pixels) hit the same L2.
constexpr int S2D_XCDS = 8;
static inline unsigned xcd_grid(int64_t tiles) { return (unsigned)(ceil_div(tiles, S2D_XCDS) * S2D_XCDS); }
#if defined(__HIPCC__)
__device__ __forceinline__ int xcd_tile(int block_id, int grid) { return (block_id % S2D_XCDS) * (grid / S2D_XCDS) + block_id / S2D_XCDS; }
#endif

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
