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

// Zero-fill by a KERNEL (csrc/common.hip).  Entry points that can be recorded into a HIP graph (the dense segment, graphed.py) must not
// use hipMemsetAsync: a memset node followed by an atomically accumulating kernel node stopped being ordered after a few replays of the
// graph (ROCm 7.2, r05: the regression-loss scatter added onto the previous replay's gradient map from the fourth replay on).
int zero_async(void *p, size_t bytes, hipStream_t st);

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

// Division of 32-bit unsigned values by a divisor known at launch time (tile index -> coordinates): the multiplier is made on the host,
// the device does a mul-hi, an add and two shifts instead of the ~30-instruction expansion of an integer division.
struct FastDiv {
    uint32_t d, m, s;
};
static inline FastDiv fastdiv_make(uint32_t d) {
    FastDiv f{d, 0u, 0u};
    if (d <= 1) return f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;   // ceil(log2 d)
    f.m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    f.s = l;
    return f;
}
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t fastdiv(uint32_t x, const FastDiv &f) {
    if (f.s == 0) return x;
    const uint32_t t = __umulhi(x, f.m);
    return (t + ((x - t) >> 1)) >> (f.s - 1);
}
#endif

#if defined(__HIPCC__)
// exact (erf) GELU = nn.GELU() and its derivative, as fused behind the row-major batch norm (features.hip) and evaluated by the conv
// epilogue that produces that batch norm's backward sums (conv2d_nhwc.hip): one definition, the two must agree on every element
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
    return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}

// A data-gradient conv whose output IS the gradient dY of a batch-norm + activation layer y = act(z * scale + shift) (act: 1 ReLU, 2 GELU)
// emits that layer's two backward sums per tile from its epilogue - sum g and sum g * z with g = dY * act'(z * scale + shift), over the
// STORED bf16 dY - in the [tile][2][c] slab layout the statistics epilogue uses; z == nullptr: plain epilogue.
struct BnBwd {
    const __bf16 *z = nullptr;
    const float *scale = nullptr, *shift = nullptr;
    int act = 0;
};

// Planar (channel-major) tensors through buffer instructions: ONE per-lane 32-bit byte offset plus a scalar offset per access,
// where flat global pointers cost a 64-bit per-lane address per plane (two VGPRs each: 160 for a 32 + 16 + 32-plane kernel), and
// a per-lane offset >= num_records reads as zero / drops the store (masked lanes need no branch).  The range check looks at
// the PER-LANE offset only: the scalar offset must always be valid.  The b64/b128 builtins return clang-internal vector types
// that do not convert to ext_vector types element-wise (an implicit conversion splats the first element): always bit_cast.
typedef float buf_f32x2 __attribute__((ext_vector_type(2)));
typedef float buf_f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned BUF_OOB = 0x80000000u;   // per-lane offset that is out of range for every resource made below (< 2 GB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);   // raw buffer
}
__device__ __forceinline__ buf_f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(buf_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ buf_f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(buf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
    using raw_t = decltype(__builtin_amdgcn_raw_buffer_load_b32(r, 0u, 0u, 0));
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(raw_t, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, buf_f32x2 v) {
    using raw_t = decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0u, 0u, 0));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(raw_t, v), r, voff, soff, 0);
}
// 16-byte stores fold the scalar offset into the per-lane one: with a REGISTER scalar offset hipcc assumes that a >64-bit buffer
// store may be followed at once by a write of its data registers (no wait state inserted) - measured on gfx950: the next channel's
// v_pk_fma overwrote the data of `buffer_store_dwordx4 ..., s22 offen` before the store had read it.  With an immediate scalar
// offset the hazard recogniser inserts the wait state.  (An out-of-range per-lane offset stays out of range: soff < 2^31.)
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, buf_f32x4 v) {
    using raw_t = decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0u, 0u, 0));
    const unsigned off = voff >= BUF_OOB ? voff : voff + soff;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(raw_t, v), r, off, 0, 0);
}
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
