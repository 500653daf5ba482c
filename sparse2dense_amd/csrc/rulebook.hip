// Rulebook construction for submanifold and strided sparse 3-D convolution on gfx950.
//
// r03: (1) SubM builds on grids of <= 8 M cells (stages 2 and 3) use a direct int32 cell -> row table: memset + scatter + probe, no
// ranking.  (2) Measured and dropped for the large stages: a hash table of the N sites.  Open addressing with Fibonacci hashing,
// load <= 0.5: insert 15 us + probe 77 us at stage 0 against 35 us for the rank/select probe (a miss walks to the next empty slot,
// the three x-neighbours no longer share a word) = 98 us per build, no better than the ~100 us it replaced; a blocked variant (a
// 2x4x4-cell block owning a 32-slot window so that a site's 27 neighbours share cache lines) took 450-900 us: occupied blocks are
// dense, windows overflow into each other and linear probing degenerates into long chains.  (3) Also measured and dropped: fusing the scan launches (a reduce launch
// whose last block - release fence + ticket - scans the block sums: 33 us against 8 + 5 for the two launches it replaced, the
// per-block L2 write-back is the cost) and prefix + decode in one launch (82 us against 17 + 18).
//
// Replaces spconv's get_indice_pairs (dense int32 grid of B*D*H*W cells, 371 MB per Waymo sample,
// or a sort+unique) with a rank/select OCCUPANCY INDEX: one bit per cell plus an exclusive
// popcount prefix per 32-cell word, interleaved as uint2{bits,prefix}.  A coordinate lookup is a
// single 8-byte load + v_bcnt; the rank of a set bit IS its row number in sorted linear (b,z,y,x)
// order, so strided-conv outputs come out canonically numbered with no sort, and the x-1/x/x+1
// probes of a 3x3x3 stencil share one word.  The index of the stage-0 grid is 1/32 the size of
// spconv's grid (11.6 MB of bits per sample) and is L2/MALL resident for every later stage.
//
// Outputs are dense gather maps (see include/s2d.h): nbr_out[k][o] / nbr_in[k][j], written
// coalesced (thread = row, loop = offset).
#include "s2d_common.h"
#include "scan.h"
#include <cstdlib>

namespace s2d {

struct Geo {
    int batch;
    int shape[3];   // D,H,W of the indexed grid
    int ksize[3];
    int stride[3];
    int pad[3];
    int dil[3];
    int oshape[3];  // output grid (== shape for subm)
    int kvol;
};

__device__ __forceinline__ int64_t lin_index(int b, int z, int y, int x, const int *shape) {
    return (((int64_t)b * shape[0] + z) * shape[1] + y) * shape[2] + x;
}

// ---- occupancy index ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void occ_set_kernel(const int32_t *__restrict__ coors, int64_t n, Geo g,
                                                      uint2 *__restrict__ occ) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coors)[i];  // b,z,y,x
    if ((unsigned)c.x >= (unsigned)g.batch || (unsigned)c.y >= (unsigned)g.shape[0] ||
        (unsigned)c.z >= (unsigned)g.shape[1] || (unsigned)c.w >= (unsigned)g.shape[2])
        return;  // out-of-range rows are ignored (they get rank -1 below)
    const int64_t lin = lin_index(c.x, c.y, c.z, c.w, g.shape);
    atomicOr(&occ[lin >> 5].x, 1u << (lin & 31));
}

struct PopcIn {
    const uint2 *occ;
    __device__ int operator()(int64_t w) const { return __popc(occ[w].x); }
};
struct PrefixOut {
    uint2 *occ;
    __device__ void operator()(int64_t w, int, int prefix) const { occ[w].y = (uint32_t)prefix; }
};

__device__ __forceinline__ int occ_rank(const uint2 *__restrict__ occ, int64_t lin) {
    const uint2 w = occ[lin >> 5];
    const uint32_t bit = 1u << (lin & 31);
    if (!(w.x & bit)) return -1;
    return (int)(w.y + __popc(w.x & (bit - 1)));
}

// rank -> input row (inputs may arrive in any order, e.g. the voxelizer's first-seen order)
__global__ __launch_bounds__(256) void occ_perm_kernel(const int32_t *__restrict__ coors, int64_t n, Geo g,
                                                       const uint2 *__restrict__ occ, int32_t *__restrict__ row_of_rank) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coors)[i];
    if ((unsigned)c.x >= (unsigned)g.batch || (unsigned)c.y >= (unsigned)g.shape[0] ||
        (unsigned)c.z >= (unsigned)g.shape[1] || (unsigned)c.w >= (unsigned)g.shape[2])
        return;
    const int r = occ_rank(occ, lin_index(c.x, c.y, c.z, c.w, g.shape));
    if (r >= 0) row_of_rank[r] = (int32_t)i;  // duplicate coordinates: last writer wins (undefined in spconv too)
}

// ---- SubM probe --------------------------------------------------------------------------------
template <int KVOL_MAX>
__global__ __launch_bounds__(256) void subm_probe_kernel(const int32_t *__restrict__ coors, int64_t n, Geo g,
                                                         const uint2 *__restrict__ occ,
                                                         const int32_t *__restrict__ row_of_rank,
                                                         int32_t *__restrict__ nbr_out, int32_t *__restrict__ pair_count) {
    __shared__ int cnt[KVOL_MAX];
    for (int k = threadIdx.x; k < g.kvol; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int4 c = reinterpret_cast<const int4 *>(coors)[i];
        const bool valid = (unsigned)c.x < (unsigned)g.batch && (unsigned)c.y < (unsigned)g.shape[0] &&
                           (unsigned)c.z < (unsigned)g.shape[1] && (unsigned)c.w < (unsigned)g.shape[2];
        int k = 0;
        for (int kz = 0; kz < g.ksize[0]; ++kz) {
            const int z = c.y + (kz - g.ksize[0] / 2) * g.dil[0];
            for (int ky = 0; ky < g.ksize[1]; ++ky) {
                const int y = c.z + (ky - g.ksize[1] / 2) * g.dil[1];
                for (int kx = 0; kx < g.ksize[2]; ++kx, ++k) {
                    const int x = c.w + (kx - g.ksize[2] / 2) * g.dil[2];
                    int j = -1;
                    if (valid && (unsigned)z < (unsigned)g.shape[0] && (unsigned)y < (unsigned)g.shape[1] &&
                        (unsigned)x < (unsigned)g.shape[2]) {
                        const int r = occ_rank(occ, lin_index(c.x, z, y, x, g.shape));
                        if (r >= 0) j = row_of_rank[r];
                    }
                    nbr_out[(int64_t)k * n + i] = j;
                    // one LDS atomic per wave and offset (the compiler folds the active-lane count)
                    if (j >= 0) atomicAdd(&cnt[k], 1);
                }
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < g.kvol; k += blockDim.x)
        if (cnt[k]) atomicAdd(&pair_count[k], cnt[k]);
}

// ---- strided conv ------------------------------------------------------------------------------
// for input position p and offset k: output o = (p + pad - k*dil) / stride when divisible & in range
__device__ __forceinline__ bool conv_out_coord(int p, int k, int pad, int dil, int stride, int olim, int *o) {
    const int num = p + pad - k * dil;
    if (num < 0) return false;
    const int q = num / stride;
    if (q * stride != num || q >= olim) return false;
    *o = q;
    return true;
}

__global__ __launch_bounds__(256) void conv_mark_kernel(const int32_t *__restrict__ coors, int64_t n, Geo g,
                                                        uint2 *__restrict__ occ_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coors)[i];
    if ((unsigned)c.x >= (unsigned)g.batch || (unsigned)c.y >= (unsigned)g.shape[0] ||
        (unsigned)c.z >= (unsigned)g.shape[1] || (unsigned)c.w >= (unsigned)g.shape[2])
        return;
    for (int kz = 0; kz < g.ksize[0]; ++kz) {
        int oz;
        if (!conv_out_coord(c.y, kz, g.pad[0], g.dil[0], g.stride[0], g.oshape[0], &oz)) continue;
        for (int ky = 0; ky < g.ksize[1]; ++ky) {
            int oy;
            if (!conv_out_coord(c.z, ky, g.pad[1], g.dil[1], g.stride[1], g.oshape[1], &oy)) continue;
            for (int kx = 0; kx < g.ksize[2]; ++kx) {
                int ox;
                if (!conv_out_coord(c.w, kx, g.pad[2], g.dil[2], g.stride[2], g.oshape[2], &ox)) continue;
                const int64_t lin = lin_index(c.x, oz, oy, ox, g.oshape);
                const uint32_t bit = 1u << (lin & 31);
                uint32_t *wp = &occ_out[lin >> 5].x;
                // most candidates are already set by a neighbour: test before the atomic
                if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(wp, bit);
            }
        }
    }
}

// one thread per 32-cell word: decode the set bits into out_coors[rank] = (b,z,y,x)
__global__ __launch_bounds__(256) void conv_decode_kernel(const uint2 *__restrict__ occ_out, int64_t n_words, Geo g,
                                                          int32_t *__restrict__ out_coors) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint2 e = occ_out[w];
    uint32_t bits = e.x;
    int r = (int)e.y;
    while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        int64_t lin = (w << 5) + b;
        int4 c;
        c.w = (int)(lin % g.oshape[2]); lin /= g.oshape[2];
        c.z = (int)(lin % g.oshape[1]); lin /= g.oshape[1];
        c.y = (int)(lin % g.oshape[0]); lin /= g.oshape[0];
        c.x = (int)lin;
        reinterpret_cast<int4 *>(out_coors)[r++] = c;
    }
}

template <int KVOL_MAX>
__global__ __launch_bounds__(256) void conv_fill_kernel(const int32_t *__restrict__ coors, int64_t n, Geo g,
                                                        const uint2 *__restrict__ occ_out, int64_t n_out,
                                                        int32_t *__restrict__ nbr_out, int32_t *__restrict__ nbr_in,
                                                        int32_t *__restrict__ pair_count) {
    __shared__ int cnt[KVOL_MAX];
    for (int k = threadIdx.x; k < g.kvol; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int4 c = reinterpret_cast<const int4 *>(coors)[i];
        const bool valid = (unsigned)c.x < (unsigned)g.batch && (unsigned)c.y < (unsigned)g.shape[0] &&
                           (unsigned)c.z < (unsigned)g.shape[1] && (unsigned)c.w < (unsigned)g.shape[2];
        int k = 0;
        for (int kz = 0; kz < g.ksize[0]; ++kz) {
            int oz = 0;
            const bool vz = valid && conv_out_coord(c.y, kz, g.pad[0], g.dil[0], g.stride[0], g.oshape[0], &oz);
            for (int ky = 0; ky < g.ksize[1]; ++ky) {
                int oy = 0;
                const bool vy = vz && conv_out_coord(c.z, ky, g.pad[1], g.dil[1], g.stride[1], g.oshape[1], &oy);
                for (int kx = 0; kx < g.ksize[2]; ++kx, ++k) {
                    int ox = 0;
                    const bool vx = vy && conv_out_coord(c.w, kx, g.pad[2], g.dil[2], g.stride[2], g.oshape[2], &ox);
                    int o = -1;
                    if (vx) o = occ_rank(occ_out, lin_index(c.x, oz, oy, ox, g.oshape));
                    nbr_in[(int64_t)k * n + i] = o;
                    if (o >= 0) {
                        nbr_out[(int64_t)k * n_out + o] = (int32_t)i;  // unique writer per (k,o)
                        atomicAdd(&cnt[k], 1);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < g.kvol; k += blockDim.x)
        if (cnt[k]) atomicAdd(&pair_count[k], cnt[k]);
}

// ---- SubM on small grids: direct cell -> row table ---------------------------------------------------
// grids of <= 8 M cells (stages 2 and 3 of the Waymo backbone: 6.2 M / 0.85 M cells at batch 4): an int32 per cell, no ranking
__global__ __launch_bounds__(256) void subm_direct_scatter_kernel(const int32_t *__restrict__ coors, int64_t n, Geo g, int32_t *__restrict__ table,
                                                                  int32_t *__restrict__ pair_count) {
    if (blockIdx.x == 0 && threadIdx.x < g.kvol) pair_count[threadIdx.x] = 0;   // the probe launch accumulates into it
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coors)[i];  // b,z,y,x
    if ((unsigned)c.x >= (unsigned)g.batch || (unsigned)c.y >= (unsigned)g.shape[0] || (unsigned)c.z >= (unsigned)g.shape[1] ||
        (unsigned)c.w >= (unsigned)g.shape[2])
        return;  // out-of-range rows are ignored (nothing can find them)
    atomicMax(&table[lin_index(c.x, c.y, c.z, c.w, g.shape)], (int32_t)i);   // duplicate coordinate: the highest row wins
}

template <int KVOL_MAX>
__global__ __launch_bounds__(256) void subm_direct_probe_kernel(const int32_t *__restrict__ coors, int64_t n, Geo g, const int32_t *__restrict__ table,
                                                                int32_t *__restrict__ nbr_out, int32_t *__restrict__ pair_count) {
    __shared__ int cnt[KVOL_MAX];
    for (int k = threadIdx.x; k < g.kvol; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int4 c = reinterpret_cast<const int4 *>(coors)[i];
        const bool valid = (unsigned)c.x < (unsigned)g.batch && (unsigned)c.y < (unsigned)g.shape[0] &&
                           (unsigned)c.z < (unsigned)g.shape[1] && (unsigned)c.w < (unsigned)g.shape[2];
        int k = 0;
        for (int kz = 0; kz < g.ksize[0]; ++kz) {
            const int z = c.y + (kz - g.ksize[0] / 2) * g.dil[0];
            for (int ky = 0; ky < g.ksize[1]; ++ky) {
                const int y = c.z + (ky - g.ksize[1] / 2) * g.dil[1];
                for (int kx = 0; kx < g.ksize[2]; ++kx, ++k) {
                    const int x = c.w + (kx - g.ksize[2] / 2) * g.dil[2];
                    int j = -1;
                    if (valid && (unsigned)z < (unsigned)g.shape[0] && (unsigned)y < (unsigned)g.shape[1] && (unsigned)x < (unsigned)g.shape[2])
                        j = table[lin_index(c.x, z, y, x, g.shape)];
                    nbr_out[(int64_t)k * n + i] = j;
                    if (j >= 0) atomicAdd(&cnt[k], 1);   // one LDS atomic per wave and offset (the compiler folds the active-lane count)
                }
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < g.kvol; k += blockDim.x)
        if (cnt[k]) atomicAdd(&pair_count[k], cnt[k]);
}

// ---- host side ---------------------------------------------------------------------------------
struct RbWs {
    uint2 *occ;
    int32_t *row_of_rank;
    int *block_sums;
    int *total;
    int64_t n_words;
    size_t bytes;
};

static int64_t occ_words(int batch, const int *shape) {
    const int64_t cells = (int64_t)batch * shape[0] * shape[1] * shape[2];
    return (cells + 31) / 32;
}

static RbWs rb_carve(void *ws, int64_t n_words, int64_t n_rows) {
    RbWs w;
    Carver c(ws);
    w.n_words = n_words;
    w.occ = c.take<uint2>((size_t)(n_words > 0 ? n_words : 1));
    w.row_of_rank = c.take<int32_t>((size_t)(n_rows > 0 ? n_rows : 1));
    w.block_sums = c.take<int>(scan_num_blocks(n_words));
    w.total = c.take<int>(1);
    w.bytes = c.total();
    return w;
}

// SubM on a grid of at most DIRECT_CELLS cells: direct table (one int32 per cell)
constexpr int64_t DIRECT_CELLS = 8 << 20;
static bool direct_ok(int batch, const int *shape) { return (int64_t)batch * shape[0] * shape[1] * shape[2] <= DIRECT_CELLS; }

static int fill_geo(Geo *g, int batch, const int32_t shape[3], const int32_t ksize[3], const int32_t stride[3],
                    const int32_t padding[3], const int32_t dilation[3], bool subm) {
    S2D_CHECK_ARG(batch > 0 && shape && ksize, "rulebook: bad batch/shape/ksize");
    g->batch = batch;
    g->kvol = 1;
    for (int a = 0; a < 3; ++a) {
        S2D_CHECK_ARG(shape[a] > 0 && ksize[a] > 0, "rulebook: non-positive extent on axis %d", a);
        g->shape[a] = shape[a];
        g->ksize[a] = ksize[a];
        g->dil[a] = dilation ? dilation[a] : 1;
        S2D_CHECK_ARG(g->dil[a] > 0, "rulebook: dilation must be positive");
        if (subm) {
            g->stride[a] = 1;
            g->pad[a] = (ksize[a] / 2) * g->dil[a];
            g->oshape[a] = shape[a];
            S2D_CHECK_ARG(ksize[a] % 2 == 1, "rulebook: SubM needs odd kernel sizes");
        } else {
            g->stride[a] = stride ? stride[a] : 1;
            g->pad[a] = padding ? padding[a] : 0;
            S2D_CHECK_ARG(g->stride[a] > 0 && g->pad[a] >= 0, "rulebook: bad stride/padding");
            const int e = shape[a] + 2 * g->pad[a] - g->dil[a] * (ksize[a] - 1) - 1;
            S2D_CHECK_ARG(e >= 0, "rulebook: kernel larger than padded input on axis %d", a);
            g->oshape[a] = e / g->stride[a] + 1;
        }
        g->kvol *= ksize[a];
    }
    if (g->kvol > 27) {
        set_error("rulebook: kernel volume %d > 27 unsupported", g->kvol);
        return S2D_ERR_UNSUPPORTED;
    }
    const double cells_in = (double)batch * shape[0] * shape[1] * shape[2];
    const double cells_out = (double)batch * g->oshape[0] * g->oshape[1] * g->oshape[2];
    if (cells_in >= 6.8e10 || cells_out >= 6.8e10) {  // word index must fit int32 comfortably
        set_error("rulebook: grid too large for the occupancy index");
        return S2D_ERR_UNSUPPORTED;
    }
    return 0;
}

static int build_occ_prefix(const RbWs &w, hipStream_t st) {
    PopcIn pin{w.occ};
    PrefixOut pout{w.occ};
    return device_exclusive_scan(pin, pout, w.n_words, w.block_sums, w.total, st);
}

}  // namespace s2d

using namespace s2d;

extern "C" size_t s2d_rulebook_workspace_bytes(int batch, const int32_t shape[3], int64_t n_rows) {
    if (batch <= 0 || !shape || n_rows < 0) return 0;
    int s[3] = {shape[0], shape[1], shape[2]};
    if (n_rows > 0 && direct_ok(batch, s)) return align_up((size_t)batch * s[0] * s[1] * s[2] * 4, 256);   // SubM: the direct table
    return rb_carve(nullptr, occ_words(batch, s), n_rows).bytes;
}

extern "C" int s2d_rulebook_subm_build(const int32_t *coors, int64_t n, int batch, const int32_t shape[3],
                                       const int32_t ksize[3], const int32_t dilation[3], int32_t *nbr_out,
                                       int32_t *pair_count, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    Geo g;
    int rc = fill_geo(&g, batch, shape, ksize, nullptr, nullptr, dilation, true);
    if (rc) return rc;
    S2D_CHECK_ARG(n >= 0 && n < 0x7fffffff, "rulebook_subm: bad n");
    S2D_CHECK_ARG(pair_count && (n == 0 || (coors && nbr_out)), "rulebook_subm: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (direct_ok(batch, g.shape)) {
        if (n == 0) {
            S2D_HIP(hipMemsetAsync(pair_count, 0, sizeof(int32_t) * g.kvol, st));
            return S2D_OK;
        }
        const size_t bytes = (size_t)batch * g.shape[0] * g.shape[1] * g.shape[2] * 4;
        if (!ws || ws_bytes < bytes) {
            set_error("rulebook_subm: workspace too small (%zu < %zu)", ws_bytes, bytes);
            return S2D_ERR_WORKSPACE;
        }
        int32_t *table = (int32_t *)ws;
        S2D_HIP(hipMemsetAsync(table, 0xFF, bytes, st));
        const dim3 blk(256), grd((unsigned)ceil_div(n, 256));
        hipLaunchKernelGGL(subm_direct_scatter_kernel, grd, blk, 0, st, coors, n, g, table, pair_count);
        hipLaunchKernelGGL(subm_direct_probe_kernel<27>, grd, blk, 0, st, coors, n, g, table, nbr_out, pair_count);
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
    RbWs w = rb_carve(ws, occ_words(batch, g.shape), n);
    if (!ws || ws_bytes < w.bytes) {
        set_error("rulebook_subm: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return S2D_ERR_WORKSPACE;
    }
    S2D_HIP(hipMemsetAsync(pair_count, 0, sizeof(int32_t) * g.kvol, st));
    if (n == 0) return S2D_OK;
    S2D_HIP(hipMemsetAsync(w.occ, 0, sizeof(uint2) * w.n_words, st));
    const dim3 blk(256), grd((unsigned)ceil_div(n, 256));
    hipLaunchKernelGGL(occ_set_kernel, grd, blk, 0, st, coors, n, g, w.occ);
    S2D_LAUNCH_CHECK();
    rc = build_occ_prefix(w, st);
    if (rc) return rc;
    hipLaunchKernelGGL(occ_perm_kernel, grd, blk, 0, st, coors, n, g, w.occ, w.row_of_rank);
    hipLaunchKernelGGL(subm_probe_kernel<27>, grd, blk, 0, st, coors, n, g, w.occ, w.row_of_rank, nbr_out, pair_count);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_rulebook_conv_count(const int32_t *coors, int64_t n, int batch, const int32_t shape[3],
                                       const int32_t ksize[3], const int32_t stride[3], const int32_t padding[3],
                                       const int32_t dilation[3], int32_t *out_n, void *ws, size_t ws_bytes,
                                       s2d_stream_t stream) {
    Geo g;
    int rc = fill_geo(&g, batch, shape, ksize, stride, padding, dilation, false);
    if (rc) return rc;
    S2D_CHECK_ARG(n >= 0 && n < 0x7fffffff && out_n && (n == 0 || coors), "rulebook_conv_count: bad argument");
    hipStream_t st = (hipStream_t)stream;
    RbWs w = rb_carve(ws, occ_words(batch, g.oshape), 0);
    if (!ws || ws_bytes < w.bytes) {
        set_error("rulebook_conv_count: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return S2D_ERR_WORKSPACE;
    }
    S2D_HIP(hipMemsetAsync(w.occ, 0, sizeof(uint2) * w.n_words, st));
    if (n > 0) {
        hipLaunchKernelGGL(conv_mark_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, coors, n, g, w.occ);
        S2D_LAUNCH_CHECK();
    }
    rc = build_occ_prefix(w, st);
    if (rc) return rc;
    S2D_HIP(hipMemcpyAsync(out_n, w.total, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    return S2D_OK;
}

extern "C" int s2d_rulebook_conv_fill(const int32_t *coors, int64_t n, int batch, const int32_t shape[3],
                                      const int32_t ksize[3], const int32_t stride[3], const int32_t padding[3],
                                      const int32_t dilation[3], int64_t n_out, int32_t *out_coors, int32_t *nbr_out,
                                      int32_t *nbr_in, int32_t *pair_count, void *ws, size_t ws_bytes,
                                      s2d_stream_t stream) {
    Geo g;
    int rc = fill_geo(&g, batch, shape, ksize, stride, padding, dilation, false);
    if (rc) return rc;
    S2D_CHECK_ARG(n >= 0 && n_out >= 0 && pair_count, "rulebook_conv_fill: bad argument");
    S2D_CHECK_ARG(n == 0 || (coors && nbr_in), "rulebook_conv_fill: null input maps");
    S2D_CHECK_ARG(n_out == 0 || (out_coors && nbr_out), "rulebook_conv_fill: null output maps");
    hipStream_t st = (hipStream_t)stream;
    RbWs w = rb_carve(ws, occ_words(batch, g.oshape), 0);
    if (!ws || ws_bytes < w.bytes) {
        set_error("rulebook_conv_fill: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return S2D_ERR_WORKSPACE;
    }
    S2D_HIP(hipMemsetAsync(pair_count, 0, sizeof(int32_t) * g.kvol, st));
    if (n_out > 0) {
        S2D_HIP(hipMemsetAsync(nbr_out, 0xFF, sizeof(int32_t) * (size_t)g.kvol * n_out, st));
        hipLaunchKernelGGL(conv_decode_kernel, dim3((unsigned)ceil_div(w.n_words, 256)), dim3(256), 0, st, w.occ,
                           w.n_words, g, out_coors);
    }
    if (n > 0)
        hipLaunchKernelGGL(conv_fill_kernel<27>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, coors, n, g, w.occ,
                           n_out, nbr_out, nbr_in, pair_count);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
