// Hard voxelization on gfx950, bit-compatible with the reference's sequential loop
// (det3d/ops/point_cloud/point_cloud_ops.py:7-55) and fused with the mean reader
// (det3d/models/readers/voxel_encoder.py:17-24).
//
// The reference is a first-come-first-served loop; the parallel restatement is
//   1. vox_insert : per point, fp32 (p-lo)/vs with a correctly rounded divide + floor, range
//                   test, linear cell key; open-addressing hash insert (L2-resident table, 8 B
//                   per slot) and atomicMin of the point index  -> "first point of the cell"
//   2. scan       : flag[i] = (first[slot(i)] == i); exclusive scan in point order = voxel id in
//                   first-appearance order; ids >= max_voxels are dropped (cells that appear
//                   after the table is full never enter it in the reference either)
//   3. vox_ksmall : per point, cascade of atomicMin through the voxel's max_points-entry list =
//                   the max_points smallest point indices in ascending order (slot order = point
//                   order), independent of execution order
//   4. vox_fill   : gather points into voxels[M][P][ndim] (zero padded), num_points, and the
//                   per-voxel mean (slot-ordered fp32 sum / count)
// HBM-bound integer/byte work: all tables are int32, reads of `points` are the only full-size
// stream (20 B/point), everything else is L2-resident at 150 k points.
#include <limits.h>

#include "s2d_common.h"
#include "scan.h"

namespace s2d {

// r04: ONE hipMemsetAsync(0x7F) clears the hash keys, the first-point table, the k-smallest lists and the scan granules
constexpr uint32_t VOX_EMPTY = 0x7F7F7F7Fu;   // > any cell key (grids are checked against it)
constexpr int IDX_EMPTY = 0x7F7F7F7F;  // what hipMemsetAsync(.., 0x7F, ..) produces; > any point index

struct VoxParams {
    float lo[3];
    float vs[3];
    int grid[3];  // x, y, z
    int ndim;
    int max_points;
    int max_voxels;
    uint32_t table_mask;
    int table_shift;  // 32 - log2(table size)
};

__device__ __forceinline__ uint32_t vox_hash(uint32_t key, int shift) { return (key * 2654435761u) >> shift; }

// Open-addressing insert of `key` + atomicMin of the point index.  The chip retires ~21 global atomics per ns (r04 measurement), and
// two per point bounded this kernel: a relaxed load first - a slot that already holds the key needs no CAS, a first-point entry that is
// already smaller needs no atomicMin.  Keys go EMPTY -> key once and first-point entries only decrease, so a stale read can only
// make the thread take the atomic it would have taken anyway.
__device__ __forceinline__ uint32_t vox_table_insert(uint32_t *__restrict__ keys, int *__restrict__ first, uint32_t key, uint32_t h, uint32_t mask, int i) {
    while (true) {
        const uint32_t cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) break;
        if (cur == VOX_EMPTY) {
            const uint32_t old = atomicCAS(&keys[h], VOX_EMPTY, key);
            if (old == VOX_EMPTY || old == key) break;
        }
        h = (h + 1) & mask;
    }
    if (__hip_atomic_load(&first[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(&first[h], i);
    return h;
}

__global__ __launch_bounds__(256) void vox_insert_kernel(const float *__restrict__ points, int n, VoxParams p,
                                                         uint32_t *__restrict__ keys, int *__restrict__ first,
                                                         int *__restrict__ pt_slot, int *__restrict__ scan_err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *scan_err = 0;   // the look-back scan that follows sets it on a spin time-out (scan.h); read by the count kernel behind it
    if (i >= n) return;
    const float *pt = points + (int64_t)i * p.ndim;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // point_cloud_ops.py:36 — IEEE fp32 subtract, correctly rounded divide, floor
        float q = __fdiv_rn(__fsub_rn(pt[j], p.lo[j]), p.vs[j]);
        float f = floorf(q);
        ok = ok && (f >= 0.0f) && (f < (float)p.grid[j]);
        c[j] = (int)f;
    }
    int slot = -1;
    if (ok) {
        const uint32_t key = ((uint32_t)c[2] * (uint32_t)p.grid[1] + (uint32_t)c[1]) * (uint32_t)p.grid[0] + (uint32_t)c[0];
        slot = (int)vox_table_insert(keys, first, key, vox_hash(key, p.table_shift), p.table_mask, i);
    }
    pt_slot[i] = slot;
}

struct FirstFlagIn {
    const int *pt_slot;
    const int *first;
    __device__ int operator()(int64_t i) const {
        int s = pt_slot[i];
        return (s >= 0 && first[s] == (int)i) ? 1 : 0;
    }
};

struct AssignVoxelOut {
    const int *pt_slot;
    const uint32_t *keys;
    int *vid;          // per table slot
    int32_t *coors;    // [max_voxels][3] z,y,x
    VoxParams p;
    __device__ void operator()(int64_t i, int flag, int rank) const {
        if (!flag) return;
        const int s = pt_slot[i];
        if (rank < p.max_voxels) {
            vid[s] = rank;
            uint32_t key = keys[s];
            int x = key % (uint32_t)p.grid[0];
            key /= (uint32_t)p.grid[0];
            int y = key % (uint32_t)p.grid[1];
            int z = key / (uint32_t)p.grid[1];
            coors[3 * rank + 0] = z;
            coors[3 * rank + 1] = y;
            coors[3 * rank + 2] = x;
        } else {
            vid[s] = -1;
        }
    }
};

__global__ void vox_finalize_count_kernel(const int *total, int max_voxels, int32_t *out_m) {
    int t = *total;
    *out_m = total[1] == 1 ? -1 : (t < max_voxels ? t : max_voxels);   // total[1]: the scan's time-out flag -> a negative count, the host raises
}

// list[0] of a voxel is its first point, which vox_insert already knows (first[slot]): that thread stores it; the other points run
// the atomicMin cascade over list[1..] only (r04: about half the atomics of a cascade from list[0], 70 -> see DESIGN us per chain)
__global__ __launch_bounds__(256) void vox_ksmall_kernel(const int *__restrict__ pt_slot, const int *__restrict__ first, const int *__restrict__ vid,
                                                         int n, int max_points, int *__restrict__ ksmall) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = pt_slot[i];
    if (s < 0) return;
    const int v = vid[s];
    if (v < 0) return;
    int *list = ksmall + (int64_t)v * max_points;
    if (first[s] == i) {
        list[0] = i;
        return;
    }
    if (max_points < 2) return;
    int x = i;
    // cheap early-out: already larger than the current last entry (entries only decrease)
    if (__hip_atomic_load(&list[max_points - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < x) return;
    for (int r = 1; r < max_points; ++r) {
        int old = atomicMin(&list[r], x);
        if (old == IDX_EMPTY) return;  // took an empty place, nothing displaced
        x = old > x ? old : x;       // carry the larger one down the list
    }
}

// ---- dense cells (pillars: max_points = 20, ~14 points per cell on average, hundreds in the cells next to the sensor) --------------
// The atomicMin cascade above serialises on the 80-byte list of a cell: when the ~1000 points of a dense pillar arrive together each of
// them walks the whole list (0.42 ms per chain on the Waymo pillar grid, all of it same-line atomics).  For max_points >=
// VOXSEL_MIN_POINTS the points are instead bucketed per cell - count (one fire-and-forget atomic per point), exclusive scan, ticket
// scatter (one returning atomic per point) - and a wave per cell picks the max_points smallest indices in index order: cells of <= 64
// points rank every entry by counting the smaller ones (v_readlane sweep), larger cells extract minima round by round.  The counters
// start at 0x7F7F7F7F (they live in the workspace's one 0x7F memset) and are read relative to it.
constexpr int VOXSEL_MIN_POINTS = 8;

__global__ __launch_bounds__(256) void voxsel_count_kernel(const int *__restrict__ pt_slot, const int *__restrict__ vid, int n, unsigned *__restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = pt_slot[i];
    if (s < 0) return;
    const int v = vid[s];
    if (v < 0) return;
    __hip_atomic_fetch_add(&cnt[v], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct CellCountIn {
    const unsigned *cnt;
    __device__ int operator()(int64_t i) const { return (int)(cnt[i] - VOX_EMPTY); }
};
struct CellStartOut {
    int *start;
    __device__ void operator()(int64_t i, int, int run) const { start[i] = run; }
};
__global__ __launch_bounds__(256) void voxsel_scatter_kernel(const int *__restrict__ pt_slot, const int *__restrict__ vid, int n,
                                                             const int *__restrict__ start, unsigned *__restrict__ fillc, int *__restrict__ bucket) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = pt_slot[i];
    if (s < 0) return;
    const int v = vid[s];
    if (v < 0) return;
    const unsigned ticket = atomicAdd(&fillc[v], 1u) - VOX_EMPTY;
    bucket[start[v] + (int)ticket] = i;
}
__global__ __launch_bounds__(256) void voxsel_select_kernel(const unsigned *__restrict__ cnt, const int *__restrict__ start, const int *__restrict__ bucket,
                                                            int64_t rows, int max_points, int *__restrict__ ksmall) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= rows) return;
    const int np = (int)(cnt[v] - VOX_EMPTY);
    if (np <= 0) return;
    const int *ent = bucket + start[v];
    int *list = ksmall + v * max_points;
    if (np <= 64) {
        const int x = lane < np ? ent[lane] : 0x7FFFFFFF;
        int rank = 0;
        for (int j = 0; j < np; ++j) rank += __builtin_amdgcn_readlane(x, j) < x ? 1 : 0;
        if (lane < np && rank < max_points) list[rank] = x;
        return;
    }
    int prev = -1;
    const int rounds = np < max_points ? np : max_points;
    for (int r = 0; r < rounds; ++r) {
        int m = 0x7FFFFFFF;
        for (int e = lane; e < np; e += 64) {
            const int y = ent[e];
            m = (y > prev && y < m) ? y : m;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int y = __shfl_xor(m, o);
            m = y < m ? y : m;
        }
        if (lane == 0) list[r] = m;
        prev = m;
    }
}

struct VoxSelWs {
    unsigned *cnt, *fillc;        // per cell row, inside the 0x7F clear region
    unsigned long long *flags;    // scan granules of the per-cell scan, inside the clear region
    int *start, *bucket;
};
// the clear-region part; call between the other clear-region takes
static void voxsel_carve_clear(Carver &c, VoxSelWs &w, int64_t rows, int max_points) {
    w.cnt = w.fillc = nullptr; w.flags = nullptr; w.start = w.bucket = nullptr;
    if (max_points < VOXSEL_MIN_POINTS) return;
    const size_t r = (size_t)(rows > 0 ? rows : 1);
    w.cnt = c.take<unsigned>(r);
    w.fillc = c.take<unsigned>(r);
    w.flags = c.take<unsigned long long>(scan1_num_blocks((int64_t)r));
}
static void voxsel_carve_rest(Carver &c, VoxSelWs &w, int64_t rows, int64_t n_points, int max_points) {
    if (max_points < VOXSEL_MIN_POINTS) return;
    w.start = c.take<int>((size_t)(rows > 0 ? rows : 1));
    w.bucket = c.take<int>((size_t)(n_points > 0 ? n_points : 1));
}
// the max_points smallest point indices of every cell, in index order, into ksmall (pre-set to IDX_EMPTY)
static int vox_select_launch(const int *pt_slot, const int *first, const int *vid, int n, int max_points, int64_t rows, int *ksmall, const VoxSelWs &w,
                             hipStream_t st) {
    const dim3 blk(256);
    if (max_points < VOXSEL_MIN_POINTS) {
        hipLaunchKernelGGL(vox_ksmall_kernel, dim3((n + 255) / 256), blk, 0, st, pt_slot, first, vid, n, max_points, ksmall);
        S2D_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(voxsel_count_kernel, dim3((n + 255) / 256), blk, 0, st, pt_slot, vid, n, w.cnt);
    CellCountIn cin{w.cnt};
    CellStartOut cout{w.start};
    int rc = device_exclusive_scan_onepass(cin, cout, rows, w.flags, nullptr, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(voxsel_scatter_kernel, dim3((n + 255) / 256), blk, 0, st, pt_slot, vid, n, w.start, w.fillc, w.bucket);
    hipLaunchKernelGGL(voxsel_select_kernel, dim3((unsigned)ceil_div(rows, 4)), blk, 0, st, w.cnt, w.start, w.bucket, rows, max_points, ksmall);
    S2D_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void vox_fill_kernel(const float *__restrict__ points, const int *__restrict__ ksmall,
                                                       const int32_t *__restrict__ out_m, int ndim, int max_points,
                                                       float *__restrict__ voxels, int32_t *__restrict__ num_points) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (voxel, slot)
    const int m = *out_m;
    const int64_t v = t / max_points;
    const int r = (int)(t - v * max_points);
    if (v >= m) return;
    const int pid = ksmall[t];
    float *dst = voxels + t * ndim;
    if (pid != IDX_EMPTY) {
        const float *src = points + (int64_t)pid * ndim;
        for (int c = 0; c < ndim; ++c) dst[c] = src[c];
    } else {
        for (int c = 0; c < ndim; ++c) dst[c] = 0.0f;
    }
    if (r == 0) {
        int cnt = 0;
        for (int q = 0; q < max_points; ++q) cnt += (ksmall[v * max_points + q] != IDX_EMPTY) ? 1 : 0;
        num_points[v] = cnt;
    }
}

__global__ __launch_bounds__(256) void vox_mean_kernel(const float *__restrict__ points, const int *__restrict__ ksmall,
                                                       const int32_t *__restrict__ out_m, int ndim, int max_points,
                                                       float *__restrict__ mean) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (voxel, channel)
    const int m = *out_m;
    const int64_t v = t / ndim;
    const int c = (int)(t - v * ndim);
    if (v >= m) return;
    float s = 0.0f;
    int cnt = 0;
    for (int q = 0; q < max_points; ++q) {  // slot order, like features.sum(dim=1)
        int pid = ksmall[v * max_points + q];
        if (pid != IDX_EMPTY) {
            s = __fadd_rn(s, points[(int64_t)pid * ndim + c]);
            ++cnt;
        }
    }
    mean[t] = __fdiv_rn(s, (float)cnt);
}

struct VoxWs {
    uint32_t *keys;
    int *first;
    int *vid;
    int *pt_slot;
    unsigned long long *flags;   // scan granules
    int *total;
    int *ksmall;
    size_t table_size;
    size_t clear_bytes;          // keys .. flags: one 0x7F memset
    size_t bytes;
    VoxSelWs sel;
};

static VoxWs vox_carve(void *ws, int64_t n_points, int max_points, int max_voxels) {
    VoxWs w;
    size_t t = 1024;
    while (t < (size_t)(2 * (n_points > 0 ? n_points : 1))) t <<= 1;
    w.table_size = t;
    Carver c(ws);
    int64_t rows = n_points < max_voxels ? n_points : max_voxels;
    w.keys = c.take<uint32_t>(t);
    w.first = c.take<int>(t);
    w.ksmall = c.take<int>((size_t)(rows > 0 ? rows : 1) * max_points);
    w.flags = c.take<unsigned long long>(scan1_num_blocks(n_points));
    voxsel_carve_clear(c, w.sel, rows, max_points);
    w.clear_bytes = c.total();
    w.vid = c.take<int>(t);
    w.pt_slot = c.take<int>(n_points > 0 ? n_points : 1);
    w.total = c.take<int>(2);   // [1] = scan error flag
    voxsel_carve_rest(c, w.sel, rows, n_points, max_points);
    w.bytes = c.total();
    return w;
}

}  // namespace s2d

using namespace s2d;

extern "C" size_t s2d_voxelize_workspace_bytes(int64_t n_points, int max_points, int max_voxels) {
    if (n_points < 0 || max_points <= 0 || max_voxels <= 0) return 0;
    return vox_carve(nullptr, n_points, max_points, max_voxels).bytes;
}

extern "C" int s2d_voxelize_run(const float *points, int64_t n_points, int ndim, const float coors_range[6],
                                const float voxel_size[3], int max_points, int max_voxels, float *voxels,
                                int32_t *coors, int32_t *num_points, float *mean, int32_t *out_m, void *ws,
                                size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(n_points >= 0 && n_points < (int64_t)INT_MAX / 2, "voxelize: n_points=%lld out of range", (long long)n_points);
    S2D_CHECK_ARG(ndim >= 3 && max_points > 0 && max_voxels > 0, "voxelize: bad ndim/max_points/max_voxels");
    S2D_CHECK_ARG(coors_range && voxel_size && out_m && coors && voxels && num_points, "voxelize: null argument");
    S2D_CHECK_ARG(n_points == 0 || points, "voxelize: null points");
    hipStream_t st = (hipStream_t)stream;
    VoxParams p;
    double cells = 1.0;
    for (int j = 0; j < 3; ++j) {
        p.lo[j] = coors_range[j];
        p.vs[j] = voxel_size[j];
        // point_cloud_ops.py:24-27 and voxel_generator.py:10-11: fp32 divide, round half to even
        float g = (coors_range[3 + j] - coors_range[j]) / voxel_size[j];
        p.grid[j] = (int)nearbyintf(g);
        S2D_CHECK_ARG(p.grid[j] > 0, "voxelize: empty grid on axis %d", j);
        cells *= p.grid[j];
    }
    if (cells >= (double)VOX_EMPTY) {
        set_error("voxelize: grid of %.0f cells exceeds the key space", cells);
        return S2D_ERR_UNSUPPORTED;
    }
    p.ndim = ndim;
    p.max_points = max_points;
    p.max_voxels = max_voxels;
    VoxWs w = vox_carve(ws, n_points, max_points, max_voxels);
    if (ws_bytes < w.bytes || !ws) {
        set_error("voxelize: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return S2D_ERR_WORKSPACE;
    }
    int lg = 0;
    while (((size_t)1 << lg) < w.table_size) ++lg;
    p.table_mask = (uint32_t)(w.table_size - 1);
    p.table_shift = 32 - lg;

    if (n_points == 0) {
        S2D_HIP(hipMemsetAsync(out_m, 0, sizeof(int32_t), st));
        return S2D_OK;
    }
    const int n = (int)n_points;
    const int64_t rows = n_points < max_voxels ? n_points : max_voxels;
    S2D_HIP(hipMemsetAsync(w.keys, 0x7F, w.clear_bytes, st));
    const dim3 blk(256);
    hipLaunchKernelGGL(vox_insert_kernel, dim3((n + 255) / 256), blk, 0, st, points, n, p, w.keys, w.first, w.pt_slot, w.total + 1);
    S2D_LAUNCH_CHECK();
    FirstFlagIn fin{w.pt_slot, w.first};
    AssignVoxelOut fout{w.pt_slot, w.keys, w.vid, coors, p};
    int rc = device_exclusive_scan_onepass(fin, fout, n_points, w.flags, w.total, w.total + 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(vox_finalize_count_kernel, dim3(1), dim3(1), 0, st, w.total, max_voxels, out_m);
    rc = vox_select_launch(w.pt_slot, w.first, w.vid, n, max_points, rows, w.ksmall, w.sel, st);
    if (rc) return rc;
    const int64_t fill_threads = rows * max_points;
    hipLaunchKernelGGL(vox_fill_kernel, dim3((unsigned)ceil_div(fill_threads, 256)), blk, 0, st, points, w.ksmall, out_m,
                       ndim, max_points, voxels, num_points);
    if (mean) {
        const int64_t mean_threads = rows * ndim;
        hipLaunchKernelGGL(vox_mean_kernel, dim3((unsigned)ceil_div(mean_threads, 256)), blk, 0, st, points, w.ksmall,
                           out_m, ndim, max_points, mean);
    }
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// =====================================================================================================================
// Batched voxelizer: B frames in ONE launch chain, writing the collated example directly (what Voxelization.__call__ per frame +
// collate_kitti produce: /root/reference/det3d/datasets/pipelines/preprocess.py:316-345, det3d/torchie/parallel/collate.py:105-144):
// points of all frames concatenated frame after frame, voxels numbered frame by frame in first-seen order, coordinates with the
// batch index prepended (b, z, y, x).  Per-frame hash tables (key = cell inside the frame) share one allocation; `first` holds
// GLOBAL point indices, so one scan over all points yields global first-seen ranks; per-frame starts S_b turn them into local
// ranks for the max_voxels cut (new cells past the cap are dropped, existing ones keep filling: point_cloud_ops.py:44-54).
// =====================================================================================================================
namespace s2d {

constexpr int VOXB_MAX_FRAMES = 64;

struct VoxBatch {
    int frames;
    int offs[VOXB_MAX_FRAMES + 1];        // point offsets of the frames
    int table_base[VOXB_MAX_FRAMES];      // first hash slot of the frame
    uint32_t table_mask[VOXB_MAX_FRAMES];
    int table_shift[VOXB_MAX_FRAMES];
};

__device__ __forceinline__ int voxb_frame_of(const VoxBatch &vb, int i) {
    int b = 0;
    while (b + 1 < vb.frames && i >= vb.offs[b + 1]) ++b;   // <= 64 frames: a short scalar-friendly scan
    return b;
}

__global__ __launch_bounds__(256) void voxb_insert_kernel(const float *__restrict__ points, int n, VoxParams p, VoxBatch vb,
                                                          uint32_t *__restrict__ keys, int *__restrict__ first, int *__restrict__ pt_slot,
                                                          int *__restrict__ scan_err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *scan_err = 0;   // see vox_insert_kernel
    if (i >= n) return;
    const int b = voxb_frame_of(vb, i);
    const float *pt = points + (int64_t)i * p.ndim;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float q = __fdiv_rn(__fsub_rn(pt[j], p.lo[j]), p.vs[j]);
        float f = floorf(q);
        ok = ok && (f >= 0.0f) && (f < (float)p.grid[j]);
        c[j] = (int)f;
    }
    int slot = -1;
    if (ok) {
        const uint32_t key = ((uint32_t)c[2] * (uint32_t)p.grid[1] + (uint32_t)c[1]) * (uint32_t)p.grid[0] + (uint32_t)c[0];
        slot = vb.table_base[b] + (int)vox_table_insert(keys + vb.table_base[b], first + vb.table_base[b], key, vox_hash(key, vb.table_shift[b]),
                                                        vb.table_mask[b], i);
    }
    pt_slot[i] = slot;
}

// scan output: global first-seen rank per table slot, and the rank at every frame start (S_b)
struct VoxbRankOut {
    const int *pt_slot;
    int *rank_of_slot;
    int *frame_start_rank;   // [frames + 1]
    VoxBatch vb;
    int n;
    __device__ void operator()(int64_t i, int flag, int rank) const {
        for (int b = 0; b < vb.frames; ++b)
            if ((int)i == vb.offs[b]) frame_start_rank[b] = rank;
        if ((int)i == n - 1) frame_start_rank[vb.frames] = rank + flag;
        if (flag) rank_of_slot[pt_slot[i]] = rank;
    }
};

// one block: per-frame voxel counts (capped), their exclusive prefix (output row base), totals
__global__ void voxb_frame_counts_kernel(const int *__restrict__ frame_start_rank, VoxBatch vb, int max_voxels, int32_t *__restrict__ out_m /*[frames]*/,
                                         int *__restrict__ out_base /*[frames + 1]*/, int32_t *__restrict__ out_base_user /*[frames + 1]*/,
                                         const int *__restrict__ scan_err) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (*scan_err == 1) {   // the look-back scan timed out: its ranks are wrong - no rows, and a negative total that the host turns into an error
        for (int b = 0; b < vb.frames; ++b) out_m[b] = out_base[b] = out_base_user[b] = 0;
        out_base[vb.frames] = 0;
        out_base_user[vb.frames] = -1;
        return;
    }
    int base = 0;
    for (int b = 0; b < vb.frames; ++b) {
        // a frame start at or past the last point was never visited by the scan: its rank is the total
        const int n = vb.offs[vb.frames];
        const int s0 = vb.offs[b] >= n ? frame_start_rank[vb.frames] : frame_start_rank[b];
        const int s1 = vb.offs[b + 1] >= n ? frame_start_rank[vb.frames] : frame_start_rank[b + 1];
        int m = s1 - s0;
        m = m < max_voxels ? m : max_voxels;
        out_m[b] = m;
        out_base[b] = out_base_user[b] = base;
        base += m;
    }
    out_base[vb.frames] = out_base_user[vb.frames] = base;
}

// first points only: voxel row (or -1 past the cap) per table slot + the collated coordinate row
__global__ __launch_bounds__(256) void voxb_assign_kernel(const int *__restrict__ pt_slot, const int *__restrict__ first, const uint32_t *__restrict__ keys,
                                                          const int *__restrict__ rank_of_slot, const int *__restrict__ frame_start_rank,
                                                          const int *__restrict__ out_base, int n, VoxParams p, VoxBatch vb, int *__restrict__ vid,
                                                          int32_t *__restrict__ coors4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = pt_slot[i];
    if (s < 0 || first[s] != i) return;
    const int b = voxb_frame_of(vb, i);
    int start = frame_start_rank[b];
    const int local = rank_of_slot[s] - start;
    if (local < p.max_voxels) {
        const int row = out_base[b] + local;
        vid[s] = row;
        uint32_t key = keys[s];
        const int x = key % (uint32_t)p.grid[0];
        key /= (uint32_t)p.grid[0];
        const int y = key % (uint32_t)p.grid[1];
        const int z = key / (uint32_t)p.grid[1];
        reinterpret_cast<int4 *>(coors4)[row] = int4{b, z, y, x};
    } else {
        vid[s] = -1;
    }
}

__global__ __launch_bounds__(256) void voxb_fill_kernel(const float *__restrict__ points, const int *__restrict__ ksmall, const int *__restrict__ out_base,
                                                        int frames, int ndim, int max_points, float *__restrict__ voxels,
                                                        int32_t *__restrict__ num_points) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (voxel, slot)
    const int m = out_base[frames];
    const int64_t v = t / max_points;
    const int r = (int)(t - v * max_points);
    if (v >= m) return;
    const int pid = ksmall[t];
    float *dst = voxels + t * ndim;
    if (pid != IDX_EMPTY) {
        const float *src = points + (int64_t)pid * ndim;
        for (int c = 0; c < ndim; ++c) dst[c] = src[c];
    } else {
        for (int c = 0; c < ndim; ++c) dst[c] = 0.0f;
    }
    if (r == 0) {
        int cnt = 0;
        for (int q = 0; q < max_points; ++q) cnt += (ksmall[v * max_points + q] != IDX_EMPTY) ? 1 : 0;
        num_points[v] = cnt;
    }
}

__global__ __launch_bounds__(256) void voxb_mean_kernel(const float *__restrict__ points, const int *__restrict__ ksmall, const int *__restrict__ out_base,
                                                        int frames, int ndim, int max_points, float *__restrict__ mean) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (voxel, channel)
    const int m = out_base[frames];
    const int64_t v = t / ndim;
    const int c = (int)(t - v * ndim);
    if (v >= m) return;
    float s = 0.0f;
    int cnt = 0;
    for (int q = 0; q < max_points; ++q) {
        int pid = ksmall[v * max_points + q];
        if (pid != IDX_EMPTY) {
            s = __fadd_rn(s, points[(int64_t)pid * ndim + c]);
            ++cnt;
        }
    }
    mean[t] = __fdiv_rn(s, (float)cnt);
}

struct VoxbWs {
    uint32_t *keys;
    int *first, *vid, *rank_of_slot, *pt_slot, *total, *ksmall, *frame_start_rank, *out_base;
    unsigned long long *flags;   // scan granules
    size_t table_total;
    size_t clear_bytes;          // keys .. flags: one 0x7F memset
    size_t bytes;
    VoxSelWs sel;
};

static size_t voxb_table(int64_t n) {
    size_t t = 1024;
    while (t < (size_t)(2 * (n > 0 ? n : 1))) t <<= 1;
    return t;
}

static VoxbWs voxb_carve(void *ws, int frames, const int64_t *offsets, int max_points, int max_voxels) {
    VoxbWs w;
    size_t tt = 0;
    int64_t rows = 0;
    for (int b = 0; b < frames; ++b) {
        const int64_t nb = offsets[b + 1] - offsets[b];
        tt += voxb_table(nb);
        rows += nb < max_voxels ? nb : max_voxels;
    }
    const int64_t n = offsets[frames];
    w.table_total = tt;
    Carver c(ws);
    w.keys = c.take<uint32_t>(tt);
    w.first = c.take<int>(tt);
    w.ksmall = c.take<int>((size_t)(rows > 0 ? rows : 1) * max_points);
    w.flags = c.take<unsigned long long>(scan1_num_blocks(n));
    voxsel_carve_clear(c, w.sel, rows, max_points);
    w.clear_bytes = c.total();
    w.vid = c.take<int>(tt);
    w.rank_of_slot = c.take<int>(tt);
    w.pt_slot = c.take<int>(n > 0 ? n : 1);
    w.total = c.take<int>(2);
    w.frame_start_rank = c.take<int>(frames + 1);
    w.out_base = c.take<int>(frames + 1);
    voxsel_carve_rest(c, w.sel, rows, n, max_points);
    w.bytes = c.total();
    return w;
}

}  // namespace s2d

extern "C" size_t s2d_voxelize_batch_workspace_bytes(int frames, const int64_t *point_offsets, int max_points, int max_voxels) {
    if (frames <= 0 || frames > VOXB_MAX_FRAMES || !point_offsets || max_points <= 0 || max_voxels <= 0) return 0;
    return voxb_carve(nullptr, frames, point_offsets, max_points, max_voxels).bytes;
}

extern "C" int s2d_voxelize_batch_run(const float *points, int frames, const int64_t *point_offsets, int ndim, const float coors_range[6],
                                      const float voxel_size[3], int max_points, int max_voxels, float *voxels, int32_t *coors4,
                                      int32_t *num_points, float *mean, int32_t *out_m, int32_t *out_base, void *ws, size_t ws_bytes,
                                      s2d_stream_t stream) {
    S2D_CHECK_ARG(frames > 0 && frames <= VOXB_MAX_FRAMES && point_offsets, "voxelize_batch: 1..%d frames", VOXB_MAX_FRAMES);
    S2D_CHECK_ARG(ndim >= 3 && max_points > 0 && max_voxels > 0, "voxelize_batch: bad ndim/max_points/max_voxels");
    S2D_CHECK_ARG(coors_range && voxel_size && out_m && out_base && coors4 && voxels && num_points, "voxelize_batch: null argument");
    const int64_t n_points = point_offsets[frames];
    S2D_CHECK_ARG(point_offsets[0] == 0 && n_points >= 0 && n_points < (int64_t)INT_MAX / 2, "voxelize_batch: bad offsets");
    S2D_CHECK_ARG(n_points == 0 || points, "voxelize_batch: null points");
    hipStream_t st = (hipStream_t)stream;
    VoxParams p;
    double cells = 1.0;
    for (int j = 0; j < 3; ++j) {
        p.lo[j] = coors_range[j];
        p.vs[j] = voxel_size[j];
        float g = (coors_range[3 + j] - coors_range[j]) / voxel_size[j];
        p.grid[j] = (int)nearbyintf(g);
        S2D_CHECK_ARG(p.grid[j] > 0, "voxelize_batch: empty grid on axis %d", j);
        cells *= p.grid[j];
    }
    if (cells >= (double)VOX_EMPTY) {
        set_error("voxelize_batch: grid of %.0f cells exceeds the key space", cells);
        return S2D_ERR_UNSUPPORTED;
    }
    p.ndim = ndim; p.max_points = max_points; p.max_voxels = max_voxels; p.table_mask = 0; p.table_shift = 0;
    VoxbWs w = voxb_carve(ws, frames, point_offsets, max_points, max_voxels);
    if (ws_bytes < w.bytes || !ws) {
        set_error("voxelize_batch: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
        return S2D_ERR_WORKSPACE;
    }
    VoxBatch vb;
    vb.frames = frames;
    int64_t rows = 0;
    size_t tb = 0;
    for (int b = 0; b < frames; ++b) {
        S2D_CHECK_ARG(point_offsets[b + 1] >= point_offsets[b], "voxelize_batch: offsets must not decrease");
        const int64_t nb = point_offsets[b + 1] - point_offsets[b];
        const size_t t = voxb_table(nb);
        int lg = 0;
        while (((size_t)1 << lg) < t) ++lg;
        vb.offs[b] = (int)point_offsets[b];
        vb.table_base[b] = (int)tb;
        vb.table_mask[b] = (uint32_t)(t - 1);
        vb.table_shift[b] = 32 - lg;
        tb += t;
        rows += nb < max_voxels ? nb : max_voxels;
    }
    vb.offs[frames] = (int)n_points;
    if (n_points == 0) {
        S2D_HIP(hipMemsetAsync(out_m, 0, sizeof(int32_t) * frames, st));
        S2D_HIP(hipMemsetAsync(out_base, 0, sizeof(int32_t) * (frames + 1), st));
        return S2D_OK;
    }
    const int n = (int)n_points;
    S2D_HIP(hipMemsetAsync(w.keys, 0x7F, w.clear_bytes, st));
    const dim3 blk(256);
    hipLaunchKernelGGL(voxb_insert_kernel, dim3((n + 255) / 256), blk, 0, st, points, n, p, vb, w.keys, w.first, w.pt_slot, w.total + 1);
    S2D_LAUNCH_CHECK();
    FirstFlagIn fin{w.pt_slot, w.first};
    VoxbRankOut fout{w.pt_slot, w.rank_of_slot, w.frame_start_rank, vb, n};
    int rc = device_exclusive_scan_onepass(fin, fout, n_points, w.flags, w.total, w.total + 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(voxb_frame_counts_kernel, dim3(1), dim3(64), 0, st, w.frame_start_rank, vb, max_voxels, out_m, w.out_base, out_base, w.total + 1);
    hipLaunchKernelGGL(voxb_assign_kernel, dim3((n + 255) / 256), blk, 0, st, w.pt_slot, w.first, w.keys, w.rank_of_slot, w.frame_start_rank,
                       w.out_base, n, p, vb, w.vid, coors4);
    rc = vox_select_launch(w.pt_slot, w.first, w.vid, n, max_points, rows, w.ksmall, w.sel, st);
    if (rc) return rc;
    hipLaunchKernelGGL(voxb_fill_kernel, dim3((unsigned)ceil_div(rows * max_points, 256)), blk, 0, st, points, w.ksmall, w.out_base, frames, ndim,
                       max_points, voxels, num_points);
    if (mean)
        hipLaunchKernelGGL(voxb_mean_kernel, dim3((unsigned)ceil_div(rows * ndim, 256)), blk, 0, st, points, w.ksmall, w.out_base, frames, ndim,
                           max_points, mean);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
