// Rows of a submanifold rulebook grouped by neighbour mask (r06).
// (spconv.ops.get_indice_pairs builds the pairs; call sites det3d/models/backbones/scn.py:104-152)
//
// Why.  The implicit-GEMM kernels multiply all K = 27 kernel offsets for every 16-row output tile and let a missing neighbour read
// zero.  On the benchmark scene an output row has 15 of 27 neighbours (stages 2-3), so 44 % of the matrix-core issue, of the weight
// slabs pulled through L2 and of the gather instructions belong to (tile, offset) pairs with nothing in them.  In voxel order the
// OR of 16 rows' masks is ~0.9 of 27 - nothing to skip.  Sorted by the 27-bit mask, rows with the same neighbourhood SHAPE (ground
// plane, wall, isolated column ...) sit next to each other: per 16-row tile the union drops to 0.60-0.64 of 27, per 192-256-row
// workgroup of a 4-frame batch to ~0.66-0.74 (tools/mask_sort_study.py).
//
// What is built (once per rulebook, reused by every layer and both passes of the stage):
//   pmask[j]      u32   neighbour mask of the j-th row in sorted order (bit k: offset k has a neighbour); the order is ascending mask
//                       inside each chunk of rows that one XCD's workgroups consume (see rbsort_mask_kernel)
//   perm[j]       i32   that row's index in the canonical (reference) order - the kernels gather by nbr_perm and STORE to perm[j], so
//                       features, rulebooks and every tensor a caller sees stay in the reference's row order, bit for bit
//   nbr_perm[k][j] i32  nbr_out[k][perm[j]]
// The sort is a stable LSD radix sort (hipcub) on the 27 mask bits: deterministic.
#include "s2d_common.h"

#include <cstdlib>

#include <hipcub/hipcub.hpp>

namespace s2d {

// sort key = (chunk of the row) << 27 | mask: the sort stays INSIDE chunks of `chunk_rows` canonical rows - the share of the rows one XCD's
// workgroups process (xcd_tile) - because the canonical order is spatially coherent (frame-major, scan order inside a frame): a global sort
// spreads every workgroup's gathers over the whole 9-17 MB feature matrix, past the XCD's 4 MB L2 (measured r02: slower than unsorted)
__global__ __launch_bounds__(256) void rbsort_mask_kernel(const int32_t *__restrict__ nbr, int kvol, int n, int chunk_rows, uint64_t *__restrict__ key,
                                                          int32_t *__restrict__ iota) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t m = 0;
    for (int k = 0; k < kvol; ++k) m |= (nbr[(int64_t)k * n + i] >= 0 ? 1u : 0u) << k;
    key[i] = ((uint64_t)(i / chunk_rows) << 27) | m;
    iota[i] = i;
}

__global__ __launch_bounds__(256) void rbsort_permute_kernel(const int32_t *__restrict__ nbr, const int32_t *__restrict__ perm, int kvol, int n,
                                                             int32_t *__restrict__ nbr_perm, const uint64_t *__restrict__ skey,
                                                             uint32_t *__restrict__ pmask) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int src = perm[j];
    pmask[j] = (uint32_t)skey[j] & 0x07ffffffu;   // drop the chunk number: what is left is the row's neighbour mask
    for (int k = 0; k < kvol; ++k) nbr_perm[(int64_t)k * n + j] = nbr[(int64_t)k * n + src];
}

static size_t rbsort_cub_bytes(int64_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr,
                                             (int)n, 0, 48, (hipStream_t)0);
    return bytes;
}

}  // namespace s2d

using namespace s2d;

extern "C" size_t s2d_rulebook_sort_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    Carver c(nullptr);
    c.take<uint64_t>((size_t)n);
    c.take<uint64_t>((size_t)n);
    c.take<int32_t>((size_t)n);
    c.take<char>(rbsort_cub_bytes(n));
    return c.total();
}

namespace s2d {
struct RgPlan {
    int mi, waves, tiles_per_block;
    unsigned grid;
};
RgPlan rg_plan(int64_t n_out, int kvol, int cin, int cout);   // csrc/spconv_rg.hip: the launch shape of the kernel that consumes the sorted rows
}

// rows per sort chunk: what the workgroups of one XCD process in the consuming kernel (xcd_tile: XCD x owns the x-th eighth of the workgroups);
// S2D_RG_SORT_CHUNKS=1 sorts globally (A/B runs)
extern "C" int64_t s2d_rulebook_sort_chunk_rows(int64_t n) {
    if (n <= 0) return 1;
    static int chunks = -1;
    if (chunks < 0) {
        const char *e = getenv("S2D_RG_SORT_CHUNKS");
        chunks = e ? atoi(e) : S2D_XCDS;
        if (chunks != 1 && chunks != S2D_XCDS) chunks = S2D_XCDS;
    }
    if (chunks == 1) return n;
    if (const char *e = getenv("S2D_RG_SORT_CHUNK_ROWS")) {   // A/B: smaller chunks (more locality, less grouping)
        const int64_t rows = atoll(e);
        if (rows >= 16) return rows;
    }
    const RgPlan p = rg_plan(n, 27, 64, 64);
    return (int64_t)(xcd_grid(p.grid) / S2D_XCDS) * p.tiles_per_block * 16;
}

extern "C" int s2d_rulebook_sort_by_mask(const int32_t *nbr, int kvol, int64_t n, int32_t *perm, uint32_t *pmask, int32_t *nbr_perm, void *ws,
                                         size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(kvol > 0 && kvol <= 27 && n >= 0 && n < 0x7fffffff, "rulebook_sort_by_mask: bad sizes (kvol %d, n %lld)", kvol, (long long)n);
    if (n == 0) return S2D_OK;
    S2D_CHECK_ARG(nbr && perm && pmask && nbr_perm && ws, "rulebook_sort_by_mask: null argument");
    S2D_CHECK_ARG(ws_bytes >= s2d_rulebook_sort_workspace_bytes(n), "rulebook_sort_by_mask: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Carver c(ws);
    uint64_t *key = c.take<uint64_t>((size_t)n), *skey = c.take<uint64_t>((size_t)n);
    int32_t *iota = c.take<int32_t>((size_t)n);
    size_t cub_bytes = rbsort_cub_bytes(n);
    char *cub = c.take<char>(cub_bytes);
    const unsigned blocks = (unsigned)ceil_div(n, 256);
    const int64_t chunk_rows = s2d_rulebook_sort_chunk_rows(n);
    int bits = 27;
    while (((int64_t)1 << (bits - 27)) < ceil_div(n, chunk_rows)) ++bits;   // chunk-number bits above the 27 mask bits
    hipLaunchKernelGGL(rbsort_mask_kernel, dim3(blocks), dim3(256), 0, st, nbr, kvol, (int)n, (int)chunk_rows, key, iota);
    S2D_LAUNCH_CHECK();
    S2D_HIP(hipcub::DeviceRadixSort::SortPairs(cub, cub_bytes, (const uint64_t *)key, skey, (const int32_t *)iota, perm, (int)n, 0, bits, st));
    hipLaunchKernelGGL(rbsort_permute_kernel, dim3(blocks), dim3(256), 0, st, nbr, perm, kvol, (int)n, nbr_perm, (const uint64_t *)skey, pmask);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
