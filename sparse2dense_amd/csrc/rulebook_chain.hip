// Every rulebook of one backbone pass (SubM at each resolution + the strided convs between them) from the stage-0 coordinates,
// in a handful of launches around ONE host read (r04).  Replaces the per-layer builds of rulebook.hip on the backbone path
// (spconv get_indice_pairs as called by det3d/models/backbones/scn.py:104-152); the per-layer entries stay for stand-alone layers.
//
// Why: the per-layer design indexed the stage-0 grid (371 M cells at batch 4) with a 93 MB bitmap that every build memset, scanned in
// three launches and probed with one 128-byte HBM line per (z, y) neighbour line: 8 builds = 50 launches, 0.82 ms, 15x the
// algorithmic traffic (profiles/r03_final_pmc_traffic.json).  Here
//   * the index is HIERARCHICAL / block-sparse: the stage-1 occupancy index (1 bit per cell + rank prefix per 32 cells, 12 MB)
//     doubles as the coarse level of stage 0: a stage-0 cell p lives in the 2x2x2 block floor(p/2), and with the backbone's first
//     strided conv (k3, s2, p1) that block IS an active stage-1 cell.  The fine level is a child table int32[N1][8] (row id of each
//     of the 8 cells of an occupied block, 8 MB) - nothing of the size of the stage-0 grid exists.
//   * stage 1 is marked from the stage-0 rows (<= 8 atomics per row); every later stage is marked WITHOUT atomics by a gather over
//     the previous bitmap (a thread owns a 32-cell output word: dilate + down-sample of 64 input bits per neighbour line);
//     all bitmaps are numbered by ONE single-pass multi-segment scan (decoupled look-back on 8-byte {tag, value} granules,
//     MI355X_MICROARCH.md "R2"), and only the counts cross to the host - once per pass, not once per layer.
//   * with the row counts known, ONE probe launch writes every gather map: a thread owns a row of stage l and fills SubM_l, the
//     input-side map of conv l->l+1 and the output-side map of conv l-1->l.  Index words are loaded in unconditional batches
//     (clamped addresses, masks applied afterwards), every store is coalesced along rows, nothing is memset, no map is scattered.
//     Rows of stage >= 1 are in canonical (b,z,y,x) order (rank = row): neighbouring threads probe neighbouring index words.
// Index layout: every (b,z,y) line is padded to whole 32-cell words (ranks are unchanged: padding bits are never set), so the
// x-1 / x / x+1 probes of a line touch at most two adjacent words and the gather marks work on whole words.
#include "s2d_common.h"
#include "scan.h"
#include <cstdlib>

namespace s2d {
namespace chain {

constexpr int MAX_ST = 6;   // resolutions incl. stage 0
constexpr int SC_THREADS = 256;
constexpr int SC_WORDS = 16;                        // index words per thread
constexpr int SC_TILE = SC_THREADS * SC_WORDS;      // 4096 words = 131 072 cells per workgroup
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct Stage {
    int shape[3];
    int wpl;                   // index words per (b,z,y) line = ceil(W / 32)
    int n;                     // rows (known from stage 1 on only in the fill phase)
    const int32_t *coors;      // [n][4]
    uint2 *occ;                // dense rank index {bits, exclusive rank prefix} per 32 cells (stages >= 1)
    long long words;           // lines * wpl, padded to a multiple of 2
    int32_t *subm_nbr;         // [27][n] or null
    int32_t *subm_cnt;         // [27]
};
struct Conv {
    int ks[3], st[3], pad[3];
    int kvol;
    int full3s2;               // k3 s2 on every axis (any padding): the batched input-side probe applies
    int32_t *nbr_out, *nbr_in, *cnt;
};
struct Params {
    int batch, nst, mark_mode;
    long long n0;
    Stage s[MAX_ST];
    Conv c[MAX_ST - 1];        // c[l]: stage l -> l+1
    int32_t *child;            // [s[1].n][8] stage-0 row per cell of an occupied 2x2x2 block
    int tile_start[MAX_ST + 1];   // scan (or decode) tiles of stage l: [tile_start[l], tile_start[l+1]), l >= 1
    int blk_start[MAX_ST + 1];    // probe blocks of stage l
    unsigned long long *flags;    // one granule per scan tile, zeroed by the plan's memset
    int32_t *counts;              // [nst-1] rows of stages 1.. ; [nst-1] = error flag
    int32_t *zero;                // pair counters to clear (fill phase)
    int zero_n;
};

__device__ __forceinline__ long long line_of(int b, int z, int y, const Stage &S) { return ((long long)b * S.shape[0] + z) * S.shape[1] + y; }
__device__ __forceinline__ int rank_bit(const uint2 e, int bit) {
    return (e.x >> bit) & 1 ? (int)(e.y + __popc(e.x & ((1u << bit) - 1))) : -1;
}
__device__ __forceinline__ int rank_of(const Stage &S, int b, int z, int y, int x) {
    return rank_bit(S.occ[line_of(b, z, y, S) * S.wpl + (x >> 5)], x & 31);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ bool row_valid(const int4 c, int batch, const int *shape) {
    return (unsigned)c.x < (unsigned)batch && (unsigned)c.y < (unsigned)shape[0] && (unsigned)c.z < (unsigned)shape[1] &&
           (unsigned)c.w < (unsigned)shape[2];
}

// ---- 1a. mark stage 1 from the stage-0 rows ----------------------------------------------------------------------------------
// outputs of a conv axis (ks, st, pad, dilation 1) reachable from input p: o*st - pad + k = p for some k in [0, ks)
// <=> ceil((p + pad - ks + 1) / st) <= o <= floor((p + pad) / st)
// LDS-staged: the <= 8 candidate (index word, bit) pairs of the block's 256 rows are first merged in a 2048-slot LDS hash table keyed
// by the word (neighbouring rows hit the same 32-cell words), then every occupied slot issues ONE global atomicOr - the chip
// retires ~21 global atomics per ns, which bounded the direct version (1.0 M atomics, 47 us at 289 k rows).
constexpr int MK_SLOTS = 2048;
__global__ __launch_bounds__(256) void chain_mark_rows_kernel(const Params P) {
    __shared__ unsigned keys[MK_SLOTS], bitsv[MK_SLOTS];
    for (int t = threadIdx.x; t < MK_SLOTS; t += 256) { keys[t] = 0xFFFFFFFFu; bitsv[t] = 0u; }
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const Conv &cv = P.c[0];
    const Stage &so = P.s[1];
    int4 c = make_int4(0, 0, 0, 0);
    if (i < P.n0) c = reinterpret_cast<const int4 *>(P.s[0].coors)[i];
    if (i < P.n0 && row_valid(c, P.batch, P.s[0].shape)) {
        auto lo_of = [](int p, int pad, int ks, int st) { const int t = p + pad - ks + 1; return t <= 0 ? 0 : (t + st - 1) / st; };
        const int a0 = lo_of(c.y, cv.pad[0], cv.ks[0], cv.st[0]), b0 = min((c.y + cv.pad[0]) / cv.st[0], so.shape[0] - 1);
        const int a1 = lo_of(c.z, cv.pad[1], cv.ks[1], cv.st[1]), b1 = min((c.z + cv.pad[1]) / cv.st[1], so.shape[1] - 1);
        const int a2 = lo_of(c.w, cv.pad[2], cv.ks[2], cv.st[2]), b2 = min((c.w + cv.pad[2]) / cv.st[2], so.shape[2] - 1);
        for (int z = a0; z <= b0; ++z)
            for (int y = a1; y <= b1; ++y)
                for (int x = a2; x <= b2; ++x) {
                    const unsigned key = (unsigned)(line_of(c.x, z, y, so) * so.wpl + (x >> 5));   // < 6e7 words (make_plan)
                    const unsigned bit = 1u << (x & 31);
                    if (P.mark_mode == 1) { atomicOr(&so.occ[key].x, bit); continue; }
                    unsigned h = (key * 2654435761u) >> 21;
                    bool done = false;
                    for (int probe = 0; probe < 16; ++probe) {
                        const unsigned old = atomicCAS(&keys[h], 0xFFFFFFFFu, key);
                        if (old == 0xFFFFFFFFu || old == key) { atomicOr(&bitsv[h], bit); done = true; break; }
                        h = (h + 1) & (MK_SLOTS - 1);
                    }
                    if (!done) atomicOr(&so.occ[key].x, bit);   // crowded neighbourhood of the table: go direct
                }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < MK_SLOTS; t += 256)
        if (keys[t] != 0xFFFFFFFFu) atomicOr(&so.occ[keys[t]].x, bitsv[t]);
}

// ---- 1b. mark stage l+1 from the bitmap of stage l >= 1: no atomics ------------------------------------------------------------
// a thread owns one 32-cell output word of line (b, zo, yo); x axis either (k3, s2, p1): output bit j = OR of input bits 2j-1, 2j,
// 2j+1 of the 64 input cells [64 xw, 64 xw + 64) (+ the last bit of the word before), or (k1, s1, p0): a copy
__device__ __forceinline__ uint32_t even_bits(unsigned long long x) {
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)x;
}
__global__ __launch_bounds__(256) void chain_mark_gather_kernel(const Params P, int l) {
    const Stage &Si = P.s[l];
    const Stage &So = P.s[l + 1];
    const Conv &cv = P.c[l];
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= So.words) return;
    const long long lines = (long long)P.batch * So.shape[0] * So.shape[1];
    const long long line = w / So.wpl;
    const int xw = (int)(w - line * So.wpl);
    uint32_t bits = 0;
    if (line < lines) {
        const int yo = (int)(line % So.shape[1]);
        const long long t = line / So.shape[1];
        const int zo = (int)(t % So.shape[0]), b = (int)(t / So.shape[0]);
        const int z0 = zo * cv.st[0] - cv.pad[0], y0 = yo * cv.st[1] - cv.pad[1];
        const bool copy = cv.ks[2] == 1;
        // the three input words of a neighbour line, loaded for all (up to 9) lines before any is used: clamped addresses, masked values
        const int wa = clampi(copy ? xw : 2 * xw - 1, 0, Si.wpl - 1), wb = clampi(copy ? xw : 2 * xw, 0, Si.wpl - 1),
                  wc = clampi(copy ? xw : 2 * xw + 1, 0, Si.wpl - 1);
        uint32_t A[9], B[9], C[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const uint2 *in = Si.occ + line_of(b, clampi(z0 + q / 3, 0, Si.shape[0] - 1), clampi(y0 + q % 3, 0, Si.shape[1] - 1), Si) * Si.wpl;
            A[q] = in[wa].x; B[q] = in[wb].x; C[q] = in[wc].x;
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int kz = q / 3, ky = q % 3;
            if (kz >= cv.ks[0] || ky >= cv.ks[1] || (unsigned)(z0 + kz) >= (unsigned)Si.shape[0] || (unsigned)(y0 + ky) >= (unsigned)Si.shape[1]) continue;
            if (copy) {
                if (xw < Si.wpl) bits |= B[q];
            } else {
                const int w0 = 2 * xw;
                const unsigned long long lo = w0 < Si.wpl ? B[q] : 0u, hi = w0 + 1 < Si.wpl ? C[q] : 0u;
                const unsigned long long prev = (w0 > 0 && w0 - 1 < Si.wpl) ? A[q] >> 31 : 0u;
                const unsigned long long I = hi << 32 | lo;
                bits |= even_bits(I | (I >> 1) | (I << 1) | prev);
            }
        }
        const int xmax = So.shape[2] - 32 * xw;   // cells of this word inside the grid
        if (xmax < 32) bits &= xmax <= 0 ? 0u : (1u << xmax) - 1u;
    }
    So.occ[w].x = bits;
}

// ---- 2. scan: rank prefixes of every stage's bitmap in one single-pass launch ------------------------------------------------
// A workgroup owns 4096 words of one stage, publishes its popcount as an 8-byte {1, count} granule and sums the granules of ALL
// earlier tiles of its stage (they are published before any look-back, so there is no serial chain; <= 2 polls per thread at the
// 370 tiles of the stage-1 grid).  Earlier tiles are dispatched earlier: a resident workgroup never waits on an undispatched one.
__global__ __launch_bounds__(SC_THREADS) void chain_scan_kernel(const Params P) {
    __shared__ int lds[4];
    int seg = 1;
    while (seg + 1 < P.nst && (int)blockIdx.x >= P.tile_start[seg + 1]) ++seg;
    const int tile = (int)blockIdx.x - P.tile_start[seg];
    uint2 *occ = P.s[seg].occ;
    const long long words = P.s[seg].words;
    const long long base = (long long)tile * SC_TILE + (long long)threadIdx.x * SC_WORDS;
    uint4 v[SC_WORDS / 2];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SC_WORDS / 2; ++j) {
        const long long w = base + 2 * j;
        v[j] = w < words ? *reinterpret_cast<const uint4 *>(occ + w) : make_uint4(0u, 0u, 0u, 0u);
        s += __popc(v[j].x) + __popc(v[j].z);
    }
    int tot;
    const int ex = block_exclusive_scan(s, &tot, lds);
    unsigned long long *flags = P.flags + P.tile_start[seg];
    if (threadIdx.x == 0) __hip_atomic_store(&flags[tile], (1ull << 32) | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int before = 0;
    for (int t = threadIdx.x; t < tile; t += SC_THREADS) {
        unsigned spins = 0;
        unsigned long long f;
        while (!((f = __hip_atomic_load(&flags[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) {   // never observed; bounded so that a broken dispatch-order assumption cannot hang the device
                atomicExch(&P.counts[P.nst - 1], 1);
                break;
            }
        }
        before += (int)(unsigned)f;
    }
    int prev;
    block_exclusive_scan(before, &prev, lds);
    if (tile == P.tile_start[seg + 1] - P.tile_start[seg] - 1 && threadIdx.x == 0) P.counts[seg - 1] = prev + tot;
    int run = prev + ex;
#pragma unroll
    for (int j = 0; j < SC_WORDS / 2; ++j) {
        const long long w = base + 2 * j;
        const int p0 = __popc(v[j].x);
        if (v[j].x | v[j].z) {   // prefixes of empty words are never read
            v[j].y = (uint32_t)run;
            v[j].w = (uint32_t)(run + p0);
            *reinterpret_cast<uint4 *>(occ + w) = v[j];
        }
        run += p0 + __popc(v[j].z);
    }
}

// ---- 3. decode: coordinates of the rows of stages >= 1 (rank order) + cleared child rows -----------------------------------
__global__ __launch_bounds__(256) void chain_decode_kernel(const Params P) {
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < P.zero_n; t += blockDim.x) P.zero[t] = 0;
    int seg = 1;
    while (seg + 1 < P.nst && (int)blockIdx.x >= P.tile_start[seg + 1]) ++seg;   // tile_start: decode tiles of 256 words here
    const long long w = (long long)((int)blockIdx.x - P.tile_start[seg]) * 256 + threadIdx.x;
    const Stage &S = P.s[seg];
    if (w >= S.words) return;
    const uint2 e = S.occ[w];
    uint32_t bits = e.x;
    if (!bits) return;
    int r = (int)e.y;
    int4 *out = reinterpret_cast<int4 *>(const_cast<int32_t *>(S.coors));
    int4 *child = reinterpret_cast<int4 *>(P.child);
    const long long line = w / S.wpl;
    int4 c;
    c.w = (int)(w - line * S.wpl) * 32;
    c.z = (int)(line % S.shape[1]);
    const long long t = line / S.shape[1];
    c.y = (int)(t % S.shape[0]);
    c.x = (int)(t / S.shape[0]);
    const int x0 = c.w;
    while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        c.w = x0 + b;
        if (r < S.n) {   // r >= n only if the host's counts were stale: never write past the caller's buffers
            out[r] = c;
            if (seg == 1) {
                child[2 * (long long)r] = make_int4(-1, -1, -1, -1);
                child[2 * (long long)r + 1] = make_int4(-1, -1, -1, -1);
            }
        }
        ++r;
    }
}

// ---- 4. child table: stage-0 row of each cell of an occupied 2x2x2 block ----------------------------------------------------
__global__ __launch_bounds__(256) void chain_child_kernel(const Params P) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n0) return;
    const int4 c = reinterpret_cast<const int4 *>(P.s[0].coors)[i];
    if (!row_valid(c, P.batch, P.s[0].shape)) return;
    const int r1 = rank_of(P.s[1], c.x, c.y >> 1, c.z >> 1, c.w >> 1);
    if (r1 < 0 || r1 >= P.s[1].n) return;   // cannot happen: the block of an active cell is an active stage-1 cell
    // plain store (289 k atomics cost 15 us): rows with a DUPLICATE coordinate - which the voxelizer never emits and spconv leaves
    // undefined - race for the cell, one of them wins
    P.child[(long long)r1 * 8 + ((c.y & 1) << 2 | (c.z & 1) << 1 | (c.w & 1))] = (int32_t)i;
}

// ---- 5. probe: every gather map -------------------------------------------------------------------------------------------
// the (at most two adjacent) index words of one line that hold the cells xa <= xb <= xa + 2, loaded without a condition
struct LineW {
    uint2 a, b;
};
__device__ __forceinline__ LineW load_line(const Stage &S, int b, int z, int y, int xa, int xb) {
    const uint2 *p = S.occ + line_of(b, z, y, S) * S.wpl;
    LineW L;
    L.a = p[xa >> 5];
    L.b = p[xb >> 5];
    return L;
}
__device__ __forceinline__ int rank_in(const LineW &L, int xa, int x) { return rank_bit((x >> 5) == (xa >> 5) ? L.a : L.b, x & 31); }

__device__ __forceinline__ void count_wave(int *cnt, bool hit) {
    const unsigned long long m = __ballot(hit);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(cnt, __popcll(m));
}

// ranks of the 27 cells around (z, y, x) in the dense index of S (stages >= 1); c must be a valid cell (callers substitute 0,0,0,0)
__device__ __forceinline__ void ranks27(const Stage &S, const int4 c, bool valid, int *r) {
    LineW L[9];
    const int xa = max(c.w - 1, 0), xb = min(c.w + 1, S.shape[2] - 1);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int z = clampi(c.y + q / 3 - 1, 0, S.shape[0] - 1), y = clampi(c.z + q % 3 - 1, 0, S.shape[1] - 1);
        L[q] = load_line(S, c.x, z, y, xa, xb);
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int z = c.y + q / 3 - 1, y = c.z + q % 3 - 1;
        const bool vl = valid && (unsigned)z < (unsigned)S.shape[0] && (unsigned)y < (unsigned)S.shape[1];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int x = c.w + dx - 1;
            r[q * 3 + dx] = (vl && (unsigned)x < (unsigned)S.shape[2]) ? rank_in(L[q], xa, clampi(x, 0, S.shape[2] - 1)) : -1;
        }
    }
}

// SubM 3x3x3 at stage 0 through the stage-1 index + child table: the 27 neighbours of a cell lie in 2x2x2 blocks.  Along an axis the
// neighbours p-1, p, p+1 of an EVEN p are (block 0, upper child), (block 1, lower), (block 1, upper) of the two blocks
// floor((p-1)/2), floor((p-1)/2)+1; of an ODD p: (0, lower), (0, upper), (1, lower).  All register indices are static; the parity
// picks among 8 candidates per output.
__device__ __forceinline__ void subm_child(const int4 c, bool valid, long long i, long long n, const Stage &S0, const Stage &S1,
                                           const int32_t *__restrict__ child, int *cnt) {
    const int Bz = (c.y - 1) >> 1, By = (c.z - 1) >> 1, Bx = (c.w - 1) >> 1;   // arithmetic shifts: -1 in front of the grid
    const int4 *child4 = reinterpret_cast<const int4 *>(child);
    int r1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int zb = Bz + (q >> 2), yb = By + ((q >> 1) & 1), xb = Bx + (q & 1);
        const bool vb = valid && (unsigned)zb < (unsigned)S1.shape[0] && (unsigned)yb < (unsigned)S1.shape[1] && (unsigned)xb < (unsigned)S1.shape[2];
        const int r = rank_of(S1, c.x, clampi(zb, 0, S1.shape[0] - 1), clampi(yb, 0, S1.shape[1] - 1), clampi(xb, 0, S1.shape[2] - 1));
        r1[q] = (vb && r < S1.n) ? r : -1;
    }
    int V[8][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const long long rr = r1[q] < 0 ? 0 : r1[q];
        const int4 lo = child4[2 * rr], hi = child4[2 * rr + 1];
        const bool ok = r1[q] >= 0;
        V[q][0] = ok ? lo.x : -1; V[q][1] = ok ? lo.y : -1; V[q][2] = ok ? lo.z : -1; V[q][3] = ok ? lo.w : -1;
        V[q][4] = ok ? hi.x : -1; V[q][5] = ok ? hi.y : -1; V[q][6] = ok ? hi.z : -1; V[q][7] = ok ? hi.w : -1;
    }
    const int pz = c.y & 1, py = c.z & 1, px = c.w & 1;
    // (block, child) of neighbour d in {0,1,2} (= -1, 0, +1) for parity p: even -> (0,1) (1,0) (1,1); odd -> (0,0) (0,1) (1,0)
#define S2D_BSEL(p, d) ((p) ? ((d) == 2 ? 1 : 0) : ((d) == 0 ? 0 : 1))
#define S2D_CSEL(p, d) ((p) ? ((d) == 1 ? 1 : 0) : ((d) == 1 ? 0 : 1))
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int dz = k / 9, dy = (k / 3) % 3, dx = k % 3;
        int v = -1;
#pragma unroll
        for (int par = 0; par < 8; ++par) {
            const int qz = par >> 2, qy = (par >> 1) & 1, qx = par & 1;
            const int blk = S2D_BSEL(qz, dz) * 4 + S2D_BSEL(qy, dy) * 2 + S2D_BSEL(qx, dx);
            const int ch = S2D_CSEL(qz, dz) * 4 + S2D_CSEL(qy, dy) * 2 + S2D_CSEL(qx, dx);
            v = (pz == qz && py == qy && px == qx) ? V[blk][ch] : v;
        }
        if (!valid) v = -1;
        if (i < n) S0.subm_nbr[(long long)k * n + i] = v;
        count_wave(&cnt[k], v >= 0);
    }
#undef S2D_BSEL
#undef S2D_CSEL
}

// input-side map of a k3 s2 conv l -> l+1 for an input row of stage l: along an axis n = p + pad is reached by k = n & 1 (output
// (n - k) / 2) and, for even n, also by k = 2 (output n / 2 - 1): 2 x 2 line slots x 2 x-slots = 8 ranks serve the 27 offsets
__device__ __forceinline__ void conv_in_k3s2(const int4 c, bool valid, long long i, long long n, const Conv &cv, const Stage &So, int *cnt) {
    const int nz = c.y + cv.pad[0], ny = c.z + cv.pad[1], nx = c.w + cv.pad[2];
    // slot 0: k = n & 1, o = (n - k) / 2 = n >> 1; slot 1: k = 2 (even n only), o = (n >> 1) - 1
    const int oz[2] = {nz >> 1, (nz >> 1) - 1}, oy[2] = {ny >> 1, (ny >> 1) - 1}, ox[2] = {nx >> 1, (nx >> 1) - 1};
    const bool vz[2] = {oz[0] < So.shape[0], !(nz & 1) && oz[1] >= 0 && oz[1] < So.shape[0]};
    const bool vy[2] = {oy[0] < So.shape[1], !(ny & 1) && oy[1] >= 0 && oy[1] < So.shape[1]};
    const bool vx[2] = {ox[0] < So.shape[2], !(nx & 1) && ox[1] >= 0 && ox[1] < So.shape[2]};
    const int xa = clampi(ox[1], 0, So.shape[2] - 1), xb = clampi(ox[0], 0, So.shape[2] - 1);
    LineW L[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        L[q] = load_line(So, c.x, clampi(oz[q >> 1], 0, So.shape[0] - 1), clampi(oy[q & 1], 0, So.shape[1] - 1), xa, xb);
    int R[2][2][2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int sz = q >> 2, sy = (q >> 1) & 1, sx = q & 1;
        int r = -1;
        if (valid && vz[sz] && vy[sy] && vx[sx]) r = rank_in(L[sz * 2 + sy], xa, ox[sx]);
        R[sz][sy][sx] = r < So.n ? r : -1;
    }
    const int pz = nz & 1, py = ny & 1, px = nx & 1;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int kz = k / 9, ky = (k / 3) % 3, kx = k % 3;
        // offset 0 / 1 are slot 0 for even / odd n, offset 2 is slot 1
        const bool ok = (kz == 2 ? !pz : kz == pz) && (ky == 2 ? !py : ky == py) && (kx == 2 ? !px : kx == px);
        const int o = ok ? R[kz == 2][ky == 2][kx == 2] : -1;
        if (i < n) cv.nbr_in[(long long)k * n + i] = o;
        count_wave(&cnt[k], o >= 0);
    }
}

// input-side map, any geometry (the (3,1,1) / (2,1,1) conv in front of the BEV map): one lookup per offset
__device__ __forceinline__ void conv_in_generic(const int4 c, bool valid, long long i, long long n, const Conv &cv, const Stage &So, int *cnt) {
    int k = 0;
    for (int kz = 0; kz < cv.ks[0]; ++kz) {
        const int nz = c.y + cv.pad[0] - kz, oz = nz / cv.st[0];
        const bool vz = valid && nz >= 0 && oz * cv.st[0] == nz && oz < So.shape[0];
        for (int ky = 0; ky < cv.ks[1]; ++ky) {
            const int ny = c.z + cv.pad[1] - ky, oy = ny / cv.st[1];
            const bool vy = vz && ny >= 0 && oy * cv.st[1] == ny && oy < So.shape[1];
            for (int kx = 0; kx < cv.ks[2]; ++kx, ++k) {
                const int nx = c.w + cv.pad[2] - kx, ox = nx / cv.st[2];
                const bool vx = vy && nx >= 0 && ox * cv.st[2] == nx && ox < So.shape[2];
                int o = -1;
                if (vx) o = rank_of(So, c.x, oz, oy, ox);
                if (o >= So.n) o = -1;
                if (i < n) cv.nbr_in[(long long)k * n + i] = o;
                count_wave(&cnt[k], o >= 0);
            }
        }
    }
}

// output-side map of a conv with a 3 x 3 x 3 kernel (any stride / padding) for an output row of stage l >= 2: nine lines of the
// stage-(l-1) index, three consecutive cells each
__device__ __forceinline__ void conv_out_k3(const int4 c, long long i, long long n, const Conv &cv, const Stage &Si) {
    const int z0 = c.y * cv.st[0] - cv.pad[0], y0 = c.z * cv.st[1] - cv.pad[1], x0 = c.w * cv.st[2] - cv.pad[2];
    const int xa = clampi(x0, 0, Si.shape[2] - 1), xb = clampi(x0 + 2, 0, Si.shape[2] - 1);
    LineW L[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
        L[q] = load_line(Si, c.x, clampi(z0 + q / 3, 0, Si.shape[0] - 1), clampi(y0 + q % 3, 0, Si.shape[1] - 1), xa, xb);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const bool vl = (unsigned)(z0 + q / 3) < (unsigned)Si.shape[0] && (unsigned)(y0 + q % 3) < (unsigned)Si.shape[1];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int x = x0 + kx;
            int r = (vl && (unsigned)x < (unsigned)Si.shape[2]) ? rank_in(L[q], xa, clampi(x, 0, Si.shape[2] - 1)) : -1;
            cv.nbr_out[(long long)(q * 3 + kx) * n + i] = r < Si.n ? r : -1;
        }
    }
}

__device__ __forceinline__ void conv_out_generic(const int4 c, long long i, long long n, const Conv &cv, const Stage &Si) {
    int k = 0;
    for (int kz = 0; kz < cv.ks[0]; ++kz) {
        const int z = c.y * cv.st[0] - cv.pad[0] + kz;
        for (int ky = 0; ky < cv.ks[1]; ++ky) {
            const int y = c.z * cv.st[1] - cv.pad[1] + ky;
            const bool vl = (unsigned)z < (unsigned)Si.shape[0] && (unsigned)y < (unsigned)Si.shape[1];
            for (int kx = 0; kx < cv.ks[2]; ++kx, ++k) {
                const int x = c.w * cv.st[2] - cv.pad[2] + kx;
                int r = -1;
                if (vl && (unsigned)x < (unsigned)Si.shape[2]) r = rank_of(Si, c.x, z, y, x);
                cv.nbr_out[(long long)k * n + i] = r < Si.n ? r : -1;
            }
        }
    }
}

// output-side map of conv 0 -> 1 (k3 s2 p1) for a stage-1 row o: input cell 2o-1+k lies in block o-1 (k = 0, its upper child) or in
// block o (k = 1, 2: lower / upper child).  The 8 blocks (o-1 | o)^3 are 8 of the row's 27 SubM neighbours: r27 holds their ranks.
__device__ __forceinline__ void conv_out_child(long long i, long long n, const Conv &cv, const int *r27, const int32_t *__restrict__ child) {
    const int4 *child4 = reinterpret_cast<const int4 *>(child);
    int V[8][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int sz = q >> 2, sy = (q >> 1) & 1, sx = q & 1;   // 0: block o-1, 1: block o
        const int r = r27[sz * 9 + sy * 3 + sx];
        const long long rr = r < 0 ? 0 : r;
        const int4 lo = child4[2 * rr], hi = child4[2 * rr + 1];
        const bool ok = r >= 0;
        V[q][0] = ok ? lo.x : -1; V[q][1] = ok ? lo.y : -1; V[q][2] = ok ? lo.z : -1; V[q][3] = ok ? lo.w : -1;
        V[q][4] = ok ? hi.x : -1; V[q][5] = ok ? hi.y : -1; V[q][6] = ok ? hi.z : -1; V[q][7] = ok ? hi.w : -1;
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int kz = k / 9, ky = (k / 3) % 3, kx = k % 3;
        const int blk = (kz > 0) * 4 + (ky > 0) * 2 + (kx > 0);
        const int ch = (kz != 1) * 4 + (ky != 1) * 2 + (kx != 1);
        cv.nbr_out[(long long)k * n + i] = V[blk][ch];
    }
}

__global__ __launch_bounds__(256) void chain_probe_kernel(const Params P) {
    __shared__ int cnt_s[27], cnt_c[27];
    if (threadIdx.x < 27) cnt_s[threadIdx.x] = cnt_c[threadIdx.x] = 0;
    __syncthreads();
    int l = 0;
    while (l + 1 < P.nst && (int)blockIdx.x >= P.blk_start[l + 1]) ++l;
    const Stage &S = P.s[l];
    const long long n = S.n;
    const long long i = (long long)((int)blockIdx.x - P.blk_start[l]) * 256 + threadIdx.x;
    int4 c = make_int4(0, 0, 0, 0);
    if (i < n) c = reinterpret_cast<const int4 *>(S.coors)[i];
    const bool valid = i < n && row_valid(c, P.batch, S.shape);
    if (!valid) c = make_int4(0, 0, 0, 0);   // addresses stay inside the index; results are masked
    if (l == 0) {
        if (S.subm_nbr) subm_child(c, valid, i, n, S, P.s[1], P.child, cnt_s);
    } else if (S.subm_nbr || l == 1) {
        int r27[27];
        ranks27(S, c, valid, r27);
        if (S.subm_nbr) {
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                if (i < n) S.subm_nbr[(long long)k * n + i] = r27[k];
                count_wave(&cnt_s[k], r27[k] >= 0);
            }
        }
        if (l == 1 && i < n) conv_out_child(i, n, P.c[0], r27, P.child);
    }
    if (l >= 2 && i < n) {
        if (P.c[l - 1].kvol == 27) conv_out_k3(c, i, n, P.c[l - 1], P.s[l - 1]);
        else conv_out_generic(c, i, n, P.c[l - 1], P.s[l - 1]);
    }
    if (l + 1 < P.nst) {
        if (P.c[l].full3s2) conv_in_k3s2(c, valid, i, n, P.c[l], P.s[l + 1], cnt_c);
        else conv_in_generic(c, valid, i, n, P.c[l], P.s[l + 1], cnt_c);
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        if (S.subm_nbr && cnt_s[threadIdx.x]) atomicAdd(&S.subm_cnt[threadIdx.x], cnt_s[threadIdx.x]);
        if (l + 1 < P.nst && threadIdx.x < P.c[l].kvol && cnt_c[threadIdx.x]) atomicAdd(&P.c[l].cnt[threadIdx.x], cnt_c[threadIdx.x]);
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------------
struct Plan {
    Params p;
    char *zero_base;      // region the plan phase clears: stage-1 bitmap + flags
    size_t zero_bytes;
    size_t ws_bytes;
    int scan_tiles;
};

static int make_plan(Plan *pl, void *ws, int batch, const int32_t shape0[3], int n_strided, const int32_t *ksize, const int32_t *stride,
                     const int32_t *padding) {
    S2D_CHECK_ARG(batch > 0 && shape0 && ksize && stride && padding, "rulebook_chain: null geometry");
    if (n_strided < 1 || n_strided > MAX_ST - 1) {
        set_error("rulebook_chain: %d strided stages unsupported (1..%d)", n_strided, MAX_ST - 1);
        return S2D_ERR_UNSUPPORTED;
    }
    Params &p = pl->p;
    memset(&p, 0, sizeof(p));
    p.batch = batch;
    p.nst = n_strided + 1;
    static const int mark_mode = getenv("S2D_CHAIN_MARK") ? atoi(getenv("S2D_CHAIN_MARK")) : 0;
    p.mark_mode = mark_mode;
    for (int a = 0; a < 3; ++a) {
        S2D_CHECK_ARG(shape0[a] > 0, "rulebook_chain: non-positive extent");
        p.s[0].shape[a] = shape0[a];
    }
    for (int l = 0; l < n_strided; ++l) {
        Conv &c = p.c[l];
        c.kvol = 1;
        c.full3s2 = 1;
        for (int a = 0; a < 3; ++a) {
            c.ks[a] = ksize[3 * l + a]; c.st[a] = stride[3 * l + a]; c.pad[a] = padding[3 * l + a];
            if (c.ks[a] < 1 || c.ks[a] > 3 || c.st[a] < 1 || c.pad[a] < 0 || c.pad[a] >= c.ks[a]) {
                set_error("rulebook_chain: conv %d axis %d (k %d, s %d, p %d) unsupported", l, a, c.ks[a], c.st[a], c.pad[a]);
                return S2D_ERR_UNSUPPORTED;
            }
            const int e = p.s[l].shape[a] + 2 * c.pad[a] - (c.ks[a] - 1) - 1;
            if (e < 0) {
                set_error("rulebook_chain: kernel larger than padded input at conv %d", l);
                return S2D_ERR_UNSUPPORTED;
            }
            p.s[l + 1].shape[a] = e / c.st[a] + 1;
            c.kvol *= c.ks[a];
            if (c.ks[a] != 3 || c.st[a] != 2) c.full3s2 = 0;
        }
        if (l == 0 && !(c.full3s2 && c.pad[0] == 1 && c.pad[1] == 1 && c.pad[2] == 1)) {
            set_error("rulebook_chain: the first strided conv must be k3 s2 p1 (its outputs index the stage-0 blocks)");
            return S2D_ERR_UNSUPPORTED;
        }
        if (l >= 1 && !((c.ks[2] == 3 && c.st[2] == 2 && c.pad[2] == 1) || (c.ks[2] == 1 && c.st[2] == 1 && c.pad[2] == 0))) {
            set_error("rulebook_chain: conv %d: x axis must be k3 s2 p1 or k1 s1 p0 (word-wise marking)", l);
            return S2D_ERR_UNSUPPORTED;
        }
    }
    Carver cv(ws);
    int tiles = 0;
    for (int l = 1; l < p.nst; ++l) {
        p.s[l].wpl = (p.s[l].shape[2] + 31) / 32;
        const double w = (double)batch * p.s[l].shape[0] * p.s[l].shape[1] * p.s[l].wpl;
        if (w >= 6.0e7) {
            set_error("rulebook_chain: stage %d grid too large for the dense index", l);
            return S2D_ERR_UNSUPPORTED;
        }
        const long long words = ((long long)w + 1) / 2 * 2;
        p.s[l].words = words;
        p.tile_start[l] = tiles;
        tiles += (int)ceil_div(words, SC_TILE);
    }
    p.tile_start[p.nst] = tiles;
    pl->scan_tiles = tiles;
    p.s[1].occ = cv.take<uint2>((size_t)p.s[1].words);
    p.flags = cv.take<unsigned long long>((size_t)tiles);
    pl->zero_base = (char *)p.s[1].occ;
    pl->zero_bytes = cv.total();
    for (int l = 2; l < p.nst; ++l) p.s[l].occ = cv.take<uint2>((size_t)p.s[l].words);   // fully written by the gather marks
    pl->ws_bytes = cv.total();
    return 0;
}

}  // namespace chain
}  // namespace s2d

using namespace s2d;
using namespace s2d::chain;

extern "C" int s2d_rulebook_chain_supported(int batch, const int32_t shape0[3], int n_strided, const int32_t *ksize, const int32_t *stride,
                                            const int32_t *padding) {
    Plan pl;
    return make_plan(&pl, nullptr, batch, shape0, n_strided, ksize, stride, padding) == 0 ? 1 : 0;
}

extern "C" size_t s2d_rulebook_chain_workspace_bytes(int batch, const int32_t shape0[3], int n_strided, const int32_t *ksize,
                                                     const int32_t *stride, const int32_t *padding) {
    Plan pl;
    if (make_plan(&pl, nullptr, batch, shape0, n_strided, ksize, stride, padding)) return 0;
    return pl.ws_bytes;
}

extern "C" int s2d_rulebook_chain_plan(const int32_t *coors0, int64_t n0, int batch, const int32_t shape0[3], int n_strided,
                                       const int32_t *ksize, const int32_t *stride, const int32_t *padding, int32_t *counts, void *ws,
                                       size_t ws_bytes, s2d_stream_t stream) {
    Plan pl;
    int rc = make_plan(&pl, ws, batch, shape0, n_strided, ksize, stride, padding);
    if (rc) return rc;
    S2D_CHECK_ARG(n0 >= 0 && n0 < 0x7fffffff && counts && (n0 == 0 || coors0), "rulebook_chain_plan: bad argument");
    if (!ws || ws_bytes < pl.ws_bytes) {
        set_error("rulebook_chain_plan: workspace too small (%zu < %zu)", ws_bytes, pl.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    pl.p.n0 = n0;
    pl.p.s[0].coors = coors0;
    pl.p.counts = counts;
    S2D_HIP(hipMemsetAsync(pl.zero_base, 0, pl.zero_bytes, st));
    S2D_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)(n_strided + 1), st));
    if (n0 > 0) hipLaunchKernelGGL(chain_mark_rows_kernel, dim3((unsigned)ceil_div(n0, 256)), dim3(256), 0, st, pl.p);
    for (int l = 1; l + 1 < pl.p.nst; ++l)
        hipLaunchKernelGGL(chain_mark_gather_kernel, dim3((unsigned)ceil_div(pl.p.s[l + 1].words, 256)), dim3(256), 0, st, pl.p, l);
    hipLaunchKernelGGL(chain_scan_kernel, dim3((unsigned)pl.scan_tiles), dim3(SC_THREADS), 0, st, pl.p);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_rulebook_chain_fill(const int32_t *coors0, int64_t n0, int batch, const int32_t shape0[3], int n_strided,
                                       const int32_t *ksize, const int32_t *stride, const int32_t *padding, const int64_t *n_rows,
                                       int32_t *const *out_coors, int32_t *const *subm_nbr, int32_t *const *conv_nbr_out,
                                       int32_t *const *conv_nbr_in, int32_t *pair_counts, int32_t *child, void *ws, size_t ws_bytes,
                                       s2d_stream_t stream) {
    Plan pl;
    int rc = make_plan(&pl, ws, batch, shape0, n_strided, ksize, stride, padding);
    if (rc) return rc;
    S2D_CHECK_ARG(n0 >= 0 && n0 < 0x7fffffff && n_rows && out_coors && subm_nbr && conv_nbr_out && conv_nbr_in && pair_counts,
                  "rulebook_chain_fill: null argument");
    if (!ws || ws_bytes < pl.ws_bytes) {
        set_error("rulebook_chain_fill: workspace too small (%zu < %zu)", ws_bytes, pl.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    Params &p = pl.p;
    hipStream_t st = (hipStream_t)stream;
    p.n0 = n0;
    p.s[0].coors = coors0;
    p.s[0].n = (int)n0;
    p.child = child;
    int blocks = 0, dtiles = 0;
    for (int l = 0; l < p.nst; ++l) {
        if (l >= 1) {
            S2D_CHECK_ARG(n_rows[l - 1] >= 0 && n_rows[l - 1] < 0x7fffffff, "rulebook_chain_fill: bad row count at stage %d", l);
            p.s[l].n = (int)n_rows[l - 1];
            p.s[l].coors = out_coors[l - 1];
            S2D_CHECK_ARG(p.s[l].n == 0 || p.s[l].coors, "rulebook_chain_fill: null out_coors at stage %d", l);
        }
        p.s[l].subm_nbr = p.s[l].n > 0 ? subm_nbr[l] : nullptr;
        p.s[l].subm_cnt = pair_counts + 27 * l;
        if (l + 1 < p.nst) {
            p.c[l].nbr_in = conv_nbr_in[l];
            p.c[l].nbr_out = conv_nbr_out[l];
            p.c[l].cnt = pair_counts + 27 * (p.nst + l);
            S2D_CHECK_ARG(p.s[l].n == 0 || p.c[l].nbr_in, "rulebook_chain_fill: null nbr_in at conv %d", l);
            S2D_CHECK_ARG(n_rows[l] == 0 || p.c[l].nbr_out, "rulebook_chain_fill: null nbr_out at conv %d", l);
        }
        p.blk_start[l] = blocks;
        blocks += (int)ceil_div(p.s[l].n, 256);
    }
    p.blk_start[p.nst] = blocks;
    S2D_CHECK_ARG(p.s[1].n == 0 || child, "rulebook_chain_fill: null child table");
    p.zero = pair_counts;
    p.zero_n = 27 * (2 * p.nst - 1);
    // decode tiles of 256 words per stage (tile_start is re-purposed for this launch)
    Params pd = p;
    for (int l = 1; l < p.nst; ++l) {
        pd.tile_start[l] = dtiles;
        dtiles += (int)ceil_div(p.s[l].words, 256);
    }
    pd.tile_start[p.nst] = dtiles;
    hipLaunchKernelGGL(chain_decode_kernel, dim3((unsigned)dtiles), dim3(256), 0, st, pd);
    if (n0 > 0 && p.s[1].n > 0) hipLaunchKernelGGL(chain_child_kernel, dim3((unsigned)ceil_div(n0, 256)), dim3(256), 0, st, p);
    static const int split = getenv("S2D_CHAIN_SPLIT") ? atoi(getenv("S2D_CHAIN_SPLIT")) : 0;
    if (split) {   // profiling aid: one probe launch per stage (identical results)
        for (int l = 0; l < p.nst; ++l) {
            Params q = p;
            const int nb = p.blk_start[l + 1] - p.blk_start[l];
            for (int m = 0; m <= p.nst; ++m) q.blk_start[m] = m <= l ? 0 : (m == l + 1 ? nb : 0x7fffffff);
            if (nb > 0) hipLaunchKernelGGL(chain_probe_kernel, dim3((unsigned)nb), dim3(256), 0, st, q);
        }
    } else if (blocks > 0) {
        hipLaunchKernelGGL(chain_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    }
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
