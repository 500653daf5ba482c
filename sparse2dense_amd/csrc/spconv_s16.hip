// Sparse-conv implicit GEMM with bf16 feature storage ("s16" path): features, gathered rows and the
// output are bf16 [n][C] in HBM, weights a pre-packed bf16 LDS image, accumulation fp32 on
// v_mfma_f32_16x16x32_bf16.  Used for the forward (nbr = output->input map) and, with the
// transposed / mirrored weight image, for the data gradient
// (spconv.ops.indice_conv / indice_conv_backward; call sites det3d/models/backbones/scn.py:104-152).
//
// The GEMM's K axis is the concatenation over kernel offsets of the input channels.  One K-step is
// 64 K-elements: 64/CIN offsets when CIN < 64, one offset's 64-channel chunk otherwise.  A workgroup
// (4 waves) owns BM output rows x all COUT columns.  Per K-step the A tile [BM rows][64] is gathered
// row by row with global_load_lds (16 B per lane; a missing neighbour reads a zero page) and the B
// tile (64 x COUT, pre-packed in fragment order) is a linear global_load_lds copy; both are double
// buffered with one barrier per K-step.  The workgroup's slice of the gather map (all offsets x BM rows)
// is copied to LDS once up front, so a step's gathers hang off an LDS read, not a second global trip.  The A image
// is [row][8 parts of 16 B] with the part index XOR-swizzled by (row & 7) so that the 16-lane groups
// of a ds_read_b128 fragment read touch 16 distinct bank slots.  A table built in the prologue says which
// 16-row MFMA tiles have a neighbour at which K-step; waves skip the MFMAs of the others.  Nothing is stored to
// LDS with ds_write inside the K loop (see s16_fill_maps).
#include "s2d_common.h"
#include <cstdio>
#include <cstdlib>

namespace s2d {

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));

__host__ __device__ inline int s16_steps(int cin, int kvol) { return cin < 64 ? (kvol + 64 / cin - 1) / (64 / cin) : kvol * (cin / 64); }
// kernel-offset slots covered by the K-steps (>= kvol; the phantom slots read zero rows and zero weights)
__host__ __device__ inline int s16_kslots(int cin, int kvol) { return cin < 64 ? s16_steps(cin, kvol) * (64 / cin) : kvol; }

template <int CIN, int COUT, int BM>
struct S16Cfg {
    static constexpr int OPS = CIN < 64 ? 64 / CIN : 1;   // kernel offsets per K-step
    static constexpr int CPO = CIN > 64 ? CIN / 64 : 1;   // K-steps per kernel offset
    static constexpr int PPO = CIN < 64 ? CIN / 8 : 8;    // 16-byte parts per offset inside a K-step
    static constexpr int WN = COUT == 128 ? 2 : ((COUT == 64 && BM == 64) ? 2 : 1);
    static constexpr int WM = 4 / WN;
    static constexpr int MI = BM / WM / 16;               // 16-row MFMA tiles per wave
    static constexpr int NJ = COUT / WN / 16;             // 16-col MFMA tiles per wave
    static constexpr int A_BYTES = BM * 64 * 2;
    static constexpr int B_BYTES = 64 * COUT * 2;
    static constexpr int A_LOADS = BM * 8 / 256;          // 16-byte chunks per thread per K-step
    static constexpr int TILES = BM / 16;
    static constexpr size_t LDS_FIXED = 2 * (size_t)A_BYTES + 2 * (size_t)B_BYTES;
    // + the gather-map slice [kslots][BM] and the tile-skip table [kslots][TILES]
    static size_t lds_bytes(int kvol) { return LDS_FIXED + (size_t)s16_kslots(CIN, kvol) * (BM + TILES) * sizeof(int); }
    static_assert(MI >= 1 && NJ >= 1, "bad tiling");
};

// packed image: [step][h (2)][nt = cout/16][q (4)][c (16)][e (8)]   (one K-step = 64*cout bf16, contiguous)
//   K element kk = 32 h + 8 q + e ;  column co = wn*(cout/WN) + c*NJ + n  with nt = wn*NJ + n
// source w: [K][cin][cout] fp32, or [K][cout][cin] when transpose (data gradient), offsets mirrored when flip.
__device__ __forceinline__ void s16_pack_element(const float *__restrict__ w, int kvol, int cin, int cout, int wn_count, int transpose, int flip,
                                                 int64_t i, __bf16 *__restrict__ out) {
    const int steps = s16_steps(cin, kvol);
    const int64_t total = (int64_t)steps * 64 * cout;
    if (i >= total) return;
    const int ntile = cout / 16, nj = ntile / wn_count;
    int64_t r = i;
    const int e = r % 8; r /= 8;
    const int c = r % 16; r /= 16;
    const int q = r % 4; r /= 4;
    const int nt = r % ntile; r /= ntile;
    const int h = r % 2; r /= 2;
    const int step = (int)r;
    const int kk = 32 * h + 8 * q + e;
    int k, ch;
    if (cin < 64) {
        k = step * (64 / cin) + kk / cin;
        ch = kk % cin;
    } else {
        k = step / (cin / 64);
        ch = (step % (cin / 64)) * 64 + kk;
    }
    const int co = (nt / nj) * (cout / wn_count) + c * nj + (nt % nj);
    float v = 0.f;
    if (k < kvol) {
        const int ks = flip ? kvol - 1 - k : k;
        v = transpose ? w[((int64_t)ks * cout + co) * cin + ch] : w[((int64_t)ks * cin + ch) * cout + co];
    }
    out[i] = (__bf16)v;
}

__global__ __launch_bounds__(256) void s16_pack_kernel(const float *__restrict__ w, int kvol, int cin, int cout, int wn_count,
                                                       int transpose, int flip, __bf16 *__restrict__ out) {
    s16_pack_element(w, kvol, cin, cout, wn_count, transpose, flip, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, out);
}

// both operands of a layer in one launch: blockIdx.y = 0 the forward image [cin -> cout], 1 the data-gradient image [cout -> cin]
// (transposed, offsets mirrored when flip_d) - the weight changes once per optimizer step and both are needed in every training step
__global__ __launch_bounds__(256) void s16_pack_pair_kernel(const float *__restrict__ w, int kvol, int cin, int cout, int wn_f, int wn_d, int flip_d,
                                                            __bf16 *__restrict__ out_f, __bf16 *__restrict__ out_d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) s16_pack_element(w, kvol, cin, cout, wn_f, 0, 0, i, out_f);
    else s16_pack_element(w, kvol, cout, cin, wn_d, 1, flip_d, i, out_d);
}

template <int CIN, int COUT, int BM, int MI_, int NJ_>
__device__ __forceinline__ void s16_epilogue(f32x4s (&acc)[MI_][NJ_], const float *__restrict__ bias, __bf16 *__restrict__ out,
                                             int row0, int row_end, int wm, int wn, int r, int q, float *__restrict__ stats_partial,
                                             int tile, char *smem) {
    typedef S16Cfg<CIN, COUT, BM> C;
    static_assert(MI_ == C::MI && NJ_ == C::NJ, "tile shape");
    // epilogue: C/D layout row = 4*(lane>>4)+reg, col = lane&15 -> columns co_base + r*NJ + jn (NJ consecutive)
    // stats_partial (optional): per-workgroup (sum, sum of squares) of the STORED bf16 rows per output channel, [tile][2][COUT] - the
    // partial-sum rows the batch-norm finalize kernel consumes, so the BatchNorm1d behind this conv skips its statistics pass
    // (same scheme as the dense 3x3 kernel's epilogue; fixed summation order)
    const int co_base = wn * (COUT / C::WN);
    float bv[C::NJ], s1[C::NJ], s2[C::NJ];
#pragma unroll
    for (int jn = 0; jn < C::NJ; ++jn) {
        bv[jn] = bias ? bias[co_base + r * C::NJ + jn] : 0.f;
        s1[jn] = 0.f;
        s2[jn] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + wm * 16 * C::MI + 16 * i + 4 * q + reg;
            if (row < row_end) {
                __bf16 v[C::NJ];
#pragma unroll
                for (int jn = 0; jn < C::NJ; ++jn) {
                    v[jn] = (__bf16)(acc[i][jn][reg] + bv[jn]);
                    const float f = (float)v[jn];
                    s1[jn] += f;
                    s2[jn] += f * f;
                }
                __bf16 *dst = out + (int64_t)row * COUT + co_base + r * C::NJ;
                if (C::NJ == 4) {
                    bf16x4s o;
                    o[0] = v[0]; o[1] = v[1 % C::NJ]; o[2] = v[2 % C::NJ]; o[3] = v[3 % C::NJ];
                    *reinterpret_cast<bf16x4s *>(dst) = o;
                } else if (C::NJ == 2) {
                    bf16x2s o;
                    o[0] = v[0]; o[1] = v[1 % C::NJ];
                    *reinterpret_cast<bf16x2s *>(dst) = o;
                } else {
                    dst[0] = v[0];
                }
            }
        }
    if (stats_partial) {   // block-uniform
#pragma unroll
        for (int jn = 0; jn < C::NJ; ++jn) {   // lanes with the same r hold the same columns: fold the four q groups
            s1[jn] += __shfl_xor(s1[jn], 16, 64); s1[jn] += __shfl_xor(s1[jn], 32, 64);
            s2[jn] += __shfl_xor(s2[jn], 16, 64); s2[jn] += __shfl_xor(s2[jn], 32, 64);
        }
        float *red = reinterpret_cast<float *>(smem);   // [WM][2][COUT]; every wave is out of the K loop: its buffers are free
        __syncthreads();
        if (q == 0) {
#pragma unroll
            for (int jn = 0; jn < C::NJ; ++jn) {
                red[(wm * 2 + 0) * COUT + co_base + r * C::NJ + jn] = s1[jn];
                red[(wm * 2 + 1) * COUT + co_base + r * C::NJ + jn] = s2[jn];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * COUT; e += 256) {
            const int which = e / COUT, col = e - which * COUT;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < C::WM; ++w) s += red[(w * 2 + which) * COUT + col];
            stats_partial[((int64_t)tile * 2 + which) * COUT + col] = s;
        }
    }
}

// Prologue shared by the forward kernels: the workgroup's slice of the gather map -> idx_lds [kslots][BM] (one coalesced
// pass; -1 for rows past the block and phantom offsets), then the tile-skip table skip_lds [steps or kvol][BM/16]
// (1 = some row of that 16-row MFMA tile has a neighbour at that K-step).  The table is written here, before the first
// LDS-DMA: a ds_write after a global_load_lds makes the compiler drain vmcnt(0) first (it cannot tell the DMA's
// destination from the store's), which serialised the gathers of a step in the first version of these kernels.
template <int CIN, int BM>
__device__ __forceinline__ void s16_fill_maps(const int32_t *__restrict__ nbr, int n_out, int kvol, int row0, int row_end,
                                              int *idx_lds, int *skip_lds) {
    constexpr int OPS = CIN < 64 ? 64 / CIN : 1, TILES = BM / 16;
    const int t = threadIdx.x;
    const int kslots = s16_kslots(CIN, kvol);
    for (int e = t; e < kslots * BM; e += 256) {
        const int k = e / BM, row = row0 + (e - k * BM);
        idx_lds[e] = (k < kvol && row < row_end) ? nbr[(int64_t)k * n_out + row] : -1;
    }
    __syncthreads();
    const int groups = kslots / OPS;
    for (int e = t; e < groups * TILES; e += 256) {
        const int g = e / TILES, i = e - g * TILES;
        int any = 0;
        for (int o = 0; o < OPS; ++o)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) any |= idx_lds[(g * OPS + o) * BM + 16 * i + rr] >= 0;
        skip_lds[e] = any;
    }
    __syncthreads();
}

template <int CIN, int COUT, int BM>
__global__ __launch_bounds__(256) void spconv_fwd_s16_kernel(const __bf16 *__restrict__ in, const __bf16 *__restrict__ wpack,
                                                             const float *__restrict__ bias, const int32_t *__restrict__ nbr,
                                                             const __bf16 *__restrict__ zero_page, int n_out, int kvol,
                                                             int rows_per_block, __bf16 *__restrict__ out, float *__restrict__ stats_partial) {
    // rows_per_block <= BM (multiple of 16): the launcher shrinks it so that the grid fills whole rounds of resident
    // workgroups; the tiles past it stay unmarked and are skipped.
    typedef S16Cfg<CIN, COUT, BM> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto abuf = [&](int b) -> char * { return smem + b * C::A_BYTES; };
    auto bbuf = [&](int b) -> char * { return smem + 2 * C::A_BYTES + b * C::B_BYTES; };
    int *idx_lds = reinterpret_cast<int *>(smem + 2 * C::A_BYTES + 2 * C::B_BYTES);   // [kslots][BM]
    int *skip_lds = idx_lds + s16_kslots(CIN, kvol) * BM;                              // [steps or kvol][TILES]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid / C::WN, wn = wid % C::WN;
    const int r = lane & 15, q = lane >> 4;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int row0 = tile * rows_per_block;
    if (row0 >= n_out) return;
    const int row_end = min(n_out, row0 + rows_per_block);
    const int steps = s16_steps(CIN, kvol);

    // staging role of this thread: LDS slot (row = id>>3, s = id&7) of chunk ids id = t + 256u holds part s ^ (row&7)
    const int prt = (t & 7) ^ ((t >> 3) & 7);
    const int oslot = CIN < 64 ? prt / C::PPO : 0;
    const int choff = (CIN < 64 ? prt % C::PPO : prt) * 8;

    // the block's whole gather-map slice lives in LDS (one coalesced pass up front): the per-step gathers then
    // depend on an LDS read, not on a second global round trip
    auto load_idx = [&](int s, int (&j)[C::A_LOADS]) {
        const int k = CIN < 64 ? s * C::OPS + oslot : s / C::CPO;
#pragma unroll
        for (int u = 0; u < C::A_LOADS; ++u) j[u] = idx_lds[k * BM + (t >> 3) + 32 * u];
    };
    auto stage = [&](int s, int buf, const int (&j)[C::A_LOADS]) {
        const int chunk = CIN > 64 ? (s % C::CPO) * 64 : 0;
#pragma unroll
        for (int u = 0; u < C::A_LOADS; ++u) {
            const __bf16 *src = j[u] >= 0 ? in + (int64_t)j[u] * CIN + chunk + choff : zero_page;
            char *dst = abuf(buf) + (size_t)(t - lane + 256 * u) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
        constexpr int B_UNITS = C::B_BYTES / 1024;
        const char *wsrc = reinterpret_cast<const char *>(wpack) + (int64_t)s * C::B_BYTES;
#pragma unroll
        for (int u = 0; u < (B_UNITS + 3) / 4; ++u) {
            const int unit = u * 4 + wid;
            if (unit < B_UNITS) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + unit * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(bbuf(buf) + unit * 1024), 16, 0, 0);
            }
        }
    };

    f32x4s acc[C::MI][C::NJ];
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int jn = 0; jn < C::NJ; ++jn) acc[i][jn] = f32x4s{0.f, 0.f, 0.f, 0.f};

    s16_fill_maps<CIN, BM>(nbr, n_out, kvol, row0, row_end, idx_lds, skip_lds);
    int jn_[C::A_LOADS];
    load_idx(0, jn_);
    stage(0, 0, jn_);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        if (s + 1 < steps) {
            load_idx(s + 1, jn_);
            stage(s + 1, cur ^ 1, jn_);
        }
        int on[C::MI];
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
            on[i] = __builtin_amdgcn_readfirstlane(skip_lds[(CIN < 64 ? s : s / C::CPO) * C::TILES + wm * C::MI + i]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8s b[C::NJ];
#pragma unroll
            for (int jn = 0; jn < C::NJ; ++jn)
                b[jn] = *reinterpret_cast<const bf16x8s *>(bbuf(cur) + ((h * (COUT / 16) + wn * C::NJ + jn) * 64 + lane) * 16);
#pragma unroll
            for (int i = 0; i < C::MI; ++i) {
                if (on[i]) {
                    const bf16x8s a = *reinterpret_cast<const bf16x8s *>(
                        abuf(cur) + ((wm * 16 * C::MI + 16 * i + r) * 8 + ((4 * h + q) ^ (r & 7))) * 16);
#pragma unroll
                    for (int jn = 0; jn < C::NJ; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[jn], acc[i][jn], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    s16_epilogue<CIN, COUT, BM, C::MI, C::NJ>(acc, bias, out, row0, row_end, wm, wn, r, q, stats_partial, tile, smem);
}

// Measured and dropped (r01): a variant with 32-channel K-steps on a 3-slot ring (manual vmcnt waits, three workgroups
// per CU - what won for the dense 3x3 kernel) is 30-55 % slower here: the gather then moves 64 B per row and request
// instead of 128 B, and the gather rate, not the barrier round trip, is what bounds this kernel.  A 3-slot ring of the
// 64-deep K-steps (two K-steps of gathers in flight, one 128-row workgroup per CU) was worse still: 128->128 62 -> 114 us.

// register-gather kernel (spconv_rg.hip): the default; S2D_S16_KERNEL=lds selects the LDS-staged kernel of this file (A/B runs)
struct RgPlan {
    int mi, waves, tiles_per_block;
    unsigned grid;
};
RgPlan rg_plan(int64_t n_out, int kvol, int cin, int cout);
int rg_run(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias, const int32_t *nbr, int64_t n_out, int kvol, int cin,
           int cout, void *out_feat, float *stats_partial, hipStream_t st);
size_t rg_packed_elems(int kvol, int cin, int cout);
int rg_pack(const float *weight, int kvol, int cin, int cout, int transpose, int flip, void *packed, hipStream_t st);
int rg_pack_pair(const float *weight, int kvol, int cin, int cout, int flip_dgrad, void *packed_fwd, void *packed_dgrad, hipStream_t st);
void rg_set_trace(void *p);
// sorted-row mode (see use_rg): initial value from S2D_RG_SORTED, changed by s2d_spconv_s16_set_sorted_rows
static int &g_sorted_rows_mode() {
    static int mode = (getenv("S2D_RG_SORTED") && atoi(getenv("S2D_RG_SORTED")) != 0) ? 1 : 0;
    return mode;
}
static bool use_rg(int cin, int cout) {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("S2D_S16_KERNEL");
        v = !(e && strcmp(e, "lds") == 0);
    }
    // 64 -> 64 (48 vs 52 us) and narrower stay on the LDS-staged kernel.  r03 experiment: the register-gather template instantiated for
    // 32 channels (four offsets per 128-deep K-step): 32 -> 32 61 vs 56 us, 64 -> 32 62 vs 68 us - no case for it; tile heights of the
    // LDS kernel re-swept at the same time (S2D_S16_PLAN): 64 rows still beat 128 / 256 at every 16...64-channel shape
    // r06: with the sorted-row form switched on (s2d_spconv_s16_set_sorted_rows: opt-in, it measured SLOWER - see csrc/rulebook_sort.hip) 64 -> 64
    // runs on the register-gather kernel too: one weight image must serve the plain and the sorted launch of a layer
    return v != 0 && cin >= 64 && cout >= 64 && (cin == 128 || cout == 128 || g_sorted_rows_mode() != 0);
}
bool rg_sorted_supported(int kvol, int cin, int cout);
int rg_run_sorted(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias, const int32_t *nbr_perm, const int32_t *perm,
                  const uint32_t *pmask, int64_t n_out, int kvol, int cin, int cout, void *out_feat, float *stats_partial, hipStream_t st);

static int s16_wn(int cout, int bm) { return cout == 128 ? 2 : ((cout == 64 && bm == 64) ? 2 : 1); }

// Launch plan (tile template height and rows per workgroup).
struct S16Plan {
    int bm, rows_per_block;
    unsigned grid;
};
static S16Plan s16_plan(int64_t n_out, int kvol, int cin, int cout) {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    if (const char *ov = getenv("S2D_S16_PLAN")) {   // tuning hook: "bm,rows_per_block"
        int bm = 0, rpb = 0;
        if (sscanf(ov, "%d,%d", &bm, &rpb) == 2 && (bm == 64 || bm == 128 || (bm == 256 && cout != 128)) && rpb >= 16 && rpb <= bm &&
            rpb % 16 == 0)
            return S16Plan{bm, rpb, (unsigned)ceil_div(n_out, rpb)};
    }
    // measured on MI355X over the stages of the 150k-point scene (scratch sweep, r01): the K loop is bound by gather
    // round trips, so what counts is rows resident per CU; 64-row tiles (3-4 workgroups per CU) win everywhere except
    // at 128 output channels, where the 16 KiB weight tile per step makes 128-row tiles cheaper per row.  Shrinking the
    // rows per workgroup below the template height never paid.
    (void)cus; (void)kvol; (void)cin;
    const int bm = cout == 128 ? 128 : 64;
    S16Plan best{bm, bm, (unsigned)ceil_div(n_out, bm)};
    return best;
}

template <int CIN, int COUT, int BM>
static int s16_launch(const S16Plan &p, const __bf16 *in, const __bf16 *wpack, const float *bias, const int32_t *nbr,
                      const __bf16 *zero_page, int n_out, int kvol, __bf16 *out, float *stats, hipStream_t st) {
    typedef S16Cfg<CIN, COUT, BM> C;
    auto kern = spconv_fwd_s16_kernel<CIN, COUT, BM>;
    static size_t attr_bytes = 48 * 1024;   // per instantiation; raising the limit is idempotent if raced
    const size_t lds = C::lds_bytes(kvol);
    if (lds > attr_bytes) {
        S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_bytes = lds;
    }
    hipLaunchKernelGGL(kern, dim3(xcd_grid(p.grid)), dim3(256), lds, st, in, wpack, bias, nbr, zero_page, n_out, kvol,
                       p.rows_per_block, out, stats);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

template <int CIN, int COUT>
static int s16_dispatch_bm(const S16Plan &p, const __bf16 *in, const __bf16 *wpack, const float *bias, const int32_t *nbr,
                           const __bf16 *zero_page, int n_out, int kvol, __bf16 *out, float *stats, hipStream_t st) {
    if (p.bm == 64) return s16_launch<CIN, COUT, 64>(p, in, wpack, bias, nbr, zero_page, n_out, kvol, out, stats, st);
    if (p.bm == 128) return s16_launch<CIN, COUT, 128>(p, in, wpack, bias, nbr, zero_page, n_out, kvol, out, stats, st);
    if constexpr (COUT != 128) return s16_launch<CIN, COUT, 256>(p, in, wpack, bias, nbr, zero_page, n_out, kvol, out, stats, st);
    return S2D_ERR_UNSUPPORTED;
}

template <int CIN>
static int s16_dispatch_cout(int cout, const S16Plan &p, const __bf16 *in, const __bf16 *wpack, const float *bias,
                             const int32_t *nbr, const __bf16 *zero_page, int n_out, int kvol, __bf16 *out, float *stats, hipStream_t st) {
    switch (cout) {
        case 16: return s16_dispatch_bm<CIN, 16>(p, in, wpack, bias, nbr, zero_page, n_out, kvol, out, stats, st);
        case 32: return s16_dispatch_bm<CIN, 32>(p, in, wpack, bias, nbr, zero_page, n_out, kvol, out, stats, st);
        case 64: return s16_dispatch_bm<CIN, 64>(p, in, wpack, bias, nbr, zero_page, n_out, kvol, out, stats, st);
        case 128: return s16_dispatch_bm<CIN, 128>(p, in, wpack, bias, nbr, zero_page, n_out, kvol, out, stats, st);
    }
    return S2D_ERR_UNSUPPORTED;
}

static bool s16_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }

}  // namespace s2d

using namespace s2d;

/* tuning aid (tools/spconv_kernel_bench.py --trace): device buffer int64[grid][64] that the ablation build of the register-gather
 * kernel fills with per-step s_memtime stamps when S2D_RG_DEBUG has bit 32 set; nullptr switches it off */
extern "C" void s2d_debug_rg_trace(void *buf) { rg_set_trace(buf); }

extern "C" int s2d_spconv_s16_supported(int cin, int cout) { return s16_ok(cin) && s16_ok(cout); }

extern "C" size_t s2d_spconv_s16_packed_elems(int kvol, int cin, int cout) {
    if (!s2d_spconv_s16_supported(cin, cout) || kvol <= 0) return 0;
    if (use_rg(cin, cout)) return rg_packed_elems(kvol, cin, cout);
    return (size_t)s16_steps(cin, kvol) * 64 * cout;
}

extern "C" int s2d_spconv_s16_pack_weights(const float *weight, int kvol, int cin, int cout, int transpose, int flip,
                                           int64_t n_out, void *packed, s2d_stream_t stream) {
    S2D_CHECK_ARG(weight && packed && kvol > 0 && n_out >= 0, "spconv_s16_pack: bad argument");
    if (!s2d_spconv_s16_supported(cin, cout)) {
        set_error("spconv_s16_pack: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    if (use_rg(cin, cout)) return rg_pack(weight, kvol, cin, cout, transpose, flip, packed, (hipStream_t)stream);
    const int64_t total = (int64_t)s2d_spconv_s16_packed_elems(kvol, cin, cout);
    hipLaunchKernelGGL(s16_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, kvol, cin,
                       cout, s16_wn(cout, s16_plan(n_out, kvol, cin, cout).bm), transpose, flip, (__bf16 *)packed);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* weight fp32 [kvol][cin][cout] -> packed_fwd (operand of a forward launch over n_out_fwd rows) and packed_dgrad (operand [cout -> cin] of the
 * data-gradient launch over n_out_dgrad rows, offsets mirrored when flip_dgrad) in ONE launch */
extern "C" int s2d_spconv_s16_pack_weights_pair(const float *weight, int kvol, int cin, int cout, int flip_dgrad, int64_t n_out_fwd,
                                                int64_t n_out_dgrad, void *packed_fwd, void *packed_dgrad, s2d_stream_t stream) {
    S2D_CHECK_ARG(weight && packed_fwd && packed_dgrad && kvol > 0 && n_out_fwd >= 0 && n_out_dgrad >= 0, "spconv_s16_pack_pair: bad argument");
    if (!s2d_spconv_s16_supported(cin, cout)) {
        set_error("spconv_s16_pack_pair: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    if (use_rg(cin, cout)) return rg_pack_pair(weight, kvol, cin, cout, flip_dgrad, packed_fwd, packed_dgrad, (hipStream_t)stream);
    const int64_t total = std::max<int64_t>((int64_t)s2d_spconv_s16_packed_elems(kvol, cin, cout), (int64_t)s2d_spconv_s16_packed_elems(kvol, cout, cin));
    hipLaunchKernelGGL(s16_pack_pair_kernel, dim3((unsigned)ceil_div(total, 256), 2), dim3(256), 0, (hipStream_t)stream, weight, kvol, cin, cout,
                       s16_wn(cout, s16_plan(n_out_fwd, kvol, cin, cout).bm), s16_wn(cin, s16_plan(n_out_dgrad, kvol, cout, cin).bm), flip_dgrad,
                       (__bf16 *)packed_fwd, (__bf16 *)packed_dgrad);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* rows of the per-workgroup statistics s2d_spconv_s16_fwd_stats writes for a launch over n_out rows */
extern "C" int64_t s2d_spconv_s16_stats_tiles(int64_t n_out, int kvol, int cin, int cout) {
    if (n_out <= 0 || kvol <= 0 || !s2d_spconv_s16_supported(cin, cout)) return 0;
    if (use_rg(cin, cout)) return (int64_t)rg_plan(n_out, kvol, cin, cout).grid;
    return (int64_t)s16_plan(n_out, kvol, cin, cout).grid;
}

extern "C" int s2d_spconv_s16_fwd_stats(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias, const int32_t *nbr,
                                        int64_t n_out, int kvol, int cin, int cout, const void *zero_page, void *out_feat,
                                        float *stats_partial, s2d_stream_t stream);

extern "C" int s2d_spconv_s16_fwd(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias,
                                  const int32_t *nbr, int64_t n_out, int kvol, int cin, int cout, const void *zero_page,
                                  void *out_feat, s2d_stream_t stream) {
    return s2d_spconv_s16_fwd_stats(in_feat, n_in, packed_weight, bias, nbr, n_out, kvol, cin, cout, zero_page, out_feat, nullptr, stream);
}

/* stats_partial (optional, fp32 [s2d_spconv_s16_stats_tiles][2][cout]): per-workgroup (sum, sum of squares) per output channel of the
 * stored rows - the statistics pass of the BatchNorm1d that follows (s2d_bn_partials_finalize_f32 / _sum_f32 fold them) */
extern "C" int s2d_spconv_s16_fwd_stats(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias, const int32_t *nbr,
                                        int64_t n_out, int kvol, int cin, int cout, const void *zero_page, void *out_feat,
                                        float *stats_partial, s2d_stream_t stream) {
    S2D_CHECK_ARG(n_in >= 0 && n_out >= 0 && n_out < 0x7fffffff && kvol > 0, "spconv_s16_fwd: bad sizes");
    if (!s2d_spconv_s16_supported(cin, cout)) {
        set_error("spconv_s16_fwd: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    if (n_out == 0) return S2D_OK;
    S2D_CHECK_ARG(in_feat && packed_weight && nbr && out_feat && zero_page && n_in > 0, "spconv_s16_fwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (use_rg(cin, cout)) {
        S2D_CHECK_ARG(n_in * cin * 2 < (int64_t)BUF_OOB, "spconv_s16_fwd: feature matrix of %lld rows exceeds the 2 GiB buffer window", (long long)n_in);
        return rg_run(in_feat, n_in, packed_weight, bias, nbr, n_out, kvol, cin, cout, out_feat, stats_partial, st);
    }
    const __bf16 *in = (const __bf16 *)in_feat, *wp = (const __bf16 *)packed_weight, *zp = (const __bf16 *)zero_page;
    __bf16 *out = (__bf16 *)out_feat;
    const S16Plan plan = s16_plan(n_out, kvol, cin, cout);
    switch (cin) {
        case 16: return s16_dispatch_cout<16>(cout, plan, in, wp, bias, nbr, zp, (int)n_out, kvol, out, stats_partial, st);
        case 32: return s16_dispatch_cout<32>(cout, plan, in, wp, bias, nbr, zp, (int)n_out, kvol, out, stats_partial, st);
        case 64: return s16_dispatch_cout<64>(cout, plan, in, wp, bias, nbr, zp, (int)n_out, kvol, out, stats_partial, st);
        case 128: return s16_dispatch_cout<128>(cout, plan, in, wp, bias, nbr, zp, (int)n_out, kvol, out, stats_partial, st);
    }
    return S2D_ERR_UNSUPPORTED;
}

/* rows grouped by neighbour mask (s2d_rulebook_sort_by_mask): see include/s2d.h */
extern "C" int s2d_spconv_s16_set_sorted_rows(int on) {
    const int was = g_sorted_rows_mode();
    g_sorted_rows_mode() = on ? 1 : 0;
    return was;
}

extern "C" int s2d_spconv_s16_sorted_supported(int kvol, int cin, int cout) {
    return g_sorted_rows_mode() != 0 && use_rg(cin, cout) && rg_sorted_supported(kvol, cin, cout);
}

extern "C" int s2d_spconv_s16_fwd_sorted(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias, const int32_t *nbr_perm,
                                         const int32_t *perm, const uint32_t *pmask, int64_t n_out, int kvol, int cin, int cout, void *out_feat,
                                         float *stats_partial, s2d_stream_t stream) {
    S2D_CHECK_ARG(n_in >= 0 && n_out >= 0 && n_out < 0x7fffffff && kvol > 0, "spconv_s16_fwd_sorted: bad sizes");
    if (!s2d_spconv_s16_sorted_supported(kvol, cin, cout)) {
        set_error("spconv_s16_fwd_sorted: unsupported layer %d -> %d, kvol %d", cin, cout, kvol);
        return S2D_ERR_UNSUPPORTED;
    }
    if (n_out == 0) return S2D_OK;
    S2D_CHECK_ARG(in_feat && packed_weight && nbr_perm && perm && pmask && out_feat && n_in > 0, "spconv_s16_fwd_sorted: null argument");
    S2D_CHECK_ARG(n_in * cin * 2 < (int64_t)BUF_OOB, "spconv_s16_fwd_sorted: feature matrix of %lld rows exceeds the 2 GiB buffer window", (long long)n_in);
    return rg_run_sorted(in_feat, n_in, packed_weight, bias, nbr_perm, perm, pmask, n_out, kvol, cin, cout, out_feat, stats_partial, (hipStream_t)stream);
}
