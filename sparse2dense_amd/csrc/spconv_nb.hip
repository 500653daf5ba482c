// Submanifold sparse-conv implicit GEMM, "neighbourhood-resident" formulation (bf16 feature storage, fp32 accumulate).
// Same contract as the gather kernel of spconv_s16.hip (out[o] = sum_k in[nbr[k][o]] W[k] + bias; forward, and the data
// gradient with the mirrored / transposed weight image; spconv.ops.indice_conv[_backward], call sites
// /root/reference/det3d/models/backbones/scn.py:104-152) - but built around what bounded that kernel (r01: 0.13 of the MFMA
// peak at 128 channels): it re-gathered every input row once per (offset, output row) pair (~15x) through the ~16 B/clk/CU
// LDS-DMA path and streamed the whole weight tensor through LDS for every 128 rows.
//
//   * TILE PLAN (built once per rulebook, shared by the 4-5 convs of a stage, forward and data gradient):
//     rows are grouped into tiles of T rows that are SPATIALLY compact (block-sorted order supplied by the host as `perm`),
//     and for each tile the set of DISTINCT input rows its 27 offsets touch is listed (`in_rows`, U ~ 1.3-2 T instead of
//     ~15 T gathered rows), with a local gather map lnbr[k][t] -> slot.
//   * the kernel loads the tile's U input rows into LDS ONCE (LDS-DMA, XOR-swizzled 16-byte pieces); A fragments are then
//     read straight out of the resident rows with per-lane addresses (slot of the lane's row at offset k) - no A tile is
//     materialised, no barrier inside the K loop.  Missing neighbours point at a zero row.
//   * the weight image of a K-step (C x C bf16, fragment order) is copied ONCE per workgroup into a double-buffered LDS slab
//     and shared by the four waves (each wave covers 64 rows x all C columns); measured r02: what bounds these kernels is the
//     bytes a CU pulls from L2 (~16 B/clk/CU whether by LDS-DMA or by plain loads - a first version that loaded B fragments
//     per wave straight into registers moved 2-4x the weight bytes and was slower than the gather kernel), so the design
//     minimises fetched bytes per CU: 256-row tiles, resident rows, shared weights, and K-steps no row of the tile uses
//     are skipped entirely (no copy, no MFMA).
//   * 16-row MFMA tiles with no neighbour at an offset are skipped (bit table per tile and offset); spatial tiles make
//     neighbouring rows share their occupied offsets, so the skip removes most of the zero work.
//   * tiles whose neighbourhood exceeds the LDS budget are processed in several phases (slots [p*UMAX, (p+1)*UMAX)).
//   * optional epilogue: per-tile (sum, sum of squares) per output channel of the stored bf16 values = the statistics
//     pass of the BatchNorm that follows.
#include "s2d_common.h"

namespace s2d {

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4n __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2n __attribute__((ext_vector_type(2)));

constexpr int NB_K = 27;          // kernel volume handled by this path (3x3x3 SubM)
constexpr int NB_KS = 28;         // offset slots (even: the 16-channel kernel pairs offsets)

template <int C>
struct NbCfg {   // CIN == COUT == C (SubM layers of the backbone)
    static constexpr int WM = 4;                           // the 4 waves split the rows; every wave covers all C columns
    static constexpr int MI = 4;
    static constexpr int T = WM * MI * 16;                 // 256 rows per tile
    static constexpr int NJ = C / 16;                      // 16-column MFMA tiles
    static constexpr int PARTS = C / 8;                    // 16-byte pieces per row
    static constexpr int ROWB = C * 2;
    static constexpr int CH = C >= 32 ? C / 32 : 1;        // 32-deep K chunks per offset (C = 16: two offsets share one)
    static constexpr int KSTEPS = C == 16 ? NB_KS / 2 : NB_K;   // weight images per launch (offset pairs for C = 16)
    static constexpr int BBYTES = CH * NJ * 1024;          // one weight image (K-step) in LDS
    // resident slots per phase, sized so that 1 (C = 128), 2 (C = 64), 3 (C = 32) or 4 (C = 16) workgroups fit a CU's 160 KiB
    static constexpr int UMAX = C == 128 ? 416 : (C == 64 ? 432 : (C == 32 ? 544 : 720));
    static constexpr int RPB = 256 / ROWB;                 // rows per 256-byte LDS bank row
    static constexpr int UCAP = NB_K * T;                  // slots reserved per tile in the plan (worst case)
    static constexpr size_t LDS = (size_t)(UMAX + 1) * ROWB + (size_t)BBYTES + (size_t)NB_KS * T * 2 + (size_t)T * 4 + NB_KS * 4 + 16;
    __host__ __device__ static int key(int slot) { return (slot / RPB) & (PARTS - 1); }
};

// ---- tile plan -----------------------------------------------------------------------------------------------------
// plan arrays (device): rows[tiles][T] i32 (-1 pad), u[tiles] i32, in_rows[tiles][UCAP] i32, lnbr[tiles][NB_KS][T] u16 (0xFFFF none),
// act[tiles][NB_KS] u32 (bit i: 16-row tile i has a neighbour at offset k)
template <int T>
__global__ __launch_bounds__(256) void nb_plan_kernel(const int32_t *__restrict__ nbr, const int32_t *__restrict__ perm, int n, int32_t *__restrict__ rows_out,
                                                      int32_t *__restrict__ u_out, int32_t *__restrict__ in_rows, uint16_t *__restrict__ lnbr,
                                                      uint32_t *__restrict__ act) {
    constexpr int HC = 8192;   // hash capacity: worst case 27 * 256 = 6912 distinct rows (84 % load); typical neighbourhoods fill < 10 %
    extern __shared__ int32_t sm[];
    int32_t *hkey = sm;                 // [HC] row id or -1
    int32_t *hslot = sm + HC;           // [HC] slot of the key
    int32_t *rows = sm + 2 * HC;        // [T]
    int32_t *cnt = rows + T;            // [1] + wave scan scratch [4]
    uint32_t *actl = reinterpret_cast<uint32_t *>(cnt + 8);   // [NB_KS]
    const int tile = blockIdx.x, t = threadIdx.x;
    for (int e = t; e < HC; e += 256) hkey[e] = -1;
    for (int e = t; e < T; e += 256) {
        const int i = tile * T + e;
        rows[e] = i < n ? (perm ? perm[i] : i) : -1;
    }
    if (t < NB_KS) actl[t] = 0u;
    if (t == 0) cnt[0] = 0;
    __syncthreads();
    // pass 1: insert every neighbour row into the hash set
    for (int e = t; e < NB_K * T; e += 256) {
        const int k = e / T, r = e - k * T;
        const int row = rows[r];
        if (row < 0) continue;
        const int j = nbr[(int64_t)k * n + row];
        if (j < 0) continue;
        atomicOr(&actl[k], 1u << (r >> 4));
        uint32_t h = ((uint32_t)j * 2654435761u) >> 19;   // top log2(HC) bits
        while (true) {
            const int prev = atomicCAS(&hkey[h], -1, j);
            if (prev == -1 || prev == j) break;
            h = (h + 1) & (HC - 1);
        }
    }
    __syncthreads();
    // pass 2: number the occupied hash entries (block-wide exclusive scan in table order) -> slots, in_rows
    {
        constexpr int PER = HC / 256;
        int local = 0;
        for (int e = 0; e < PER; ++e) local += hkey[t * PER + e] >= 0;
        // wave scan + cross-wave offsets
        int incl = local;
        const int lane = t & 63, wid = t >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) cnt[1 + wid] = incl;
        __syncthreads();
        int base = incl - local;
        for (int w = 0; w < wid; ++w) base += cnt[1 + w];
        if (t == 255) cnt[0] = base + local;
        for (int e = 0; e < PER; ++e) {
            const int j = hkey[t * PER + e];
            if (j >= 0) {
                hslot[t * PER + e] = base;
                in_rows[(int64_t)tile * (NB_K * T) + base] = j;
                ++base;
            }
        }
    }
    __syncthreads();
    if (t == 0) u_out[tile] = cnt[0];
    if (t < NB_KS) act[(int64_t)tile * NB_KS + t] = actl[t];
    for (int e = t; e < T; e += 256) rows_out[(int64_t)tile * T + e] = rows[e];
    // pass 3: local gather map
    for (int e = t; e < NB_KS * T; e += 256) {
        const int k = e / T, r = e - k * T;
        const int row = rows[r];
        uint16_t s = 0xFFFFu;
        if (k < NB_K && row >= 0) {
            const int j = nbr[(int64_t)k * n + row];
            if (j >= 0) {
                uint32_t h = ((uint32_t)j * 2654435761u) >> 19;
                while (hkey[h] != j) h = (h + 1) & (HC - 1);
                s = (uint16_t)hslot[h];
            }
        }
        lnbr[(int64_t)tile * NB_KS * T + e] = s;
    }
}

// ---- weight image ----------------------------------------------------------------------------------------------------
// [kstep][wn][chunk][jn][lane 64][8 bf16]; K element of lane (q = lane>>4, e): kk = 8q + e within a 32-deep chunk.
//   C >= 32: kstep = offset k, input channel ci = 32*chunk + kk ; C = 16: kstep = offset pair, k = 2*kstep + (kk >> 4), ci = kk & 15
// column of lane r = lane & 15 in n-tile jn: co = wn*(C/WN) + r*NJ + jn   (NJ consecutive channels per lane: contiguous stores)
// source w fp32 [27][cin][cout]; transpose: w is the forward layer's [27][cout][cin] (data gradient); flip: offsets mirrored.
__global__ __launch_bounds__(256) void nb_pack_kernel(const float *__restrict__ w, int c, int wn_count, int transpose, int flip, __bf16 *__restrict__ out) {
    const int ksteps = c == 16 ? NB_KS / 2 : NB_K, ch = c >= 32 ? c / 32 : 1, nj = c / wn_count / 16;
    const int64_t total = (int64_t)ksteps * wn_count * ch * nj * 512;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int64_t r = i;
    const int e = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int jn = r % nj; r /= nj;
    const int chunk = r % ch; r /= ch;
    const int wn = r % wn_count; r /= wn_count;
    const int ks = (int)r;
    const int kk = 8 * (lane >> 4) + e;
    const int k = c == 16 ? 2 * ks + (kk >> 4) : ks;
    const int ci = c == 16 ? (kk & 15) : 32 * chunk + kk;
    const int co = wn * (c / wn_count) + (lane & 15) * nj + jn;
    float v = 0.f;
    if (k < NB_K) {
        const int ksrc = flip ? NB_K - 1 - k : k;
        v = transpose ? w[((int64_t)ksrc * c + co) * c + ci] : w[((int64_t)ksrc * c + ci) * c + co];
    }
    out[i] = (__bf16)v;
}

// ---- the kernel ------------------------------------------------------------------------------------------------------
template <int C, bool STATS>
__global__ __launch_bounds__(256) void spconv_nb_kernel(const __bf16 *__restrict__ in, const __bf16 *__restrict__ wpack, const float *__restrict__ bias,
                                                        const int32_t *__restrict__ p_rows, const int32_t *__restrict__ p_u,
                                                        const int32_t *__restrict__ p_in, const uint16_t *__restrict__ p_lnbr,
                                                        const uint32_t *__restrict__ p_act, const __bf16 *__restrict__ zero_page, int n_tiles,
                                                        __bf16 *__restrict__ out, float *__restrict__ stats) {
    typedef NbCfg<C> F;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *rowsb = smem;                                                         // [(UMAX+1)][ROWB] resident input rows (+ zero row)
    char *bbuf = smem + (size_t)(F::UMAX + 1) * F::ROWB;                        // [BBYTES] weight image of the current K-step
    uint16_t *lnbr = reinterpret_cast<uint16_t *>(bbuf + F::BBYTES);            // [NB_KS][T]
    int32_t *orow = reinterpret_cast<int32_t *>(lnbr + NB_KS * F::T);           // [T]
    uint32_t *act = reinterpret_cast<uint32_t *>(orow + F::T);                  // [NB_KS]
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid;
    const int r = lane & 15, q = lane >> 4;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    if (tile >= n_tiles) return;
    const int u_total = p_u[tile];
    const int32_t *tin = p_in + (int64_t)tile * F::UCAP;

    // descriptors -> LDS (before any LDS-DMA is in flight)
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p_lnbr + (int64_t)tile * NB_KS * F::T);
        uint32_t *dst = reinterpret_cast<uint32_t *>(lnbr);
        for (int e = t; e < NB_KS * F::T / 2; e += 256) dst[e] = src[e];
        for (int e = t; e < F::T; e += 256) orow[e] = p_rows[(int64_t)tile * F::T + e];
        if (t < NB_KS) act[t] = p_act[(int64_t)tile * NB_KS + t];
        for (int e = t; e < F::ROWB / 4; e += 256) reinterpret_cast<uint32_t *>(rowsb + (size_t)F::UMAX * F::ROWB)[e] = 0u;   // the zero row
    }

    f32x4n acc[F::MI][F::NJ];
#pragma unroll
    for (int i = 0; i < F::MI; ++i)
#pragma unroll
        for (int jn = 0; jn < F::NJ; ++jn) acc[i][jn] = f32x4n{0.f, 0.f, 0.f, 0.f};

    // weight image of K-step ks -> the LDS slab (linear copy, 1 KiB per wave instruction, spread over the waves)
    auto stage_b = [&](int ks) {
        const char *wsrc = reinterpret_cast<const char *>(wpack) + (int64_t)ks * F::BBYTES;
        constexpr int UNITS = F::BBYTES / 1024;
#pragma unroll
        for (int u0 = 0; u0 < (UNITS + 3) / 4; ++u0) {
            const int unit = u0 * 4 + wid;
            if (unit < UNITS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + unit * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(bbuf + unit * 1024), 16, 0, 0);
        }
    };

    __syncthreads();   // descriptors visible
    // K-steps with at least one pair in this tile (flat neighbourhoods use few of the 27): the others cost neither a weight
    // copy nor MFMAs.  One bit per K-step, wave-uniform.
    uint32_t live_all = 0u;
#pragma unroll 1
    for (int ks = 0; ks < F::KSTEPS; ++ks) {
        const uint32_t m = C == 16 ? (act[2 * ks] | act[2 * ks + 1]) : act[ks];
        live_all |= (m != 0u ? 1u : 0u) << ks;
    }
    live_all = __builtin_amdgcn_readfirstlane(live_all);

    const int phases = (u_total + F::UMAX - 1) / F::UMAX;
    for (int ph = 0; ph < max(phases, 1); ++ph) {
        const int sbase = ph * F::UMAX;
        const int cnt = min(F::UMAX, u_total - sbase);
        if (ph) __syncthreads();   // previous phase's reads of the resident rows are done
        // gather the phase's rows: piece p = slot*PARTS + part', source part = part' ^ key(slot)
        for (int p0 = wid * 64; p0 < cnt * F::PARTS; p0 += 256) {
            const int p = p0 + lane;
            const int slot = p / F::PARTS, pp = p % F::PARTS;
            const __bf16 *src = zero_page;
            if (slot < cnt) src = in + (int64_t)tin[sbase + slot] * C + ((pp ^ F::key(slot)) * 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(rowsb + (size_t)p0 * 16), 16, 0, 0);
        }
        uint32_t live = live_all;
        if (live) stage_b(__builtin_ctz(live));
#pragma unroll 1
        while (live) {
            const int ks = __builtin_ctz(live);
            live &= live - 1u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                 // the weight image of ks (and, first time round, the resident rows) has landed
            bf16x8n b[F::CH][F::NJ];         // this wave's copy of the image: every wave covers all C columns
#pragma unroll
            for (int c = 0; c < F::CH; ++c)
#pragma unroll
                for (int jn = 0; jn < F::NJ; ++jn) b[c][jn] = *reinterpret_cast<const bf16x8n *>(bbuf + (c * F::NJ + jn) * 1024 + lane * 16);
            uint32_t on = C == 16 ? (act[2 * ks] | act[2 * ks + 1]) : act[ks];
            on = (__builtin_amdgcn_readfirstlane(on) >> (wm * F::MI)) & ((1u << F::MI) - 1u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                 // every wave holds the image in registers: the slab can be refilled ...
            if (live) stage_b(__builtin_ctz(live));   // ... with the next live K-step while this one's MFMAs run
            const int koff = C == 16 ? 2 * ks + (q >> 1) : ks;
#pragma unroll
            for (int i = 0; i < F::MI; ++i) {
                if (!((on >> i) & 1u)) continue;   // 16-row tile without a neighbour at this offset (wave-uniform)
                int sl = (int)lnbr[koff * F::T + (wm * F::MI + i) * 16 + r] - sbase;
                sl = (unsigned)sl < (unsigned)cnt ? sl : F::UMAX;
                const char *rowp = rowsb + (size_t)sl * F::ROWB;
                const int key = F::key(sl);
                bf16x8n a[F::CH];
#pragma unroll
                for (int c = 0; c < F::CH; ++c) {
                    const int part = C == 16 ? (q & 1) : 4 * c + q;
                    a[c] = *reinterpret_cast<const bf16x8n *>(rowp + ((part ^ key) << 4));
                }
#pragma unroll
                for (int c = 0; c < F::CH; ++c)
#pragma unroll
                    for (int jn = 0; jn < F::NJ; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c], b[c][jn], acc[i][jn], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // epilogue: C/D layout row = 4*(lane>>4)+reg, col = lane&15 -> columns co_base + r*NJ + jn (NJ consecutive)
    const int co_base = 0;
    float bv[F::NJ];
#pragma unroll
    for (int jn = 0; jn < F::NJ; ++jn) bv[jn] = bias ? bias[co_base + r * F::NJ + jn] : 0.f;
    float s1[F::NJ], s2[F::NJ];
#pragma unroll
    for (int jn = 0; jn < F::NJ; ++jn) s1[jn] = s2[jn] = 0.f;
#pragma unroll
    for (int i = 0; i < F::MI; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = orow[(wm * F::MI + i) * 16 + 4 * q + reg];
            if (row < 0) continue;
            __bf16 v[F::NJ];
#pragma unroll
            for (int jn = 0; jn < F::NJ; ++jn) {
                v[jn] = (__bf16)(acc[i][jn][reg] + bv[jn]);
                if (STATS) {
                    const float f = (float)v[jn];
                    s1[jn] += f;
                    s2[jn] += f * f;
                }
            }
            __bf16 *dst = out + (int64_t)row * C + co_base + r * F::NJ;
            if (F::NJ == 8) {
                bf16x8n o;
#pragma unroll
                for (int jn = 0; jn < 8; ++jn) o[jn] = v[jn % F::NJ];
                *reinterpret_cast<bf16x8n *>(dst) = o;
            } else if (F::NJ == 4) {
                bf16x4n o; o[0] = v[0]; o[1] = v[1 % F::NJ]; o[2] = v[2 % F::NJ]; o[3] = v[3 % F::NJ];
                *reinterpret_cast<bf16x4n *>(dst) = o;
            } else if (F::NJ == 2) {
                bf16x2n o; o[0] = v[0]; o[1] = v[1 % F::NJ];
                *reinterpret_cast<bf16x2n *>(dst) = o;
            } else {
                dst[0] = v[0];
            }
        }
    if (STATS) {
        // fold the 4 q-lanes of a column group, then the WM row waves through LDS -> stats[tile][2][C]
        __syncthreads();   // all A reads done: the resident rows can be overwritten
        float *red = reinterpret_cast<float *>(rowsb);   // [WM][2][C]
#pragma unroll
        for (int jn = 0; jn < F::NJ; ++jn) {
            float a = s1[jn], b = s2[jn];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (q == 0) {
                red[(wm * 2 + 0) * C + co_base + r * F::NJ + jn] = a;
                red[(wm * 2 + 1) * C + co_base + r * F::NJ + jn] = b;
            }
        }
        __syncthreads();
        for (int e = t; e < 2 * C; e += 256) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < F::WM; ++w) v += red[(w * 2 + e / C) * C + e % C];
            stats[(int64_t)tile * 2 * C + e] = v;
        }
    }
}

static bool nb_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }
static int nb_tile_rows(int) { return 256; }

template <int C>
static int nb_launch(const __bf16 *in, const __bf16 *wp, const float *bias, const int32_t *p_rows, const int32_t *p_u, const int32_t *p_in,
                     const uint16_t *p_lnbr, const uint32_t *p_act, const __bf16 *zp, int n_tiles, __bf16 *out, float *stats, hipStream_t st) {
    typedef NbCfg<C> F;
    static bool attr_done = false;
    if (!attr_done) {
        S2D_HIP(hipFuncSetAttribute((const void *)spconv_nb_kernel<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F::LDS));
        S2D_HIP(hipFuncSetAttribute((const void *)spconv_nb_kernel<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F::LDS));
        attr_done = true;
    }
    const dim3 grid(xcd_grid(n_tiles)), blk(256);
    if (stats)
        hipLaunchKernelGGL((spconv_nb_kernel<C, true>), grid, blk, F::LDS, st, in, wp, bias, p_rows, p_u, p_in, p_lnbr, p_act, zp, n_tiles, out, stats);
    else
        hipLaunchKernelGGL((spconv_nb_kernel<C, false>), grid, blk, F::LDS, st, in, wp, bias, p_rows, p_u, p_in, p_lnbr, p_act, zp, n_tiles, out, stats);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_spconv_nb_supported(int channels, int kvol) { return nb_ok(channels) && kvol == NB_K; }
extern "C" int s2d_spconv_nb_tile_rows(int channels) { return nb_ok(channels) ? nb_tile_rows(channels) : 0; }

/* element counts of the plan arrays for n rows: [0] tiles, [1] rows i32, [2] u i32, [3] in_rows i32, [4] lnbr u16, [5] act u32 */
extern "C" int s2d_spconv_nb_plan_sizes(int channels, int64_t n, int64_t sizes[6]) {
    S2D_CHECK_ARG(nb_ok(channels) && n >= 0 && sizes, "spconv_nb_plan_sizes: bad argument");
    const int64_t t = nb_tile_rows(channels), tiles = ceil_div(n, t);
    sizes[0] = tiles; sizes[1] = tiles * t; sizes[2] = tiles; sizes[3] = tiles * NB_K * t; sizes[4] = tiles * NB_KS * t; sizes[5] = tiles * NB_KS;
    return S2D_OK;
}

extern "C" int s2d_spconv_nb_plan_build(const int32_t *nbr, const int32_t *perm, int64_t n, int channels, int32_t *rows, int32_t *u,
                                        int32_t *in_rows, uint16_t *lnbr, uint32_t *act, s2d_stream_t stream) {
    S2D_CHECK_ARG(nb_ok(channels) && n >= 0 && n < 0x7fffffff, "spconv_nb_plan_build: bad argument");
    if (n == 0) return S2D_OK;
    S2D_CHECK_ARG(nbr && rows && u && in_rows && lnbr && act, "spconv_nb_plan_build: null argument");
    const int t = nb_tile_rows(channels);
    const int tiles = (int)ceil_div(n, t);
    hipStream_t st = (hipStream_t)stream;
    {
        const size_t lds = (size_t)(2 * 8192 + 256 + 8 + NB_KS) * 4;
        static bool done = false;
        if (!done) { S2D_HIP(hipFuncSetAttribute((const void *)nb_plan_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
        hipLaunchKernelGGL(nb_plan_kernel<256>, dim3(tiles), dim3(256), lds, st, nbr, perm, (int)n, rows, u, in_rows, lnbr, act);
    }
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" size_t s2d_spconv_nb_packed_elems(int channels) {
    if (!nb_ok(channels)) return 0;
    const int ch = channels >= 32 ? channels / 32 : 1, nj = channels / 16;
    return (size_t)(channels == 16 ? NB_KS / 2 : NB_K) * ch * nj * 512;
}

extern "C" int s2d_spconv_nb_pack_weights(const float *weight, int channels, int transpose, int flip, void *packed, s2d_stream_t stream) {
    S2D_CHECK_ARG(weight && packed && nb_ok(channels), "spconv_nb_pack: bad argument");
    const int64_t total = (int64_t)s2d_spconv_nb_packed_elems(channels);
    hipLaunchKernelGGL(nb_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, channels, 1, transpose,
                       flip, (__bf16 *)packed);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_spconv_nb_fwd(const void *in_feat, const void *packed_weight, const float *bias, const int32_t *plan_rows, const int32_t *plan_u,
                                 const int32_t *plan_in, const uint16_t *plan_lnbr, const uint32_t *plan_act, int64_t n_tiles, int channels,
                                 const void *zero_page, void *out_feat, float *stats, s2d_stream_t stream) {
    S2D_CHECK_ARG(nb_ok(channels) && n_tiles >= 0 && n_tiles < 0x7fffffff, "spconv_nb_fwd: bad argument");
    if (n_tiles == 0) return S2D_OK;
    S2D_CHECK_ARG(in_feat && packed_weight && plan_rows && plan_u && plan_in && plan_lnbr && plan_act && zero_page && out_feat,
                  "spconv_nb_fwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    const __bf16 *in = (const __bf16 *)in_feat, *wp = (const __bf16 *)packed_weight, *zp = (const __bf16 *)zero_page;
    __bf16 *out = (__bf16 *)out_feat;
    switch (channels) {
        case 16: return nb_launch<16>(in, wp, bias, plan_rows, plan_u, plan_in, plan_lnbr, plan_act, zp, (int)n_tiles, out, stats, st);
        case 32: return nb_launch<32>(in, wp, bias, plan_rows, plan_u, plan_in, plan_lnbr, plan_act, zp, (int)n_tiles, out, stats, st);
        case 64: return nb_launch<64>(in, wp, bias, plan_rows, plan_u, plan_in, plan_lnbr, plan_act, zp, (int)n_tiles, out, stats, st);
        case 128: return nb_launch<128>(in, wp, bias, plan_rows, plan_u, plan_in, plan_lnbr, plan_act, zp, (int)n_tiles, out, stats, st);
    }
    return S2D_ERR_UNSUPPORTED;
}
