// Sparse-conv implicit GEMM, bf16 storage, REGISTER-GATHER form (r03; replaces the LDS-staged A tile of spconv_s16.hip).
// (spconv.ops.indice_conv / indice_conv_backward; call sites det3d/models/backbones/scn.py:104-152)
//
// What bounded the LDS-staged kernel was not the L2 but the issue cost of the LDS-DMA pieces: a 128-row x 128-column
// workgroup issued 32 one-KiB `global_load_lds` per 64-deep K-step (16 KiB gathered A + 16 KiB weight slab) next to 32 MFMAs
// per wave, at 60-185 cycles of issue per piece (MI355X_MICROARCH.md, "LDS-DMA piece issue cost") - the step took ~1270
// cycles against 544 of MFMA.  Here the A operand never touches LDS:
//   * a wave owns MI 16-row MFMA tiles x ALL output columns.  Lane (r = lane & 15, q = lane >> 4) of v_mfma_f32_16x16x32_bf16
//     holds A[row r][k = 8q..8q+7] = 16 contiguous bytes of the gathered input row, so the fragment IS one
//     `buffer_load_dwordx4` from feature row nbr[offset][row r]; a missing neighbour is an out-of-range buffer offset
//     (reads zero, moves no data, needs no zero page and no branch).  Four lanes cover 64 contiguous bytes of a row.
//   * one step = one kernel offset (CIN >= 64) or 64 / CIN offsets (CIN < 64): the A registers of step s+1 are loaded while
//     step s multiplies (register double buffer, loop unrolled by two), the gather indices one step further ahead.
//   * only the weight slab of a step (K x COUT, 2..32 KiB, packed in fragment order) goes through LDS: global -> registers
//     during step s-1 -> ds_write at the top of step s -> read by every wave in step s+1; two LDS stages, one barrier per step.
//   * a workgroup is 8 waves (2 per SIMD) working on `tiles_per_block` <= 8 MI consecutive 16-row tiles dealt round-robin
//     to the waves; the launcher sizes tiles_per_block so that the grid is a whole number of rounds of the 256 CUs.
//   * 16-row tiles with no neighbour at a step skip their MFMAs (wave-uniform ballot of the indices).
// The epilogue is the one of the LDS-staged kernel: bias, one bf16 rounding, 2*NJ-byte row stores, optional per-workgroup
// (sum, sum of squares) rows for the BatchNorm1d that follows.
#include "s2d_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace s2d {

typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));

__host__ __device__ inline int rg_steps(int cin, int kvol) { return cin < 64 ? (kvol + 64 / cin - 1) / (64 / cin) : kvol; }

template <int CIN, int COUT, int MI_, int WAVES_>
struct RgCfg {
    static constexpr int MI = MI_, WAVES = WAVES_;
    static constexpr int OPS = CIN < 64 ? 64 / CIN : 1;      // kernel offsets per step
    static constexpr int KC = CIN <= 64 ? 2 : CIN / 32;      // 32-deep MFMA K chunks per step
    static constexpr int NIDX = CIN < 64 ? 2 : 1;            // gather indices per lane, tile and step (chunk c uses index c % NIDX)
    static constexpr int NJ = COUT / 16;                     // 16-column MFMA tiles (a wave owns all of them)
    static constexpr int THREADS = WAVES * 64;
    static constexpr int B_BYTES = KC * 32 * COUT * 2;       // weight slab of one step
    static constexpr int B_PIECES = B_BYTES / 16;
    static constexpr int B_LOADS = (B_PIECES + THREADS - 1) / THREADS;
    static constexpr int CAP = WAVES * MI;                   // 16-row tiles per workgroup at most
    static constexpr size_t LDS = 2 * (size_t)B_BYTES > (size_t)WAVES * 2 * COUT * 4 ? 2 * (size_t)B_BYTES : (size_t)WAVES * 2 * COUT * 4;
};

// The packed weight image is the one of spconv_s16.hip's s16_pack_element with wn_count = 1:
//   [64-deep K-step][h (2)][nt = COUT/16][lane = q*16 + r][e (8)],  K element 32 h + 8 q + e, column r * NJ + nt
// (a lane's NJ accumulator columns are consecutive in the output row).  For CIN = 128 an offset is two consecutive K-steps.

template <int CIN, int COUT, int MI, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void spconv_rg_kernel(const __bf16 *__restrict__ in, unsigned in_bytes, const __bf16 *__restrict__ wpack,
                                                               const float *__restrict__ bias, const int32_t *__restrict__ nbr, int n_out,
                                                               int kvol, int tiles_per_block, __bf16 *__restrict__ out,
                                                               float *__restrict__ stats_partial, int dbg) {
    typedef RgCfg<CIN, COUT, MI, WAVES> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int block = xcd_tile(blockIdx.x, gridDim.x);
    const int total_tiles = (n_out + 15) >> 4;
    const int tile0 = block * tiles_per_block;
    if (tile0 >= total_tiles) return;
    const int tile_end = min(total_tiles, tile0 + tiles_per_block);
    const int steps = rg_steps(CIN, kvol);

    // this lane's output row per tile slot (-1: no such tile / row past the end)
    int row[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int tl = tile0 + wid + WAVES * i;
        const int rw = tl * 16 + r;
        row[i] = (tl < tile_end && rw < n_out) ? rw : -1;
    }
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(in, in_bytes);
    // byte offset of this lane's 16 bytes inside a gathered row for chunk c, and which of the step's offsets it belongs to
    auto chunk_byte = [&](int c) -> unsigned { return CIN >= 64 ? (unsigned)(64 * c + 16 * q) : (CIN == 32 ? (unsigned)(16 * q) : (unsigned)(16 * (q & 1))); };
    auto chunk_off = [&](int s, int j) -> int { return CIN >= 64 ? s : (CIN == 32 ? 2 * s + j : 4 * s + 2 * j + (q >> 1)); };

    // A fragments travel as raw 4-dword vectors (a bf16x8 value carried around the loop is rebuilt half by half by hipcc, which
    // puts a wait on the load right behind its issue); gather indices are loaded unconditionally from a clamped address and
    // masked when they are consumed, so that no step of the loop has a conditional memory operation: hipcc's vmcnt bookkeeping
    // then keeps every load class one full step in flight.
    typedef float f4 __attribute__((ext_vector_type(4)));
    int rowc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rowc[i] = row[i] < 0 ? 0 : row[i];
    auto load_idx = [&](int s, int (&idx)[MI][C::NIDX]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NIDX; ++j) idx[i][j] = nbr[(int64_t)min(chunk_off(s, j), kvol - 1) * n_out + rowc[i]];
    };
    auto mask_idx = [&](int s, int (&idx)[MI][C::NIDX]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NIDX; ++j) {
                if (dbg & 1) idx[i][j] = rowc[i];   // ablation: sequential rows, every neighbour present
                if (!(row[i] >= 0 && chunk_off(s, j) < kvol)) idx[i][j] = -1;
            }
    };
    auto load_a = [&](const int (&idx)[MI][C::NIDX], f4 (&a)[MI][C::KC]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int c = 0; c < C::KC; ++c) {
                const int j = idx[i][c % C::NIDX];
                const unsigned off = (j >= 0 && !(dbg & 4)) ? (unsigned)j * (unsigned)(CIN * 2) + chunk_byte(c) : BUF_OOB;
                a[i][c] = buf_load4(rin, off, 0);
            }
    };
    auto tile_flags = [&](const int (&idx)[MI][C::NIDX], bool (&on)[MI][C::NIDX]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NIDX; ++j) on[i][j] = __ballot(idx[i][j] >= 0) != 0ull;
    };
    auto load_b = [&](int s, f4 (&breg)[C::B_LOADS]) {
        const char *src = reinterpret_cast<const char *>(wpack) + (int64_t)((dbg & 8) ? 0 : s) * C::B_BYTES;
#pragma unroll
        for (int u = 0; u < C::B_LOADS; ++u) {
            const int piece = t + C::THREADS * u;
            if (C::B_PIECES % C::THREADS == 0 || piece < C::B_PIECES) breg[u] = *reinterpret_cast<const f4 *>(src + (size_t)piece * 16);
        }
    };
    auto store_b = [&](int stage, const f4 (&breg)[C::B_LOADS]) {
#pragma unroll
        for (int u = 0; u < C::B_LOADS; ++u) {
            const int piece = t + C::THREADS * u;
            if (C::B_PIECES % C::THREADS == 0 || piece < C::B_PIECES) *reinterpret_cast<f4 *>(smem + stage * C::B_BYTES + (size_t)piece * 16) = breg[u];
        }
    };

    f32x4r acc[MI][C::NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int n = 0; n < C::NJ; ++n) acc[i][n] = f32x4r{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int stage, const f4 (&a)[MI][C::KC], const bool (&on)[MI][C::NIDX]) {
        const char *bs = smem + stage * C::B_BYTES;
        bool any = false;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NIDX; ++j) any |= on[i][j];
        if (any && !(dbg & 2)) {
            // ONE straight-line block per step (a second, per-tile-branched path made hipcc copy the accumulators between the
            // paths): tiles without a neighbour multiply the zeros their out-of-range loads returned.  The weight fragments
            // travel in groups of FG column tiles, group g+1 is read from LDS while group g multiplies.
            constexpr int FG = C::NJ < 4 ? C::NJ : 4, GPC = C::NJ / FG, NG = C::KC * GPC;
            bf16x8r b[2][FG];
            auto read_g = [&](int g, bf16x8r (&bb)[FG]) {
#pragma unroll
                for (int n = 0; n < FG; ++n)
                    bb[n] = *reinterpret_cast<const bf16x8r *>(bs + (((g / GPC) * C::NJ + (g % GPC) * FG + n) * 64 + lane) * 16);
            };
            read_g(0, b[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) read_g(g + 1, b[(g + 1) & 1]);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const bf16x8r av = __builtin_bit_cast(bf16x8r, a[i][g / GPC]);
#pragma unroll
                    for (int n = 0; n < FG; ++n)
                        acc[i][(g % GPC) * FG + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[g & 1][n], acc[i][(g % GPC) * FG + n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    int idx0[MI][C::NIDX], idx1[MI][C::NIDX];
    bool on0[MI][C::NIDX], on1[MI][C::NIDX];
    f4 a0[MI][C::KC], a1[MI][C::KC];
    f4 breg[C::B_LOADS];

    // prologue: B(0) -> stage 0, idx(0) -> A(0), idx(1), B(1) -> registers
    load_idx(0, idx0);
    load_b(0, breg);
    if (steps > 1) load_idx(1, idx1);
    mask_idx(0, idx0);
    load_a(idx0, a0);
    tile_flags(idx0, on0);
    store_b(0, breg);
    if (steps > 1) load_b(1, breg);

    // Step s (unrolled by two; slot = parity):  barrier | idx(s+2) -> the slot idx(s) came from | A(s+1) from idx(s+1) | multiply
    // step s | ds_write B(s+1) | load B(s+2).  Every load is consumed one step after its issue: idx(s+2) at the top of step s+1,
    // A(s+1) by the MFMAs of step s+1, B(s+2) by the ds_write at the end of step s+1.
    auto step = [&](int s, auto has1, auto has2, const f4 (&acur)[MI][C::KC], f4 (&anxt)[MI][C::KC], const bool (&oncur)[MI][C::NIDX],
                    bool (&onnxt)[MI][C::NIDX], int (&idxcur)[MI][C::NIDX], int (&idxnxt)[MI][C::NIDX]) {
        const int stage = s & 1;
        __syncthreads();   // B(s) visible in `stage`; every wave is done reading the other stage (step s-1)
        if constexpr (decltype(has2)::value) load_idx(s + 2, idxcur);
        if constexpr (decltype(has1)::value) {
            mask_idx(s + 1, idxnxt);
            load_a(idxnxt, anxt);
            tile_flags(idxnxt, onnxt);
        }
        compute(stage, acur, oncur);
        if constexpr (decltype(has1)::value) store_b(stage ^ 1, breg);
        if constexpr (decltype(has2)::value) load_b(s + 2, breg);
    };
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    int s = 0;
    for (; s + 3 < steps; s += 2) {
        step(s, T, T, a0, a1, on0, on1, idx0, idx1);
        step(s + 1, T, T, a1, a0, on1, on0, idx1, idx0);
    }
    const int rem = steps - s;   // 1..3 (steps >= 1)
    if (rem == 3) {
        step(s, T, T, a0, a1, on0, on1, idx0, idx1);
        step(s + 1, T, F, a1, a0, on1, on0, idx1, idx0);
        step(s + 2, F, F, a0, a1, on0, on1, idx0, idx1);
    } else if (rem == 2) {
        step(s, T, F, a0, a1, on0, on1, idx0, idx1);
        step(s + 1, F, F, a1, a0, on1, on0, idx1, idx0);
    } else {
        step(s, F, F, a0, a1, on0, on1, idx0, idx1);
    }

    // epilogue: C/D layout row = 4 q + reg, column (of tile n) = r  ->  output column r * NJ + n
    float bv[C::NJ], s1[C::NJ], s2[C::NJ];
#pragma unroll
    for (int n = 0; n < C::NJ; ++n) {
        bv[n] = bias ? bias[r * C::NJ + n] : 0.f;
        s1[n] = 0.f;
        s2[n] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int tl = tile0 + wid + WAVES * i;
        if (tl >= tile_end) continue;   // wave-uniform
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rw = tl * 16 + 4 * q + reg;
            if (rw < n_out) {
                __bf16 v[C::NJ];
#pragma unroll
                for (int n = 0; n < C::NJ; ++n) {
                    v[n] = (__bf16)(acc[i][n][reg] + bv[n]);
                    const float f = (float)v[n];
                    s1[n] += f;
                    s2[n] += f * f;
                }
                __bf16 *dst = out + (int64_t)rw * COUT + r * C::NJ;
                if constexpr (C::NJ == 8) {
                    bf16x8r o;
#pragma unroll
                    for (int n = 0; n < 8; ++n) o[n] = v[n];
                    *reinterpret_cast<bf16x8r *>(dst) = o;
                } else if constexpr (C::NJ == 4) {
                    typedef __bf16 bf16x4r __attribute__((ext_vector_type(4)));
                    bf16x4r o;
#pragma unroll
                    for (int n = 0; n < 4; ++n) o[n] = v[n];
                    *reinterpret_cast<bf16x4r *>(dst) = o;
                } else if constexpr (C::NJ == 2) {
                    typedef __bf16 bf16x2r __attribute__((ext_vector_type(2)));
                    bf16x2r o;
                    o[0] = v[0]; o[1] = v[1];
                    *reinterpret_cast<bf16x2r *>(dst) = o;
                } else {
                    dst[0] = v[0];
                }
            }
        }
    }
    if (stats_partial) {   // block-uniform; fixed summation order: q groups, then waves 0..WAVES-1
#pragma unroll
        for (int n = 0; n < C::NJ; ++n) {
            s1[n] += __shfl_xor(s1[n], 16, 64); s1[n] += __shfl_xor(s1[n], 32, 64);
            s2[n] += __shfl_xor(s2[n], 16, 64); s2[n] += __shfl_xor(s2[n], 32, 64);
        }
        float *red = reinterpret_cast<float *>(smem);   // [WAVES][2][COUT]
        __syncthreads();                                // every wave is out of the step loop: the weight stages are free
        if (q == 0) {
#pragma unroll
            for (int n = 0; n < C::NJ; ++n) {
                red[(wid * 2 + 0) * COUT + r * C::NJ + n] = s1[n];
                red[(wid * 2 + 1) * COUT + r * C::NJ + n] = s2[n];
            }
        }
        __syncthreads();
        for (int e = t; e < 2 * COUT; e += C::THREADS) {
            const int which = e / COUT, col = e - which * COUT;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s += red[(w * 2 + which) * COUT + col];
            stats_partial[((int64_t)block * 2 + which) * COUT + col] = s;
        }
    }
}

// ---- launch plan -------------------------------------------------------------------------------------------------
struct RgPlan {
    int mi, waves, tiles_per_block;
    unsigned grid;
};

static int rg_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

RgPlan rg_plan(int64_t n_out, int kvol, int cin, int cout) {
    (void)kvol; (void)cin;
    int mi = cout == 128 ? 2 : 4, waves = 8, tpb = 0;
    if (const char *ov = getenv("S2D_RG_PLAN")) {   // tuning hook: "mi,waves[,tiles_per_block]"
        int a = 0, b = 0, c = 0;
        const int got = sscanf(ov, "%d,%d,%d", &a, &b, &c);
        if (got >= 2 && (a == 1 || a == 2 || a == 4) && (b == 4 || b == 8) && !(cout == 128 && a == 4)) {
            mi = a; waves = b;
            if (got == 3 && c >= 1 && c <= a * b) tpb = c;
        }
    }
    const int64_t tiles = ceil_div(n_out, 16);
    const int cap = mi * waves;
    if (!tpb) {
        // whole rounds of the chip: the smallest number of rounds that fits, then equal shares
        const int64_t per_round = (int64_t)rg_cus() * (8 / waves);
        const int64_t rounds = std::max<int64_t>(1, ceil_div(tiles, per_round * cap));
        tpb = (int)std::min<int64_t>(cap, std::max<int64_t>(1, ceil_div(tiles, per_round * rounds)));
    }
    return RgPlan{mi, waves, tpb, (unsigned)std::max<int64_t>(1, ceil_div(tiles, tpb))};
}

template <int CIN, int COUT, int MI, int WAVES>
static int rg_launch(const RgPlan &p, const __bf16 *in, int64_t n_in, const __bf16 *wpack, const float *bias, const int32_t *nbr, int n_out, int kvol,
                     __bf16 *out, float *stats, hipStream_t st) {
    typedef RgCfg<CIN, COUT, MI, WAVES> C;
    auto kern = spconv_rg_kernel<CIN, COUT, MI, WAVES>;
    static bool attr_done = false;
    if (!attr_done && C::LDS > 48 * 1024) {
        S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr_done = true;
    }
    static int dbg = -1;
    if (dbg < 0) dbg = getenv("S2D_RG_DEBUG") ? atoi(getenv("S2D_RG_DEBUG")) : 0;   // ablation switches for tools/spconv_kernel_bench.py
    hipLaunchKernelGGL(kern, dim3(xcd_grid(p.grid)), dim3(C::THREADS), C::LDS, st, in, (unsigned)(n_in * CIN * 2), wpack, bias, nbr, n_out, kvol,
                       p.tiles_per_block, out, stats, dbg);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

template <int CIN, int COUT>
static int rg_dispatch_plan(const RgPlan &p, const __bf16 *in, int64_t n_in, const __bf16 *wpack, const float *bias, const int32_t *nbr, int n_out,
                            int kvol, __bf16 *out, float *stats, hipStream_t st) {
#define RG_CASE(MI_, W_) \
    if (p.mi == MI_ && p.waves == W_) return rg_launch<CIN, COUT, MI_, W_>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st)
    RG_CASE(2, 8);
    RG_CASE(2, 4);
    RG_CASE(1, 8);
    if constexpr (COUT != 128) {
        RG_CASE(4, 8);
        RG_CASE(4, 4);
    }
#undef RG_CASE
    return S2D_ERR_UNSUPPORTED;
}

template <int CIN>
static int rg_dispatch_cout(int cout, const RgPlan &p, const __bf16 *in, int64_t n_in, const __bf16 *wpack, const float *bias, const int32_t *nbr,
                            int n_out, int kvol, __bf16 *out, float *stats, hipStream_t st) {
    switch (cout) {
        case 16: return rg_dispatch_plan<CIN, 16>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
        case 32: return rg_dispatch_plan<CIN, 32>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
        case 64: return rg_dispatch_plan<CIN, 64>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
        case 128: return rg_dispatch_plan<CIN, 128>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
    }
    return S2D_ERR_UNSUPPORTED;
}

int rg_run(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias, const int32_t *nbr, int64_t n_out, int kvol, int cin,
           int cout, void *out_feat, float *stats_partial, hipStream_t st) {
    const RgPlan plan = rg_plan(n_out, kvol, cin, cout);
    const __bf16 *in = (const __bf16 *)in_feat, *wp = (const __bf16 *)packed_weight;
    __bf16 *out = (__bf16 *)out_feat;
    switch (cin) {
        case 16: return rg_dispatch_cout<16>(cout, plan, in, n_in, wp, bias, nbr, (int)n_out, kvol, out, stats_partial, st);
        case 32: return rg_dispatch_cout<32>(cout, plan, in, n_in, wp, bias, nbr, (int)n_out, kvol, out, stats_partial, st);
        case 64: return rg_dispatch_cout<64>(cout, plan, in, n_in, wp, bias, nbr, (int)n_out, kvol, out, stats_partial, st);
        case 128: return rg_dispatch_cout<128>(cout, plan, in, n_in, wp, bias, nbr, (int)n_out, kvol, out, stats_partial, st);
    }
    return S2D_ERR_UNSUPPORTED;
}

}  // namespace s2d
