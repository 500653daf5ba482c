// Sparse-conv implicit GEMM, bf16 storage, REGISTER-GATHER form (r03; replaces the LDS-staged A tile of spconv_s16.hip).
// (spconv.ops.indice_conv / indice_conv_backward; call sites det3d/models/backbones/scn.py:104-152)
//
// What bounds these kernels (r03 measurements, tools/spconv_kernel_bench.py with the S2D_RG_DEBUG ablation switches and per-step
// s_memtime stamps; 128 -> 128 channels, 47 890 rows, 738 674 pairs): seven structurally different versions - A through LDS-DMA,
// A in registers with one or two waves per SIMD, one/two/three-slot rings, loads clustered or threaded through the MFMAs - all
// took 51-57 us at an effective 1.9 GHz, while the MFMAs alone need 22 us.  The common term is the traffic every CU pulls
// through its L1 from the XCD's L2: the gathered rows (189 MB) plus the weight image once per workgroup (250 x 884 KB = 221 MB):
// 410 MB / 52 us = 7.9 TB/s; the 64 -> 128 strided layer (165 MB, 27 us) and the 64 -> 64 SubM layer (354 MB, 53 us) land at
// 6-7 TB/s too (the same "~10 TB/s of L2 -> CU" r01 measured on the LDS-DMA path).  Within that bound this kernel is what wins
// at 128 channels:
//   * the A operand never touches LDS.  Lane (r = lane & 15, q = lane >> 4) of v_mfma_f32_16x16x32_bf16 holds
//     A[row r][k = 8q..8q+7] = 16 contiguous bytes of the gathered input row, so the fragment IS one `buffer_load_dwordx4` from
//     feature row nbr[offset][row r]; a missing neighbour is an out-of-range buffer offset (reads zero, moves no data, needs
//     no zero page and no branch).
//   * a wave owns MI 16-row MFMA tiles x ALL output columns; a workgroup is 8 waves (two per SIMD: one wave's memory-issue
//     back-pressure and LDS waits are covered by its partner's MFMAs) working on tiles_per_block consecutive tiles dealt
//     round-robin; waves with fewer tiles run a leaner instantiation (rg_wave<NT>).
//   * one step = 128 K-elements = 128 / CIN kernel offsets.  Two register slots (step parity) refilled IN PLACE: as soon as
//     the MFMAs of a K chunk of step s have been issued its registers are reloaded with the chunk of step s+2; gather
//     indices are requested four steps ahead.
//   * only the weight slab of a step (128 x COUT bf16, packed in fragment order) goes through LDS: global -> registers in
//     step s -> ds_write in step s+1 -> read by every wave in step s+2; two LDS stages, one barrier per step.
//   * no conditional memory operation and no peeled tail: the step count is padded to even with phantom steps (indices -1,
//     clamped slab), so hipcc's vmcnt bookkeeping sees one straight loop body and keeps every load class in flight.
//   * r03: the accumulators were updated by asm MFMAs ("+a") because the builtin made hipcc keep a second accumulator set and copy ~120
//     registers per loop iteration.  End of r04: the asm form is wrong in one instantiation (RG_ASM_MFMA below) and the builtin is no
//     longer slower; scheduling barriers pin the ds_read / memory / MFMA interleave either way.
// The epilogue: bias, one bf16 rounding, 2*NJ-byte row stores, optional per-workgroup (sum, sum of squares) rows for the
// BatchNorm1d that follows.
#include "s2d_common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace s2d {

typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));

constexpr int RG_KSTEP = 128;   // K elements per step
#ifndef RG_ASM_MFMA
// 0 (default since the end of r04): the compiler's MFMA builtin.  1: the r03 inline-asm MFMAs with the accumulator tied in place ("+a").  The asm form is WRONG
// in the <128, 128, MI = 2, WAVES = 8> instantiation: the second tile of every wave comes out wrong above 32 768 output rows (DESIGN rule 31) - an inline asm
// statement is opaque to hipcc's hazard recognizer, which therefore cannot keep a later write of an operand register (fragment reads, in-place refills) away from
// an MFMA that is still reading it; the instantiation with the highest register pressure is where it bites.  With the builtin every instantiation is right at
// the benchmark's row counts and no slower (ROCm 7.2's hipcc no longer copies the accumulator set per iteration, the reason r03 went to asm): 128 -> 128 at
// 47 890 rows 43.9 us (asm: 45.7 us and wrong), 64 -> 128 26.1 (28.0), 128 -> 64 61.8 (63.6).
#define RG_ASM_MFMA 0
#endif
#ifndef RG_BBURST
#define RG_BBURST 1   // 1: all weight-slab pieces of a step are stored / re-requested in its first group (all 32 KiB in flight at once)
#endif
__host__ __device__ inline int rg_steps(int cin, int kvol) { return (kvol * cin + RG_KSTEP - 1) / RG_KSTEP; }

template <int CIN, int COUT, int MI_, int WAVES_>
struct RgCfg {
    static constexpr int MI = MI_, WAVES = WAVES_;
    static constexpr int OPS = RG_KSTEP / CIN;                // kernel offsets per step (1, 2, 4, 8)
    static constexpr int KC = RG_KSTEP / 32;                  // 32-deep MFMA K chunks per step
    static constexpr int NIDX = CIN == 128 ? 1 : (CIN == 64 ? 2 : 4);   // gather indices per lane, tile and step
    static constexpr int NJ = COUT / 16;                      // 16-column MFMA tiles (a wave owns all of them)
    static constexpr int THREADS = WAVES * 64;
    static constexpr int B_BYTES = RG_KSTEP * COUT * 2;       // weight slab of one step
    static constexpr int B_PIECES = B_BYTES / 16;
    static constexpr int B_LOADS = (B_PIECES + THREADS - 1) / THREADS;   // 16-byte pieces per thread
    static constexpr bool B_FULL = B_PIECES % THREADS == 0;
    static constexpr int CAP = WAVES * MI;                    // 16-row tiles per workgroup
    static constexpr size_t LDS = 2 * (size_t)B_BYTES > (size_t)WAVES * 2 * COUT * 4 ? 2 * (size_t)B_BYTES : (size_t)WAVES * 2 * COUT * 4;
};

// packed weight image: [step][c (4)][nt = COUT/16][lane = q*16 + r][e (8)]   (one step = 128*COUT bf16, contiguous)
//   K element kk = 32 c + 8 q + e  ->  kernel offset step*(128/CIN) + kk / CIN, input channel kk % CIN;  column r * NJ + nt
// (a lane's NJ accumulator columns are consecutive in the output row).  Offsets past kvol are zero.
// source w: [K][cin][cout] fp32, or [K][cout][cin] when transpose (data gradient), offsets mirrored when flip.
__device__ __forceinline__ void rg_pack_element(const float *__restrict__ w, int kvol, int cin, int cout, int transpose, int flip, int64_t i,
                                                __bf16 *__restrict__ out) {
    const int64_t total = (int64_t)rg_steps(cin, kvol) * RG_KSTEP * cout;
    if (i >= total) return;
    const int nj = cout / 16;
    int64_t x = i;
    const int e = x % 8; x /= 8;
    const int r = x % 16; x /= 16;
    const int q = x % 4; x /= 4;
    const int nt = x % nj; x /= nj;
    const int c = x % 4; x /= 4;
    const int step = (int)x;
    const int kk = 32 * c + 8 * q + e;
    const int k = step * (RG_KSTEP / cin) + kk / cin, ch = kk % cin;
    const int co = r * nj + nt;
    float v = 0.f;
    if (k < kvol) {
        const int ks = flip ? kvol - 1 - k : k;
        v = transpose ? w[((int64_t)ks * cout + co) * cin + ch] : w[((int64_t)ks * cin + ch) * cout + co];
    }
    out[i] = (__bf16)v;
}

__global__ __launch_bounds__(256) void rg_pack_kernel(const float *__restrict__ w, int kvol, int cin, int cout, int transpose, int flip,
                                                      __bf16 *__restrict__ out) {
    rg_pack_element(w, kvol, cin, cout, transpose, flip, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, out);
}

// both operands of a layer in one launch: blockIdx.y = 0 the forward image [cin -> cout], 1 the data-gradient image [cout -> cin]
__global__ __launch_bounds__(256) void rg_pack_pair_kernel(const float *__restrict__ w, int kvol, int cin, int cout, int flip_d,
                                                           __bf16 *__restrict__ out_f, __bf16 *__restrict__ out_d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) rg_pack_element(w, kvol, cin, cout, 0, 0, i, out_f);
    else rg_pack_element(w, kvol, cout, cin, 1, flip_d, i, out_d);
}

// The work of one wave with MI valid tile slots (the workgroup's waves may run different instantiations: a workgroup of
// tiles_per_block < WAVES * MI_max tiles deals its waves unequal tile counts; every instantiation executes the same barriers)
// SORTED (r06): the rows arrive grouped by neighbour mask (csrc/rulebook_sort.hip): `nbr` is the permuted map nbr_perm[k][j], `pmask[j]` the
// mask of sorted row j, `perm[j]` its canonical row (where the result is stored).  The workgroup multiplies only the kernel offsets that
// occur in the OR of its rows' masks - its step list is the set bits of that union, walked by two scalar cursors (one for the gather
// indices, four steps ahead, one for the weight slabs, two steps ahead) - and a wave skips the gathers and MFMAs of a tile whose own
// 16-row union lacks the step's offset(s).
template <int CIN, int COUT, int MI, int WAVES, int DBG, bool SORTED>
__device__ __forceinline__ void rg_wave(const __bf16 *__restrict__ in, unsigned in_bytes, const __bf16 *__restrict__ wpack,
                                                        const float *__restrict__ bias, const int32_t *__restrict__ nbr, int n_out, int kvol,
                                                        int tiles_per_block, __bf16 *__restrict__ out, float *__restrict__ stats_partial,
                                                        int dbg_arg, long long *__restrict__ trace, const int32_t *__restrict__ perm,
                                                        const uint32_t *__restrict__ pmask) {
    typedef RgCfg<CIN, COUT, MI, WAVES> C;
    const int dbg = DBG == 1 ? dbg_arg : 0;   // DBG: 0 production, 1 ablation switches + stamps, 2 stamps only (the production instruction stream)
    // dbg & 32: wave 0 of every workgroup records s_memtime at kernel entry, after the prologue, after every step and at the end
    auto stamp = [&](int k) {
        if (DBG && (dbg_arg & 32) && trace && threadIdx.x == 0) {
            trace[(int64_t)blockIdx.x * 64 + k] = (long long)__builtin_readcyclecounter();
            if (k == 0) trace[(int64_t)blockIdx.x * 64 + 60] = (long long)__builtin_amdgcn_s_memrealtime();   // 100 MHz, chip-wide
            if (k == 63) trace[(int64_t)blockIdx.x * 64 + 61] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);   // ablation switches (tools/spconv_kernel_bench.py): compiled in for two shapes only
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
    // SORTED: the union of the workgroup's row masks (every wave computes it for itself: <= 4 rows per lane, no barrier) and of each of
    // this wave's tiles; cnt = offsets to multiply
    uint32_t bmask = 0, tmask[MI];
    if constexpr (SORTED) {
        const int r0 = tile0 * 16, r1 = min(tile_end * 16, n_out);
        for (int rr = r0 + lane; rr < r1; rr += 64) bmask |= pmask[rr];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) bmask |= (uint32_t)__shfl_xor((int)bmask, o, 64);
        bmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)bmask);
    }
    const int cnt = SORTED ? __builtin_popcount(bmask) : kvol;
    const int steps = SORTED ? (cnt + C::OPS - 1) / C::OPS : rg_steps(CIN, kvol);

    // this lane's output row per tile slot (-1: no such tile / row past the end); tiles are dealt round-robin to the waves
    int row[MI], rowc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int tl = tile0 + wid + WAVES * i;
        const int rw = tl * 16 + r;
        row[i] = (tl < tile_end && rw < n_out) ? rw : -1;
        rowc[i] = row[i] < 0 ? 0 : row[i];
        if constexpr (SORTED) {
            uint32_t m = row[i] >= 0 ? pmask[rowc[i]] : 0u;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) m |= (uint32_t)__shfl_xor((int)m, o, 64);
            tmask[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
        } else {
            tmask[i] = 0xffffffffu;
        }
    }
    // SORTED: the offsets of the step list, handed out in order by two cursors (remaining bits of the union)
    uint32_t cur_idx = bmask, cur_b = bmask;
    auto next_off = [&](uint32_t &cur) -> int {   // next offset of the list, -1 past its end (wave-uniform)
        if (cur == 0u) return -1;
        const int k = __builtin_ctz(cur);
        cur &= cur - 1u;
        return k;
    };
    const __amdgpu_buffer_rsrc_t rin = buf_rsrc(in, in_bytes);
    // chunk c of a step: which of the step's NIDX index registers it gathers with, its kernel offset, and the byte offset of this
    // lane's 16 bytes inside the gathered row
    auto chunk_j = [&](int c) -> int { return CIN == 128 ? 0 : (CIN == 64 ? c >> 1 : c); };
    auto idx_off = [&](int s, int j) -> int { return CIN == 16 ? 8 * s + 2 * j + (q >> 1) : s * C::OPS + j; };
    auto chunk_byte = [&](int c) -> unsigned {
        return CIN == 128 ? (unsigned)(64 * c + 16 * q) : (CIN == 64 ? (unsigned)(64 * (c & 1) + 16 * q) : (CIN == 32 ? (unsigned)(16 * q) : (unsigned)(16 * (q & 1))));
    };

    typedef float f4 __attribute__((ext_vector_type(4)));
    // koff[j]: the kernel offset behind index register j of the slot (SORTED; -1 = past the list), kept beside the indices
    auto load_idx = [&](int s, int (&idx)[MI][C::NIDX], int (&koff)[C::NIDX]) {   // unconditional, clamped; masked by mask_idx when consumed
#pragma unroll
        for (int j = 0; j < C::NIDX; ++j) {
            int k;
            if constexpr (SORTED) {
                k = koff[j] = next_off(cur_idx);
                k = k < 0 ? 0 : k;
            } else {
                k = min(idx_off(s, j), kvol - 1);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) idx[i][j] = nbr[(int64_t)k * n_out + rowc[i]];
        }
    };
    auto mask_idx = [&](int s, int (&idx)[MI][C::NIDX], const int (&koff)[C::NIDX]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NIDX; ++j) {
                if (dbg & 1) idx[i][j] = rowc[i];   // ablation: sequential rows, every neighbour present
                const bool live = SORTED ? koff[j] >= 0 : idx_off(s, j) < kvol;
                if (!(row[i] >= 0 && live)) idx[i][j] = -1;
            }
    };
    // does tile slot i need the gathers / MFMAs of K chunk c of a step whose offsets are koff?  (wave-uniform; always for the plain kernel)
    auto tile_live = [&](int i, int c, const int (&koff)[C::NIDX]) -> bool {
        if constexpr (!SORTED) return true;
        const int k = koff[chunk_j(c)];
        return k >= 0 && ((tmask[i] >> k) & 1u);
    };
    if constexpr (SORTED) {
        if (dbg_arg & 128) {   // A/B switch (S2D_RG_SORTED_DEBUG=128): no per-tile skipping, the workgroup's step list only
#pragma unroll
            for (int i = 0; i < MI; ++i) tmask[i] = 0xffffffffu;
        }
    }
    auto load_a = [&](const int (&idx)[MI][C::NIDX], const int (&koff)[C::NIDX], f4 (&a)[MI][C::KC]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int c = 0; c < C::KC; ++c) {
                if (!tile_live(i, c, koff)) continue;   // (its MFMAs are skipped as well)
                const int j = idx[i][chunk_j(c)];
                const unsigned off = (j >= 0 && !(dbg & 4)) ? (unsigned)j * (unsigned)(CIN * 2) + chunk_byte(c) : BUF_OOB;
                a[i][c] = buf_load4(rin, off, 0);
            }
    };
    // weight slab of a step: plain kernel = slab `s` of the image; SORTED = the (half-)slabs of the step's offsets, taken from the slab cursor
    // (the image is [offset][c][nt][lane][e] for every CIN: a step of OPS offsets is OPS consecutive pieces of B_BYTES / OPS bytes)
    constexpr int SUB_BYTES = C::B_BYTES / C::OPS, SUB_PIECES = C::B_PIECES / C::OPS;
    const char *bsub[C::OPS];
    auto b_sources = [&](int s) {
#pragma unroll
        for (int j = 0; j < C::OPS; ++j) {
            int k;
            if constexpr (SORTED) {
                k = next_off(cur_b);
                k = k < 0 ? 0 : k;
            } else {
                k = min(s, steps - 1) * C::OPS + j;
            }
            bsub[j] = reinterpret_cast<const char *>(wpack) + (int64_t)((dbg & 8) ? j : k) * SUB_BYTES;
        }
    };
    auto b_piece_src = [&](int u) -> const char * {
        const int p = t + C::THREADS * u;
        if constexpr (C::OPS == 1) return bsub[0] + (size_t)p * 16;
        else {
            const int j = p / SUB_PIECES;
            const char *base = bsub[0];
#pragma unroll
            for (int jj = 1; jj < C::OPS; ++jj) base = j == jj ? bsub[jj] : base;
            return base + (size_t)(p - j * SUB_PIECES) * 16;
        }
    };
    auto load_b = [&](int s, f4 (&breg)[C::B_LOADS]) {
        b_sources(s);
#pragma unroll
        for (int u = 0; u < C::B_LOADS; ++u)
            if (C::B_FULL || t + C::THREADS * u < C::B_PIECES) breg[u] = *reinterpret_cast<const f4 *>(b_piece_src(u));
    };
    auto store_b = [&](int stage, const f4 (&breg)[C::B_LOADS]) {
#pragma unroll
        for (int u = 0; u < C::B_LOADS; ++u)
            if (C::B_FULL || t + C::THREADS * u < C::B_PIECES) *reinterpret_cast<f4 *>(smem + stage * C::B_BYTES + (size_t)(t + C::THREADS * u) * 16) = breg[u];
    };

    f32x4r acc[MI][C::NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int n = 0; n < C::NJ; ++n) acc[i][n] = f32x4r{0.f, 0.f, 0.f, 0.f};

    // Two ring slots (step parity), refilled IN PLACE: slot p holds A(s) while step s (parity p) multiplies; as soon as the MFMAs
    // of K chunk c have been issued, its registers are reloaded with A(s+2)[.][c], so every fragment is requested two steps
    // before its use without a third register set (a three-slot ring made hipcc shuffle 116 registers between the VGPR and AGPR
    // files per loop iteration).  idx_p holds the gather indices of step s+2 during step s and is reloaded with those of step
    // s+4 after the last refill.  The loop is unrolled by two = the parity of the LDS weight stage.
    int idx0[MI][C::NIDX], idx1[MI][C::NIDX];
    // offsets behind the slot's CURRENT fragments (kc*) and behind the indices waiting in idx* for the refill (kn*)
    int kc0[C::NIDX], kc1[C::NIDX], kn0[C::NIDX], kn1[C::NIDX];
#pragma unroll
    for (int j = 0; j < C::NIDX; ++j) kc0[j] = kc1[j] = kn0[j] = kn1[j] = 0;
    f4 a0[MI][C::KC], a1[MI][C::KC];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int c = 0; c < C::KC; ++c) a0[i][c] = a1[i][c] = f4{0.f, 0.f, 0.f, 0.f};   // (SORTED: a skipped gather leaves its registers as they are)
    f4 breg[C::B_LOADS];
    const int steps2 = (steps + 1) / 2 * 2;

    // prologue: idx(0), idx(1); B(0) -> stage 0; A(0), A(1); B(1) -> registers; idx(2), idx(3)
    load_idx(0, idx0, kc0);
    load_idx(1, idx1, kc1);
    load_b(0, breg);
    mask_idx(0, idx0, kc0);
    load_a(idx0, kc0, a0);
    mask_idx(1, idx1, kc1);
    load_a(idx1, kc1, a1);
    store_b(0, breg);
    load_b(1, breg);
    load_idx(2, idx0, kn0);
    load_idx(3, idx1, kn1);
    stamp(1);

    // Step s (slot = parity):  barrier | NG groups, group g = { ds_read the weight fragments of group g+1 | its share of the
    // step's memory tasks: ds_write B(s+1) + reload B(s+2) pieces, refill of the A chunk whose MFMAs were issued in the groups
    // before | MI*FG MFMAs on the fragments read during group g-1 } closed by a scheduling barrier | refill of the last chunk,
    // idx(s+4).  With one wave per SIMD nothing else fills the matrix pipe while this wave issues memory instructions or waits
    // for a fragment, and hipcc left to itself puts a step's loads in front of its MFMAs and every fragment read right in front
    // of its first use; the scheduling barriers pin the interleave written here.
    constexpr int FG = C::NJ < 4 ? C::NJ : 4, GPC = C::NJ / FG, NG = C::KC * GPC;
    constexpr int BPER = (C::B_LOADS + NG - 1) / NG;   // B pieces per group
    auto step = [&](int s, auto parity, f4 (&a)[MI][C::KC], int (&idx)[MI][C::NIDX], int (&kc)[C::NIDX], int (&kn)[C::NIDX]) {
        constexpr int stage = decltype(parity)::value;
        __syncthreads();   // B(s) visible in `stage`; every wave is done reading the other stage (step s-1)
        mask_idx(s + 2, idx, kn);
        const char *bs = smem + stage * C::B_BYTES;
        char *bw = smem + (stage ^ 1) * C::B_BYTES;
        b_sources(s + 2);
        auto b_task = [&](int u) {
            if (C::B_FULL || t + C::THREADS * u < C::B_PIECES) {
                *reinterpret_cast<f4 *>(bw + (size_t)(t + C::THREADS * u) * 16) = breg[u];                 // B(s+1) -> LDS
                if (!(dbg & 64)) breg[u] = *reinterpret_cast<const f4 *>(b_piece_src(u));                  // B(s+2) -> registers
            }
        };
        // which (tile, chunk) MFMAs this step issues: decided from the offsets behind the CURRENT fragments, before the refills retarget kc
        bool live[MI][C::KC];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int c = 0; c < C::KC; ++c) live[i][c] = tile_live(i, c, kc);
        auto refill = [&](int c) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (!tile_live(i, c, kn)) continue;
                const int j = idx[i][chunk_j(c)];
                const unsigned off = (j >= 0 && !(dbg & 4)) ? (unsigned)j * (unsigned)(CIN * 2) + chunk_byte(c) : BUF_OOB;
                a[i][c] = buf_load4(rin, off, 0);
            }
        };
        f4 b[2][FG];
        auto read_g = [&](int g, f4 (&bb)[FG]) {
#pragma unroll
            for (int n = 0; n < FG; ++n)
                bb[n] = *reinterpret_cast<const f4 *>(bs + (((g / GPC) * C::NJ + (g % GPC) * FG + n) * 64 + lane) * 16);
        };
        read_g(0, b[0]);
        if (dbg & 16) read_g(0, b[1]);   // ablation: one fragment group per step
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG && !(dbg & 16)) read_g(g + 1, b[(g + 1) & 1]);
#pragma unroll
            for (int u = 0; u < C::B_LOADS; ++u)
                if ((RG_BBURST ? 0 : u / BPER) == g) b_task(u);
            if (g > 0 && g % GPC == 0) refill(g / GPC - 1);
            if (!(dbg & 2)) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (SORTED && !live[i][g / GPC]) continue;   // wave-uniform: this tile has no neighbour at the chunk's offset
#pragma unroll
                    for (int n = 0; n < FG; ++n)
                        // in-place accumulate in the AGPR file, written as asm: with the builtin hipcc gave the MFMAs a second
                        // accumulator set (dst != src C) and copied ~120 registers back per loop iteration
#if RG_ASM_MFMA
                        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][(g % GPC) * FG + n]) : "v"(a[i][g / GPC]), "v"(b[g & 1][n]));
#else
                        acc[i][(g % GPC) * FG + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8r, a[i][g / GPC]), __builtin_bit_cast(bf16x8r, b[g & 1][n]),
                                                                                               acc[i][(g % GPC) * FG + n], 0, 0, 0);
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        refill(C::KC - 1);
#pragma unroll
        for (int j = 0; j < C::NIDX; ++j) kc[j] = kn[j];   // the slot's fragments now belong to step s + 2
        load_idx(s + 4, idx, kn);
        if (s + 2 < 60) stamp(s + 2);
    };
    for (int s = 0; s < steps2; s += 2) {
        step(s, std::integral_constant<int, 0>{}, a0, idx0, kc0, kn0);
        step(s + 1, std::integral_constant<int, 1>{}, a1, idx1, kc1, kn1);
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the accumulators were written by asm MFMAs: no hazard bookkeeping by the compiler
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
                const int64_t orow = SORTED ? (int64_t)perm[rw] : (int64_t)rw;   // sorted position -> canonical row
                __bf16 *dst = out + orow * COUT + r * C::NJ;
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
    stamp(63);
}

template <int CIN, int COUT, int MI, int WAVES, int DBG, bool SORTED = false>
__global__ __launch_bounds__(WAVES * 64) void spconv_rg_kernel(const __bf16 *__restrict__ in, unsigned in_bytes, const __bf16 *__restrict__ wpack,
                                                               const float *__restrict__ bias, const int32_t *__restrict__ nbr, int n_out,
                                                               int kvol, int tiles_per_block, __bf16 *__restrict__ out,
                                                               float *__restrict__ stats_partial, int dbg_arg, long long *__restrict__ trace,
                                                               const int32_t *__restrict__ perm, const uint32_t *__restrict__ pmask) {
    static_assert(!SORTED || RgCfg<CIN, COUT, MI, WAVES>::NIDX == RgCfg<CIN, COUT, MI, WAVES>::OPS, "sorted rows: 64 / 128 input channels");
    // valid tile slots of this wave (tiles are dealt round-robin: slot i = tile0 + wave + WAVES * i)
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total_tiles = (n_out + 15) >> 4;
    const int tile0 = xcd_tile(blockIdx.x, gridDim.x) * tiles_per_block;
    if (tile0 >= total_tiles) return;
    const int tile_end = min(total_tiles, tile0 + tiles_per_block);
    int nt = 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) nt += (tile0 + wid + WAVES * i < tile_end) ? 1 : 0;
    nt = __builtin_amdgcn_readfirstlane(nt);
#define RG_ARGS in, in_bytes, wpack, bias, nbr, n_out, kvol, tiles_per_block, out, stats_partial, dbg_arg, trace, perm, pmask
    if constexpr (MI >= 4) { if (nt == 4) { rg_wave<CIN, COUT, 4, WAVES, DBG, SORTED>(RG_ARGS); return; } }
    if constexpr (MI >= 3) { if (nt == 3) { rg_wave<CIN, COUT, 3, WAVES, DBG, SORTED>(RG_ARGS); return; } }
    if constexpr (MI >= 2) { if (nt == 2) { rg_wave<CIN, COUT, 2, WAVES, DBG, SORTED>(RG_ARGS); return; } }
    rg_wave<CIN, COUT, 1, WAVES, DBG, SORTED>(RG_ARGS);   // also a wave without a tile: one phantom slot, same barriers
#undef RG_ARGS
}

static long long *g_rg_trace = nullptr;   // device buffer [grid][64] for the dbg & 32 timestamps (s2d_debug_rg_trace)
void rg_set_trace(void *p) { g_rg_trace = (long long *)p; }

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

// rows per workgroup: the fewest whole rounds of one workgroup per CU that cover the tiles, then the tile count per workgroup
// that spreads them evenly (a 47 890-row stage = 2 994 tiles runs as 250 workgroups of 12 tiles, not 187 of 16)
RgPlan rg_plan(int64_t n_out, int kvol, int cin, int cout) {
    (void)kvol; (void)cin;
    const int64_t tiles = ceil_div(n_out, 16);
    // measured on the bench scene (tools/spconv_kernel_bench.py, r03): two waves per SIMD with two tiles each win or tie at every
    // shape with 128 input or output channels (128->128: 51 us vs 54 with one wave of three tiles per SIMD; 64->128: 26 vs 29;
    // 128->64: 55-61 vs 63)
    int mi = 2, waves = 8, tpb = 0;
    if (const char *ov = getenv("S2D_RG_PLAN")) {   // tuning hook: "mi,waves[,tiles_per_block]"
        int a = 0, b = 0, c = 0;
        const int got = sscanf(ov, "%d,%d,%d", &a, &b, &c);
        if (got >= 2 && a >= 1 && a <= 4 && (b == 4 || b == 8) && !(b == 8 && a > 2 && cout > 64)) {
            mi = a; waves = b;
            if (got == 3 && c >= 1 && c <= a * b) tpb = c;
        }
    }
    if (!tpb) {   // the fewest whole rounds of one workgroup per CU, then equal shares (unequal tile counts per wave: rg_wave<NT>)
        const int64_t per_round = (int64_t)rg_cus();
        const int64_t rounds = std::max<int64_t>(1, ceil_div(tiles, per_round * waves * mi));
        tpb = (int)std::min<int64_t>((int64_t)waves * mi, std::max<int64_t>(1, ceil_div(tiles, per_round * rounds)));
    }
    return RgPlan{mi, waves, tpb, (unsigned)std::max<int64_t>(1, ceil_div(tiles, tpb))};
}

template <int CIN, int COUT, int MI, int WAVES>
static int rg_launch(const RgPlan &p, const __bf16 *in, int64_t n_in, const __bf16 *wpack, const float *bias, const int32_t *nbr, int n_out, int kvol,
                     __bf16 *out, float *stats, hipStream_t st) {
    typedef RgCfg<CIN, COUT, MI, WAVES> C;
    static int dbg = -1;
    if (dbg < 0) dbg = getenv("S2D_RG_DEBUG") ? atoi(getenv("S2D_RG_DEBUG")) : 0;   // ablation switches for tools/spconv_kernel_bench.py
    auto kern = spconv_rg_kernel<CIN, COUT, MI, WAVES, 0>;
    if constexpr (CIN == 128 && COUT == 128 && (MI == 3 || (WAVES == 8 && MI == 2))) {
        if (dbg == 32) kern = spconv_rg_kernel<CIN, COUT, MI, WAVES, 2>;
        else if (dbg) kern = spconv_rg_kernel<CIN, COUT, MI, WAVES, 1>;
    }
    static bool attr_done[2] = {false, false};
    if (!attr_done[dbg != 0] && C::LDS > 48 * 1024) {
        S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr_done[dbg != 0] = true;
    }
    hipLaunchKernelGGL(kern, dim3(xcd_grid(p.grid)), dim3(C::THREADS), C::LDS, st, in, (unsigned)(n_in * CIN * 2), wpack, bias, nbr, n_out, kvol,
                       p.tiles_per_block, out, stats, dbg, g_rg_trace, (const int32_t *)nullptr, (const uint32_t *)nullptr);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// rows grouped by neighbour mask (csrc/rulebook_sort.hip): one plan shape (two tiles per wave, eight waves), 64 -> 64 and 128 -> 128 - the
// submanifold layers of the two wide stages
template <int CIN, int COUT>
static int rg_launch_sorted(const RgPlan &p, const __bf16 *in, int64_t n_in, const __bf16 *wpack, const float *bias, const int32_t *nbr_perm,
                            const int32_t *perm, const uint32_t *pmask, int n_out, int kvol, __bf16 *out, float *stats, hipStream_t st) {
    typedef RgCfg<CIN, COUT, 2, 8> C;
    auto kern = spconv_rg_kernel<CIN, COUT, 2, 8, 0, true>;
    static bool attr_done = false;
    if (!attr_done && C::LDS > 48 * 1024) {
        S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr_done = true;
    }
    static int sdbg = -1;
    if (sdbg < 0) sdbg = getenv("S2D_RG_SORTED_DEBUG") ? atoi(getenv("S2D_RG_SORTED_DEBUG")) : 0;
    hipLaunchKernelGGL(kern, dim3(xcd_grid(p.grid)), dim3(C::THREADS), C::LDS, st, in, (unsigned)(n_in * CIN * 2), wpack, bias, nbr_perm, n_out, kvol,
                       p.tiles_per_block, out, stats, sdbg, (long long *)nullptr, perm, pmask);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

template <int CIN, int COUT>
static int rg_dispatch_plan(const RgPlan &p, const __bf16 *in, int64_t n_in, const __bf16 *wpack, const float *bias, const int32_t *nbr, int n_out,
                            int kvol, __bf16 *out, float *stats, hipStream_t st) {
    if (p.waves == 8) {
        switch (p.mi) {
            case 1: return rg_launch<CIN, COUT, 1, 8>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
            case 2: return rg_launch<CIN, COUT, 2, 8>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
        }
        if constexpr (COUT <= 64) {
            switch (p.mi) {
                case 3: return rg_launch<CIN, COUT, 3, 8>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
                case 4: return rg_launch<CIN, COUT, 4, 8>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
            }
        }
        return S2D_ERR_UNSUPPORTED;
    }
    switch (p.mi) {
        case 1: return rg_launch<CIN, COUT, 1, 4>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
        case 2: return rg_launch<CIN, COUT, 2, 4>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
        case 3: return rg_launch<CIN, COUT, 3, 4>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
        case 4: return rg_launch<CIN, COUT, 4, 4>(p, in, n_in, wpack, bias, nbr, n_out, kvol, out, stats, st);
    }
    return S2D_ERR_UNSUPPORTED;
}

template <int CIN>
static int rg_dispatch_cout(int cout, const RgPlan &p, const __bf16 *in, int64_t n_in, const __bf16 *wpack, const float *bias, const int32_t *nbr,
                            int n_out, int kvol, __bf16 *out, float *stats, hipStream_t st) {
    switch (cout) {
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
    switch (cin) {   // 16 / 32 channels: the LDS-staged kernel of spconv_s16.hip (64-byte rows: latency hiding by occupancy wins there)
        case 64: return rg_dispatch_cout<64>(cout, plan, in, n_in, wp, bias, nbr, (int)n_out, kvol, out, stats_partial, st);
        case 128: return rg_dispatch_cout<128>(cout, plan, in, n_in, wp, bias, nbr, (int)n_out, kvol, out, stats_partial, st);
    }
    return S2D_ERR_UNSUPPORTED;
}

bool rg_sorted_supported(int kvol, int cin, int cout) { return kvol == 27 && cin == cout && (cin == 64 || cin == 128); }

// the sorted-row launch: the plan is rg_plan's tile count with the fixed (2, 8) shape, so stats_partial has rg_plan(...).grid rows as usual
int rg_run_sorted(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias, const int32_t *nbr_perm, const int32_t *perm,
                  const uint32_t *pmask, int64_t n_out, int kvol, int cin, int cout, void *out_feat, float *stats_partial, hipStream_t st) {
    RgPlan plan = rg_plan(n_out, kvol, cin, cout);
    if (plan.mi != 2 || plan.waves != 8) return S2D_ERR_UNSUPPORTED;   // (S2D_RG_PLAN tuning override in force)
    const __bf16 *in = (const __bf16 *)in_feat, *wp = (const __bf16 *)packed_weight;
    __bf16 *out = (__bf16 *)out_feat;
    if (cin == 64 && cout == 64) return rg_launch_sorted<64, 64>(plan, in, n_in, wp, bias, nbr_perm, perm, pmask, (int)n_out, kvol, out, stats_partial, st);
    if (cin == 128 && cout == 128) return rg_launch_sorted<128, 128>(plan, in, n_in, wp, bias, nbr_perm, perm, pmask, (int)n_out, kvol, out, stats_partial, st);
    return S2D_ERR_UNSUPPORTED;
}

size_t rg_packed_elems(int kvol, int cin, int cout) { return (size_t)rg_steps(cin, kvol) * RG_KSTEP * cout; }

int rg_pack(const float *weight, int kvol, int cin, int cout, int transpose, int flip, void *packed, hipStream_t st) {
    const int64_t total = (int64_t)rg_packed_elems(kvol, cin, cout);
    hipLaunchKernelGGL(rg_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, weight, kvol, cin, cout, transpose, flip, (__bf16 *)packed);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

int rg_pack_pair(const float *weight, int kvol, int cin, int cout, int flip_dgrad, void *packed_fwd, void *packed_dgrad, hipStream_t st) {
    const int64_t total = (int64_t)std::max(rg_packed_elems(kvol, cin, cout), rg_packed_elems(kvol, cout, cin));
    hipLaunchKernelGGL(rg_pack_pair_kernel, dim3((unsigned)ceil_div(total, 256), 2), dim3(256), 0, st, weight, kvol, cin, cout, flip_dgrad,
                       (__bf16 *)packed_fwd, (__bf16 *)packed_dgrad);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

}  // namespace s2d
