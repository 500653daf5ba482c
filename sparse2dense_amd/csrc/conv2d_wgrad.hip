// Weight gradient of the dense 3x3 stride-1 convolution on NHWC bf16 activations (BEV neck / head):
//   dW[co][ci][ky][kx] = sum over output pixels m of dY[m][co] * X[pix(m) + (ky - pad, kx - pad)][ci]
// (/root/reference/det3d/models/necks/rpn.py:126-145, bbox_heads/center_head.py:209-232; replaces the cuDNN
// backward-filter call behind nn.Conv2d there).
//
// The contraction runs over pixels, which is the slow axis of both NHWC operands, so both MFMA operands need a
// transpose.  It is done by the LDS transpose read of gfx950 (ds_read_b64_tr_b16): a 16-lane group reads a
// [4 pixel][16 ch] block, each lane receiving the 4 pixels of its channel (the row stride of the block is free).
// Tiles are staged with global_load_lds pixel-major, [pixel][channels + 32 B pad]: whole 128/256-byte pixel rows
// are fetched by consecutive lanes (full cache lines; a 16-channel slab layout fetched 32-byte pieces and ran 4x
// over-fetch through the L1), and the 32-byte pad (lanes that land on it read a zero page) shifts consecutive rows
// by 8 banks so that the 8 rows a 32-lane half touches cover all 64 banks.  Lane group g holds pixels
// {4g..4g+3, 16+4g..16+4g+3} of a 32-pixel K-step for A (dY) and B (X) alike — as good a K order as any.
//
// One K-step = 32 consecutive output pixels of one image row.  A workgroup (8 waves) owns one kernel row ky, a
// TCO x TCI channel tile and a contiguous range of K-steps; per K-step it stages dY[32 px][TCO] and
// X[34 px][TCI] of input row y+ky-pad once and uses the X tile for all three kx taps (the tap is a row offset of
// the transpose read).  Out-of-image pixels read a zero page.  Partial sums go to fp32 slabs
// [split][ky][kx][co][ci]; a second kernel reduces the splits in a fixed order (deterministic) into the
// torch layout [cout][cin][3][3].
//
// Status (r01, MI355X, 4x188x188, reduce included): 128->128 69 us and 256->256@94 71 us against MIOpen's 119 / 114 us
// (plus its zeroing / cast launches), 512->64 146 us against 227 us: every stride-1 3x3 weight gradient runs here.
// History of the 128->128 case: 133 us with the K loop at 79 us (loads alone 32, transpose reads + MFMA alone 44, not
// overlapping), partial-slab stores 37 us (4-byte stores of 50 MB), reduce 19 us.  Then: operands swapped (D[ci][co], a
// lane's four results are consecutive ci), staging roles hoisted out of the K loop, fragments double buffered in
// registers, slabs stored through LDS as whole row segments, split-parallel float4 reduce -> 116 us; and the one that
// mattered most: the grid was 258 workgroups for 256 CUs with one resident workgroup per CU (203 VGPRs), i.e. two
// rounds - rounding the split count down gave 69 us.
#include "s2d_common.h"

namespace s2d {

typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef short s16x8w __attribute__((ext_vector_type(8)));

// STRIDE = 2 (pad 1): the contraction of a [Ho][Wo] tensor (A) with the [H][W] = [2 Ho][2 Wo] tensor (B) it is strided over,
//   dW[a ch][b ch][ky][kx] = sum A[n][i][j][a ch] * B[n][2 i - 1 + ky][2 j - 1 + kx][b ch]
// = the weight gradient of ConvTranspose2d(4,2,1) (A = its input, B = dY: [Cin][Cout][4][4]) and of a stride-2 3x3 conv (A = dY,
// B = its input: [Cout][Cin][3][3]).  The B tile holds 31 * 2 + KS pixels and the transpose reads walk it with a two-pixel row stride;
// its rows are padded by 16 B instead of 32 so that the doubled stride still spreads eight rows over all banks.
template <int TCO, int TCI, int KS = 3, int STRIDE = 1, bool DEEP = false>
struct WgCfg {
    static constexpr int WCI = TCI >= 128 ? (TCO >= 128 ? 4 : 8) : 4;   // waves across ci
    static constexpr int WCO = 8 / WCI;                                 // waves across co
    static constexpr int MI = TCO / WCO / 16;                           // 16-channel co tiles per wave
    static constexpr int NJ = TCI / WCI / 16;                           // 16-channel ci tiles per wave
    static constexpr int CPR_A = TCO / 8 + 2, CPR_B = TCI / 8 + (STRIDE == 2 ? 1 : 2);   // 16-byte chunks per pixel row (data + pad)
    static constexpr int RS_A = CPR_A * 16, RS_B = CPR_B * 16;          // row strides in bytes
    static constexpr int A_CHUNKS = (32 * CPR_A + 63) / 64 * 64;        // 32 output pixels
    static constexpr int NB = 31 * STRIDE + (KS == 1 ? 3 : KS);         // input pixels of a K-step (34 for the 3x3 and 1x1 stride-1 cases)
    static constexpr int B_CHUNKS = (NB * CPR_B + 63) / 64 * 64;
    static constexpr int A_BYTES = A_CHUNKS * 16, B_BYTES = B_CHUNKS * 16;
    static constexpr int CHUNKS = A_CHUNKS + B_CHUNKS;
    // LDS ring depth: NS - 1 K-steps staged ahead, NS - 2 still in flight behind the wait.  DEEP: as many slots as the 160 KB hold (<= 8)
    static constexpr int NS_FIT = (160 * 1024) / (A_BYTES + B_BYTES);
    static constexpr int NS = DEEP ? (NS_FIT > 8 ? 8 : NS_FIT) : 4;
    static constexpr size_t LDS = NS * (size_t)(A_BYTES + B_BYTES);
    static_assert(NS >= 4, "ring too shallow");
    static_assert(MI >= 1 && NJ >= 1, "bad tiling");
};

typedef int i32x2w __attribute__((ext_vector_type(2)));
typedef int i32x4w __attribute__((ext_vector_type(4)));

// Transpose reads are issued through inline asm: hipcc cannot see through the ds_read_tr builtin's pointer and, to
// be safe against the in-flight global_load_lds of the NEXT ring slots, puts s_waitcnt vmcnt(0) in front of it
// (which serialises every K-step on its own prefetch).  The asm is opaque to that tracking; the caller issues all
// reads of a K-step, then tr_wait() (lgkmcnt(0)) and a scheduling barrier before the first MFMA.
template <int HI_OFF>   // byte offset of the second read (16 pixel rows further)
__device__ __forceinline__ void tr_issue(i32x2w &lo, i32x2w &hi, unsigned lds_addr) {
    // early-clobber outputs: the address register is read again by the second instruction
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(lo), "=&v"(hi) : "v"(lds_addr), "n"(HI_OFF) : "memory");
}
__device__ __forceinline__ void tr_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8w tr_pack(const i32x2w &lo, const i32x2w &hi) {
    i32x4w v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(bf16x8w, v);
}

// KS = kernel size: 3, or 1 (the 1x1 convs: one "kernel row", one kx tap, the same transposed-operand pipeline)
// KXN = kx taps per workgroup (blockIdx.y = ky * (KS / KXN) + kx group): 4x4 kernels run two per workgroup to stay in registers
template <int N>
__device__ __forceinline__ void wg_wait_vm() {   // N younger loads may stay in flight; own LDS reads done
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

template <int TCO, int TCI, int KS = 3, int STRIDE = 1, int KXN = KS, bool DEEP = false>
__global__ __launch_bounds__(512) void conv3x3_wgrad_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy,
                                                            const __bf16 *__restrict__ zero_page, int n_img, int H, int W, int cin,
                                                            int cout, int pad, int steps_per_block, float *__restrict__ partial,
                                                            float *__restrict__ dbias_partial) {
    typedef WgCfg<TCO, TCI, KS, STRIDE, DEEP> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto abuf = [&](int b) -> char * { return smem + b * (C::A_BYTES + C::B_BYTES); };
    const unsigned smem_addr = (unsigned)(size_t)((__attribute__((address_space(3))) char *)smem);

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wco = wid / C::WCI, wci = wid % C::WCI;
    const int g = lane >> 4, c16 = lane & 15;
    const int Ho = (H + 2 * pad - KS) / STRIDE + 1, Wo = (W + 2 * pad - KS) / STRIDE + 1;
    const int XC = (Wo + 31) / 32;
    const int total = n_img * Ho * XC;
    constexpr int KXG = KS / KXN;
    const int ky = blockIdx.y / KXG, kx0 = (blockIdx.y % KXG) * KXN;
    const int cit_n = cin / TCI;
    const int cot = blockIdx.z / cit_n, cit = blockIdx.z % cit_n;
    const int s0 = blockIdx.x * steps_per_block;
    const int s1 = min(total, s0 + steps_per_block);

    f32x4w acc[C::MI][C::NJ][KXN];
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int j = 0; j < C::NJ; ++j)
#pragma unroll
            for (int k = 0; k < KXN; ++k) acc[i][j][k] = f32x4w{0.f, 0.f, 0.f, 0.f};

    // Bias gradient for free (dbias_partial != NULL): the dY fragments of the workgroups of kernel row 0 / input-channel tile 0 pass
    // every output pixel exactly once, so their first ci wave column sums them per channel (8 pixels per lane and K-step) -
    // the separate statistics pass over dY that produced nn.Conv2d's bias gradient is gone (39 launches of ~11 us per step).
    const bool do_db = dbias_partial != nullptr && blockIdx.y == 0 && cit == 0 && wci == 0;   // wave-uniform
    float dbs[C::MI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i) dbs[i] = 0.f;

    // (n, y, xc) of step s, advanced incrementally
    int sn = s0 / (Ho * XC), sy = (s0 / XC) % Ho, sxc = s0 % XC;
    auto advance = [&]() {
        if (++sxc == XC) {
            sxc = 0;
            if (++sy == Ho) { sy = 0; ++sn; }
        }
    };
    // Staging roles are fixed per thread: chunk c = t + 512 u of the slot is an A (dY) chunk or a B (X) chunk (wave-uniform,
    // the chunk counts are multiples of 64) at a fixed (pixel row, 16-byte column) of its tile.  Everything that depends
    // on the thread is computed once - the source pointer at step origin and the pixel row (a huge value for pad columns
    // and filler rows, which then always fail the range test and read the zero page); per K-step only wave-uniform
    // offsets and limits change.  (The first version redid the chunk -> (row, col) division and the address products
    // in every K-step: ~200 staging instructions per 24 MFMAs.)
    constexpr int NU = (C::CHUNKS + 511) / 512;
    const __bf16 *ptr_u[NU];
    int row_u[NU];
    bool isa_u[NU], has_u[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c = t + 512 * u;
        has_u[u] = __builtin_amdgcn_readfirstlane(c < C::CHUNKS ? 1 : 0) != 0;
        isa_u[u] = __builtin_amdgcn_readfirstlane(c < C::A_CHUNKS ? 1 : 0) != 0;
        if (c < C::A_CHUNKS) {
            const int row = c / C::CPR_A, col = c % C::CPR_A;
            row_u[u] = (col < TCO / 8 && row < 32) ? row : 0x40000000;
            ptr_u[u] = dy + (int64_t)row * cout + cot * TCO + col * 8;
        } else {
            const int q = c - C::A_CHUNKS;
            const int row = q / C::CPR_B, col = q % C::CPR_B;
            row_u[u] = (col < TCI / 8 && row < 31 * STRIDE + KS) ? row : 0x40000000;
            ptr_u[u] = x + (int64_t)row * cin + cit * TCI + col * 8;
        }
    }
    auto stage = [&](int buf) {   // stages the tiles of step (sn, sy, sxc)
        const int x0 = sxc * 32;
        const int yin = sy * STRIDE + ky - pad;
        const int64_t ua = (((int64_t)sn * Ho + sy) * Wo + x0) * cout;          // element offset of dY[sn][sy][x0][0]
        const int64_t ub = (((int64_t)sn * H + yin) * W + x0 * STRIDE - pad) * cin;   // of X[sn][yin][x0 * STRIDE - pad][0]
        const unsigned lim_b = (unsigned)yin < (unsigned)H ? (unsigned)W : 0u;  // a kernel row outside the image: all zero
        char *slot = abuf(buf);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (has_u[u]) {   // wave-uniform
                const int64_t uo = isa_u[u] ? ua : ub;
                const unsigned lim = isa_u[u] ? (unsigned)Wo : lim_b;
                const unsigned xv = (unsigned)((isa_u[u] ? x0 : x0 * STRIDE - pad) + row_u[u]);
                const __bf16 *src = xv < lim ? ptr_u[u] + uo : zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(slot + (size_t)(t + 512 * u - lane) * 16), 16, 0, 0);
            }
        }
    };

    // global_load_lds instructions this wave issues per K-step (wave-uniform): the ring is advanced with
    // "s_waitcnt vmcnt(2 * nl)" = the two youngest K-steps may stay in flight
    int nl = 0;
#pragma unroll
    for (int u = 0; u < NU; ++u) nl += has_u[u] ? 1 : 0;

    // per-lane position of the transpose reads inside a tile: pixel row 4g + c16/4 (+16 for the second read), 8-byte column c16 % 4
    const int tr_row = 4 * g + (c16 >> 2), tr_col = (c16 & 3) * 8;
    // Fragments are double buffered in registers: iteration i issues the transpose reads of step i+1 and then runs the
    // MFMAs of step i, whose fragments were read during iteration i-1 (three LDS round trips per K-step - one per kx tap -
    // used to sit in front of the MFMAs).  Ring protocol (slot of step i = i % NS), iteration i:
    //   stage step i+NS-1 into the slot of step i-1 (its fragment reads were waited for before the last barrier);
    //   wait: own loads of step i+1 landed, own fragment reads of step i done;  barrier;
    //   issue the reads of step i+1;  MFMAs of step i.
    // A kernel row outside the image stages zero tiles (lim_b = 0), so no step is skipped.
    struct Frags {
        i32x2w alo[C::MI], ahi[C::MI], blo[KXN][C::NJ], bhi[KXN][C::NJ];
    };
    const unsigned a_off = tr_row * C::RS_A + wco * C::MI * 32 + tr_col;
    const unsigned b_off = C::A_BYTES + (tr_row * STRIDE + kx0) * C::RS_B + wci * C::NJ * 32 + tr_col;
    auto read_frags = [&](int slot, Frags &f) {
        const unsigned base = smem_addr + slot * (C::A_BYTES + C::B_BYTES);
#pragma unroll
        for (int i = 0; i < C::MI; ++i) tr_issue<16 * C::RS_A>(f.alo[i], f.ahi[i], base + a_off + i * 32);
#pragma unroll
        for (int kx = 0; kx < KXN; ++kx)
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) tr_issue<16 * STRIDE * C::RS_B>(f.blo[kx][j], f.bhi[kx][j], base + b_off + j * 32 + kx * C::RS_B);
    };
    auto mfmas = [&](const Frags &f) {
        if (do_db) {
#pragma unroll
            for (int i = 0; i < C::MI; ++i) {
                const bf16x8w a = tr_pack(f.alo[i], f.ahi[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) dbs[i] += (float)a[e];
            }
        }
#pragma unroll
        for (int kx = 0; kx < KXN; ++kx)
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) {
                const bf16x8w xf = tr_pack(f.blo[kx][j], f.bhi[kx][j]);
#pragma unroll
                for (int i = 0; i < C::MI; ++i)   // D[ci][co]: the X fragment is the row operand, so a lane's 4 results are 4 consecutive ci
                    acc[i][j][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, tr_pack(f.alo[i], f.ahi[i]), acc[i][j][kx], 0, 0, 0);
            }
    };
    const int n = s1 - s0;
    const int n_pre = min(C::NS - 1, n);
    for (int p = 0; p < n_pre; ++p) {
        stage(p);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (simple: the whole prologue lands before the first step)
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    read_frags(0, f0);
    int slot_next = 1, slot_stage = C::NS - 1;   // slots of steps i+1 and i+NS-1
    auto iteration = [&](int i, const Frags &cur, Frags &nxt) {
        const bool more = i + C::NS - 1 < n;
        if (more) {
            stage(slot_stage);
            advance();
        }
        // step i+1 landed (in every wave after the barrier); the two younger steps stay in flight
        if (more) {
            constexpr int FL = C::NS - 2;   // K-steps left in flight
            if (nl == 4) wg_wait_vm<4 * FL>();
            else if (nl == 3) wg_wait_vm<3 * FL>();
            else if (nl == 2) wg_wait_vm<2 * FL>();
            else wg_wait_vm<FL>();
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < n) read_frags(slot_next, nxt);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(cur);
        __builtin_amdgcn_sched_barrier(0);
        slot_next = slot_next == C::NS - 1 ? 0 : slot_next + 1;
        slot_stage = slot_stage == C::NS - 1 ? 0 : slot_stage + 1;
    };
    for (int i = 0; i < n; i += 2) {
        iteration(i, f0, f1);
        if (i + 1 < n) iteration(i + 1, f1, f0);
    }

    // partial[split][ky][kx][co][ci].  C/D layout: row (ci) = 4*(lane>>4)+reg, col (co) = lane&15: a direct store writes
    // 64-byte pieces of 16 different co rows per instruction (measured: 37 us for the 50 MB of slabs of a 128->128 layer).
    // Each wave instead passes its [MI*16 co][NJ*16 ci] tile of one kx through LDS (the ring is free now) and stores it
    // as whole NJ*64-byte row segments, 16 bytes per lane.
    constexpr int EP_ROWB = C::NJ * 64 + 16;              // padded LDS row (bytes)
    constexpr int EP_BYTES = C::MI * 16 * EP_ROWB;
    static_assert(8 * EP_BYTES <= (int)C::LDS, "epilogue tiles exceed the ring");
    if (do_db) {   // lane (g, c16) holds channel c16 of co tile i over the pixels of its group: fold the four groups
#pragma unroll
        for (int i = 0; i < C::MI; ++i) {
            float v = dbs[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) dbias_partial[(int64_t)blockIdx.x * cout + cot * TCO + wco * C::MI * 16 + i * 16 + c16] = v;
        }
    }
    __syncthreads();   // every wave is out of the K loop
    char *ep = smem + wid * EP_BYTES;
    float *dst = partial + ((int64_t)blockIdx.x * (KS * KS) + ky * KS + kx0) * (int64_t)cout * cin;
    const int co0 = cot * TCO + wco * C::MI * 16, ci0 = cit * TCI + wci * C::NJ * 16;
#pragma unroll
    for (int kx = 0; kx < KXN; ++kx) {
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NJ; ++j)
                *reinterpret_cast<f32x4w *>(ep + (16 * i + c16) * EP_ROWB + (16 * j + 4 * g) * 4) = acc[i][j][kx];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave: its LDS ops are ordered, this pins the compiler
        constexpr int CPRW = C::NJ * 4;                       // 16-byte chunks per row
#pragma unroll
        for (int it = 0; it < C::MI * 16 * CPRW / 64; ++it) {
            const int id = lane + 64 * it, row = id / CPRW, ch = id % CPRW;
            const f32x4w v = *reinterpret_cast<const f32x4w *>(ep + row * EP_ROWB + ch * 16);
            *reinterpret_cast<f32x4w *>(dst + ((int64_t)kx * cout + co0 + row) * cin + ci0 + ch * 4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next kx overwrites the tile
    }
}

// dw[co][ci][ky][kx] = sum_split partial[split][ky*3+kx][co][ci]   (fixed order -> deterministic)
// A workgroup owns 64 float4 columns of the [tap][co][ci] plane; 4 thread rows stride over the splits (more loads in
// flight than one thread per element walking all splits), fixed-order fold through LDS.  cin % 4 == 0.
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float4 *__restrict__ partial, int splits, int cin, int cout,
                                                                   int taps, float *__restrict__ dw, int main_blocks = 0x7fffffff,
                                                                   const float *__restrict__ dbias_partial = nullptr,
                                                                   float *__restrict__ dbias = nullptr) {
    __shared__ float4 red[4][64];
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    if ((int)blockIdx.x >= main_blocks) {   // trailing blocks: the bias gradient, dbias[co] = sum over splits of dbias_partial[split][co]
        const int co = ((int)blockIdx.x - main_blocks) * 64 + col;
        float s = 0.f;
        if (co < cout)
            for (int k = sl; k < splits; k += 4) s += dbias_partial[(int64_t)k * cout + co];
        red[sl][col].x = s;
        __syncthreads();
        if (sl == 0 && co < cout) dbias[co] = (red[0][col].x + red[1][col].x) + (red[2][col].x + red[3][col].x);
        return;
    }
    const int64_t plane = (int64_t)cout * cin;          // floats per tap
    const int64_t size4 = taps * plane / 4;
    const int64_t i4 = (int64_t)blockIdx.x * 64 + col;  // float4 index over [tap][co][ci]
    float4 s = float4{0.f, 0.f, 0.f, 0.f};
    if (i4 < size4) {
        for (int k = sl; k < splits; k += 4) {
            const float4 v = partial[(int64_t)k * size4 + i4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[sl][col] = s;
    __syncthreads();
    if (sl == 0 && i4 < size4) {
        float4 t = red[0][col];
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            const float4 v = red[r][col];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        const int64_t i = i4 * 4;
        const int tap = (int)(i / plane);
        const int64_t r = i - tap * plane;               // co * cin + ci, 4 consecutive ci
        dw[r * taps + tap] = t.x;
        dw[(r + 1) * taps + tap] = t.y;
        dw[(r + 2) * taps + tap] = t.z;
        dw[(r + 3) * taps + tap] = t.w;
    }
}

struct WgPlan {
    int tco, tci, splits, steps_per_block;
    size_t ws_bytes;
};
static bool wg_supported(int cin, int cout) { return cin % 64 == 0 && cout % 64 == 0 && cin >= 64 && cout >= 64; }
static WgPlan wg_plan(int n_img, int h, int w, int cin, int cout, int pad, int ks = 3, int stride = 1, int kxg = 1) {
    WgPlan p;
    p.tco = cout % 128 == 0 ? 128 : 64;
    p.tci = cin % 128 == 0 ? 128 : 64;
    const int ho = (h + 2 * pad - ks) / stride + 1, wo = (w + 2 * pad - ks) / stride + 1;
    const int64_t total = (int64_t)n_img * ho * ((wo + 31) / 32);
    const int tiles = (cout / p.tco) * (cin / p.tci);
    // one 8-wave workgroup per CU (203 VGPRs: a second one does not fit), and never more workgroups than CUs: 258
    // workgroups on 256 CUs ran as two rounds (rocprofv3: SQ busy 51 us of a 103 us launch) - round DOWN
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    int64_t splits = cus / (ks * kxg * tiles);
    if (splits > total) splits = total;
    if (splits < 1) splits = 1;
    p.steps_per_block = (int)ceil_div(total, splits);
    p.splits = (int)ceil_div(total, p.steps_per_block);
    p.ws_bytes = (size_t)p.splits * ks * ks * cout * cin * sizeof(float) + (size_t)p.splits * cout * sizeof(float);   // + bias-gradient rows
    return p;
}

// S2D_WG_RING_DEEP=0: the four-slot ring everywhere (the r01-r05 kernel)
static bool wg_ring_deep() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("S2D_WG_RING_DEEP");
        v = e ? (atoi(e) != 0) : 1;
    }
    return v != 0;
}

template <int TCO, int TCI, int KS = 3, int STRIDE = 1, int KXN = KS, bool DEEP = false>
static int wg_launch_ring(const WgPlan &p, const __bf16 *x, const __bf16 *dy, const __bf16 *zero_page, int n_img, int h, int w, int cin,
                     int cout, int pad, float *partial, hipStream_t st, float *dbias_partial = nullptr) {
    typedef WgCfg<TCO, TCI, KS, STRIDE, DEEP> C;
    auto kern = conv3x3_wgrad_kernel<TCO, TCI, KS, STRIDE, KXN, DEEP>;
    static bool attr_set = false;   // once per instantiation (idempotent if raced)
    if (C::LDS > 48 * 1024 && !attr_set) {
        S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr_set = true;
    }
    const dim3 grid(p.splits, KS * (KS / KXN), (cout / TCO) * (cin / TCI));
    hipLaunchKernelGGL(kern, grid, dim3(512), C::LDS, st, x, dy, zero_page, n_img, h, w, cin, cout, pad, p.steps_per_block, partial,
                       dbias_partial);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

template <int TCO, int TCI, int KS = 3, int STRIDE = 1, int KXN = KS>
static int wg_launch(const WgPlan &p, const __bf16 *x, const __bf16 *dy, const __bf16 *zero_page, int n_img, int h, int w, int cin,
                     int cout, int pad, float *partial, hipStream_t st, float *dbias_partial = nullptr) {
    if (wg_ring_deep())
        return wg_launch_ring<TCO, TCI, KS, STRIDE, KXN, true>(p, x, dy, zero_page, n_img, h, w, cin, cout, pad, partial, st, dbias_partial);
    return wg_launch_ring<TCO, TCI, KS, STRIDE, KXN, false>(p, x, dy, zero_page, n_img, h, w, cin, cout, pad, partial, st, dbias_partial);
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_conv2d3x3_wgrad_supported(int cin, int cout) { return wg_supported(cin, cout); }

extern "C" size_t s2d_conv2d3x3_wgrad_workspace_bytes(int n_img, int h, int w, int cin, int cout, int pad) {
    if (!wg_supported(cin, cout) || n_img <= 0 || h + 2 * pad - 2 <= 0 || w + 2 * pad - 2 <= 0) return 0;
    return wg_plan(n_img, h, w, cin, cout, pad).ws_bytes;
}

extern "C" int s2d_conv2d3x3_wgrad_nhwc_bf16(const void *x, const void *dy, const void *zero_page, int n_img, int h, int w, int cin,
                                             int cout, int pad, float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && dy && zero_page && dweight && n_img > 0 && h > 0 && w > 0 && (pad == 0 || pad == 1), "conv2d3x3_wgrad: bad argument");
    if (!wg_supported(cin, cout)) {
        set_error("conv2d3x3_wgrad: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    S2D_CHECK_ARG(h + 2 * pad - 2 > 0 && w + 2 * pad - 2 > 0, "conv2d3x3_wgrad: empty output");
    const WgPlan p = wg_plan(n_img, h, w, cin, cout, pad);
    if (!ws || ws_bytes < p.ws_bytes) {
        set_error("conv2d3x3_wgrad: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const __bf16 *xp = (const __bf16 *)x, *dp = (const __bf16 *)dy, *zp = (const __bf16 *)zero_page;
    float *partial = (float *)ws;
    float *dbp = dbias ? partial + (size_t)p.splits * 9 * cout * cin : nullptr;   // [splits][cout] behind the weight slabs
    int rc;
    if (p.tco == 128 && p.tci == 128) rc = wg_launch<128, 128>(p, xp, dp, zp, n_img, h, w, cin, cout, pad, partial, st, dbp);
    else if (p.tco == 64 && p.tci == 128) rc = wg_launch<64, 128>(p, xp, dp, zp, n_img, h, w, cin, cout, pad, partial, st, dbp);
    else if (p.tco == 128 && p.tci == 64) rc = wg_launch<128, 64>(p, xp, dp, zp, n_img, h, w, cin, cout, pad, partial, st, dbp);
    else rc = wg_launch<64, 64>(p, xp, dp, zp, n_img, h, w, cin, cout, pad, partial, st, dbp);
    if (rc) return rc;
    const int64_t total = (int64_t)9 * cin * cout;
    const unsigned main_blocks = (unsigned)ceil_div(total / 4, 64);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(main_blocks + (dbias ? (unsigned)ceil_div(cout, 64) : 0u)), dim3(256), 0, st,
                       (const float4 *)partial, p.splits, cin, cout, 9, dweight, (int)main_blocks, (const float *)dbp, dbias);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* 1x1 convolution (stride 1): dW[co][ci] = sum over pixels of dY[m][co] * X[m][ci], same kernel with one tap */
extern "C" size_t s2d_conv2d1x1_wgrad_workspace_bytes(int n_img, int h, int w, int cin, int cout) {
    if (!wg_supported(cin, cout) || n_img <= 0 || h <= 0 || w <= 0) return 0;
    return wg_plan(n_img, h, w, cin, cout, 0, 1).ws_bytes;
}

extern "C" int s2d_conv2d1x1_wgrad_nhwc_bf16(const void *x, const void *dy, const void *zero_page, int n_img, int h, int w, int cin, int cout,
                                             float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && dy && zero_page && dweight && n_img > 0 && h > 0 && w > 0, "conv2d1x1_wgrad: bad argument");
    if (!wg_supported(cin, cout)) {
        set_error("conv2d1x1_wgrad: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const WgPlan p = wg_plan(n_img, h, w, cin, cout, 0, 1);
    if (!ws || ws_bytes < p.ws_bytes) {
        set_error("conv2d1x1_wgrad: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const __bf16 *xp = (const __bf16 *)x, *dp = (const __bf16 *)dy, *zp = (const __bf16 *)zero_page;
    float *partial = (float *)ws;
    float *dbp = dbias ? partial + (size_t)p.splits * cout * cin : nullptr;
    int rc;
    if (p.tco == 128 && p.tci == 128) rc = wg_launch<128, 128, 1>(p, xp, dp, zp, n_img, h, w, cin, cout, 0, partial, st, dbp);
    else if (p.tco == 64 && p.tci == 128) rc = wg_launch<64, 128, 1>(p, xp, dp, zp, n_img, h, w, cin, cout, 0, partial, st, dbp);
    else if (p.tco == 128 && p.tci == 64) rc = wg_launch<128, 64, 1>(p, xp, dp, zp, n_img, h, w, cin, cout, 0, partial, st, dbp);
    else rc = wg_launch<64, 64, 1>(p, xp, dp, zp, n_img, h, w, cin, cout, 0, partial, st, dbp);
    if (rc) return rc;
    const int64_t total = (int64_t)cin * cout;
    const unsigned main_blocks = (unsigned)ceil_div(total / 4, 64);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(main_blocks + (dbias ? (unsigned)ceil_div(cout, 64) : 0u)), dim3(256), 0, st,
                       (const float4 *)partial, p.splits, cin, cout, 1, dweight, (int)main_blocks, (const float *)dbp, dbias);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* Stride-2 (pad 1) contraction, ks = 3 or 4 (see WgCfg):  dW[a ch][b ch][ky][kx] = sum a[n][i][j][a ch] * b[n][2 i - 1 + ky][2 j - 1 + kx][b ch]
   with a = [n][h/2][w/2][ca] and b = [n][h][w][cb]: ConvTranspose2d(4,2,1) weight gradient (a = input, b = dY; rpn.py:217-231) and
   the stride-2 3x3 conv's (a = dY, b = input; rpn.py:126-133). */
extern "C" int s2d_conv2d_s2_wgrad_supported(int ca, int cb, int ks) { return wg_supported(cb, ca) && (ks == 3 || ks == 4); }

extern "C" size_t s2d_conv2d_s2_wgrad_workspace_bytes(int n_img, int h, int w, int ca, int cb, int ks) {
    if (!s2d_conv2d_s2_wgrad_supported(ca, cb, ks) || n_img <= 0 || h < 2 || w < 2 || h % 2 || w % 2) return 0;
    return wg_plan(n_img, h, w, cb, ca, 1, ks, 2, ks == 4 ? 2 : 1).ws_bytes;
}

extern "C" int s2d_conv2d_s2_wgrad_nhwc_bf16(const void *a, const void *b, const void *zero_page, int n_img, int h, int w, int ca, int cb,
                                             int ks, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(a && b && zero_page && dweight && n_img > 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "conv2d_s2_wgrad: bad argument");
    if (!s2d_conv2d_s2_wgrad_supported(ca, cb, ks)) {
        set_error("conv2d_s2_wgrad: unsupported channels %d x %d, kernel %d", ca, cb, ks);
        return S2D_ERR_UNSUPPORTED;
    }
    const WgPlan p = wg_plan(n_img, h, w, cb, ca, 1, ks, 2, ks == 4 ? 2 : 1);
    if (!ws || ws_bytes < p.ws_bytes) {
        set_error("conv2d_s2_wgrad: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const __bf16 *bp = (const __bf16 *)b, *ap = (const __bf16 *)a, *zp = (const __bf16 *)zero_page;
    float *partial = (float *)ws;
    // kernel roles: "x" = the strided-over tensor b (cin = cb), "dy" = a (cout = ca)
    int rc;
#define S2D_WG2(TCO_, TCI_)                                                                                      \
    (ks == 4 ? wg_launch<TCO_, TCI_, 4, 2, 2>(p, bp, ap, zp, n_img, h, w, cb, ca, 1, partial, st)                \
             : wg_launch<TCO_, TCI_, 3, 2, 3>(p, bp, ap, zp, n_img, h, w, cb, ca, 1, partial, st))
    if (p.tco == 128 && p.tci == 128) rc = S2D_WG2(128, 128);
    else if (p.tco == 64 && p.tci == 128) rc = S2D_WG2(64, 128);
    else if (p.tco == 128 && p.tci == 64) rc = S2D_WG2(128, 64);
    else rc = S2D_WG2(64, 64);
#undef S2D_WG2
    if (rc) return rc;
    const int64_t total = (int64_t)ks * ks * ca * cb;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)ceil_div(total / 4, 64)), dim3(256), 0, st, (const float4 *)partial,
                       p.splits, cb, ca, ks * ks, dweight);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
