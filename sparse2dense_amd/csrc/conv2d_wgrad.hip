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
// Status (r01, MI355X, 4x188x188): 128->128 133 us and 256->256@94 123 us against MIOpen's 118 / 115 us; 512->64
// 187 us against 226 us, so the host side (dense2d.py) only routes cin >= 512 here for now.  Measured split of the
// 128->128 case before the last two changes: K loop 79 us (loads alone 32, transpose reads + MFMA alone 44, not
// overlapping), partial-slab stores 37 us (4-byte stores of 50 MB), reduce 19 us.  Since then the X reads of tap kx+1
// are issued ahead of the MFMAs of tap kx, and the operands are swapped (D[ci][co]) so that a lane's four results are
// consecutive ci = one 16-byte slab store.
#include "s2d_common.h"

namespace s2d {

typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef short s16x8w __attribute__((ext_vector_type(8)));

template <int TCO, int TCI>
struct WgCfg {
    static constexpr int WCI = TCI >= 128 ? (TCO >= 128 ? 4 : 8) : 4;   // waves across ci
    static constexpr int WCO = 8 / WCI;                                 // waves across co
    static constexpr int MI = TCO / WCO / 16;                           // 16-channel co tiles per wave
    static constexpr int NJ = TCI / WCI / 16;                           // 16-channel ci tiles per wave
    static constexpr int CPR_A = TCO / 8 + 2, CPR_B = TCI / 8 + 2;      // 16-byte chunks per pixel row (data + 32 B pad)
    static constexpr int RS_A = CPR_A * 16, RS_B = CPR_B * 16;          // row strides in bytes
    static constexpr int A_CHUNKS = (32 * CPR_A + 63) / 64 * 64;        // 32 output pixels
    static constexpr int B_CHUNKS = (34 * CPR_B + 63) / 64 * 64;        // 34 input pixels (three kx taps)
    static constexpr int A_BYTES = A_CHUNKS * 16, B_BYTES = B_CHUNKS * 16;
    static constexpr int CHUNKS = A_CHUNKS + B_CHUNKS;
    static constexpr int NS = 4;                                        // LDS ring depth: 3 K-steps of loads in flight
    static constexpr size_t LDS = NS * (size_t)(A_BYTES + B_BYTES);
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

template <int TCO, int TCI>
__global__ __launch_bounds__(512) void conv3x3_wgrad_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy,
                                                            const __bf16 *__restrict__ zero_page, int n_img, int H, int W, int cin,
                                                            int cout, int pad, int steps_per_block, float *__restrict__ partial) {
    typedef WgCfg<TCO, TCI> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto abuf = [&](int b) -> char * { return smem + b * (C::A_BYTES + C::B_BYTES); };
    auto bbuf = [&](int b) -> char * { return smem + b * (C::A_BYTES + C::B_BYTES) + C::A_BYTES; };
    const unsigned smem_addr = (unsigned)(size_t)((__attribute__((address_space(3))) char *)smem);

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wco = wid / C::WCI, wci = wid % C::WCI;
    const int g = lane >> 4, c16 = lane & 15;
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    const int XC = (Wo + 31) / 32;
    const int total = n_img * Ho * XC;
    const int ky = blockIdx.y;
    const int cit_n = cin / TCI;
    const int cot = blockIdx.z / cit_n, cit = blockIdx.z % cit_n;
    const int s0 = blockIdx.x * steps_per_block;
    const int s1 = min(total, s0 + steps_per_block);

    f32x4w acc[C::MI][C::NJ][3];
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int j = 0; j < C::NJ; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[i][j][k] = f32x4w{0.f, 0.f, 0.f, 0.f};

    // (n, y, xc) of step s, advanced incrementally
    int sn = s0 / (Ho * XC), sy = (s0 / XC) % Ho, sxc = s0 % XC;
    auto advance = [&]() {
        if (++sxc == XC) {
            sxc = 0;
            if (++sy == Ho) { sy = 0; ++sn; }
        }
    };
    auto stage = [&](int buf) {   // stages the tiles of step (sn, sy, sxc): chunk c of the slot = A chunks, then B chunks
        const int x0 = sxc * 32;
        const int yin = sy + ky - pad;
        char *slot = abuf(buf);
#pragma unroll
        for (int u = 0; u < (C::CHUNKS + 511) / 512; ++u) {
            const int c = t + 512 * u;
            if (c < C::CHUNKS) {   // wave-uniform: chunk counts are multiples of 64
                const __bf16 *src = zero_page;
                if (c < C::A_CHUNKS) {   // wave-uniform
                    const int row = c / C::CPR_A, col = c % C::CPR_A, xo = x0 + row;
                    if (col < TCO / 8 && row < 32 && xo < Wo)
                        src = dy + ((int64_t)(sn * Ho + sy) * Wo + xo) * cout + cot * TCO + col * 8;
                } else {
                    const int q = c - C::A_CHUNKS;
                    const int row = q / C::CPR_B, col = q % C::CPR_B, xin = x0 - pad + row;
                    if (col < TCI / 8 && row < 34 && (unsigned)yin < (unsigned)H && (unsigned)xin < (unsigned)W)
                        src = x + ((int64_t)(sn * H + yin) * W + xin) * cin + cit * TCI + col * 8;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(slot + (size_t)(c - lane) * 16), 16, 0, 0);
            }
        }
    };

    // global_load_lds instructions this wave issues per K-step (wave-uniform): the ring is advanced with
    // "s_waitcnt vmcnt(2 * nl)" = the two youngest K-steps may stay in flight
    int nl = 0;
#pragma unroll
    for (int u = 0; u < (C::CHUNKS + 511) / 512; ++u) nl += (t + 512 * u < C::CHUNKS) ? 1 : 0;
    nl = __builtin_amdgcn_readfirstlane(nl);

    // per-lane position of the transpose reads inside a tile: pixel row 4g + c16/4 (+16 for the second read), 8-byte column c16 % 4
    const int tr_row = 4 * g + (c16 >> 2), tr_col = (c16 & 3) * 8;
    // prologue: fill the ring (the staging cursor runs ahead of the compute cursor; rows are tracked separately)
    int cy = sy;                                  // image row of the step being computed
    int cxc = sxc;
    const int n_pre = min(C::NS - 1, s1 - s0);
    for (int p = 0; p < n_pre; ++p) {
        stage(p);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (simple: the whole prologue lands before the first step)
    __builtin_amdgcn_s_barrier();
    for (int s = s0; s < s1; ++s) {
        const int cur = (s - s0) % C::NS;
        const int yin = cy + ky - pad;
        const bool row_ok = (unsigned)yin < (unsigned)H;   // block-uniform: a kernel row outside the image adds nothing
        if (++cxc == XC) {
            cxc = 0;
            if (++cy == Ho) cy = 0;
        }
        const bool more = s + C::NS - 1 < s1;
        if (more) {   // slot (s-1) % NS was last read in the previous iteration, before its barrier
            stage((s - s0 + C::NS - 1) % C::NS);
            advance();
        }
        if (row_ok) {
            // dY fragments and the kx = 0 X fragments first; then the X reads of tap kx+1 are issued ahead of the MFMAs of
            // tap kx (LDS returns in order: "lgkmcnt(2*NJ)" = everything but the youngest tap's reads has arrived)
            i32x2w alo[C::MI], ahi[C::MI], blo[3][C::NJ], bhi[3][C::NJ];
            const unsigned abase = smem_addr + cur * (C::A_BYTES + C::B_BYTES) + tr_row * C::RS_A + wco * C::MI * 32 + tr_col;
            const unsigned bbase = smem_addr + cur * (C::A_BYTES + C::B_BYTES) + C::A_BYTES + tr_row * C::RS_B + wci * C::NJ * 32 + tr_col;
#pragma unroll
            for (int i = 0; i < C::MI; ++i) tr_issue<16 * C::RS_A>(alo[i], ahi[i], abase + i * 32);
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) tr_issue<16 * C::RS_B>(blo[0][j], bhi[0][j], bbase + j * 32);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if (kx < 2) {
#pragma unroll
                    for (int j = 0; j < C::NJ; ++j) tr_issue<16 * C::RS_B>(blo[kx + 1][j], bhi[kx + 1][j], bbase + j * 32 + (kx + 1) * C::RS_B);
                    if (C::NJ == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < C::NJ; ++j) {
                    const bf16x8w xf = tr_pack(blo[kx][j], bhi[kx][j]);
#pragma unroll
                    for (int i = 0; i < C::MI; ++i)   // D[ci][co]: the X fragment is the row operand, so a lane's 4 results are 4 consecutive ci
                        acc[i][j][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, tr_pack(alo[i], ahi[i]), acc[i][j][kx], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // next step's tiles must have landed (in every wave) before anyone reads them; the two younger steps stay in flight
        if (more) {
            if (nl == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (nl == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (nl == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }

    // partial[split][ky][kx][co][ci] ; C/D layout: row (ci) = 4*(lane>>4)+reg -> one 16-byte store, col (co) = lane&15
    float *dst = partial + ((int64_t)blockIdx.x * 9 + ky * 3) * (int64_t)cout * cin;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) {
                const int co = cot * TCO + (wco * C::MI + i) * 16 + c16;
                const int ci = cit * TCI + (wci * C::NJ + j) * 16 + 4 * g;
                *reinterpret_cast<f32x4w *>(dst + ((int64_t)kx * cout + co) * cin + ci) = acc[i][j][kx];
            }
}

// dw[co][ci][ky][kx] = sum_split partial[split][ky*3+kx][co][ci]   (fixed order -> deterministic)
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float *__restrict__ partial, int splits, int cin, int cout,
                                                                   float *__restrict__ dw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [tap][co][ci], ci fastest (coalesced reads)
    const int64_t plane = (int64_t)cout * cin;
    if (i >= 9 * plane) return;
    const int tap = (int)(i / plane);
    const int64_t r = i - tap * plane;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += partial[(int64_t)k * 9 * plane + i];
    dw[r * 9 + tap] = s;
}

struct WgPlan {
    int tco, tci, splits, steps_per_block;
    size_t ws_bytes;
};
static bool wg_supported(int cin, int cout) { return cin % 64 == 0 && cout % 64 == 0 && cin >= 64 && cout >= 64; }
static WgPlan wg_plan(int n_img, int h, int w, int cin, int cout, int pad) {
    WgPlan p;
    p.tco = cout % 128 == 0 ? 128 : 64;
    p.tci = cin % 128 == 0 ? 128 : 64;
    const int ho = h + 2 * pad - 2, wo = w + 2 * pad - 2;
    const int64_t total = (int64_t)n_img * ho * ((wo + 31) / 32);
    const int tiles = (cout / p.tco) * (cin / p.tci);
    int64_t splits = ceil_div(256, 3 * tiles);   // one 8-wave workgroup per CU
    if (splits > total) splits = total;
    if (splits < 1) splits = 1;
    p.steps_per_block = (int)ceil_div(total, splits);
    p.splits = (int)ceil_div(total, p.steps_per_block);
    p.ws_bytes = (size_t)p.splits * 9 * cout * cin * sizeof(float);
    return p;
}

template <int TCO, int TCI>
static int wg_launch(const WgPlan &p, const __bf16 *x, const __bf16 *dy, const __bf16 *zero_page, int n_img, int h, int w, int cin,
                     int cout, int pad, float *partial, hipStream_t st) {
    typedef WgCfg<TCO, TCI> C;
    auto kern = conv3x3_wgrad_kernel<TCO, TCI>;
    static bool attr_set = false;   // once per instantiation (idempotent if raced)
    if (C::LDS > 48 * 1024 && !attr_set) {
        S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        attr_set = true;
    }
    const dim3 grid(p.splits, 3, (cout / TCO) * (cin / TCI));
    hipLaunchKernelGGL(kern, grid, dim3(512), C::LDS, st, x, dy, zero_page, n_img, h, w, cin, cout, pad, p.steps_per_block, partial);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_conv2d3x3_wgrad_supported(int cin, int cout) { return wg_supported(cin, cout); }

extern "C" size_t s2d_conv2d3x3_wgrad_workspace_bytes(int n_img, int h, int w, int cin, int cout, int pad) {
    if (!wg_supported(cin, cout) || n_img <= 0 || h + 2 * pad - 2 <= 0 || w + 2 * pad - 2 <= 0) return 0;
    return wg_plan(n_img, h, w, cin, cout, pad).ws_bytes;
}

extern "C" int s2d_conv2d3x3_wgrad_nhwc_bf16(const void *x, const void *dy, const void *zero_page, int n_img, int h, int w, int cin,
                                             int cout, int pad, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream) {
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
    int rc;
    if (p.tco == 128 && p.tci == 128) rc = wg_launch<128, 128>(p, xp, dp, zp, n_img, h, w, cin, cout, pad, partial, st);
    else if (p.tco == 64 && p.tci == 128) rc = wg_launch<64, 128>(p, xp, dp, zp, n_img, h, w, cin, cout, pad, partial, st);
    else if (p.tco == 128 && p.tci == 64) rc = wg_launch<128, 64>(p, xp, dp, zp, n_img, h, w, cin, cout, pad, partial, st);
    else rc = wg_launch<64, 64>(p, xp, dp, zp, n_img, h, w, cin, cout, pad, partial, st);
    if (rc) return rc;
    const int64_t total = (int64_t)9 * cin * cout;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, (const float *)partial,
                       p.splits, cin, cout, dweight);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
