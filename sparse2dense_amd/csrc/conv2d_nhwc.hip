// Dense 3x3 convolution (stride 1) for the BEV neck / head on the gfx950 matrix cores:
// NHWC bf16 activations, bf16 weights, fp32 accumulate, bf16 output — an implicit GEMM with
//   M = N*Ho*Wo output pixels, N = Cout, K = 9 taps x Cin.
// (/root/reference/det3d/models/necks/rpn.py:126-145 `_make_layer`, bbox_heads/center_head.py:209-232.)
//
// Workgroup = 256 threads = 2(M) x 2(N) waves, tile 128 pixels x BN couts (BN = 128 or 64); each
// wave owns 64 pixels x BN/2 couts = 4 x (BN/32) MFMA tiles of v_mfma_f32_16x16x32_bf16.  One K-step
// = one tap x 64 input channels: the A tile (128 px x 64 ch) and the pre-packed B tile are brought
// in with global_load_lds (16 B per lane, lane-linear LDS image), double buffered, one barrier per
// K-step.  B fragments are contiguous 1 KiB reads; the A image is [pixel][8 parts of 16 B] with the
// part index XOR-swizzled by (pixel & 7) — chosen by which global chunk each lane fetches — so that
// the 16-lane groups of a ds_read_b128 fragment read hit 16 distinct bank slots (unswizzled rows
// 128 B apart are 4-way conflicted and the kernel becomes LDS-bound).  Border taps read a zero page instead of branching.  The data gradient is the same
// kernel with the flipped / transposed weight image (conv2d_pack_weights_bf16).
#include "s2d_common.h"
#include <cstdlib>

namespace s2d {

typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4c __attribute__((ext_vector_type(4)));

// weight image: [kstep = tap*(Cin/64)+chunk][co block (BN)][half h = k32 within the 64-chunk (2)][nt (BN/16)][q][c][8]
//   ci = chunk*64 + h*32 + 8q + e ;  co = blk*BN + wn*(BN/2) + c*(BN/32) + n  with nt = wn*(BN/32) + n
// source w[Cout][Cin][3][3] (torch layout).  transpose_flip=1 builds the data-gradient operand:
//   packed "cin" runs over the source's Cout, packed "cout" over its Cin, taps mirrored.
__device__ __forceinline__ void conv2d_pack_element(const float *__restrict__ w, int cin, int cout, int bn, int transpose_flip, int w_nhwc, int taps,
                                                    int64_t i, __bf16 *__restrict__ out) {
    const int64_t total = (int64_t)taps * cin * cout;
    if (i >= total) return;
    const int ntile = bn / 16, per_wave = bn / 32;
    int64_t r = i;
    const int e = r % 8; r /= 8;
    const int c = r % 16; r /= 16;
    const int q = r % 4; r /= 4;
    const int nt = r % ntile; r /= ntile;
    const int h = r % 2; r /= 2;
    const int blk = r % (cout / bn); r /= (cout / bn);
    const int chunk = r % (cin / 64); r /= (cin / 64);
    const int tap = (int)r;
    const int ci = chunk * 64 + h * 32 + 8 * q + e;
    const int co = blk * bn + (nt / per_wave) * (bn / 2) + c * per_wave + (nt % per_wave);
    // source element (o, c, t) of the forward weight; w_nhwc: memory order [o][t][c] (torch channels_last) instead of [o][c][t]
    const int so = transpose_flip ? ci : co, sc = transpose_flip ? co : ci, st = transpose_flip ? taps - 1 - tap : tap;
    const int src_c = transpose_flip ? cout : cin;
    const float v = w_nhwc ? w[((int64_t)so * taps + st) * src_c + sc] : w[((int64_t)so * src_c + sc) * taps + st];
    out[i] = (__bf16)v;
}

// the eight consecutive packed elements [8 i8, 8 i8 + 8) (the e index: eight consecutive input channels of one (tap, co)): one 16-byte
// store and one index decomposition instead of eight (the batched re-pack after the optimizer step: 150 -> see DESIGN us per step)
__device__ __forceinline__ void conv2d_pack_element8(const float *__restrict__ w, int cin, int cout, int bn, int transpose_flip, int w_nhwc, int taps,
                                                     int64_t i8, __bf16 *__restrict__ out) {
    const int64_t total8 = (int64_t)taps * cin * cout / 8;
    if (i8 >= total8) return;
    const int ntile = bn / 16, per_wave = bn / 32;
    int64_t r = i8;
    const int c = r % 16; r /= 16;
    const int q = r % 4; r /= 4;
    const int nt = r % ntile; r /= ntile;
    const int h = r % 2; r /= 2;
    const int blk = r % (cout / bn); r /= (cout / bn);
    const int chunk = r % (cin / 64); r /= (cin / 64);
    const int tap = (int)r;
    const int ci0 = chunk * 64 + h * 32 + 8 * q;
    const int co = blk * bn + (nt / per_wave) * (bn / 2) + c * per_wave + (nt % per_wave);
    const int st = transpose_flip ? taps - 1 - tap : tap;
    const int src_c = transpose_flip ? cout : cin;
    typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));
    bf16x8p o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = ci0 + e;
        const int so = transpose_flip ? ci : co, sc = transpose_flip ? co : ci;
        o[e] = (__bf16)(w_nhwc ? w[((int64_t)so * taps + st) * src_c + sc] : w[((int64_t)so * src_c + sc) * taps + st]);
    }
    *reinterpret_cast<bf16x8p *>(out + i8 * 8) = o;
}

__global__ __launch_bounds__(256) void conv2d_pack_kernel(const float *__restrict__ w, int cin, int cout, int bn,
                                                          int transpose_flip, int w_nhwc, int taps, __bf16 *__restrict__ out) {
    conv2d_pack_element(w, cin, cout, bn, transpose_flip, w_nhwc, taps, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, out);
}

// forward operand [cin -> cout] (blockIdx.y = 0) and data-gradient operand [cout -> cin], taps mirrored (1) of one layer in one launch
__global__ __launch_bounds__(256) void conv2d_pack_pair_kernel(const float *__restrict__ w, int cin, int cout, int bn_f, int bn_d, int w_nhwc, int taps,
                                                               __bf16 *__restrict__ out_f, __bf16 *__restrict__ out_d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) conv2d_pack_element(w, cin, cout, bn_f, 0, w_nhwc, taps, i, out_f);
    else conv2d_pack_element(w, cout, cin, bn_d, 1, w_nhwc, taps, i, out_d);
}

// Weight image of the stride-2 transposed kernels (conv3x3_k32_nhwc_bf16_kernel<.., UP = true>): source w[K ch][N ch][KS][KS] (the
// layout of a ConvTranspose2d weight [Cin][Cout][kh][kw], and of a Conv2d weight [Cout][Cin][kh][kw] seen from its data gradient).
// Packed tap t runs over the four parity classes (p, q) in order, each class's (KS + p) / 2 x (KS + q) / 2 window row-major from its
// top-left source pixel: window position (ty, tx) is the tap ky = 2 (TY - 1 - ty) + 1 - p, kx = 2 (TX - 1 - tx) + 1 - q.
__global__ __launch_bounds__(256) void conv2d_pack_up_kernel(const float *__restrict__ w, int kc, int nc, int bn, int ks,
                                                             __bf16 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int taps = ks * ks;
    if (i >= (int64_t)taps * kc * nc) return;
    const int ntile = bn / 16, per_wave = bn / 32;
    int64_t r = i;
    const int e = r % 8; r /= 8;
    const int c = r % 16; r /= 16;
    const int q4 = r % 4; r /= 4;
    const int nt = r % ntile; r /= ntile;
    const int h = r % 2; r /= 2;
    const int blk = r % (nc / bn); r /= (nc / bn);
    const int chunk = r % (kc / 64); r /= (kc / 64);
    int t = (int)r, p = 0, q = 0;
    for (int cls = 0; cls < 4; ++cls) {
        p = cls >> 1; q = cls & 1;
        const int n = ((ks + p) / 2) * ((ks + q) / 2);
        if (t < n) break;
        t -= n;
    }
    const int TY = (ks + p) / 2, TX = (ks + q) / 2;
    const int ty = t / TX, tx = t % TX;
    const int ky = 2 * (TY - 1 - ty) + 1 - p, kx = 2 * (TX - 1 - tx) + 1 - q;
    const int k = chunk * 64 + h * 32 + 8 * q4 + e;
    const int n_ch = blk * bn + (nt / per_wave) * (bn / 2) + c * per_wave + (nt % per_wave);
    out[i] = (__bf16)w[(((int64_t)k * nc + n_ch) * ks + ky) * ks + kx];
}

// Shared epilogue.  C/D layout: row = 4*(lane>>4)+reg (pixel), col = lane&15 -> couts co_base + r*NT + j (NT consecutive).
// stats_partial (optional): per-tile (sum, sum of squares) of the STORED bf16 outputs per output channel, laid out
// [tile][2][cout] — exactly the partial-sum slabs the batch-norm finalize kernel consumes, so the BatchNorm that
// follows this conv skips its statistics pass over the tensor.  Fixed summation order (deterministic).
// UP (the stride-2 transposed kernels below): row m = (n, i, j) of the [up_h][up_w] source grid is stored at output pixel
// (n, 2 i + up_p, 2 j + up_q) of the [2 up_h][2 up_w] image.
template <int BN, int MI = 4, bool UP = false>
__device__ __forceinline__ void conv_epilogue(f32x4c (&acc)[MI][BN / 32], const float *__restrict__ bias, __bf16 *__restrict__ y,
                                              int64_t m0, int64_t m_total, int cout, int blk_n, int wm, int wn, int r, int q,
                                              char *smem, float *__restrict__ stats_partial, int tile, int up_h = 0, int up_w = 0,
                                              int up_p = 0, int up_q = 0, const BnBwd bb = BnBwd{}) {
    constexpr int NT = BN / 32;
    const int co_base = blk_n * BN + wn * (BN / 2);
    float bv[NT], s1[NT], s2[NT], bsc[NT], bsh[NT];
    const bool bnb = bb.z != nullptr;   // block-uniform: the sums are the backward sums of the batch norm whose output gradient this is
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bv[j] = bias ? bias[co_base + r * NT + j] : 0.f;
        s1[j] = 0.f;
        s2[j] = 0.f;
        bsc[j] = bnb ? bb.scale[co_base + r * NT + j] : 0.f;
        bsh[j] = bnb ? bb.shift[co_base + r * NT + j] : 0.f;
    }
    // batch-norm backward mode: every z piece of the tile is requested before the first one is used (one load latency, not 4 MI of them:
    // the first version loaded inside the store loop and a 128 x 128 tile's epilogue went from ~3 to ~20 us)
    constexpr int ZW = NT / 2;               // 32-bit words of z per (pixel, lane)
    unsigned zw[MI][4][ZW];
    if (bnb) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t m = m0 + 16 * MI * wm + 16 * i + 4 * q + reg;
                const unsigned *zp = reinterpret_cast<const unsigned *>(bb.z + (m < m_total ? m : m_total - 1) * cout + co_base + r * NT);
                if (ZW == 2) {
                    const uint2 t2 = *reinterpret_cast<const uint2 *>(zp);
                    zw[i][reg][0] = t2.x; zw[i][reg][ZW - 1] = t2.y;
                } else {
                    zw[i][reg][0] = zp[0];
                }
            }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t m = m0 + 16 * MI * wm + 16 * i + 4 * q + reg;
            if (m < m_total) {
                __bf16 v[NT];
                int64_t mo = m;
                if (UP) {
                    const int j = (int)(m % up_w);
                    const int64_t t = m / up_w;
                    const int i = (int)(t % up_h);
                    mo = ((t / up_h * 2 * up_h) + 2 * i + up_p) * (2 * up_w) + 2 * j + up_q;
                }
                if (bnb) {   // (never with UP: the launchers of the transposed forms pass no BnBwd)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        v[j] = (__bf16)(acc[i][j][reg] + bv[j]);
                        const unsigned w = zw[i][reg][j / 2];
                        const float zf = __uint_as_float((j & 1) ? (w & 0xffff0000u) : (w << 16));   // bf16 -> fp32
                        const float pre = fmaf(zf, bsc[j], bsh[j]);
                        float g = (float)v[j];
                        if (bb.act == 1) g = pre > 0.f ? g : 0.f;
                        else if (bb.act == 2) g *= gelu_grad_f(pre);
                        s1[j] += g;
                        s2[j] += g * zf;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        v[j] = (__bf16)(acc[i][j][reg] + bv[j]);
                        const float f = (float)v[j];
                        s1[j] += f;
                        s2[j] += f * f;
                    }
                }
                __bf16 *dst = y + mo * cout + co_base + r * NT;
                if (NT == 4) {
                    bf16x4c o;
                    o[0] = v[0]; o[1] = v[1 % NT]; o[2] = v[2 % NT]; o[3] = v[3 % NT];
                    *reinterpret_cast<bf16x4c *>(dst) = o;
                } else {
                    dst[0] = v[0];
                    dst[1] = v[1 % NT];
                }
            }
        }
    if (stats_partial) {   // block-uniform
        // lanes with the same r hold the same columns: fold the four q groups, then the two M waves through LDS
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            s1[j] += __shfl_xor(s1[j], 16, 64); s1[j] += __shfl_xor(s1[j], 32, 64);
            s2[j] += __shfl_xor(s2[j], 16, 64); s2[j] += __shfl_xor(s2[j], 32, 64);
        }
        float *red = reinterpret_cast<float *>(smem);   // [wm][2][BN]
        __syncthreads();                                // every wave is out of the K loop: its buffers are free
        if (q == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                red[(wm * 2 + 0) * BN + wn * (BN / 2) + r * NT + j] = s1[j];
                red[(wm * 2 + 1) * BN + wn * (BN / 2) + r * NT + j] = s2[j];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * BN; e += 256) {
            const int which = e / BN, col = e - which * BN;
            stats_partial[((int64_t)tile * 2 + which) * cout + blk_n * BN + col] = red[which * BN + col] + red[(2 + which) * BN + col];
        }
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NS = depth of the LDS ring.  NS = 2: double buffer, one __syncthreads() per K-step (whose release waits for every
// outstanding load).  NS >= 3: the loads of K-step s+NS-1 are issued while step s is computed and only the NEXT step's
// loads are waited for (s_waitcnt vmcnt(N) with N = the younger steps' load count, then a raw s_barrier), so NS-2
// K-steps of LDS-DMA stay in flight across the barrier.  Measured (r01, MI355X, 128->128 @ 4x188x188): NS=2 563 TFLOP/s,
// NS=3 360, NS=4 365 — the deeper rings cost the second resident workgroup per CU (96 / 128 KiB of LDS), and the four
// extra waves hide more latency than the extra K-steps in flight; the host launches NS=2 unless S2D_CONV_NS says otherwise.
// KS = kernel size (3, or 1: the 1x1 convs of the S2D module run the same tile pipeline with cin/64 K-steps per tile)
template <int BN, int NS, int KS = 3>
__global__ __launch_bounds__(256) void conv3x3_nhwc_bf16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ wpack,
                                                                const float *__restrict__ bias,
                                                                const __bf16 *__restrict__ zero_page, int n_img, int H, int W,
                                                                int cin, int cout, int pad, int stride, __bf16 *__restrict__ y,
                                                                float *__restrict__ stats_partial) {
    constexpr int NT = BN / 32;            // N tiles per wave
    constexpr int A_BYTES = 128 * 64 * 2;  // 16 KiB: [128 px][64 ch]
    constexpr int B_BYTES = 64 * BN * 2;   // [2][BN/16][64 lanes][8]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto abuf = [&](int b) -> char * { return smem + b * (A_BYTES + B_BYTES); };
    auto bbuf = [&](int b) -> char * { return smem + b * (A_BYTES + B_BYTES) + A_BYTES; };
    constexpr int LPS = 4 + (B_BYTES / 1024) / 4;   // global_load_lds instructions per wave and K-step (A: 4, B: 4 or 2)

    const int Ho = (H + 2 * pad - KS) / stride + 1, Wo = (W + 2 * pad - KS) / stride + 1;
    const int64_t m_total = (int64_t)n_img * Ho * Wo;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 15, q = lane >> 4;
    const int64_t m0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * 128;
    if (m0 >= m_total) return;
    const int blk_n = blockIdx.y;
    const int chunks = cin / 64;
    const int ksteps = KS * KS * chunks;

    // A staging: thread t moves 16-byte chunks t, t+256, t+512, t+768 of the [128][64ch] tile: pixel = id/8, part = id%8
    int a_img[4], a_y[4], a_x[4];
    bool a_ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int id = threadIdx.x + 256 * u;
        const int64_t m = m0 + id / 8;
        a_ok[u] = m < m_total;
        const int64_t mm = a_ok[u] ? m : 0;
        a_x[u] = (int)(mm % Wo) * stride;            // input coordinates of tap (0,0) + pad
        a_y[u] = (int)((mm / Wo) % Ho) * stride;
        a_img[u] = (int)(mm / ((int64_t)Wo * Ho));
    }
    const char *wsrc = reinterpret_cast<const char *>(wpack) + (int64_t)blk_n * B_BYTES;
    const int64_t wstep = (int64_t)(cout / BN) * B_BYTES;

    auto stage = [&](int s, int buf) {
        const int tap = s / chunks, chunk = s - tap * chunks;
        const int dy = tap / KS - pad, dx = tap % KS - pad;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = threadIdx.x + 256 * u;
            const int part = (id & 7) ^ ((id >> 3) & 7);   // XOR swizzle: LDS slot (row, s) holds channel part s ^ (row & 7)
            const int yy = a_y[u] + dy, xx = a_x[u] + dx;
            const bool ok = a_ok[u] && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const __bf16 *src = ok ? x + (((int64_t)a_img[u] * H + yy) * W + xx) * cin + chunk * 64 + part * 8 : zero_page;
            // LDS destination is wave-uniform base + lane*16: chunk ids of a wave are consecutive (id = 64*w' + lane)
            char *dst = abuf(buf) + (size_t)(id - lane) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
        constexpr int B_UNITS = B_BYTES / 1024;   // wave-instructions
#pragma unroll
        for (int u = 0; u < (B_UNITS + 3) / 4; ++u) {
            const int unit = u * 4 + wid;
            if (unit < B_UNITS) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(wsrc + (int64_t)s * wstep + unit * 1024 + lane * 16),
                    (__attribute__((address_space(3))) void *)(bbuf(buf) + unit * 1024), 16, 0, 0);
            }
        }
    };

    f32x4c acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4c{0.f, 0.f, 0.f, 0.f};

    if (NS == 2) {
        stage(0, 0);
        __syncthreads();
    } else {
        const int npre = ksteps < NS - 1 ? ksteps : NS - 1;
        for (int p = 0; p < npre; ++p) stage(p, p);
        if (npre >= 3) wait_vmcnt<2 * LPS>();
        else if (npre == 2) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    for (int s = 0; s < ksteps; ++s) {
        const int cur = NS == 2 ? (s & 1) : s % NS;
        if (NS == 2) {
            if (s + 1 < ksteps) stage(s + 1, cur ^ 1);
        } else if (s + NS - 1 < ksteps) {
            stage(s + NS - 1, (s + NS - 1) % NS);   // the slot read in iteration s-1
        }
        // A tile image: 16-byte chunk id = pixel*8 + part ; a lane's fragment for half h: pixel = 64*wm + 16*i + r, part = 4*h + q
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8c a[4], b[NT];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const bf16x8c *>(abuf(cur) + ((64 * wm + 16 * i + r) * 8 + ((4 * h + q) ^ (r & 7))) * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                b[j] = *reinterpret_cast<const bf16x8c *>(bbuf(cur) + ((h * (BN / 16) + wn * NT + j) * 64 + lane) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (NS == 2) {
            __syncthreads();   // tile s consumed by every wave; tile s+1 landed (the barrier's release drains the LDS-DMA)
        } else {
            if (s + 1 < ksteps) {   // step s+1 must have landed; the younger steps (at most NS-2) stay in flight
                const int ahead = ksteps - 2 - s < NS - 2 ? ksteps - 2 - s : NS - 2;
                if (ahead >= 2) wait_vmcnt<2 * LPS>();
                else if (ahead == 1) wait_vmcnt<LPS>();
                else wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
        }
    }

    conv_epilogue<BN>(acc, bias, y, m0, m_total, cout, blk_n, wm, wn, r, q, smem, stats_partial, (int)(m0 / 128));
}


// ---- 32-deep K-steps on a 3-slot ring -------------------------------------------------------------------------------
// The kernels above are bound by the load -> barrier round trip of a K-step, and what hides it is resident waves, not
// ring depth (see the NS note).  Halving the K-step (one tap x 32 channels: A 8 KiB + B 8 KiB) makes a 3-slot ring cost
// 48 KiB, so three workgroups (12 waves) stay resident per CU, each with two K-steps of LDS-DMA in flight across its
// barrier.  A image: [128 px][4 parts of 16 B], part index XOR-swizzled by (row >> 1) & 3 (checked against the
// ds_read_b128 lane groups: 16 distinct bank slots).  B = the h-th half of the 64-deep packed slab (same weight image).
// KS = kernel size of the direct form (3; 4 with stride 2 = the data gradient of ConvTranspose2d(4,2,1)).  UP = the stride-2
// TRANSPOSED form (pad 1) split into its four output parity classes (blockIdx.z = 2 p + q): output pixel (2 i + p, 2 j + q) reads
// the source pixels (i + p - a, j + q - b) with the taps ky = 2 a + 1 - p < KS, kx = 2 b + 1 - q < KS - (KS + p) / 2 x (KS + q) / 2
// dense taps per class, no zero taps.  KS = 4: the forward of ConvTranspose2d(4,2,1) (2 x 2 taps per class); KS = 3: the data
// gradient of a stride-2 3x3 conv (1, 2, 2, 4 taps).  The weight image lists the classes one after another, each in the window's
// row-major order (conv2d_pack_up_kernel); H, W are the SOURCE grid in that mode (pad / stride unused).
// r04 measurements on this kernel (128 -> 128 @ 4 x 188^2, 59.6 us = 700 TFLOP/s), ablations built as template flags and removed again:
//   MFMAs only (no staging, no fragment reads) 32.3 us | + fragment reads 40.9 | + staging (no reads) 48.8 | staging only 42.4 | no MFMAs 42.8
// i.e. the three streams add up instead of overlapping, and the MFMA-only floor is already 2 x the 17 us of the 2.5 PFLOP/s peak
// (clock ~1.9 GHz under this load, 4.3 tiles per CU = 5 rounds, per-tile prologue / epilogue).  Tried and dropped (all bit-identical):
//   * taller tiles on this kernel, MI = 6 / 8 (192 / 256 pixels, 2 workgroups per CU): 61 / 67 us;
//   * a tall-tile kernel sharing the A tile between the three kx taps (288 pixels per workgroup, A staged 3 x instead of 9 x, weights
//     once per 288 pixels: 250 MB of LDS-DMA per launch instead of 636 MB): 58-60 us; with all 13 fragment reads of a sub-step issued
//     up front (hipcc had paired them with lgkmcnt(0) waits) 58 us; with the LDS-DMA pieces interleaved one per MFMA row group 61 us.
//     Fewer staged bytes and fewer pieces per MFMA did not move the launch: the limit is not the L2 -> LDS byte rate.
// r05: (a) counters of this instantiation alone (profiles/r05_conv3x3_fwd_wgrad_sq_lds_tcp_counters.txt): matrix pipe 36 % busy, 2.2 resident waves per SIMD
//     on average, 48 % issue stalls / 25 % parked at s_waitcnt + barrier, no LDS conflicts (LDS array 8 % active), L2 hit rate 92 %: no unit saturates.
//   (b) tile height (tools/conv_bm_bench.py, back to back): 128 / 96 / 64 rows = 55.9 / 56.7 / 58.8 us - the 1 112-workgroup tail on 768 slots is not it.
//   (c) register staging instead of LDS-DMA (global_load_dwordx4 -> VGPR -> ds_write_b128, same ring and LDS image, loads of sub-step t+4 issued at
//     iteration t, 165 VGPRs, bit-identical, built as a template flag and removed again): 60.1 us against 55.4 us - the ~20 B/clk/CU the LDS-DMA path
//     delivers here (636 MB per launch) is not what holds the kernel either; the vector path only adds four 13-cycle ds_write_b128 per sub-step.
template <int BN, int MI, int KS = 3, bool UP = false>
__global__ __launch_bounds__(256, 3) void conv3x3_k32_nhwc_bf16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ wpack,
                                                                    const float *__restrict__ bias,
                                                                    const __bf16 *__restrict__ zero_page, int n_img, int H, int W,
                                                                    int cin, int cout, int pad, int stride, __bf16 *__restrict__ y,
                                                                    float *__restrict__ stats_partial, const BnBwd bb) {
    constexpr int NT = BN / 32;
    constexpr int BM = 32 * MI;                // output pixels per workgroup (2 x 2 waves, MI 16-row tiles per wave)
    constexpr int A_BYTES = BM * 32 * 2;       // 8 KiB at MI = 4
    constexpr int A_LOADS = (BM * 4 + 255) / 256;
    constexpr int B_HALF = 32 * BN * 2;        // half of a packed K-step slab: [BN/16][64 lanes][8]
    constexpr int LPS = A_LOADS + (B_HALF / 1024) / 4;   // LDS-DMA instructions per wave and sub-step (waves 2, 3: one less at MI = 3)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto abuf = [&](int b) -> char * { return smem + b * (A_BYTES + B_HALF); };
    auto bbuf = [&](int b) -> char * { return smem + b * (A_BYTES + B_HALF) + A_BYTES; };

    const int Ho = UP ? H : (H + 2 * pad - KS) / stride + 1, Wo = UP ? W : (W + 2 * pad - KS) / stride + 1;
    const int64_t m_total = (int64_t)n_img * Ho * Wo;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 15, q = lane >> 4;
    const int64_t m0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * BM;
    if (m0 >= m_total) return;
    const int blk_n = blockIdx.y;
    const int chunks = cin / 64;
    // tap window of this launch (UP: of this parity class) and the K-steps of the classes before it in the weight image
    const int up_p = UP ? (int)(blockIdx.z >> 1) : 0, up_q = UP ? (int)(blockIdx.z & 1) : 0;
    const int TY = UP ? (KS + up_p) / 2 : KS, TX = UP ? (KS + up_q) / 2 : KS;
    constexpr int N00 = (KS / 2) * (KS / 2), N01 = (KS / 2) * ((KS + 1) / 2);
    const int tap_off = !UP ? 0 : blockIdx.z == 0 ? 0 : blockIdx.z == 1 ? N00 : blockIdx.z == 2 ? N00 + N01 : N00 + 2 * N01;
    const int T = 2 * TY * TX * chunks;   // sub-steps: (tap, 64-channel chunk, half)

    // A staging: thread t moves the 16-byte chunks t and t+256 of the [128][32 ch] tile: row = id/4, slot = id%4.
    // Everything that depends on the thread is computed once: the address of the window's corner pixel (tap 0) and a
    // 9-bit mask of the taps that fall inside the image.  The K loop then only adds wave-uniform offsets kept by a
    // scalar cursor (tap, channel offset, weight pointer) - the first version recomputed tap / chunk / bounds with
    // divisions in every sub-step and spent 190 instructions per 16 MFMAs, i.e. ~760 issue cycles per wave and
    // sub-step against 256 MFMA cycles (rocprofv3 SQ_ACTIVE_INST_ANY vs SQ_VALU_MFMA_BUSY_CYCLES, profiles/).
    const bool short_wave = MI == 3 && wid >= 2;   // 384 chunks: the second A load only exists for waves 0, 1
    const __bf16 *a_base[A_LOADS];
    unsigned a_mask[A_LOADS];
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
        const int id = threadIdx.x + 256 * u;
        const int row = id >> 2;
        const int part = (id & 3) ^ ((row >> 1) & 3);
        const int64_t m = m0 + row;
        const bool ok = m < m_total;
        const int64_t mm = ok ? m : 0;
        const int ix0 = UP ? (int)(mm % Wo) + up_q - (TX - 1) : (int)(mm % Wo) * stride - pad;
        const int iy0 = UP ? (int)((mm / Wo) % Ho) + up_p - (TY - 1) : (int)((mm / Wo) % Ho) * stride - pad;
        const int img = (int)(mm / ((int64_t)Wo * Ho));
        a_base[u] = x + (((int64_t)img * H + iy0) * W + ix0) * cin + part * 8;
        unsigned mask = 0;
        constexpr int TAPS_MAX = UP ? ((KS + 1) / 2) * ((KS + 1) / 2) : KS * KS;
#pragma unroll
        for (int tap = 0; tap < TAPS_MAX; ++tap)
            if (ok && tap < TY * TX && (unsigned)(iy0 + tap / TX) < (unsigned)H && (unsigned)(ix0 + tap % TX) < (unsigned)W) mask |= 1u << tap;
        a_mask[u] = mask;
    }
    const int64_t wstep = (int64_t)(cout / BN) * (2 * B_HALF);
    // stage cursor (wave-uniform)
    const char *st_w = reinterpret_cast<const char *>(wpack) + (int64_t)tap_off * chunks * wstep + (int64_t)blk_n * (2 * B_HALF) + wid * 1024;
    int st_tap = 0, st_kx = 0, st_coff = 0, st_off = 0, st_h = 0;

    auto stage_next = [&](int buf) {
#pragma unroll
        for (int u = 0; u < A_LOADS; ++u) {
            if (u == 1 && short_wave) break;
            const int id = threadIdx.x + 256 * u;
            const bool ok = (a_mask[u] >> st_tap) & 1u;
            const __bf16 *src = ok ? a_base[u] + (st_off + st_coff) : zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(abuf(buf) + (size_t)(id - lane) * 16), 16, 0, 0);
        }
        constexpr int B_UNITS = B_HALF / 1024;
#pragma unroll
        for (int u = 0; u < B_UNITS / 4; ++u) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(st_w + u * 4096 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(bbuf(buf) + (u * 4 + wid) * 1024), 16, 0, 0);
        }
        // advance: (tap, chunk, half) order, the half is the fast index
        st_w += st_h ? wstep - B_HALF : (int64_t)B_HALF;
        st_h ^= 1;
        st_coff += 32;
        if (st_coff == cin) {
            st_coff = 0;
            ++st_tap;
            ++st_kx;
            st_off += cin;
            if (st_kx == TX) {
                st_kx = 0;
                st_off += (W - TX) * cin;
            }
        }
    };

    f32x4c acc[MI][NT];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4c{0.f, 0.f, 0.f, 0.f};

    // Fragments are double buffered in registers: the ds_reads of sub-step t+1 are issued right after the barrier and
    // fly under the 16 MFMAs of sub-step t, and so does the staging of sub-step t+3, so a wave's MFMA stream is only
    // interrupted by the barrier itself.  Ring protocol (slot of sub-step k = k % 3), iteration t:
    //   wait: own loads of t+1 landed (those of t+2 stay in flight), own fragment reads of t done;  barrier
    //   read fragments of t+1;  stage t+3 into the slot of t (its fragments are in registers);  MFMAs of t.
    const int a_rd = ((16 * MI * wm + r) * 4 + (q ^ ((r >> 1) & 3))) * 16;   // rows 16 MI wm + 16 i + r: + 1024 i bytes
    const int b_rd = (wn * NT * 64 + lane) * 16;
    auto read_frags = [&](int slot, bf16x8c (&a)[MI], bf16x8c (&b)[NT]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8c *>(abuf(slot) + a_rd + i * 1024);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8c *>(bbuf(slot) + b_rd + j * 1024);
    };
    auto mfmas = [&](const bf16x8c (&a)[MI], const bf16x8c (&b)[NT]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    // the waits are the builtin, not inline asm: the compiler's own waitcnt bookkeeping then knows the fragment reads
    // are done and does not put a second lgkmcnt(0) between the next reads and the MFMAs
    auto sync_in_flight = [&]() {   // vmcnt(one sub-step of this wave's loads) lgkmcnt(0)
        if (short_wave) __builtin_amdgcn_s_waitcnt(((LPS - 1) & 15) | 0x70 | (((LPS - 1) >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt((LPS & 15) | 0x70 | ((LPS >> 4) << 14));
        __builtin_amdgcn_s_barrier();
    };
    auto sync_all = [&]() {
        __builtin_amdgcn_s_waitcnt(0x70);   // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    };
    int rd = 1, st = 0;   // slots of sub-steps t+1 and t+3 (= t)
    auto rot = [&]() {
        rd = rd == 2 ? 0 : rd + 1;
        st = st == 2 ? 0 : st + 1;
    };
    bf16x8c fa[2][MI], fb[2][NT];
    stage_next(0);
    stage_next(1);
    stage_next(2);
    if (short_wave) __builtin_amdgcn_s_waitcnt(((2 * LPS - 2) & 15) | 0x70 | (((2 * LPS - 2) >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(((2 * LPS) & 15) | 0x70 | (((2 * LPS) >> 4) << 14));   // sub-step 0 landed
    __builtin_amdgcn_s_barrier();
    read_frags(0, fa[0], fb[0]);
    int t = 0;
    for (; t + 4 < T; t += 2) {   // T is even and >= 4 (host check); no conditional waits inside the loop
        sync_in_flight();
        read_frags(rd, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        stage_next(st);
        mfmas(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        rot();
        sync_in_flight();
        read_frags(rd, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        stage_next(st);
        mfmas(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        rot();
    }
    // t = T-4 .. T-1
    sync_in_flight();
    read_frags(rd, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    stage_next(st);   // sub-step T-1
    mfmas(fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    rot();
    sync_in_flight();   // T-2 landed, T-1 in flight
    read_frags(rd, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    rot();
    sync_all();
    read_frags(rd, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(fa[0], fb[0]);
    mfmas(fa[1], fb[1]);

    conv_epilogue<BN, MI, UP>(acc, bias, y, m0, m_total, cout, blk_n, wm, wn, r, q, smem, stats_partial,
                              (int)(m0 / BM) + (UP ? (int)(blockIdx.z * ((m_total + BM - 1) / BM)) : 0), H, W, up_p, up_q, bb);
}

// ---- warp-specialised form of the kernel above (r06 experiment) ------------------------------------------------------
// Same tiles, same LDS images, same 3-slot ring of 32-deep sub-steps, same MFMA order (bit-identical results) - but the staging has a wave of
// its own: a workgroup is FIVE (or six: NP = 2) waves, waves 0-3 (consumers: 2 x 2 tiles) only read fragments and issue MFMAs, wave 4 (and 5) issues every
// LDS-DMA piece of the workgroup (A: BM * 4 / 64 pieces, B: B_HALF / 1024 pieces per sub-step) and nothing else.  The consumers' instruction
// stream then holds no address arithmetic, no VMEM issue and no vmcnt wait: what r04's ablation showed "adding up instead of overlapping"
// (MFMAs only 32 us, + fragment reads 41, staging only 42, all together 56-59) sits in different waves.  Price on gfx950: registers are
// allocated per KERNEL, so the producer wave holds a consumer's ~130 VGPRs as well, and 5-wave workgroups fit two per CU where the 4-wave
// form fits three.  Ring protocol per iteration t (slot of sub-step k = k % 3):
//   producer:  wait vmcnt(pieces of one sub-step): sub-step t+1 has landed, t+2 in flight;  barrier;  stage t+3 into the slot of t
//   consumer:  wait lgkmcnt(0): own fragment reads of t+1 are done...;                        barrier;  read fragments of t+1; MFMAs of t
// Measured: see tools/conv_bm_bench.py (S2D_CONV_WS=1) and profiles/r06_conv3x3_warp_specialised_ab.txt.
template <int BN, int MI, int NP>   // NP producer waves (1 or 2): the pieces of a sub-step are dealt round-robin to them
__global__ __launch_bounds__(256 + 64 * NP, 2) void conv3x3_k32ws_nhwc_bf16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ wpack,
                                                                        const float *__restrict__ bias, const __bf16 *__restrict__ zero_page,
                                                                        int n_img, int H, int W, int cin, int cout, int pad, int stride,
                                                                        __bf16 *__restrict__ y, float *__restrict__ stats_partial) {
    constexpr int KS = 3;
    constexpr int NT = BN / 32;
    constexpr int BM = 32 * MI;
    constexpr int A_BYTES = BM * 32 * 2;
    constexpr int B_HALF = 32 * BN * 2;
    constexpr int A_PIECES = BM * 4 / 64;        // 1 KiB LDS-DMA pieces (64 lanes x 16 B) of the A tile of a sub-step
    constexpr int B_PIECES = B_HALF / 1024;
    static_assert(A_PIECES % NP == 0 && B_PIECES % NP == 0, "equal shares per producer wave");
    constexpr int LPS = (A_PIECES + B_PIECES) / NP;   // one producer wave's loads per sub-step (16 / NP at BN = 128, MI = 4)
    static_assert(2 * LPS <= 63, "vmcnt is six bits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto abuf = [&](int b) -> char * { return smem + b * (A_BYTES + B_HALF); };
    auto bbuf = [&](int b) -> char * { return smem + b * (A_BYTES + B_HALF) + A_BYTES; };

    const int Ho = (H + 2 * pad - KS) / stride + 1, Wo = (W + 2 * pad - KS) / stride + 1;
    const int64_t m_total = (int64_t)n_img * Ho * Wo;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t m0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * BM;
    if (m0 >= m_total) return;
    const int blk_n = blockIdx.y;
    const int chunks = cin / 64;
    const int T = 2 * KS * KS * chunks;   // sub-steps: (tap, 64-channel chunk, half)

    if (wid >= 4) {
        // ---------------- producer(s) ----------------
        const int pw = wid - 4;   // this producer's pieces: u = pw, pw + NP, ...
        const __bf16 *a_base[A_PIECES / NP];
        unsigned a_mask[A_PIECES / NP];
#pragma unroll
        for (int uu = 0; uu < A_PIECES / NP; ++uu) {
            const int u = pw + NP * uu;
            const int id = lane + 64 * u;
            const int row = id >> 2;
            const int part = (id & 3) ^ ((row >> 1) & 3);
            const int64_t m = m0 + row;
            const bool ok = m < m_total;
            const int64_t mm = ok ? m : 0;
            const int ix0 = (int)(mm % Wo) * stride - pad;
            const int iy0 = (int)((mm / Wo) % Ho) * stride - pad;
            const int img = (int)(mm / ((int64_t)Wo * Ho));
            a_base[uu] = x + (((int64_t)img * H + iy0) * W + ix0) * cin + part * 8;
            unsigned mask = 0;
#pragma unroll
            for (int tap = 0; tap < KS * KS; ++tap)
                if (ok && (unsigned)(iy0 + tap / KS) < (unsigned)H && (unsigned)(ix0 + tap % KS) < (unsigned)W) mask |= 1u << tap;
            a_mask[uu] = mask;
        }
        const int64_t wstep = (int64_t)(cout / BN) * (2 * B_HALF);
        const char *st_w = reinterpret_cast<const char *>(wpack) + (int64_t)blk_n * (2 * B_HALF);
        int st_tap = 0, st_kx = 0, st_coff = 0, st_off = 0, st_h = 0;
        auto stage_next = [&](int buf) {
#pragma unroll
            for (int uu = 0; uu < A_PIECES / NP; ++uu) {
                const int u = pw + NP * uu;
                const bool ok = (a_mask[uu] >> st_tap) & 1u;
                const __bf16 *src = ok ? a_base[uu] + (st_off + st_coff) : zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(abuf(buf) + (size_t)u * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int uu = 0; uu < B_PIECES / NP; ++uu) {
                const int u = pw + NP * uu;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(st_w + u * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(bbuf(buf) + u * 1024), 16, 0, 0);
            }
            st_w += st_h ? wstep - B_HALF : (int64_t)B_HALF;
            st_h ^= 1;
            st_coff += 32;
            if (st_coff == cin) {
                st_coff = 0;
                ++st_tap;
                ++st_kx;
                st_off += cin;
                if (st_kx == KS) {
                    st_kx = 0;
                    st_off += (W - KS) * cin;
                }
            }
        };
        int st = 0;
        stage_next(0);
        stage_next(1);
        stage_next(2);
        wait_vmcnt<2 * LPS>();             // sub-step 0 landed
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t + 3 < T; ++t) {  // iterations 0 .. T-4: the last stages sub-step T-1
            wait_vmcnt<LPS>();
            __builtin_amdgcn_s_barrier();
            stage_next(st);
            st = st == 2 ? 0 : st + 1;
        }
        wait_vmcnt<LPS>();                 // T-2 landed, T-1 in flight
        __builtin_amdgcn_s_barrier();
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (stats_partial) {               // the statistics fold of conv_epilogue: two workgroup barriers
            __syncthreads();
            __syncthreads();
        }
        return;
    }

    // ---------------- consumers ----------------
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 15, q = lane >> 4;
    f32x4c acc[MI][NT];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4c{0.f, 0.f, 0.f, 0.f};
    const int a_rd = ((16 * MI * wm + r) * 4 + (q ^ ((r >> 1) & 3))) * 16;
    const int b_rd = (wn * NT * 64 + lane) * 16;
    auto read_frags = [&](int slot, bf16x8c (&a)[MI], bf16x8c (&b)[NT]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8c *>(abuf(slot) + a_rd + i * 1024);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8c *>(bbuf(slot) + b_rd + j * 1024);
    };
    auto mfmas = [&](const bf16x8c (&a)[MI], const bf16x8c (&b)[NT]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    auto sync = [&]() {
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): own fragment reads done (no vector-memory loads in this wave)
        __builtin_amdgcn_s_barrier();
    };
    int rd = 1;
    auto rot = [&]() { rd = rd == 2 ? 0 : rd + 1; };
    bf16x8c fa[2][MI], fb[2][NT];
    __builtin_amdgcn_s_barrier();            // sub-step 0 landed (the producer waited for it)
    read_frags(0, fa[0], fb[0]);
    int t = 0;
    for (; t + 4 < T; t += 2) {
        sync();
        read_frags(rd, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        rot();
        sync();
        read_frags(rd, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        rot();
    }
    sync();                                  // t = T-4
    read_frags(rd, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    rot();
    sync();                                  // t = T-3
    read_frags(rd, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    rot();
    sync();                                  // t = T-2: everything landed
    read_frags(rd, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(fa[0], fb[0]);
    mfmas(fa[1], fb[1]);
    conv_epilogue<BN, MI, false>(acc, bias, y, m0, m_total, cout, blk_n, wm, wn, r, q, smem, stats_partial, (int)(m0 / BM), H, W, 0, 0);
}

// ---- pad = 1 variant with the A tile shared by the three kx taps ------------------------------------------------------
// With padding 1 the input pixel of output pixel m at tap (ky, kx) is the ky-row centre
// pixel of output pixel m + kx - 1, so one [130 px][64 ch] tile per (ky, channel chunk) serves all three kx taps: the tap
// is a row offset of the fragment read, and the two pixels per image row whose neighbour falls off the row are zeroed
// in the fragment (they would otherwise pick up the previous / next image row).  Staged bytes per three taps drop from
// 96 KiB to 64.6 KiB.  K order: (ky, chunk, kx); the weight image is the one of the kernel above (step = tap*chunks+chunk).
template <int BN>
__global__ __launch_bounds__(256) void conv3x3_p1_nhwc_bf16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ wpack,
                                                                   const float *__restrict__ bias,
                                                                   const __bf16 *__restrict__ zero_page, int n_img, int H, int W,
                                                                   int cin, int cout, __bf16 *__restrict__ y,
                                                                   float *__restrict__ stats_partial) {
    constexpr int NT = BN / 32;
    constexpr int A_ROWS = 136;                  // 130 used: output pixels m0-1 .. m0+128
    constexpr int A_BYTES = A_ROWS * 64 * 2;
    constexpr int A_CHUNKS = A_ROWS * 8;
    constexpr int A_LOADS = (A_CHUNKS + 255) / 256;
    constexpr int B_BYTES = 64 * BN * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto abuf = [&](int b) -> char * { return smem + b * A_BYTES; };
    auto bbuf = [&](int b) -> char * { return smem + 2 * A_BYTES + b * B_BYTES; };

    const int64_t m_total = (int64_t)n_img * H * W;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 15, q = lane >> 4;
    const int64_t m0 = (int64_t)xcd_tile(blockIdx.x, gridDim.x) * 128;
    if (m0 >= m_total) return;
    const int blk_n = blockIdx.y;
    const int chunks = cin / 64;
    const int T = 9 * chunks;

    // staging roles: chunk id = threadIdx.x + 256u -> tile row j = id/8 (output pixel m0 - 1 + j), LDS slot id%8
    int a_img[A_LOADS], a_y[A_LOADS], a_x[A_LOADS];
    bool a_ok[A_LOADS];
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
        const int id = threadIdx.x + 256 * u;
        const int64_t m = m0 - 1 + id / 8;
        a_ok[u] = id < 130 * 8 && m >= 0 && m < m_total;
        const int64_t mm = a_ok[u] ? m : 0;
        a_x[u] = (int)(mm % W);
        a_y[u] = (int)((mm / W) % H);
        a_img[u] = (int)(mm / ((int64_t)W * H));
    }
    const char *wsrc = reinterpret_cast<const char *>(wpack) + (int64_t)blk_n * B_BYTES;
    const int64_t wstep = (int64_t)(cout / BN) * B_BYTES;

    auto stage_a = [&](int g, int buf) {   // g = ky*chunks + chunk
        const int ky = g / chunks, chunk = g - ky * chunks;
#pragma unroll
        for (int u = 0; u < A_LOADS; ++u) {
            const int id = threadIdx.x + 256 * u;
            if (id < A_CHUNKS) {   // wave-uniform (A_CHUNKS is a multiple of 64)
                const int part = (id & 7) ^ ((id >> 3) & 7);
                const int yy = a_y[u] + ky - 1;
                const bool ok = a_ok[u] && (unsigned)yy < (unsigned)H;
                const __bf16 *src = ok ? x + (((int64_t)a_img[u] * H + yy) * W + a_x[u]) * cin + chunk * 64 + part * 8 : zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(abuf(buf) + (size_t)(id - lane) * 16), 16, 0, 0);
            }
        }
    };
    auto stage_b = [&](int t, int buf) {   // t = (ky*chunks + chunk)*3 + kx  ->  packed step (ky*3+kx)*chunks + chunk
        const int g = t / 3, kx = t - 3 * g;
        const int ky = g / chunks, chunk = g - ky * chunks;
        const int s = (ky * 3 + kx) * chunks + chunk;
        constexpr int B_UNITS = B_BYTES / 1024;
#pragma unroll
        for (int u = 0; u < (B_UNITS + 3) / 4; ++u) {
            const int unit = u * 4 + wid;
            if (unit < B_UNITS) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(wsrc + (int64_t)s * wstep + unit * 1024 + lane * 16),
                    (__attribute__((address_space(3))) void *)(bbuf(buf) + unit * 1024), 16, 0, 0);
            }
        }
    };

    // per-lane border flags of the 4 M tiles: bit i of edge_l / edge_r = this lane's pixel of tile i sits at x == 0 / x == W-1
    unsigned edge_l = 0, edge_r = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + 64 * wm + 16 * i + r;
        const int xx = (int)((m < m_total ? m : 0) % W);
        edge_l |= (xx == 0 ? 1u : 0u) << i;
        edge_r |= (xx == W - 1 ? 1u : 0u) << i;
    }

    f32x4c acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4c{0.f, 0.f, 0.f, 0.f};

    stage_a(0, 0);
    stage_b(0, 0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int g = t / 3, kx = t - 3 * g;
        if (t + 1 < T) {
            stage_b(t + 1, (t + 1) & 1);
            if (kx == 2) stage_a(g + 1, (g + 1) & 1);   // buffer last read in group g-1
        }
        const char *ab = abuf(g & 1), *bb = bbuf(t & 1);
        const unsigned dead = kx == 0 ? edge_l : (kx == 2 ? edge_r : 0u);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8c a[4], b[NT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 64 * wm + 16 * i + r + kx;   // tile row of output pixel (64wm+16i+r) shifted by kx-1, +1 halo
                a[i] = *reinterpret_cast<const bf16x8c *>(ab + (row * 8 + ((4 * h + q) ^ (row & 7))) * 16);
                if ((dead >> i) & 1u) a[i] = bf16x8c{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
                b[j] = *reinterpret_cast<const bf16x8c *>(bb + ((h * (BN / 16) + wn * NT + j) * 64 + lane) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    conv_epilogue<BN>(acc, bias, y, m0, m_total, cout, blk_n, wm, wn, r, q, smem, stats_partial, (int)(m0 / 128));
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_conv2d3x3_supported(int cin, int cout) { return cin >= 64 && cin % 64 == 0 && cout >= 64 && cout % 64 == 0; }

static int conv_bn(int cout) { return cout % 128 == 0 ? 128 : 64; }

extern "C" int s2d_conv2d3x3_pack_weights_bf16(const float *weight, int cin, int cout, int transpose_flip, int weight_nhwc,
                                               void *packed, s2d_stream_t stream) {
    // (cin, cout) = dimensions of the PACKED operand; weight is torch [Cout][Cin][3][3] of the forward conv
    S2D_CHECK_ARG(weight && packed, "conv2d_pack: null argument");
    if (!s2d_conv2d3x3_supported(cin, cout)) {
        set_error("conv2d_pack: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t total = (int64_t)9 * cin * cout;
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, cin,
                       cout, conv_bn(cout), transpose_flip, weight_nhwc, 9, (__bf16 *)packed);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* both operands of a conv layer in one launch: weight = torch [cout][cin][k][k] of the forward conv (k = 3: taps 9, k = 1: taps 1);
 * packed_fwd = the [cin -> cout] image, packed_dgrad = the [cout -> cin] image with the taps mirrored */
static int conv2d_pack_pair(const float *weight, int cin, int cout, int weight_nhwc, int taps, void *packed_fwd, void *packed_dgrad,
                            s2d_stream_t stream) {
    S2D_CHECK_ARG(weight && packed_fwd && packed_dgrad, "conv2d_pack_pair: null argument");
    if (!s2d_conv2d3x3_supported(cin, cout) || !s2d_conv2d3x3_supported(cout, cin)) {
        set_error("conv2d_pack_pair: unsupported channels %d <-> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t total = (int64_t)taps * cin * cout;
    hipLaunchKernelGGL(conv2d_pack_pair_kernel, dim3((unsigned)ceil_div(total, 256), 2), dim3(256), 0, (hipStream_t)stream, weight, cin, cout,
                       conv_bn(cout), conv_bn(cin), weight_nhwc, taps, (__bf16 *)packed_fwd, (__bf16 *)packed_dgrad);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_conv2d3x3_pack_weights_pair_bf16(const float *weight, int cin, int cout, int weight_nhwc, void *packed_fwd, void *packed_dgrad,
                                                    s2d_stream_t stream) {
    return conv2d_pack_pair(weight, cin, cout, weight_nhwc, 9, packed_fwd, packed_dgrad, stream);
}

extern "C" int s2d_conv2d1x1_pack_weights_pair_bf16(const float *weight, int cin, int cout, void *packed_fwd, void *packed_dgrad,
                                                    s2d_stream_t stream) {
    return conv2d_pack_pair(weight, cin, cout, 0, 1, packed_fwd, packed_dgrad, stream);
}

// ---- r04: every dense-conv weight image of the model in ONE launch -------------------------------------------------------------------
// After an optimizer step ~40 layers re-pack their bf16 images (forward + data-gradient operand): 38 launches of ~5 us.  The batch
// entry takes a host table of the same arguments the single entries take and launches one kernel whose blocks find their layer by a
// block-uniform scan of the table (the pattern of the fused Adam, optim.hip).
constexpr int PACK_MAX = 64;
struct PackTable {
    const float *w[PACK_MAX];
    __bf16 *out_f[PACK_MAX];
    __bf16 *out_d[PACK_MAX];       // null: single image (out_f, built with tf[i])
    int cin[PACK_MAX], cout[PACK_MAX], taps[PACK_MAX], nhwc[PACK_MAX], tf[PACK_MAX];
    int first_block[PACK_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void conv2d_pack_batch_kernel(const PackTable tb) {
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.first_block[ti + 1]) ++ti;
    const int64_t i = (int64_t)((int)blockIdx.x - tb.first_block[ti]) * 256 + threadIdx.x;
    const int cin = tb.cin[ti], cout = tb.cout[ti];
    if (tb.out_d[ti]) {   // pair: y = 0 forward [cin -> cout], y = 1 data gradient [cout -> cin], taps mirrored
        if (blockIdx.y == 0) conv2d_pack_element8(tb.w[ti], cin, cout, cout % 128 == 0 ? 128 : 64, 0, tb.nhwc[ti], tb.taps[ti], i, tb.out_f[ti]);
        else conv2d_pack_element8(tb.w[ti], cout, cin, cin % 128 == 0 ? 128 : 64, 1, tb.nhwc[ti], tb.taps[ti], i, tb.out_d[ti]);
    } else if (blockIdx.y == 0) {
        conv2d_pack_element8(tb.w[ti], cin, cout, cout % 128 == 0 ? 128 : 64, tb.tf[ti], tb.nhwc[ti], tb.taps[ti], i, tb.out_f[ti]);
    }
}

/* n <= 64 layers; per layer the arguments of s2d_conv2d{3x3,1x1}_pack_weights[_pair]_bf16: weight, (cin, cout) of the PACKED operand for
 * single images / of the forward conv for pairs, taps (9 or 1), weight_nhwc, transpose_flip (single images), packed_fwd, packed_dgrad
 * (NULL = single image).  All arrays live on the host. */
extern "C" int s2d_conv2d_pack_batch_bf16(int n, const float *const *weights, const int32_t *cin, const int32_t *cout, const int32_t *taps,
                                          const int32_t *weight_nhwc, const int32_t *transpose_flip, void *const *packed_fwd,
                                          void *const *packed_dgrad, s2d_stream_t stream) {
    S2D_CHECK_ARG(n >= 0 && n <= PACK_MAX && (n == 0 || (weights && cin && cout && taps && weight_nhwc && transpose_flip && packed_fwd && packed_dgrad)),
                  "conv2d_pack_batch: 0..%d layers", PACK_MAX);
    if (n == 0) return S2D_OK;
    PackTable tb;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        S2D_CHECK_ARG(weights[i] && packed_fwd[i] && (taps[i] == 9 || taps[i] == 1), "conv2d_pack_batch: bad layer %d", i);
        if (!s2d_conv2d3x3_supported(cin[i], cout[i]) || (packed_dgrad[i] && !s2d_conv2d3x3_supported(cout[i], cin[i]))) {
            set_error("conv2d_pack_batch: unsupported channels %d <-> %d (layer %d)", cin[i], cout[i], i);
            return S2D_ERR_UNSUPPORTED;
        }
        tb.w[i] = weights[i]; tb.out_f[i] = (__bf16 *)packed_fwd[i]; tb.out_d[i] = (__bf16 *)packed_dgrad[i];
        tb.cin[i] = cin[i]; tb.cout[i] = cout[i]; tb.taps[i] = taps[i]; tb.nhwc[i] = weight_nhwc[i]; tb.tf[i] = transpose_flip[i];
        tb.first_block[i] = blocks;
        blocks += (int)ceil_div((int64_t)taps[i] * cin[i] * cout[i] / 8, 256);   // a thread packs eight consecutive elements
    }
    tb.first_block[n] = blocks;
    tb.count = n;
    hipLaunchKernelGGL(conv2d_pack_batch_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, (hipStream_t)stream, tb);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// ---- launch plan --------------------------------------------------------------------------------------------------
static int conv_env_int(const char *name) {
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
}
static bool conv_use_shared_a(int cin, int bn, int pad, int stride) {
    // tap-shared A tile: measured on MI355X (r01) it wins where the A tile dominates the staged bytes (64-wide column
    // blocks with many input channels: 512->64 196 -> 141 us) and is neutral-to-slightly-slower at 128-wide blocks.
    // S2D_CONV_SHARED_A=0/1 forces the choice for A/B runs.
    const char *force = getenv("S2D_CONV_SHARED_A");
    return pad == 1 && stride == 1 && (force ? force[0] == '1' : (bn == 64 && cin >= 128));
}
static bool conv_use_k32(int bn) {
    // the 32-deep / 3-slot-ring kernel is the default for both column-block widths (64-wide blocks: 64->64@4x188^2 26 us
    // against 30 us on the 64-deep double buffer since the cursor rewrite); S2D_CONV_NS=2/3/4 forces the 64-deep variants
    static const int ns_env = conv_env_int("S2D_CONV_NS");
    (void)bn;
    return ns_env == 32 || ns_env == 0;
}
// Output pixels per workgroup of the k32 kernel (128, 96 or 64).  The BEV launches are small against the chip (a
// 256->256 conv on 4 x 94 x 94 pixels is 552 tiles of 128 x 128 for 768 resident workgroups).  Fitted to the measured
// launch times (r01, 10 shapes x 3 tile heights): time ~ max(1, workgroups / resident slots) x (tile rows + 48), i.e.
// shorter tiles pay while the launch is below one round of resident workgroups, taller ones amortise the per-step
// overhead once it is above.
static int conv_k32_rows(int64_t m, int col_blocks) {
    static const int force = conv_env_int("S2D_CONV_BM");
    if (force == 64 || force == 96 || force == 128) return force;
    static int slots = 0;
    if (!slots) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        slots = 3 * n;
    }
    int best = 128;
    int64_t best_cost = -1;
    for (int bm : {128, 96, 64}) {
        const int64_t blocks = ceil_div(m, bm) * col_blocks;
        const int64_t cost = (blocks > slots ? blocks : slots) * (bm + 48);
        if (best_cost < 0 || cost < best_cost) {
            best = bm;
            best_cost = cost;
        }
    }
    return best;
}
static int conv_tile_rows(int64_t m, int cin, int cout, int pad, int stride) {
    const int bn = conv_bn(cout);
    if (conv_use_shared_a(cin, bn, pad, stride) || !conv_use_k32(bn)) return 128;
    return bn == 128 ? conv_k32_rows(m, cout / bn) : 128;
}

extern "C" int s2d_conv2d3x3_tile_rows(int n_img, int h, int w, int cin, int cout, int pad, int stride) {
    if (!s2d_conv2d3x3_supported(cin, cout) || n_img <= 0 || h + 2 * pad < 3 || w + 2 * pad < 3 || stride < 1) return 0;
    const int ho = (h + 2 * pad - 3) / stride + 1, wo = (w + 2 * pad - 3) / stride + 1;
    return conv_tile_rows((int64_t)n_img * ho * wo, cin, cout, pad, stride);
}

extern "C" int64_t s2d_conv2d3x3_stats_tiles(int n_img, int h, int w, int cin, int cout, int pad, int stride) {
    if (!s2d_conv2d3x3_supported(cin, cout) || n_img <= 0 || h + 2 * pad < 3 || w + 2 * pad < 3 || stride < 1) return 0;
    const int ho = (h + 2 * pad - 3) / stride + 1, wo = (w + 2 * pad - 3) / stride + 1;
    const int64_t m = (int64_t)n_img * ho * wo;
    return ceil_div(m, conv_tile_rows(m, cin, cout, pad, stride));
}

static int conv3x3_run(const void *x, const void *packed_weight, const float *bias, const void *zero_page, int n_img, int h, int w, int cin,
                       int cout, int pad, int stride, void *y, float *stats_partial, const BnBwd bb, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && packed_weight && zero_page && y && n_img > 0 && h > 0 && w > 0 && (pad == 0 || pad == 1) &&
                      (stride == 1 || stride == 2), "conv2d3x3: bad argument");
    if (!s2d_conv2d3x3_supported(cin, cout)) {
        set_error("conv2d3x3: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int ho = (h + 2 * pad - 3) / stride + 1, wo = (w + 2 * pad - 3) / stride + 1;
    S2D_CHECK_ARG(h + 2 * pad >= 3 && w + 2 * pad >= 3, "conv2d3x3: empty output");
    const int64_t m = (int64_t)n_img * ho * wo;
    hipStream_t st = (hipStream_t)stream;
    const int bn = conv_bn(cout);
    const dim3 grid(xcd_grid(ceil_div(m, 128)), cout / bn), blk(256);
    const bool shared_a = conv_use_shared_a(cin, bn, pad, stride);
    if (bb.z && (shared_a || !conv_use_k32(bn))) {   // (callers ask s2d_conv2d3x3_bnbwd_supported first)
        set_error("conv2d3x3: the batch-norm backward epilogue needs the 32-deep kernel (%d -> %d, pad %d, stride %d)", cin, cout, pad, stride);
        return S2D_ERR_UNSUPPORTED;
    }
    if (shared_a) {
        const size_t lds = 2 * (136 * 64 * 2) + 2 * (size_t)(64 * bn * 2);
        if (bn == 128) {
            auto kern = conv3x3_p1_nhwc_bf16_kernel<128>;
            static bool attr_set = false;
            if (!attr_set) {
                S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_set = true;
            }
            hipLaunchKernelGGL(kern, grid, blk, lds, st, (const __bf16 *)x, (const __bf16 *)packed_weight, bias,
                               (const __bf16 *)zero_page, n_img, h, w, cin, cout, (__bf16 *)y, stats_partial);
        } else {
            auto kern = conv3x3_p1_nhwc_bf16_kernel<64>;
            static bool attr_set = false;
            if (!attr_set) {
                S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_set = true;
            }
            hipLaunchKernelGGL(kern, grid, blk, lds, st, (const __bf16 *)x, (const __bf16 *)packed_weight, bias,
                               (const __bf16 *)zero_page, n_img, h, w, cin, cout, (__bf16 *)y, stats_partial);
        }
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
    // ring depth of the LDS-DMA pipeline (S2D_CONV_NS=2/3/4 forces it for A/B runs)
    static const int ns_env = [] { const char *e = getenv("S2D_CONV_NS"); return e ? atoi(e) : 0; }();
    const int ns = (ns_env >= 2 && ns_env <= 4) ? ns_env : 2;
    if (conv_use_k32(bn)) {   // 32-deep K-steps, 3-slot ring
#define S2D_CONV_K32(BN_, MI_)                                                                                                  \
    hipLaunchKernelGGL((conv3x3_k32_nhwc_bf16_kernel<BN_, MI_>), dim3(xcd_grid(ceil_div(m, 32 * MI_)), cout / bn), blk,         \
                       3 * (size_t)(32 * MI_ * 64 + 64 * BN_), st, (const __bf16 *)x, (const __bf16 *)packed_weight, bias,      \
                       (const __bf16 *)zero_page, n_img, h, w, cin, cout, pad, stride, (__bf16 *)y, stats_partial, bb)
        static const int ws_env = conv_env_int("S2D_CONV_WS");   // r06 experiment: the warp-specialised form, 1 or 2 producer waves (stride 1, 128-wide blocks)
        if (bn == 128 && (ws_env == 1 || ws_env == 2) && stride == 1 && !bb.z) {
            const int rows = conv_k32_rows(m, cout / bn);
#define S2D_CONV_K32WS(MI_, NP_)                                                                                                      \
    hipLaunchKernelGGL((conv3x3_k32ws_nhwc_bf16_kernel<128, MI_, NP_>), dim3(xcd_grid(ceil_div(m, 32 * MI_)), cout / bn),             \
                       dim3(256 + 64 * NP_), 3 * (size_t)(32 * MI_ * 64 + 64 * 128), st, (const __bf16 *)x, (const __bf16 *)packed_weight, \
                       bias, (const __bf16 *)zero_page, n_img, h, w, cin, cout, pad, stride, (__bf16 *)y, stats_partial)
            if (ws_env == 2) {
                if (rows == 128) S2D_CONV_K32WS(4, 2);
                else if (rows == 96) S2D_CONV_K32WS(3, 2);
                else S2D_CONV_K32WS(2, 2);
            } else {
                if (rows == 128) S2D_CONV_K32WS(4, 1);
                else if (rows == 96) S2D_CONV_K32WS(3, 1);
                else S2D_CONV_K32WS(2, 1);
            }
#undef S2D_CONV_K32WS
        } else if (bn == 128) {
            const int rows = conv_k32_rows(m, cout / bn);
            if (rows == 128) S2D_CONV_K32(128, 4);
            else if (rows == 96) S2D_CONV_K32(128, 3);
            else S2D_CONV_K32(128, 2);
        } else {
            S2D_CONV_K32(64, 4);
        }
#undef S2D_CONV_K32
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
    const size_t lds = (size_t)ns * (128 * 64 * 2 + 64 * bn * 2);
#define S2D_CONV_LAUNCH(BN_, NS_)                                                                                         \
    do {                                                                                                                  \
        auto kern = conv3x3_nhwc_bf16_kernel<BN_, NS_>;                                                                   \
        static bool attr_set = false; /* once per instantiation; also keeps the call out of graph captures */            \
        if (!attr_set) {                                                                                                  \
            S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
            attr_set = true;                                                                                              \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, blk, lds, st, (const __bf16 *)x, (const __bf16 *)packed_weight, bias,              \
                           (const __bf16 *)zero_page, n_img, h, w, cin, cout, pad, stride, (__bf16 *)y, stats_partial);   \
    } while (0)
    if (bn == 128) {
        if (ns == 4) S2D_CONV_LAUNCH(128, 4); else if (ns == 3) S2D_CONV_LAUNCH(128, 3); else S2D_CONV_LAUNCH(128, 2);
    } else {
        if (ns == 4) S2D_CONV_LAUNCH(64, 4); else if (ns == 3) S2D_CONV_LAUNCH(64, 3); else S2D_CONV_LAUNCH(64, 2);
    }
#undef S2D_CONV_LAUNCH
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_conv2d3x3_nhwc_bf16(const void *x, const void *packed_weight, const float *bias, const void *zero_page,
                                       int n_img, int h, int w, int cin, int cout, int pad, int stride, void *y,
                                       float *stats_partial, s2d_stream_t stream) {
    return conv3x3_run(x, packed_weight, bias, zero_page, n_img, h, w, cin, cout, pad, stride, y, stats_partial, BnBwd{}, stream);
}

/* The conv as the DATA GRADIENT of the layer behind a batch norm + activation (include/s2d.h): y = dY of that batch norm, and
   bn_partial[tile][2][cout] receives its backward sums (sum g, sum g z; g = dY act'(z scale + shift)) - the slabs
   s2d_bn_partials_bwd_finalize_f32 folds.  Tiles as s2d_conv2d3x3_stats_tiles. */
extern "C" int s2d_conv2d3x3_bnbwd_supported(int cin, int cout, int pad, int stride) {
    if (!s2d_conv2d3x3_supported(cin, cout) || stride != 1) return 0;
    const int bn = conv_bn(cout);
    return !conv_use_shared_a(cin, bn, pad, stride) && conv_use_k32(bn);
}

extern "C" int s2d_conv2d3x3_nhwc_bf16_bnbwd(const void *x, const void *packed_weight, const void *zero_page, int n_img, int h, int w, int cin,
                                             int cout, int pad, void *y, const void *bn_z, const float *bn_scale, const float *bn_shift,
                                             int bn_act, float *bn_partial, s2d_stream_t stream) {
    S2D_CHECK_ARG(bn_z && bn_scale && bn_shift && bn_partial && bn_act >= 0 && bn_act <= 2, "conv2d3x3_bnbwd: bad argument");
    BnBwd bb;
    bb.z = (const __bf16 *)bn_z; bb.scale = bn_scale; bb.shift = bn_shift; bb.act = bn_act;
    return conv3x3_run(x, packed_weight, nullptr, zero_page, n_img, h, w, cin, cout, pad, 1, y, bn_partial, bb, stream);
}


// ---- stride-2 transposed forms and the 4x4 stride-2 conv on the 32-deep kernel -----------------------------------------------------
// decoder_1 / decoder_2 of the S2D module (/root/reference/det3d/models/necks/rpn.py:217-231: nn.ConvTranspose2d(256, 256, 4, 2, 1)
// at 47 -> 94 and 94 -> 188) and the backward of the RPN's stride-2 3x3 convs (rpn.py:126-133).  "up": source [n][h][w][kc] ->
// [n][2h][2w][nc]; ks = 4: ConvTranspose2d(4,2,1) forward from its weight [kc][nc][4][4]; ks = 3: data gradient of
// Conv2d(nc -> kc, 3, stride 2, pad 1) on a [2h][2w] input from its weight [kc][nc][3][3].
static bool convup_supported(int kc, int nc, int ks) {
    return (ks == 3 || ks == 4) && s2d_conv2d3x3_supported(kc, nc) && (ks == 4 || kc >= 128);   // >= 4 sub-steps per class
}
extern "C" int s2d_convup_supported(int kc, int nc, int ks) { return convup_supported(kc, nc, ks); }

extern "C" int s2d_convup_pack_weights_bf16(const float *weight, int kc, int nc, int ks, void *packed, s2d_stream_t stream) {
    S2D_CHECK_ARG(weight && packed, "convup_pack: null argument");
    if (!convup_supported(kc, nc, ks)) {
        set_error("convup_pack: unsupported %d -> %d, kernel %d", kc, nc, ks);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t total = (int64_t)ks * ks * kc * nc;
    hipLaunchKernelGGL(conv2d_pack_up_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, kc, nc,
                       conv_bn(nc), ks, (__bf16 *)packed);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

static int convup_rows(int64_t m, int nc) { return conv_bn(nc) == 128 ? conv_k32_rows(4 * m, nc / 128) : 128; }

extern "C" int64_t s2d_convup_stats_tiles(int n_img, int h, int w, int kc, int nc, int ks) {
    if (!convup_supported(kc, nc, ks) || n_img <= 0 || h <= 0 || w <= 0) return 0;
    const int64_t m = (int64_t)n_img * h * w;
    return 4 * ceil_div(m, convup_rows(m, nc));
}

extern "C" int s2d_convup_nhwc_bf16(const void *x, const void *packed_weight, const float *bias, const void *zero_page, int n_img, int h,
                                    int w, int kc, int nc, int ks, void *y, float *stats_partial, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && packed_weight && zero_page && y && n_img > 0 && h > 0 && w > 0, "convup: bad argument");
    if (!convup_supported(kc, nc, ks)) {
        set_error("convup: unsupported %d -> %d, kernel %d", kc, nc, ks);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t m = (int64_t)n_img * h * w;
    const int rows = convup_rows(m, nc);
    hipStream_t st = (hipStream_t)stream;
    if (conv_bn(nc) == 64) {   // 64-wide column blocks (the pillar S2D module's 64-channel decoder, pillar_encoder.py:337-394)
        if (ks == 4)
            hipLaunchKernelGGL((conv3x3_k32_nhwc_bf16_kernel<64, 4, 4, true>), dim3(xcd_grid(ceil_div(m, 128)), nc / 64, 4), dim3(256),
                               3 * (size_t)(128 * 64 + 64 * 64), st, (const __bf16 *)x, (const __bf16 *)packed_weight, bias,
                               (const __bf16 *)zero_page, n_img, h, w, kc, nc, 1, 2, (__bf16 *)y, stats_partial, BnBwd{});
        else
            hipLaunchKernelGGL((conv3x3_k32_nhwc_bf16_kernel<64, 4, 3, true>), dim3(xcd_grid(ceil_div(m, 128)), nc / 64, 4), dim3(256),
                               3 * (size_t)(128 * 64 + 64 * 64), st, (const __bf16 *)x, (const __bf16 *)packed_weight, bias,
                               (const __bf16 *)zero_page, n_img, h, w, kc, nc, 1, 2, (__bf16 *)y, stats_partial, BnBwd{});
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
#define S2D_CONVUP(MI_, KS_)                                                                                                       \
    hipLaunchKernelGGL((conv3x3_k32_nhwc_bf16_kernel<128, MI_, KS_, true>), dim3(xcd_grid(ceil_div(m, 32 * MI_)), nc / 128, 4),     \
                       dim3(256), 3 * (size_t)(32 * MI_ * 64 + 64 * 128), st, (const __bf16 *)x, (const __bf16 *)packed_weight,    \
                       bias, (const __bf16 *)zero_page, n_img, h, w, kc, nc, 1, 2, (__bf16 *)y, stats_partial, BnBwd{})
    if (ks == 4) {
        if (rows == 128) S2D_CONVUP(4, 4); else if (rows == 96) S2D_CONVUP(3, 4); else S2D_CONVUP(2, 4);
    } else {
        if (rows == 128) S2D_CONVUP(4, 3); else if (rows == 96) S2D_CONVUP(3, 3); else S2D_CONVUP(2, 3);
    }
#undef S2D_CONVUP
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// Conv2d(cin -> cout, 4, stride 2, pad 1) = the data gradient of ConvTranspose2d(cout -> cin, 4, 2, 1): weight [cout][cin][4][4]
// in the forward-conv sense, i.e. the ConvTranspose2d weight [its Cin = cout here][its Cout = cin here][4][4] as stored.
extern "C" int s2d_conv2d4x4s2_pack_weights_bf16(const float *weight, int cin, int cout, void *packed, s2d_stream_t stream) {
    S2D_CHECK_ARG(weight && packed, "conv2d4x4s2_pack: null argument");
    if (!s2d_conv2d3x3_supported(cin, cout)) {
        set_error("conv2d4x4s2_pack: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t total = (int64_t)16 * cin * cout;
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, cin, cout,
                       conv_bn(cout), 0, 0, 16, (__bf16 *)packed);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_conv2d4x4s2_nhwc_bf16(const void *x, const void *packed_weight, const void *zero_page, int n_img, int h, int w, int cin,
                                         int cout, void *y, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && packed_weight && zero_page && y && n_img > 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0,
                  "conv2d4x4s2: bad argument");
    if (!s2d_conv2d3x3_supported(cin, cout)) {
        set_error("conv2d4x4s2: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t m = (int64_t)n_img * (h / 2) * (w / 2);
    hipStream_t st = (hipStream_t)stream;
    if (conv_bn(cout) == 64) {
        hipLaunchKernelGGL((conv3x3_k32_nhwc_bf16_kernel<64, 4, 4, false>), dim3(xcd_grid(ceil_div(m, 128)), cout / 64), dim3(256),
                           3 * (size_t)(128 * 64 + 64 * 64), st, (const __bf16 *)x, (const __bf16 *)packed_weight, (const float *)nullptr,
                           (const __bf16 *)zero_page, n_img, h, w, cin, cout, 1, 2, (__bf16 *)y, (float *)nullptr, BnBwd{});
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
    const int rows = conv_k32_rows(m, cout / 128);
#define S2D_CONV4(MI_)                                                                                                             \
    hipLaunchKernelGGL((conv3x3_k32_nhwc_bf16_kernel<128, MI_, 4, false>), dim3(xcd_grid(ceil_div(m, 32 * MI_)), cout / 128),       \
                       dim3(256), 3 * (size_t)(32 * MI_ * 64 + 64 * 128), st, (const __bf16 *)x, (const __bf16 *)packed_weight,    \
                       (const float *)nullptr, (const __bf16 *)zero_page, n_img, h, w, cin, cout, 1, 2, (__bf16 *)y, (float *)nullptr, BnBwd{})
    if (rows == 128) S2D_CONV4(4); else if (rows == 96) S2D_CONV4(3); else S2D_CONV4(2);
#undef S2D_CONV4
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}


// ---- 1x1 convolution (stride 1) on the same tile pipeline ------------------------------------------------------------
// The S2D module's 1x1 convs (/root/reference/det3d/models/necks/rpn.py:186-253: fusion_sparse / fusion_dense 256->256, out_conv
// 256->640 at 188 x 188, the ConvNeXt point-wise pairs 256<->1024 at 47 x 47) are GEMMs [N*H*W, Cin] x [Cin, Cout] on NHWC rows: the
// 64-deep double-buffered kernel with KS = 1 (cin/64 K-steps per 128-pixel tile), same packed-weight image, same epilogue incl.
// the per-tile batch-norm statistics.  The data gradient is the same entry on dY with the transposed image.
extern "C" int s2d_conv2d1x1_pack_weights_bf16(const float *weight, int cin, int cout, int transpose, void *packed, s2d_stream_t stream) {
    // (cin, cout) = dimensions of the PACKED operand; weight is torch [Cout][Cin][1][1] of the forward conv (any memory format)
    S2D_CHECK_ARG(weight && packed, "conv2d1x1_pack: null argument");
    if (!s2d_conv2d3x3_supported(cin, cout)) {
        set_error("conv2d1x1_pack: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t total = (int64_t)cin * cout;
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, cin, cout,
                       conv_bn(cout), transpose, 0, 1, (__bf16 *)packed);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// 1x1 launches take the 32-deep three-slot-ring kernel with one tap (three workgroups per CU; measured r03 on the nine 1x1 shapes of
// the S2D module: 393 us against 514 us for the 64-deep double buffer, e.g. 256 -> 256 @ 4 x 188^2 44.8 vs 61.3 us); S2D_CONV1X1=k64
// forces the older route.  Rows per workgroup follow conv_k32_rows (128-wide column blocks only), as for the 3x3 launches.
static bool conv1x1_use_k32(int cin) {
    static const bool k64 = [] { const char *e = getenv("S2D_CONV1X1"); return e && e[0] == 'k' && e[1] == '6'; }();
    return !k64 && cin >= 128;
}
static int conv1x1_rows(int64_t m, int cin, int cout) {
    return conv1x1_use_k32(cin) && conv_bn(cout) == 128 ? conv_k32_rows(m, cout / 128) : 128;
}

extern "C" int64_t s2d_conv2d1x1_stats_tiles(int n_img, int h, int w, int cin, int cout) {
    const int64_t m = (int64_t)n_img * h * w;
    return ceil_div(m, conv1x1_rows(m, cin, cout));
}

static int conv1x1_run(const void *x, const void *packed_weight, const float *bias, const void *zero_page, int n_img, int h, int w, int cin,
                       int cout, void *y, float *stats_partial, const BnBwd bb, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && packed_weight && zero_page && y && n_img > 0 && h > 0 && w > 0, "conv2d1x1: bad argument");
    if (!s2d_conv2d3x3_supported(cin, cout)) {
        set_error("conv2d1x1: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t m = (int64_t)n_img * h * w;
    const int bn = conv_bn(cout);
    const dim3 grid(xcd_grid(ceil_div(m, 128)), cout / bn), blk(256);
    const size_t lds = (size_t)2 * (128 * 64 * 2 + 64 * bn * 2);
    hipStream_t st = (hipStream_t)stream;
    if (bb.z && !conv1x1_use_k32(cin)) {
        set_error("conv2d1x1: the batch-norm backward epilogue needs the 32-deep kernel (%d -> %d)", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    if (conv1x1_use_k32(cin)) {
#define S2D_CONV1_K32(BN_, MI_)                                                                                                    \
    hipLaunchKernelGGL((conv3x3_k32_nhwc_bf16_kernel<BN_, MI_, 1, false>), dim3(xcd_grid(ceil_div(m, 32 * MI_)), cout / bn), blk,   \
                       3 * (size_t)(32 * MI_ * 64 + 64 * BN_), st, (const __bf16 *)x, (const __bf16 *)packed_weight, bias,         \
                       (const __bf16 *)zero_page, n_img, h, w, cin, cout, 0, 1, (__bf16 *)y, stats_partial, bb)
        if (bn == 128) {
            const int rows = conv1x1_rows(m, cin, cout);
            if (rows == 128) S2D_CONV1_K32(128, 4); else if (rows == 96) S2D_CONV1_K32(128, 3); else S2D_CONV1_K32(128, 2);
        } else {
            S2D_CONV1_K32(64, 4);
        }
#undef S2D_CONV1_K32
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
#define S2D_CONV1_LAUNCH(BN_)                                                                                             \
    do {                                                                                                                  \
        auto kern = conv3x3_nhwc_bf16_kernel<BN_, 2, 1>;                                                                  \
        static bool attr_set = false;                                                                                     \
        if (!attr_set) {                                                                                                  \
            S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
            attr_set = true;                                                                                              \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, blk, lds, st, (const __bf16 *)x, (const __bf16 *)packed_weight, bias,              \
                           (const __bf16 *)zero_page, n_img, h, w, cin, cout, 0, 1, (__bf16 *)y, stats_partial);          \
    } while (0)
    if (bn == 128) S2D_CONV1_LAUNCH(128); else S2D_CONV1_LAUNCH(64);
#undef S2D_CONV1_LAUNCH
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_conv2d1x1_nhwc_bf16(const void *x, const void *packed_weight, const float *bias, const void *zero_page, int n_img, int h,
                                       int w, int cin, int cout, void *y, float *stats_partial, s2d_stream_t stream) {
    return conv1x1_run(x, packed_weight, bias, zero_page, n_img, h, w, cin, cout, y, stats_partial, BnBwd{}, stream);
}

extern "C" int s2d_conv2d1x1_bnbwd_supported(int cin, int cout) { return s2d_conv2d3x3_supported(cin, cout) && conv1x1_use_k32(cin); }

/* 1x1 data gradient with the batch-norm backward epilogue (see s2d_conv2d3x3_nhwc_bf16_bnbwd); tiles as s2d_conv2d1x1_stats_tiles */
extern "C" int s2d_conv2d1x1_nhwc_bf16_bnbwd(const void *x, const void *packed_weight, const void *zero_page, int n_img, int h, int w, int cin,
                                             int cout, void *y, const void *bn_z, const float *bn_scale, const float *bn_shift, int bn_act,
                                             float *bn_partial, s2d_stream_t stream) {
    S2D_CHECK_ARG(bn_z && bn_scale && bn_shift && bn_partial && bn_act >= 0 && bn_act <= 2, "conv2d1x1_bnbwd: bad argument");
    BnBwd bb;
    bb.z = (const __bf16 *)bn_z; bb.scale = bn_scale; bb.shift = bn_shift; bb.act = bn_act;
    return conv1x1_run(x, packed_weight, nullptr, zero_page, n_img, h, w, cin, cout, y, bn_partial, bb, stream);
}


// ---- 2x2 stride-2 convolution (forward) ----------------------------------------------------------------------------------
// encoder_1[0] of the S2D module (rpn.py:188: nn.Conv2d(c, 256, 2, 2)): KS = 2, stride 2, no padding on the same kernel (4 taps x
// cin/64 K-steps per tile), with the batch-norm statistics epilogue.  Its backward stays with the library.
extern "C" int s2d_conv2d2x2s2_pack_weights_bf16(const float *weight, int cin, int cout, int weight_nhwc, void *packed, s2d_stream_t stream) {
    S2D_CHECK_ARG(weight && packed, "conv2d2x2s2_pack: null argument");
    if (!s2d_conv2d3x3_supported(cin, cout)) {
        set_error("conv2d2x2s2_pack: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t total = (int64_t)4 * cin * cout;
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, cin, cout,
                       conv_bn(cout), 0, weight_nhwc, 4, (__bf16 *)packed);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int64_t s2d_conv2d2x2s2_stats_tiles(int n_img, int h, int w) { return ceil_div((int64_t)n_img * (h / 2) * (w / 2), 128); }

extern "C" int s2d_conv2d2x2s2_nhwc_bf16(const void *x, const void *packed_weight, const float *bias, const void *zero_page, int n_img, int h,
                                         int w, int cin, int cout, void *y, float *stats_partial, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && packed_weight && zero_page && y && n_img > 0 && h >= 2 && w >= 2, "conv2d2x2s2: bad argument");
    if (!s2d_conv2d3x3_supported(cin, cout)) {
        set_error("conv2d2x2s2: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t m = (int64_t)n_img * (h / 2) * (w / 2);
    const int bn = conv_bn(cout);
    const dim3 grid(xcd_grid(ceil_div(m, 128)), cout / bn), blk(256);
    const size_t lds = (size_t)2 * (128 * 64 * 2 + 64 * bn * 2);
    hipStream_t st = (hipStream_t)stream;
#define S2D_CONV2_LAUNCH(BN_)                                                                                             \
    do {                                                                                                                  \
        auto kern = conv3x3_nhwc_bf16_kernel<BN_, 2, 2>;                                                                  \
        static bool attr_set = false;                                                                                     \
        if (!attr_set) {                                                                                                  \
            S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
            attr_set = true;                                                                                              \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, blk, lds, st, (const __bf16 *)x, (const __bf16 *)packed_weight, bias,              \
                           (const __bf16 *)zero_page, n_img, h, w, cin, cout, 0, 2, (__bf16 *)y, stats_partial);          \
    } while (0)
    if (bn == 128) S2D_CONV2_LAUNCH(128); else S2D_CONV2_LAUNCH(64);
#undef S2D_CONV2_LAUNCH
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
