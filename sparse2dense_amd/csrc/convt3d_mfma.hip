// ConvTranspose3d(kernel 4, stride 2, padding 1) of the PCR head (/root/reference/det3d/models/necks/rpn.py:263-296:
// 32->32 on [B,32,5,188,188] and 16->3 on [B,16,10,376,376]) on the bf16 matrix cores, for the bf16 compute mode.
// Tensors stay NCDHW fp32 in HBM (what the neighbouring layers and the loss read); operands are rounded to bf16 while
// they are staged, accumulation is fp32.  r02 baseline (fp32 VALU / fp32 MFMA kernels of dense3d.hip, B=4):
// forward 3.7 + 1.05 ms, data gradient 2 x 2.06 ms, weight gradient 10.9 + 4.9 ms per step.
//
// Geometry: out o = 2h - 1 + k per axis (k = 0..3).  Output parity p = o & 1 selects two taps per axis:
//   p = 0 (o = 2h'):   (k = 1, h = h'), (k = 3, h = h' - 1)        p = 1 (o = 2h' + 1): (k = 0, h = h' + 1), (k = 2, h = h')
// so each of the 8 output parity classes is a dense 2x2x2 convolution over the input (forward), every input cell
// gathers 4x4x4 output positions (data gradient), and dW[ci][co][k] correlates x with the stride-2 sampled dout.
//
//   ct_fwd_mfma<CIN, NT>     block = one input row (n,hz,hy) x 64 cells; the 3x3 neighbouring rows are staged in LDS as
//                            [row][x][ci] bf16 (ci innermost: an A fragment = one ds_read_b128); wave w owns the output rows
//                            (pz,py) = (w>>1, w&1), both px; B = pre-packed weight fragments straight from L2.
//   ct_dgrad_mfma<COP, NT, G> block = one input row x 64 cells; G of the 16 (kz,ky) output rows are staged per round as
//                            de-interleaved arrays T[kx][c][co] = dout[co][2c-1+kx] (co innermost); wave w owns 16 cells.
//   ct_wgrad_mfma<CIT, COT, KG> no LDS staging: M = ci, N = co, K = cells of a row; lanes load their A / B fragments
//                            straight from the planar tensors (one 18-float window of dout feeds the 4 kx fragments).
#include "s2d_common.h"
#include <algorithm>
#include <cstdlib>

namespace s2d {

typedef float f32x4m __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8m __attribute__((ext_vector_type(8)));

// planar tensors through buffer instructions (s2d_common.h)
constexpr unsigned CT_OOB = BUF_OOB;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ct_rsrc(const void *base, unsigned bytes) { return buf_rsrc(base, bytes); }
__device__ __forceinline__ f32x4m ct_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return buf_load4(r, voff, soff); }
// bf16-stored gradients (r04, the fused PCR levels write dy in bf16): raw dwords, two elements each (low half = the even element)
typedef uint32_t u32x2m __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3m __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x3m ct_load3u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x3m, __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0));
}
__device__ __forceinline__ u32x4m ct_load4u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4m, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ u32x2m ct_load2u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x2m, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ uint32_t ct_hi_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // (a.hi, b.hi)
__device__ __forceinline__ uint32_t ct_lo_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }   // (a.lo, b.lo)
// two fp32 -> one dword of two bf16 (round to nearest even; low half = a): ONE v_cvt_pk_bf16_f32.  hipcc emits the instruction with a dummy second
// source per value and merges the halves with a v_perm_b32 (three instructions per pair) when the pair is built from two (__bf16) casts.
// a * b + c as ONE scalar v_fma_f32 the SLP vectoriser cannot pair into v_pk_fma_f32: rule 36 - packed FP32 results were timing-dependent in their high
// lane when another queue's MFMA kernel ran beside the kernel (r05, `pcr_level_bwd_dense`); the kernels below run beside other streams' MFMA work
__device__ __forceinline__ float ct_fma_scalar(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t ct_cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct CtDims {
    int n, d, h, w;   // batch and INPUT extents; output is 2d x 2h x 2w
    int cin, cout;
};

// tap tables: for parity p and tap a in {0,1}: kernel index and input offset
__device__ __forceinline__ int ct_k(int p, int a) { return p ? (a ? 2 : 0) : (a ? 3 : 1); }
__device__ __forceinline__ int ct_d(int p, int a) { return p ? (a ? 0 : 1) : (a ? -1 : 0); }

// ---- weight packing ------------------------------------------------------------------------------
// forward image: [class 8 = pz*4+py*2+px][kstep][nt][lane 64][8]; K index kk = 8q+e within a k-step:
//   CIN = 32: kstep = tap (a*4+b*2+c), ci = kk ;  CIN = 16: kstep = tap >> 1, tap = 2*kstep + (kk >> 4), ci = kk & 15
// column n = lane & 15 -> co = nt*16 + n.
__global__ __launch_bounds__(256) void ct_pack_fwd_kernel(const float *__restrict__ w, int cin, int cout, int nt_count, __bf16 *__restrict__ out) {
    const int ksteps = cin == 32 ? 8 : 4;
    const int total = 8 * ksteps * nt_count * 64 * 8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int r = i;
    const int e = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int nt = r % nt_count; r /= nt_count;
    const int ks = r % ksteps; r /= ksteps;
    const int cls = r;
    const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
    const int kk = 8 * (lane >> 4) + e;
    const int tap = cin == 32 ? ks : 2 * ks + (kk >> 4);
    const int ci = cin == 32 ? kk : (kk & 15);
    const int a = tap >> 2, b = (tap >> 1) & 1, c = tap & 1;
    const int co = nt * 16 + (lane & 15);
    float v = 0.f;
    if (co < cout) v = w[((((int64_t)ci * cout + co) * 4 + ct_k(pz, a)) * 4 + ct_k(py, b)) * 4 + ct_k(px, c)];
    out[i] = (__bf16)v;
}

// data-gradient image: [kzky 16][kstep][nt (ci tiles)][lane][8]; COP = 32: kstep = kx, co = kk ; COP = 8: one k-step,
// kx = kk >> 3, co = kk & 7.  column n -> ci = nt*16 + n.
// direct = 1 (the no-LDS kernel): K index kk <-> co = 8*kstep + 2*(kk >> 3) + ((kk & 7) >> 2), kx = kk & 3: a lane's 8 elements are the four
// kx samples of two channels = two 16-byte loads at position 2c-1.
// direct = 2 (cout <= 4): entry = (kz, ky pair); K index kk <-> ky = 2*pair + (kk >> 4), co = 2*((kk >> 3) & 1) + ((kk & 7) >> 2), kx = kk & 3:
// all four lane groups carry data (with the direct = 1 mapping 5 of the 8 channel slots of a 3-channel layer are empty).
__global__ __launch_bounds__(256) void ct_pack_dgrad_kernel(const float *__restrict__ w, int cin, int cout, int cop, int nt_count, int direct,
                                                            __bf16 *__restrict__ out) {
    const int ksteps = cop == 32 ? 4 : 1;
    const int total = 16 * ksteps * nt_count * 64 * 8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int r = i;
    const int e = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int nt = r % nt_count; r /= nt_count;
    const int ks = r % ksteps; r /= ksteps;
    const int kzky = r;
    const int kk = 8 * (lane >> 4) + e;
    const int kx = direct ? (kk & 3) : (cop == 32 ? ks : (kk >> 3));
    int co = direct ? 8 * ks + 2 * (kk >> 3) + ((kk & 7) >> 2) : (cop == 32 ? kk : (kk & 7));
    int kz = kzky >> 2, ky = kzky & 3;
    if (direct == 2) {   // <= 4 output channels: K = (2 ky) x (2 channel pairs) x (4 kx); entry = kz * 2 + ky pair, 8 entries used
        kz = kzky >> 1;
        ky = 2 * (kzky & 1) + (kk >> 4);
        co = kzky < 8 ? 2 * ((kk >> 3) & 1) + ((kk & 7) >> 2) : cout;
    }
    const int ci = nt * 16 + (lane & 15);
    float v = 0.f;
    if (co < cout && ci < cin) v = w[((((int64_t)ci * cout + co) * 4 + kz) * 4 + ky) * 4 + kx];
    out[i] = (__bf16)v;
}

// ---- forward -------------------------------------------------------------------------------------
constexpr int CT_TX = 64;   // cells per block along x

struct CtTileMap {   // tile index -> (n, chunk of yc rows, z, y in chunk, x tile): divisors d*h*xt, d*yc*xt, yc*xt, xt
    FastDiv per_n, per_chunk, per_plane, per_row;
    int yc;
};
static inline CtTileMap ct_tile_map(int d, int h, int xt, int yc) {
    return CtTileMap{fastdiv_make((uint32_t)(d * h * xt)), fastdiv_make((uint32_t)(d * yc * xt)), fastdiv_make((uint32_t)(yc * xt)),
                     fastdiv_make((uint32_t)xt), yc};
}
// rows per chunk such that the rows a chunk keeps live across one z step (`live_rows(yc)` rows of `row_bytes`) stay under half an L2
static inline int ct_chunk_rows(int64_t row_bytes, int rows_per_y, int halo, int planes) {
    int yc = 16;
    while (yc > 1 && (int64_t)(rows_per_y * yc + halo) * planes * row_bytes > (2 << 20)) yc >>= 1;
    return yc;
}

// waves_per_eu(3): the 32 -> 32 instantiation otherwise takes 176 registers (2 waves per SIMD); at 156 a third workgroup per CU overlaps
// its staging round trip with the others' MFMA / store phases (372 -> 347 us)
// r04: TO = float or __bf16 - the element type of the written output.  The 724 / 543 MB raw outputs of the two up-samplers are read by
// four passes of the fused PCR level each (losses.hip): stored as bf16 they cost half the bytes; the statistics of the batch norm
// that follows are taken from the ROUNDED values, like the dense conv epilogue does.
template <typename TO> struct CtStore;
template <> struct CtStore<float> {
    static __device__ __forceinline__ float4 rnd(float4 v) { return v; }
    static __device__ __forceinline__ float rnd1(float v) { return v; }
    static __device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    static __device__ __forceinline__ void st1(float *p, float v) { *p = v; }
};
template <> struct CtStore<__bf16> {
    static __device__ __forceinline__ float rnd1(float v) { return (float)(__bf16)v; }
    static __device__ __forceinline__ float4 rnd(float4 v) { return float4{rnd1(v.x), rnd1(v.y), rnd1(v.z), rnd1(v.w)}; }
    static __device__ __forceinline__ void st4(__bf16 *p, float4 v) {   // v already rounded: the conversions are exact
        typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
        bf16x4s o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4s *>(p) = o;
    }
    static __device__ __forceinline__ void st1(__bf16 *p, float v) { *p = (__bf16)v; }
};

template <int CIN, int NT, int YR, typename TO = float>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void ct_fwd_mfma_kernel(const float *__restrict__ x, const __bf16 *__restrict__ wp, const float *__restrict__ bias,
                                                          CtDims s, int xtiles, CtTileMap map, TO *__restrict__ out,
                                                          float *__restrict__ stats_partial, const float *__restrict__ in_norm = nullptr) {
    // in_norm (r04, scale[CIN] | shift[CIN] or null): the input is the RAW tensor in front of a BatchNorm3d + ReLU, applied here while
    // staging - relu(fma(x, scale[ci], shift[ci])), the expression of s2d_bncm_apply_f32 - so the normalised tensor is never written
    // or read (cells outside the tensor stay zero: the padding is of the normalised tensor)
    constexpr int GROUPS = CIN / 8;          // 16-byte pieces per staged cell
    constexpr int XS = CT_TX + 2;            // staged columns: x0-1 .. x0+64
    constexpr int KSTEPS = CIN == 32 ? 8 : 4;
    // a block produces YR consecutive input rows y (the same z, x tile): 3 x (YR + 2) rows staged for them instead of 9 per row
    constexpr int RY = YR + 2, ROWS = 3 * RY;
    __shared__ __attribute__((aligned(16))) __bf16 xs[ROWS * XS * CIN];   // [zi * RY + yi][xx][ci]
    auto xa = [&](int cell, int g) -> int { return (cell * GROUPS + g) * 8; };
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    // Tile order (XCD-aware): every tile re-reads the rows of its 8 (z, y) neighbours.  Workgroup ids are dealt round-robin over the
    // 8 XCDs, so with tile = blockIdx the nine readers of a row sat on different XCDs and each private L2 fetched its own copy over
    // the fabric (3.3 GB for a 362 MB input: what bounded this kernel at 0.8 ms).  xcd_tile() gives an XCD a contiguous range of
    // tiles, ordered (n, chunk of yc rows, z, y in chunk, x tile): the z and y re-reads then hit that XCD's L2
    // (3 planes x (yc + 2) rows x w x CIN floats, yc chosen to keep that under 2 MB).
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int hgroups = (s.h + YR - 1) / YR;   // the tile map counts groups of YR rows
    if (tile >= s.n * s.d * hgroups * xtiles) return;
    const int n = (int)fastdiv((uint32_t)tile, map.per_n);
    int rem = tile - n * (int)map.per_n.d;
    const int chunk = (int)fastdiv((uint32_t)rem, map.per_chunk);
    rem -= chunk * (int)map.per_chunk.d;
    const int yc = min(map.yc, hgroups - chunk * map.yc);   // row groups of this chunk
    const int hz = yc == map.yc ? (int)fastdiv((uint32_t)rem, map.per_plane) : rem / (yc * xtiles);
    rem -= hz * (yc * xtiles);
    const int yr = (int)fastdiv((uint32_t)rem, map.per_row);
    const int hy0 = (chunk * map.yc + yr) * YR, xt = rem - yr * xtiles;
    const int x0 = xt * CT_TX;
    const int64_t cells = (int64_t)s.d * s.h * s.w;
    const float *xb = x + (int64_t)n * CIN * cells;
    // stage: the 64-cell body of a row as 16-byte loads (piece = (row9, group, 4 cells): 8 channel planes x float4 -> four 16-byte LDS
    // stores), the two halo columns (and everything when w % 4 != 0) with one dword per channel.  Strided dword gathers cost ~4x the
    // issue slots of 16-byte loads (DESIGN rule 7); this staging is what bounds the kernel.
    const bool vec = (s.w & 3) == 0;
    if (vec) {
        // every thread issues the loads of its halo piece (threads < 18 * GROUPS) and of its first body piece before converting any
        constexpr int HALO = ROWS * GROUPS * 2, BODY = ROWS * GROUPS * 16;
        static_assert(HALO <= 256, "one halo piece per thread");
        float hv[8];
        const bool has_halo = t < HALO;
        int h_slot = 0;
        if (has_halo) {
            const int c = t & 1, g = (t >> 1) % GROUPS, r9 = (t >> 1) / GROUPS;
            const int xx = c * (XS - 1);
            const int z = hz + r9 / RY - 1, y = hy0 + r9 % RY - 1, xp = x0 - 1 + xx;
            h_slot = xa(r9 * XS + xx, g);
            const bool ok = (unsigned)z < (unsigned)s.d && (unsigned)y < (unsigned)s.h && (unsigned)xp < (unsigned)s.w;
            const float *src = ok ? xb + ((int64_t)z * s.h + y) * s.w + xp + (int64_t)(g * 8) * cells : xb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float val = src[ok ? (int64_t)e * cells : 0];
                if (in_norm) val = fmaxf(fmaf(val, in_norm[g * 8 + e], in_norm[CIN + g * 8 + e]), 0.f);
                hv[e] = ok ? val : 0.f;
            }
        }
        // (p >> 4) advances by 16 per iteration, a multiple of GROUPS: a thread's channel group - and its eight scale / shift pairs - is fixed
        static_assert(16 % GROUPS == 0, "a thread keeps its channel group");
        float nsc[8], nsh[8];
        if (in_norm) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                nsc[e] = in_norm[((t >> 4) % GROUPS) * 8 + e];
                nsh[e] = in_norm[CIN + ((t >> 4) % GROUPS) * 8 + e];
            }
        }
        for (int p = t; p < BODY; p += 256) {
            const int xq = p & 15, g = (p >> 4) % GROUPS, r9 = (p >> 4) / GROUPS;
            const int z = hz + r9 / RY - 1, y = hy0 + r9 % RY - 1, xp = x0 + 4 * xq;
            const bool ok = (unsigned)z < (unsigned)s.d && (unsigned)y < (unsigned)s.h && xp < s.w;
            const float *src = ok ? xb + ((int64_t)z * s.h + y) * s.w + xp + (int64_t)(g * 8) * cells : xb;
            float4 f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = *reinterpret_cast<const float4 *>(src + (ok ? (int64_t)e * cells : 0));
            if (in_norm) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sc = nsc[e], sh = nsh[e];
                    f[e] = float4{fmaxf(fmaf(f[e].x, sc, sh), 0.f), fmaxf(fmaf(f[e].y, sc, sh), 0.f), fmaxf(fmaf(f[e].z, sc, sh), 0.f),
                                  fmaxf(fmaf(f[e].w, sc, sh), 0.f)};
                }
            }
            bf16x8m v[4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[0][e] = (__bf16)(ok ? f[e].x : 0.f); v[1][e] = (__bf16)(ok ? f[e].y : 0.f);
                v[2][e] = (__bf16)(ok ? f[e].z : 0.f); v[3][e] = (__bf16)(ok ? f[e].w : 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x8m *>(&xs[xa(r9 * XS + 1 + 4 * xq + j, g)]) = v[j];
        }
        if (has_halo) {
            bf16x8m v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)hv[e];
            *reinterpret_cast<bf16x8m *>(&xs[h_slot]) = v;
        }
    } else {
        for (int p = t; p < ROWS * GROUPS * XS; p += 256) {
            const int xx = p % XS, g = (p / XS) % GROUPS, r9 = p / (XS * GROUPS);
            const int z = hz + r9 / RY - 1, y = hy0 + r9 % RY - 1, xp = x0 - 1 + xx;
            bf16x8m v;
            if ((unsigned)z < (unsigned)s.d && (unsigned)y < (unsigned)s.h && (unsigned)xp < (unsigned)s.w) {
                const float *src = xb + ((int64_t)z * s.h + y) * s.w + xp + (int64_t)(g * 8) * cells;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float val = src[(int64_t)e * cells];
                    if (in_norm) val = fmaxf(fmaf(val, in_norm[g * 8 + e], in_norm[CIN + g * 8 + e]), 0.f);
                    v[e] = (__bf16)val;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            }
            *reinterpret_cast<bf16x8m *>(&xs[xa(r9 * XS + xx, g)]) = v;
        }
    }
    __syncthreads();
    const int pz = wid >> 1, py = wid & 1;
    const int r = lane & 15, q = lane >> 4;
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    const int oz = 2 * hz + pz;
    TO *ob = out + (int64_t)n * s.cout * od * oh * ow;
    using ST = CtStore<TO>;
    float st1[NT], st2[NT];   // per-channel (sum, sum of squares) of this lane's outputs: the batch norm that follows skips its statistics pass
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        st1[nt] = 0.f;
        st2[nt] = 0.f;
    }
    // <= 4 output channels (the last up-sampler, 16 -> 3): only 4 of 16 column lanes hold outputs, so storing from the accumulator
    // layout issues 32 store instructions per wave with 12 live lanes each.  The wave transposes its row through LDS instead and
    // stores whole 512-byte channel rows with every lane (bias and statistics on the float4s).
    __shared__ __attribute__((aligned(16))) float tr[NT == 1 ? 4 * 4 * 2 * CT_TX : 4];   // [wave][co < 4][2 * CT_TX]
    const bool narrow = NT == 1 && s.cout <= 4 && (ow & 3) == 0;
    __shared__ float sred[4][2][NT * 16];
    float na[2] = {0.f, 0.f}, nb[2] = {0.f, 0.f};   // narrow path: this lane's sums for channel (lane + 64 k) >> 5
    for (int yy = 0; yy < YR; ++yy) {
        const int hy = hy0 + yy;
        if (hy >= s.h) break;   // block-uniform
        const int oy = 2 * hy + py;
        f32x4m acc[2][4][NT];
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[px][mt][nt] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int cls = pz * 4 + py * 2 + px;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                bf16x8m b[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[nt] = *reinterpret_cast<const bf16x8m *>(wp + ((((int64_t)cls * KSTEPS + ks) * NT + nt) * 64 + lane) * 8);
                const int tap = CIN == 32 ? ks : 2 * ks + (q >> 1);
                const int g = CIN == 32 ? q : (q & 1);
                const int ta = tap >> 2, tb = (tap >> 1) & 1, tc = tap & 1;
                const int r9 = (ct_d(pz, ta) + 1) * RY + (ct_d(py, tb) + 1 + yy);
                const int dx = ct_d(px, tc);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int xx = mt * 16 + r + 1 + dx;
                    const bf16x8m a = *reinterpret_cast<const bf16x8m *>(&xs[xa(r9 * XS + xx, g)]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[px][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[nt], acc[px][mt][nt], 0, 0, 0);
                }
            }
        }
        // epilogue: lane holds column co = nt*16 + r and the 4 cells 4q..4q+3 of every m-tile, both px -> 8 consecutive floats
        if (NT == 1 && narrow) {
            // raw accumulators -> tr[wave][co][2 * cell + px] (the wave's own slab: LDS accesses of a wave stay in order)
            float *dst = tr + wid * 4 * 2 * CT_TX;
            if (r < s.cout) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int c0 = mt * 16 + 4 * q;
                    *reinterpret_cast<float4 *>(dst + r * 2 * CT_TX + 2 * c0) = float4{acc[0][mt][0][0], acc[1][mt][0][0], acc[0][mt][0][1], acc[1][mt][0][1]};
                    *reinterpret_cast<float4 *>(dst + r * 2 * CT_TX + 2 * c0 + 4) = float4{acc[0][mt][0][2], acc[1][mt][0][2], acc[0][mt][0][3], acc[1][mt][0][3]};
                }
            }
            __syncthreads();
            const int n4 = (min(CT_TX, s.w - x0) * 2) >> 2;   // float4 per channel row of this tile
#pragma unroll
            for (int k = 0; k < 2; ++k) {   // 4 channels x 32 float4: two per lane; a 32-lane half holds one channel
                const int i = lane + 64 * k, co = i >> 5, x4 = i & 31;
                if (co < s.cout && x4 < n4) {
                    float4 v = *reinterpret_cast<const float4 *>(dst + co * 2 * CT_TX + 4 * x4);
                    const float bv = bias ? bias[co] : 0.f;
                    v.x += bv; v.y += bv; v.z += bv; v.w += bv;
                    v = ST::rnd(v);
                    ST::st4(ob + (((int64_t)co * od + oz) * oh + oy) * ow + 2 * x0 + 4 * x4, v);
                    na[k] += (v.x + v.y) + (v.z + v.w);
                    nb[k] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
            }
            __syncthreads();   // the slab is rewritten by the next row
            continue;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 16 + r;
            if (co >= s.cout) continue;
            const float bv = bias ? bias[co] : 0.f;
            TO *orow = ob + (((int64_t)co * od + oz) * oh + oy) * ow;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int c0 = x0 + mt * 16 + 4 * q;
                if (c0 + 3 < s.w && (ow & 3) == 0) {
                    const float4 lo = ST::rnd(float4{acc[0][mt][nt][0] + bv, acc[1][mt][nt][0] + bv, acc[0][mt][nt][1] + bv, acc[1][mt][nt][1] + bv});
                    const float4 hi = ST::rnd(float4{acc[0][mt][nt][2] + bv, acc[1][mt][nt][2] + bv, acc[0][mt][nt][3] + bv, acc[1][mt][nt][3] + bv});
                    ST::st4(orow + 2 * c0, lo);
                    ST::st4(orow + 2 * c0 + 4, hi);
                    st1[nt] += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
                    st2[nt] += ((lo.x * lo.x + lo.y * lo.y) + (lo.z * lo.z + lo.w * lo.w)) + ((hi.x * hi.x + hi.y * hi.y) + (hi.z * hi.z + hi.w * hi.w));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j < s.w) {
                            const float v0 = ST::rnd1(acc[0][mt][nt][j] + bv), v1 = ST::rnd1(acc[1][mt][nt][j] + bv);
                            ST::st1(orow + 2 * (c0 + j), v0);
                            ST::st1(orow + 2 * (c0 + j) + 1, v1);
                            st1[nt] += v0 + v1;
                            st2[nt] += v0 * v0 + v1 * v1;
                        }
                }
            }
        }
    }
    if (stats_partial) {   // block partial [2][cout]: lanes of a channel, then the 4 waves (= the 4 (pz,py) classes), fixed order
        if (NT == 1 && narrow) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float a = na[k], b = nb[k];
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    a += __shfl_xor(a, d, 64);
                    b += __shfl_xor(b, d, 64);
                }
                if ((lane & 31) == 0) {
                    sred[wid][0][(lane + 64 * k) >> 5] = a;
                    sred[wid][1][(lane + 64 * k) >> 5] = b;
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float a = st1[nt], b = st2[nt];
                a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
                b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
                if (q == 0) {
                    sred[wid][0][nt * 16 + r] = a;
                    sred[wid][1][nt * 16 + r] = b;
                }
            }
        }
        __syncthreads();
        if (t < 2 * NT * 16) {
            const int which = t / (NT * 16), co = t % (NT * 16);
            if (co < s.cout)
                stats_partial[((int64_t)tile * 2 + which) * s.cout + co] =
                    (sred[0][which][co] + sred[1][which][co]) + (sred[2][which][co] + sred[3][which][co]);
        }
    }
}

// z-sliding form of ct_fwd_mfma_kernel for 16 input channels (r06): the same staging code, MFMA order and epilogue, but a block owns
// (n, YR rows, x tile) for ALL z and keeps a ring of three staged planes - 1.5 staged rows per produced row instead of 4.5; the wave's
// weight fragments (its (pz, py) class never changes) stay in registers; the narrow epilogue's LDS transpose is wave-local (no workgroup
// barriers).  16 -> 3 @ 4 x 10 x 376 x 376, fp32 output: 575 -> 457 us (z ring 526, + weights in registers 470, + wave-local epilogue 457),
// bit-identical.  Ablation of the 466 us build (template switches, removed again): without the output stores 384, without the MFMAs 413,
// without the global loads 393, without the A-fragment LDS reads 326, without all four 178 - no single stream dominates; the 32
// ds_read_b128 per wave and row (4 x redundant: both px classes and both x taps re-read the same cells) are the largest single term.
template <int CIN, int NT, int YR, typename TO = float, typename TXI = float>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void ct_fwd_zslide_kernel(const TXI *__restrict__ x, const __bf16 *__restrict__ wp, const float *__restrict__ bias,
                                                          CtDims s, int xtiles, TO *__restrict__ out,
                                                          float *__restrict__ stats_partial, const float *__restrict__ in_norm = nullptr) {
    // in_norm (r04, scale[CIN] | shift[CIN] or null): the input is the RAW tensor in front of a BatchNorm3d + ReLU, applied here while
    // staging - relu(fma(x, scale[ci], shift[ci])), the expression of s2d_bncm_apply_f32 - so the normalised tensor is never written
    // or read (cells outside the tensor stay zero: the padding is of the normalised tensor)
    constexpr int GROUPS = CIN / 8;          // 16-byte pieces per staged cell
    constexpr int XS = CT_TX + 2;            // staged columns: x0-1 .. x0+64
    constexpr int KSTEPS = CIN == 32 ? 8 : 4;
    // a block produces YR consecutive input rows y of ONE x tile for EVERY z, walking z with a ring of three staged planes of YR + 2 rows:
    // each plane is staged once per block instead of three times (by the blocks of z - 1, z, z + 1)
    constexpr int RY = YR + 2, ROWS = RY;            // rows staged per z step
    __shared__ __attribute__((aligned(16))) __bf16 xs[3 * RY * XS * CIN];   // [slot * RY + yi][xx][ci], slot of plane z = (z + 3) % 3
    auto xa = [&](int cell, int g) -> int { return (cell * GROUPS + g) * 8; };
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    // Tile order: (n, group of YR rows, x tile); xcd_tile() gives an XCD a contiguous range of tiles, so the y-halo rows two neighbouring row
    // groups share are fetched by one XCD's L2
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int hgroups = (s.h + YR - 1) / YR;
    if (tile >= s.n * hgroups * xtiles) return;
    // the batch norm's scale | shift in LDS: read from global memory inside the staging (8 channel pairs per halo cell, 16 per body thread, every z
    // step) each pair of dword loads sat between two x loads with an s_waitcnt vmcnt(0) - the halo threads' 8 x loads of a z step went out one
    // memory latency after the other, in front of the step's barrier
    __shared__ __attribute__((aligned(16))) float nrm_s[2 * CIN];
    if (in_norm) {   // block-uniform
        if (t < 2 * CIN) nrm_s[t] = in_norm[t];
        __syncthreads();
    }
    const int n = tile / (hgroups * xtiles);
    const int rem0 = tile - n * (hgroups * xtiles);
    const int hy0 = (rem0 / xtiles) * YR, xt = rem0 % xtiles;
    const int x0 = xt * CT_TX;
    const int64_t cells = (int64_t)s.d * s.h * s.w;
    const TXI *xb = x + (int64_t)n * CIN * cells;   // (TXI = __bf16, r06: the bf16-stored z - a 16-byte load carries 8 cells of a plane, w % 8 == 0)
    constexpr bool X16 = sizeof(TXI) == 2;
    constexpr int CPP = X16 ? 8 : 4;             // cells per 16-byte piece of a plane
    constexpr int PPR = CT_TX / CPP;             // pieces per 64-cell row and plane
    // stage: the 64-cell body of a row as 16-byte loads (piece = (row9, group, 4 cells): 8 channel planes x float4 -> four 16-byte LDS
    // stores), the two halo columns (and everything when w % 4 != 0) with one dword per channel.  Strided dword gathers cost ~4x the
    // issue slots of 16-byte loads (DESIGN rule 7); this staging is what bounds the kernel.
    const bool vec = (s.w & (CPP - 1)) == 0;
    auto stage_plane = [&](int zp, int slot) {   // plane zp (zeros outside the tensor) -> rows [slot * RY, slot * RY + RY) of the ring
    const int rbase = slot * RY;
    if (vec) {
        // every thread issues the loads of its halo piece (threads < 18 * GROUPS) and of its first body piece before converting any
        constexpr int HALO = ROWS * GROUPS * 2, BODY = ROWS * GROUPS * PPR;
        static_assert(HALO <= 256, "one halo piece per thread");
        float hv[8];
        const bool has_halo = t < HALO;
        int h_slot = 0;
        if (has_halo) {
            const int c = t & 1, g = (t >> 1) % GROUPS, r9 = (t >> 1) / GROUPS;
            const int xx = c * (XS - 1);
            const int z = zp, y = hy0 + r9 - 1, xp = x0 - 1 + xx;
            h_slot = xa((rbase + r9) * XS + xx, g);
            const bool ok = (unsigned)z < (unsigned)s.d && (unsigned)y < (unsigned)s.h && (unsigned)xp < (unsigned)s.w;
            const TXI *src = ok ? xb + ((int64_t)z * s.h + y) * s.w + xp + (int64_t)(g * 8) * cells : xb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float val = (float)src[ok ? (int64_t)e * cells : 0];
                if (in_norm) val = fmaxf(fmaf(val, nrm_s[g * 8 + e], nrm_s[CIN + g * 8 + e]), 0.f);
                hv[e] = ok ? val : 0.f;
            }
        }
        // (p >> 4) advances by 16 per iteration, a multiple of GROUPS: a thread's channel group - and its eight scale / shift pairs - is fixed
        static_assert((256 / PPR) % GROUPS == 0, "a thread keeps its channel group");
        float nsc[8], nsh[8];
        if (in_norm) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                nsc[e] = nrm_s[((t / PPR) % GROUPS) * 8 + e];
                nsh[e] = nrm_s[CIN + ((t / PPR) % GROUPS) * 8 + e];
            }
        }
        if constexpr (X16) {
            // bf16 planes: a 16-byte load is 8 cells of one plane; the input is always the raw tensor in front of the batch norm here (in_norm != null:
            // the launcher's contract).  Per value: one shift / and to widen, one scalar v_fma_f32 (not v_pk_fma_f32: ct_fma_scalar), half a
            // v_cvt_pk_bf16_f32 - packing the channel PAIR (2k, 2k + 1) of a cell, which is the [cell][ci] order of the LDS image: no transposition -
            // and half a v_pk_max_i16 against zero: the ReLU on the two rounded bf16 values (sign bit set -> 0; rounding keeps the sign, so
            // relu(round(v)) == round(relu(v)) bit for bit).  The first version did widen / fma / max / narrow per scalar with a run-time in_norm test
            // per element: 0.47 -> 0.53 ms (the staging is what bounds this kernel).
            // Two threads per piece (planes 0-3 / 4-7 of the group, an 8-byte half of each cell's LDS slot): 2 BODY = 192 of the 256 threads stage, as on
            // the fp32 path - with one thread per 8-cell piece only 96 were busy with twice the work each (0.47 -> 0.51 ms).
            typedef float f32x2x __attribute__((ext_vector_type(2)));
            typedef short s16x2x __attribute__((ext_vector_type(2)));
            static_assert((128 / PPR) % GROUPS == 0, "a thread keeps its channel group");
            const int half = t & 1;
            f32x2x sc2[2], sh2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c0 = ((t >> 1) / PPR % GROUPS) * 8 + 4 * half + 2 * k;
                sc2[k] = f32x2x{nrm_s[c0], nrm_s[c0 + 1]};
                sh2[k] = f32x2x{nrm_s[CIN + c0], nrm_s[CIN + c0 + 1]};
            }
            for (int p2 = t; p2 < 2 * BODY; p2 += 256) {
                const int p = p2 >> 1;
                const int xq = p % PPR, g = (p / PPR) % GROUPS, r9 = (p / PPR) / GROUPS;
                const int z = zp, y = hy0 + r9 - 1, xp = x0 + 8 * xq;
                const bool ok = (unsigned)z < (unsigned)s.d && (unsigned)y < (unsigned)s.h && xp < s.w;
                const TXI *src = ok ? xb + ((int64_t)z * s.h + y) * s.w + xp + (int64_t)(g * 8 + 4 * half) * cells : xb;
                u32x4m f[4];   // plane 4 half + e: 8 cells, two per dword (low half = the even cell)
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = *reinterpret_cast<const u32x4m *>(src + (ok ? (int64_t)e * cells : 0));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    u32x2m o;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const uint32_t wa = f[2 * k][j >> 1], wb = f[2 * k + 1][j >> 1];
                        f32x2x val = {__uint_as_float((j & 1) ? (wa & 0xffff0000u) : (wa << 16)),
                                      __uint_as_float((j & 1) ? (wb & 0xffff0000u) : (wb << 16))};
                        val = f32x2x{ct_fma_scalar(val[0], sc2[k][0], sh2[k][0]), ct_fma_scalar(val[1], sc2[k][1], sh2[k][1])};
                        const s16x2x r2 = __builtin_elementwise_max(__builtin_bit_cast(s16x2x, ct_cvt_pk_bf16(val[0], val[1])), s16x2x{0, 0});
                        o[k] = ok ? __builtin_bit_cast(uint32_t, r2) : 0u;
                    }
                    *reinterpret_cast<u32x2m *>(&xs[xa((rbase + r9) * XS + 1 + 8 * xq + j, g) + 4 * half]) = o;
                }
            }
        } else
        for (int p = t; p < BODY; p += 256) {
            const int xq = p & 15, g = (p >> 4) % GROUPS, r9 = (p >> 4) / GROUPS;
            const int z = zp, y = hy0 + r9 - 1, xp = x0 + 4 * xq;
            const bool ok = (unsigned)z < (unsigned)s.d && (unsigned)y < (unsigned)s.h && xp < s.w;
            const TXI *src = ok ? xb + ((int64_t)z * s.h + y) * s.w + xp + (int64_t)(g * 8) * cells : xb;
            float4 f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = *reinterpret_cast<const float4 *>(src + (ok ? (int64_t)e * cells : 0));
            if (in_norm) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sc = nsc[e], sh = nsh[e];
                    f[e] = float4{fmaxf(fmaf(f[e].x, sc, sh), 0.f), fmaxf(fmaf(f[e].y, sc, sh), 0.f), fmaxf(fmaf(f[e].z, sc, sh), 0.f),
                                  fmaxf(fmaf(f[e].w, sc, sh), 0.f)};
                }
            }
            bf16x8m v[4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[0][e] = (__bf16)(ok ? f[e].x : 0.f); v[1][e] = (__bf16)(ok ? f[e].y : 0.f);
                v[2][e] = (__bf16)(ok ? f[e].z : 0.f); v[3][e] = (__bf16)(ok ? f[e].w : 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x8m *>(&xs[xa((rbase + r9) * XS + 1 + 4 * xq + j, g)]) = v[j];
        }
        if (has_halo) {
            bf16x8m v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)hv[e];
            *reinterpret_cast<bf16x8m *>(&xs[h_slot]) = v;
        }
    } else {
        for (int p = t; p < ROWS * GROUPS * XS; p += 256) {
            const int xx = p % XS, g = (p / XS) % GROUPS, r9 = p / (XS * GROUPS);
            const int z = zp, y = hy0 + r9 - 1, xp = x0 - 1 + xx;
            bf16x8m v;
            if ((unsigned)z < (unsigned)s.d && (unsigned)y < (unsigned)s.h && (unsigned)xp < (unsigned)s.w) {
                const TXI *src = xb + ((int64_t)z * s.h + y) * s.w + xp + (int64_t)(g * 8) * cells;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float val = (float)src[(int64_t)e * cells];
                    if (in_norm) val = fmaxf(fmaf(val, nrm_s[g * 8 + e], nrm_s[CIN + g * 8 + e]), 0.f);
                    v[e] = (__bf16)val;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            }
            *reinterpret_cast<bf16x8m *>(&xs[xa((rbase + r9) * XS + xx, g)]) = v;
        }
    }
    };
    const int pz = wid >> 1, py = wid & 1;
    const int r = lane & 15, q = lane >> 4;
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    TO *ob = out + (int64_t)n * s.cout * od * oh * ow;
    using ST = CtStore<TO>;
    float st1[NT], st2[NT];   // per-channel (sum, sum of squares) of this lane's outputs: the batch norm that follows skips its statistics pass
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        st1[nt] = 0.f;
        st2[nt] = 0.f;
    }
    // <= 4 output channels (the last up-sampler, 16 -> 3): only 4 of 16 column lanes hold outputs, so storing from the accumulator
    // layout issues 32 store instructions per wave with 12 live lanes each.  The wave transposes its row through LDS instead and
    // stores whole 512-byte channel rows with every lane (bias and statistics on the float4s).
    __shared__ __attribute__((aligned(16))) float tr[NT == 1 ? 4 * 4 * 2 * CT_TX : 4];   // [wave][co < 4][2 * CT_TX]
    const bool narrow = NT == 1 && s.cout <= 4 && (ow & 3) == 0;
    __shared__ float sred[4][2][NT * 16];
    float na[2] = {0.f, 0.f}, nb[2] = {0.f, 0.f};   // narrow path: this lane's sums for channel (lane + 64 k) >> 5
    bf16x8m bw[2][NT == 1 ? KSTEPS : 1];   // NT == 1: the 2 x KSTEPS weight fragments of this wave's (pz, py) classes, loaded once
    if constexpr (NT == 1) {
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
                bw[px][ks] = *reinterpret_cast<const bf16x8m *>(wp + (((int64_t)(pz * 4 + py * 2 + px) * KSTEPS + ks) * 64 + lane) * 8);
    }
    stage_plane(-1, 2);
    stage_plane(0, 0);
    for (int hz = 0; hz < s.d; ++hz) {
    stage_plane(hz + 1, (hz + 1) % 3);   // overwrites plane hz - 2: every wave left its reads behind at the barrier that closed step hz - 1
    __syncthreads();
    const int oz = 2 * hz + pz;
    const int zslot[3] = {(hz + 2) % 3, hz % 3, (hz + 1) % 3};   // ring slots of the planes hz - 1, hz, hz + 1
    for (int yy = 0; yy < YR; ++yy) {
        const int hy = hy0 + yy;
        if (hy >= s.h) break;   // block-uniform
        const int oy = 2 * hy + py;
        f32x4m acc[2][4][NT];
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[px][mt][nt] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int cls = pz * 4 + py * 2 + px;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                bf16x8m b[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (NT == 1) b[nt] = bw[px][ks];   // the wave's class never changes: its weight fragments stay in registers
                    else b[nt] = *reinterpret_cast<const bf16x8m *>(wp + ((((int64_t)cls * KSTEPS + ks) * NT + nt) * 64 + lane) * 8);
                }
                const int tap = CIN == 32 ? ks : 2 * ks + (q >> 1);
                const int g = CIN == 32 ? q : (q & 1);
                const int ta = tap >> 2, tb = (tap >> 1) & 1, tc = tap & 1;
                const int r9 = zslot[ct_d(pz, ta) + 1] * RY + (ct_d(py, tb) + 1 + yy);
                const int dx = ct_d(px, tc);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int xx = mt * 16 + r + 1 + dx;
                    const bf16x8m a = *reinterpret_cast<const bf16x8m *>(&xs[xa(r9 * XS + xx, g)]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[px][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[nt], acc[px][mt][nt], 0, 0, 0);
                }
            }
        }
        // epilogue: lane holds column co = nt*16 + r and the 4 cells 4q..4q+3 of every m-tile, both px -> 8 consecutive floats
        if (NT == 1 && narrow) {
            // raw accumulators -> tr[wave][co][2 * cell + px] (the wave's own slab: LDS accesses of a wave stay in order)
            float *dst = tr + wid * 4 * 2 * CT_TX;
            if (r < s.cout) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int c0 = mt * 16 + 4 * q;
                    *reinterpret_cast<float4 *>(dst + r * 2 * CT_TX + 2 * c0) = float4{acc[0][mt][0][0], acc[1][mt][0][0], acc[0][mt][0][1], acc[1][mt][0][1]};
                    *reinterpret_cast<float4 *>(dst + r * 2 * CT_TX + 2 * c0 + 4) = float4{acc[0][mt][0][2], acc[1][mt][0][2], acc[0][mt][0][3], acc[1][mt][0][3]};
                }
            }
            // (the slab is this wave's own and a wave's LDS accesses execute in order: no workgroup barrier - r06; the per-z kernel above
            // still has two per row)
            __builtin_amdgcn_wave_barrier();
            const int n4 = (min(CT_TX, s.w - x0) * 2) >> 2;   // float4 per channel row of this tile
#pragma unroll
            for (int k = 0; k < 2; ++k) {   // 4 channels x 32 float4: two per lane; a 32-lane half holds one channel
                const int i = lane + 64 * k, co = i >> 5, x4 = i & 31;
                if (co < s.cout && x4 < n4) {
                    float4 v = *reinterpret_cast<const float4 *>(dst + co * 2 * CT_TX + 4 * x4);
                    const float bv = bias ? bias[co] : 0.f;
                    v.x += bv; v.y += bv; v.z += bv; v.w += bv;
                    v = ST::rnd(v);
                    ST::st4(ob + (((int64_t)co * od + oz) * oh + oy) * ow + 2 * x0 + 4 * x4, v);
                    na[k] += (v.x + v.y) + (v.z + v.w);
                    nb[k] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
            }
            __builtin_amdgcn_wave_barrier();   // the slab is rewritten by the next row (same wave, in order)
            continue;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 16 + r;
            if (co >= s.cout) continue;
            const float bv = bias ? bias[co] : 0.f;
            TO *orow = ob + (((int64_t)co * od + oz) * oh + oy) * ow;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int c0 = x0 + mt * 16 + 4 * q;
                if (c0 + 3 < s.w && (ow & 3) == 0) {
                    const float4 lo = ST::rnd(float4{acc[0][mt][nt][0] + bv, acc[1][mt][nt][0] + bv, acc[0][mt][nt][1] + bv, acc[1][mt][nt][1] + bv});
                    const float4 hi = ST::rnd(float4{acc[0][mt][nt][2] + bv, acc[1][mt][nt][2] + bv, acc[0][mt][nt][3] + bv, acc[1][mt][nt][3] + bv});
                    ST::st4(orow + 2 * c0, lo);
                    ST::st4(orow + 2 * c0 + 4, hi);
                    st1[nt] += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
                    st2[nt] += ((lo.x * lo.x + lo.y * lo.y) + (lo.z * lo.z + lo.w * lo.w)) + ((hi.x * hi.x + hi.y * hi.y) + (hi.z * hi.z + hi.w * hi.w));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j < s.w) {
                            const float v0 = ST::rnd1(acc[0][mt][nt][j] + bv), v1 = ST::rnd1(acc[1][mt][nt][j] + bv);
                            ST::st1(orow + 2 * (c0 + j), v0);
                            ST::st1(orow + 2 * (c0 + j) + 1, v1);
                            st1[nt] += v0 + v1;
                            st2[nt] += v0 * v0 + v1 * v1;
                        }
                }
            }
        }
    }
    __syncthreads();   // step hz done: the next step's staging may overwrite the plane behind it
    }   // hz
    if (stats_partial) {   // block partial [2][cout]: lanes of a channel, then the 4 waves (= the 4 (pz,py) classes), fixed order
        if (NT == 1 && narrow) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float a = na[k], b = nb[k];
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    a += __shfl_xor(a, d, 64);
                    b += __shfl_xor(b, d, 64);
                }
                if ((lane & 31) == 0) {
                    sred[wid][0][(lane + 64 * k) >> 5] = a;
                    sred[wid][1][(lane + 64 * k) >> 5] = b;
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float a = st1[nt], b = st2[nt];
                a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
                b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
                if (q == 0) {
                    sred[wid][0][nt * 16 + r] = a;
                    sred[wid][1][nt * 16 + r] = b;
                }
            }
        }
        __syncthreads();
        if (t < 2 * NT * 16) {
            const int which = t / (NT * 16), co = t % (NT * 16);
            if (co < s.cout)
                stats_partial[((int64_t)tile * 2 + which) * s.cout + co] =
                    (sred[0][which][co] + sred[1][which][co]) + (sred[2][which][co] + sred[3][which][co]);
        }
    }
}

// ---- data gradient -------------------------------------------------------------------------------
template <int COP, int NT, int G>
__global__ __launch_bounds__(256) void ct_dgrad_mfma_kernel(const float *__restrict__ dout, const __bf16 *__restrict__ wp, CtDims s, int xtiles,
                                                            float *__restrict__ din) {
    constexpr int GROUPS = COP / 8;
    constexpr int KSTEPS = COP == 32 ? 4 : 1;
    __shared__ __attribute__((aligned(16))) __bf16 ts[G * 4 * CT_TX * COP];   // [g][kx][c][co]
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int xt = blockIdx.x % xtiles;
    const int64_t row = blockIdx.x / xtiles;
    const int hy = (int)(row % s.h), hz = (int)((row / s.h) % s.d), n = (int)(row / ((int64_t)s.h * s.d));
    const int x0 = xt * CT_TX;
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    const int64_t oplane = (int64_t)od * oh * ow;
    const float *db = dout + (int64_t)n * s.cout * oplane;
    const int r = lane & 15, q = lane >> 4;
    f32x4m acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4m{0.f, 0.f, 0.f, 0.f};
    for (int round = 0; round < 16 / G; ++round) {
        if (round) __syncthreads();
        // staging: a wave = the 64 cells of the tile for one (output row gi, channel group g).  Each lane loads the aligned
        // pair dout[co][2c], dout[co][2c+1] of 8 planes (coalesced float2 rows) and takes positions 2c-1 / 2c+2 from its
        // neighbour lanes: the four kx samples of cell c are exactly dout[2c-1 .. 2c+2].
        for (int p = t; p < G * GROUPS * CT_TX; p += 256) {
            const int c = p % CT_TX, g = (p / CT_TX) % GROUPS, gi = p / (CT_TX * GROUPS);
            const int kzky = round * G + gi;
            const int z = 2 * hz - 1 + (kzky >> 2), y = 2 * hy - 1 + (kzky & 3);
            const bool rowok = (unsigned)z < (unsigned)od && (unsigned)y < (unsigned)oh;   // wave-uniform
            const int xe = 2 * (x0 + c);
            const bool cellok = rowok && x0 + c < s.w;
            const float *src = db + ((int64_t)(rowok ? z : 0) * oh + (rowok ? y : 0)) * ow + (int64_t)(g * 8) * oplane;
            bf16x8m v0, v1, v2, v3;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool chok = cellok && g * 8 + e < s.cout;
                float2 m = float2{0.f, 0.f};
                if (chok) m = *reinterpret_cast<const float2 *>(src + (int64_t)e * oplane + xe);
                float lo = __shfl_up(m.y, 1, 64), hi = __shfl_down(m.x, 1, 64);
                if (c == 0) lo = (chok && xe > 0) ? src[(int64_t)e * oplane + xe - 1] : 0.f;
                if (c == CT_TX - 1) hi = (chok && xe + 2 < ow) ? src[(int64_t)e * oplane + xe + 2] : 0.f;
                if (!chok) { lo = 0.f; hi = 0.f; }
                v0[e] = (__bf16)lo; v1[e] = (__bf16)m.x; v2[e] = (__bf16)m.y; v3[e] = (__bf16)hi;
            }
            __bf16 *dst = &ts[(((gi * 4 + 0) * CT_TX + c) * GROUPS + g) * 8];
            *reinterpret_cast<bf16x8m *>(dst) = v0;
            *reinterpret_cast<bf16x8m *>(dst + 1 * CT_TX * GROUPS * 8) = v1;
            *reinterpret_cast<bf16x8m *>(dst + 2 * CT_TX * GROUPS * 8) = v2;
            *reinterpret_cast<bf16x8m *>(dst + 3 * CT_TX * GROUPS * 8) = v3;
        }
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int kzky = round * G + gi;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int kx = COP == 32 ? ks : q;
                const int g = COP == 32 ? q : 0;
                const bf16x8m a = *reinterpret_cast<const bf16x8m *>(&ts[(((gi * 4 + kx) * CT_TX + wid * 16 + r) * GROUPS + g) * 8]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8m b = *reinterpret_cast<const bf16x8m *>(wp + ((((int64_t)kzky * KSTEPS + ks) * NT + nt) * 64 + lane) * 8);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[nt], 0, 0, 0);
                }
            }
        }
    }
    const int64_t cells = (int64_t)s.d * s.h * s.w;
    float *ib = din + (int64_t)n * s.cin * cells + ((int64_t)hz * s.h + hy) * s.w;
    const int c0 = x0 + wid * 16 + 4 * q;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ci = nt * 16 + r;
        if (ci >= s.cin) continue;
        float *dst = ib + (int64_t)ci * cells + c0;
        if (c0 + 3 < s.w && (s.w & 3) == 0) {
            *reinterpret_cast<float4 *>(dst) = float4{acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < s.w) dst[j] = acc[nt][j];
        }
    }
}

// ---- data gradient without LDS staging -------------------------------------------------------------
// M = 16 input cells per MFMA row tile, N = ci, K = (co, kx) per (kz,ky) output row: the four kx samples of cell c are the four
// consecutive floats dout[co][z][y][2c-1 .. 2c+2], ONE 16-byte load.  Lane (r = cell, q) holds K elements e = (co 2q + (e >> 2),
// kx e & 3) of k-step ks (channels 8ks .. 8ks+7): two 16-byte buffer loads per fragment, straight from the planar tensor (lanes
// r walk the row in 8-byte steps: one instruction covers ~140 contiguous bytes per plane); rows / columns / channels outside
// the tensor get an out-of-range per-lane offset.  A wave owns MT row tiles of one input row, so every weight fragment it reads
// (L1) feeds MT MFMAs.  The staged kernel above holds 32-64 KB of LDS per block and waits for memory once per staging round
// (1.9 ms for the 16 -> 3 layer at [4,16,10,376,376] against a 0.2 ms stream); a first direct version with dword loads (one
// per channel and kx) ran at ~70 clocks per load instruction: 1.9 / 1.5 ms.
template <int COP, int NT, int MT, bool N4 = false, typename TD = float, typename TI = float>
__global__ __launch_bounds__(256) void ct_dgrad_direct_kernel(const TD *__restrict__ dout, const __bf16 *__restrict__ wp, CtDims s, int tiles_per_row,
                                                              CtTileMap map, TI *__restrict__ din) {
    constexpr bool D16 = sizeof(TD) == 2;   // bf16 dout: the window [2c-1, 2c+2] is elements 1..4 of the three dwords from element 2c-2
    constexpr unsigned ES = sizeof(TD);
    constexpr unsigned IS = sizeof(TI);     // r06: din stored in bf16 (w % 4 == 0): a lane's four cells are one 8-byte store
    constexpr int KSTEPS = COP == 32 ? 4 : 1;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    const int64_t cells = (int64_t)s.d * s.h * s.w, oplane = (int64_t)od * oh * ow;
    const unsigned oplane_b = (unsigned)(oplane * ES), dbytes = (unsigned)(s.cout * oplane * ES), ibytes = (unsigned)(s.cin * cells * IS);
    // Item order (XCD-aware, see ct_fwd_mfma_kernel): an input row reads 4 x 4 (z, y) rows of dout and shares half of them with each
    // neighbour, so every dout row has four readers.  With items dealt to workgroups in plain order those sat on different XCDs and
    // HBM delivered every row up to 4x (PMC: 2.9 GB fetched for a 724 MB gradient - the kernel ran AT the HBM roof).  Here XCD x owns
    // the contiguous eighth [x, x+1) * items / 8 of the order (n, chunk of yc rows, z, y in chunk, x tile) and its workgroups sweep
    // that range together: the readers of a row now share an L2.
    const int items = s.n * s.d * s.h * tiles_per_row;
    const int xcd = blockIdx.x % S2D_XCDS, bx = blockIdx.x / S2D_XCDS, bpx = gridDim.x / S2D_XCDS;
    const int per_xcd = (items + S2D_XCDS - 1) / S2D_XCDS;
    const int it_hi = min(items, (xcd + 1) * per_xcd);
    // r06: the narrow layer's 8 weight fragments (one per (kz, ky pair)) are the same for every item of the wave's sweep: loaded once,
    // kept in registers (82 -> ~115 VGPRs, still four waves per SIMD), instead of 8 L1 reads in front of every item's MFMAs
    constexpr bool WREG = N4 && NT == 1 && KSTEPS == 1;
    bf16x8m wreg[WREG ? 8 : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) wreg[k8] = *reinterpret_cast<const bf16x8m *>(wp + ((int64_t)k8 * 64 + lane) * 8);
    }
    for (int item = xcd * per_xcd + bx * 4 + wid; item < it_hi; item += bpx * 4) {
        const int n = (int)fastdiv((uint32_t)item, map.per_n);
        int rem = item - n * (int)map.per_n.d;
        const int chunk = (int)fastdiv((uint32_t)rem, map.per_chunk);
        rem -= chunk * (int)map.per_chunk.d;
        const int yc = min(map.yc, s.h - chunk * map.yc);   // rows of this chunk
        const int hz = yc == map.yc ? (int)fastdiv((uint32_t)rem, map.per_plane) : rem / (yc * tiles_per_row);
        rem -= hz * (yc * tiles_per_row);
        const int yr = (int)fastdiv((uint32_t)rem, map.per_row);
        const int hy = chunk * map.yc + yr, xt = rem - yr * tiles_per_row;
        const __amdgpu_buffer_rsrc_t dr = ct_rsrc(dout + (int64_t)n * s.cout * oplane, dbytes);
        const int x0 = xt * 16 * MT;
        f32x4m acc[MT][NT];
        unsigned pos[MT];     // byte offset of the lane's window in a row (the first cell's starts at 0 instead of -4: `shifted`)
        bool shifted[MT], last[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4m{0.f, 0.f, 0.f, 0.f};
            const int c = x0 + mt * 16 + r;
            shifted[mt] = c == 0;
            last[mt] = c == s.w - 1;
            pos[mt] = c < s.w ? (D16 ? (unsigned)((c == 0 ? 0 : 2 * c - 2) * 2) : (unsigned)((2 * c - (c == 0 ? 0 : 1)) * 4)) : CT_OOB;
        }
#pragma unroll (WREG ? 8 : 2)
        for (int kzky = 0; kzky < (N4 ? 8 : 16); ++kzky) {
            // N4 (cout <= 4): the lane groups q = (ky of a pair, channel pair) - the row is per lane and goes into the lane offset
            const int z = 2 * hz - 1 + (N4 ? kzky >> 1 : kzky >> 2), y = 2 * hy - 1 + (N4 ? 2 * (kzky & 1) + (q >> 1) : kzky & 3);
            const bool rok = (unsigned)z < (unsigned)od && (unsigned)y < (unsigned)oh;   // wave-uniform unless N4
            const unsigned roff = rok ? (unsigned)(((int64_t)z * oh + y) * ow * ES) : 0u;
            const unsigned soff = N4 ? 0u : roff;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                bf16x8m bfr[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (WREG) bfr[nt] = wreg[kzky];   // (the loop is fully unrolled then: a register, not an indexed array)
                    else bfr[nt] = *reinterpret_cast<const bf16x8m *>(wp + ((((int64_t)kzky * KSTEPS + ks) * NT + nt) * 64 + lane) * 8);
                }
                const int co0 = N4 ? 2 * (q & 1) : 8 * ks + 2 * q;
                const unsigned pl0 = (rok && co0 < s.cout) ? (unsigned)co0 * oplane_b + (N4 ? roff : 0u) : CT_OOB;
                const unsigned pl1 = (rok && co0 + 1 < s.cout) ? (unsigned)(co0 + 1) * oplane_b + (N4 ? roff : 0u) : CT_OOB;
                if constexpr (D16) {
                    u32x3m u0[MT], u1[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        u0[mt] = ct_load3u(dr, (pos[mt] | pl0) >= CT_OOB ? CT_OOB : pos[mt] + pl0, soff);
                        u1[mt] = ct_load3u(dr, (pos[mt] | pl1) >= CT_OOB ? CT_OOB : pos[mt] + pl1, soff);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        // elements 1..4 of the six loaded ones; shifted lanes (cell 0) loaded from element 0: sample kx = element kx - 1, kx = 0 zero
                        const bool sh = shifted[mt];
                        const uint32_t a0 = sh ? 0u : u0[mt].x, a1 = sh ? u0[mt].x : u0[mt].y, a2 = sh ? u0[mt].y : u0[mt].z;
                        const uint32_t b0 = sh ? 0u : u1[mt].x, b1 = sh ? u1[mt].x : u1[mt].y, b2 = sh ? u1[mt].y : u1[mt].z;
                        const uint32_t lm = last[mt] ? 0x0000FFFFu : 0xFFFFFFFFu;   // position 2w of the last cell belongs to the next row
                        const u32x4m pk = {__builtin_amdgcn_alignbit(a1, a0, 16), __builtin_amdgcn_alignbit(a2, a1, 16) & lm,
                                           __builtin_amdgcn_alignbit(b1, b0, 16), __builtin_amdgcn_alignbit(b2, b1, 16) & lm};
                        const bf16x8m a = __builtin_bit_cast(bf16x8m, pk);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[nt], acc[mt][nt], 0, 0, 0);
                    }
                } else {
                f32x4m v0[MT], v1[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    // an out-of-range marker in either addend keeps the sum >= 2^31 > num_records (both are < 2^31 otherwise)
                    v0[mt] = ct_load4(dr, (pos[mt] | pl0) >= CT_OOB ? CT_OOB : pos[mt] + pl0, soff);
                    v1[mt] = ct_load4(dr, (pos[mt] | pl1) >= CT_OOB ? CT_OOB : pos[mt] + pl1, soff);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    bf16x8m a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // shifted lanes loaded positions 0..3: sample kx = element kx - 1 (kx = 0 is position -1: zero)
                        float s0 = shifted[mt] ? (e ? v0[mt][e - (e ? 1 : 0)] : 0.f) : v0[mt][e];
                        float s1 = shifted[mt] ? (e ? v1[mt][e - (e ? 1 : 0)] : 0.f) : v1[mt][e];
                        if (e == 3) {   // position 2w of the last cell belongs to the next row
                            s0 = last[mt] ? 0.f : s0;
                            s1 = last[mt] ? 0.f : s1;
                        }
                        a[e] = (__bf16)s0;
                        a[4 + e] = (__bf16)s1;
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[nt], acc[mt][nt], 0, 0, 0);
                }
                }
            }
        }
        // C/D layout: row (cell) = 4q + reg, column (ci) = r: a lane stores 4 consecutive cells of its plane
        const __amdgpu_buffer_rsrc_t ir = ct_rsrc(din + (int64_t)n * s.cin * cells, ibytes);
        const unsigned rowoff = (unsigned)(((int64_t)hz * s.h + hy) * s.w * IS);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int c0 = x0 + mt * 16 + 4 * q;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ci = nt * 16 + r;
                const unsigned plane = ci < s.cin ? (unsigned)(ci * cells * IS) : CT_OOB;
                if constexpr (IS == 2) {   // (the launcher takes this instantiation for w % 4 == 0 only)
                    typedef __bf16 bf16x4i __attribute__((ext_vector_type(4)));
                    bf16x4i o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (__bf16)acc[mt][nt][j];
                    buf_store2(ir, (c0 < s.w && ci < s.cin) ? plane + (unsigned)c0 * 2u : CT_OOB, rowoff, __builtin_bit_cast(buf_f32x2, o));
                } else if ((s.w & 3) == 0) {
                    buf_store4(ir, (c0 < s.w && ci < s.cin) ? plane + (unsigned)c0 * 4u : CT_OOB, rowoff, acc[mt][nt]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        buf_store1(ir, (c0 + j < s.w && ci < s.cin) ? plane + (unsigned)(c0 + j) * 4u : CT_OOB, rowoff, acc[mt][nt][j]);
                }
            }
        }
    }
}

// ---- weight gradient -----------------------------------------------------------------------------
// grid (row chunks, 16 / KG groups of (kz,ky)); a wave walks the rows of its chunk (wave w takes rows w, w+4, ...), keeps
// acc[KG][4 kx][CIT][COT] in registers, and the block folds its 4 waves through LDS into partial[chunk][ci][co][64].
template <int CIT, int COT, int KG>
__global__ __launch_bounds__(256) void ct_wgrad_mfma_kernel(const float *__restrict__ x, const float *__restrict__ dout, CtDims s, int rows_per_block,
                                                            float *__restrict__ partial) {
    constexpr int FR = KG * 4 * CIT * COT;
    __shared__ float red[FR * 4][64];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    const int64_t cells = (int64_t)s.d * s.h * s.w, oplane = (int64_t)od * oh * ow;
    const int64_t total_rows = (int64_t)s.n * s.d * s.h;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < total_rows ? r0 + rows_per_block : total_rows;
    const int grp = blockIdx.y;
    f32x4m acc[KG][4][CIT][COT];
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int a = 0; a < CIT; ++a)
#pragma unroll
                for (int b = 0; b < COT; ++b) acc[g][k][a][b] = f32x4m{0.f, 0.f, 0.f, 0.f};
    const int steps = (s.w + 31) / 32;
    for (int64_t row = r0 + wid; row < r1; row += 4) {
        const int hy = (int)(row % s.h), hz = (int)((row / s.h) % s.d), n = (int)(row / ((int64_t)s.h * s.d));
        const float *xrow = x + (int64_t)n * s.cin * cells + ((int64_t)hz * s.h + hy) * s.w;
        const float *dbase = dout + (int64_t)n * s.cout * oplane;
        for (int st = 0; st < steps; ++st) {
            const int c0 = st * 32 + 8 * q;   // this lane's 8 cells
            bf16x8m a[CIT];
#pragma unroll
            for (int ai = 0; ai < CIT; ++ai) {
                const int ci = ai * 16 + r;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[ai][e] = (__bf16)0.f;
                if (ci < s.cin) {
                    const float *src = xrow + (int64_t)ci * cells + c0;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (c0 + e < s.w) a[ai][e] = (__bf16)src[e];
                }
            }
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const int kzky = grp * KG + g;
                const int z = 2 * hz - 1 + (kzky >> 2), y = 2 * hy - 1 + (kzky & 3);
                if ((unsigned)z >= (unsigned)od || (unsigned)y >= (unsigned)oh) continue;   // wave-uniform
#pragma unroll
                for (int bi = 0; bi < COT; ++bi) {
                    const int co = bi * 16 + r;
                    float win[20];   // win[j] = dout[co][z][y][2*c0 - 2 + j], j = 0..19 (aligned float2 loads); tap kx of cell e = win[1 + kx + 2e]
#pragma unroll
                    for (int j = 0; j < 20; ++j) win[j] = 0.f;
                    if (co < s.cout) {
                        const float *src = dbase + (int64_t)co * oplane + ((int64_t)z * oh + y) * ow + 2 * c0 - 2;
#pragma unroll
                        for (int j = 0; j < 10; ++j) {
                            const int pos = 2 * c0 - 2 + 2 * j;
                            if (pos >= 0 && pos + 1 < ow) {
                                const float2 m = *reinterpret_cast<const float2 *>(src + 2 * j);
                                win[2 * j] = m.x; win[2 * j + 1] = m.y;
                            }
                        }
                    }
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        bf16x8m b;
#pragma unroll
                        for (int e = 0; e < 8; ++e) b[e] = (__bf16)((c0 + e < s.w) ? win[1 + kx + 2 * e] : 0.f);
#pragma unroll
                        for (int ai = 0; ai < CIT; ++ai)
                            acc[g][kx][ai][bi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ai], b, acc[g][kx][ai][bi], 0, 0, 0);
                    }
                }
            }
        }
    }
    // fold the 4 waves in a fixed order (wave 0 stores, waves 1..3 add in turn) and write this block's slab
    for (int wv = 0; wv < 4; ++wv) {
        if (wid == wv) {
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int a = 0; a < CIT; ++a)
#pragma unroll
                        for (int b = 0; b < COT; ++b)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) {
                                float *slot = &red[(((g * 4 + k) * CIT + a) * COT + b) * 4 + reg][lane];
                                *slot = wv ? *slot + acc[g][k][a][b][reg] : acc[g][k][a][b][reg];
                            }
        }
        __syncthreads();
    }
    float *dst = partial + (int64_t)blockIdx.x * s.cin * s.cout * 64;
    for (int e = t; e < FR * 4 * 64; e += 256) {
        const int ln = e % 64, f4 = e / 64, reg = f4 % 4, f = f4 / 4;
        const int b = f % COT, a = (f / COT) % CIT, k = (f / (COT * CIT)) % 4, g = f / (COT * CIT * 4);
        const float v = red[f4][ln];
        const int ci = a * 16 + 4 * (ln >> 4) + reg, co = b * 16 + (ln & 15);   // C/D layout: row = 4*(lane>>4)+reg, col = lane&15
        const int kzky = grp * KG + g;
        if (ci < s.cin && co < s.cout) dst[(((int64_t)ci * s.cout + co) * 16 + kzky) * 4 + k] = v;
    }
}

// output-row-major variant (w % 4 == 0): a block owns one output parity class (pz,py) and one tile of 16 output channels and walks
// the dout rows z = 2zz+pz, y = 2yy+py of its class.  A dout row of parity (pz,py) meets exactly the 2 x 2 taps (kz,ky) =
// (ct_k(pz,a), ct_k(py,b)) with the input rows (zz + ct_d(pz,a), yy + ct_d(py,b)), so every dout element is read ONCE (the
// input-row-major kernel above reads each 4 times, from 8 blocks with one resident wave per SIMD: 1.9 ms for 32 -> 32 at
// [4,32,5,188,188]); the small x tensor is re-read instead.  Per 32-cell step a wave issues all its loads first (five 16-byte
// pieces of the 20-float dout window = the 4 kx fragments, and the 8 cells of the four input rows), then 16*CIT MFMAs into
// acc[a][b][kx][ci tile].
template <int CIT, typename TD = float>
__global__ __launch_bounds__(256) void ct_wgrad_rows_kernel(const float *__restrict__ x, const TD *__restrict__ dout, CtDims s, int rows_per_block,
                                                            int co_tiles, int groups, int n_chunks, float *__restrict__ partial,
                                                            const float *__restrict__ in_norm = nullptr) {
    constexpr int FR = 16 * CIT;
    constexpr bool D16 = sizeof(TD) == 2;   // bf16 dout: the 20-element window is ten dwords, the fragments are byte permutes of them
    constexpr unsigned ES = sizeof(TD);
    __shared__ float red[FR * 4][64];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int r = lane & 15, q = lane >> 4;
    // 1-D grid, XCD-aware: tile = row chunk * groups + group and consecutive tiles run on one XCD, so the groups of a chunk (which re-read
    // the same input rows) and neighbouring chunks share an L2 instead of fetching their own copies (PMC: 1.9 GB for 0.81 GB of operands).
    // group = (input-channel group, output-channel tile, parity class); a channel group = CIT tiles of 16 input channels
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int chunk = tile / groups, grp = tile - chunk * groups;
    if (chunk >= n_chunks) return;   // grid padding (an empty chunk below n_chunks still writes its zero slab)
    const int cls = grp & 3, bt = (grp >> 2) % co_tiles, ci_base = (grp >> 2) / co_tiles * 16 * CIT;
    const int pz = cls >> 1, py = cls & 1;
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    const int64_t cells = (int64_t)s.d * s.h * s.w, oplane = (int64_t)od * oh * ow;
    const int64_t total_rows = (int64_t)s.n * s.d * s.h;
    const int64_t r0 = (int64_t)chunk * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < total_rows ? r0 + rows_per_block : total_rows;
    f32x4m acc[2][2][4][CIT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < CIT; ++i) acc[a][b][k][i] = f32x4m{0.f, 0.f, 0.f, 0.f};
    const int steps = (s.w + 31) / 32;
    const unsigned xbytes = (unsigned)(s.cin * cells * 4), dbytes = (unsigned)(s.cout * oplane * ES);
    const int co = bt * 16 + r;
    const unsigned dlane = co < s.cout ? (unsigned)(co * oplane * ES) : CT_OOB;
    for (int64_t row = r0 + wid; row < r1; row += 4) {
        const int yy = (int)(row % s.h), zz = (int)((row / s.h) % s.d), n = (int)(row / ((int64_t)s.h * s.d));
        const __amdgpu_buffer_rsrc_t xr = ct_rsrc(x + (int64_t)n * s.cin * cells, xbytes);
        const __amdgpu_buffer_rsrc_t dr = ct_rsrc(dout + (int64_t)n * s.cout * oplane, dbytes);
        const unsigned drow = (unsigned)(((int64_t)(2 * zz + pz) * oh + (2 * yy + py)) * ow * ES);
        unsigned xrow[2][2];
        bool xok[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int hz = zz + ct_d(pz, a), hy = yy + ct_d(py, b);
                xok[a][b] = (unsigned)hz < (unsigned)s.d && (unsigned)hy < (unsigned)s.h;   // wave-uniform
                xrow[a][b] = xok[a][b] ? (unsigned)(((int64_t)hz * s.h + hy) * s.w * 4) : 0u;
            }
        for (int st = 0; st < steps; ++st) {
            const int c0 = st * 32 + 8 * q;
            const bool lo_ok = c0 < s.w, hi_ok = c0 + 4 < s.w;   // w % 4 == 0
            // dout window: floats [2*c0 - 2, 2*c0 + 18) of the row; tap kx of cell c0 + e = element 1 + kx + 2e.  The first cells'
            // window would start at -2 (out of range as a whole for plane 0): those lanes start at 0 and index two elements earlier.
            const bool shifted = c0 == 0;
            const unsigned wl = lo_ok ? dlane + (unsigned)((2 * c0 - (shifted ? 0 : 2)) * ES) : CT_OOB;
            f32x4m wv[D16 ? 1 : 5];
            uint32_t ww[10];
            if constexpr (D16) {
                const u32x4m w0 = ct_load4u(dr, wl, drow), w1 = ct_load4u(dr, wl, drow + 16u);
                const u32x2m w2 = ct_load2u(dr, wl, drow + 32u);
                ww[0] = w0.x; ww[1] = w0.y; ww[2] = w0.z; ww[3] = w0.w; ww[4] = w1.x; ww[5] = w1.y; ww[6] = w1.z; ww[7] = w1.w; ww[8] = w2.x; ww[9] = w2.y;
            } else {
#pragma unroll
                for (int j = 0; j < 5; ++j) wv[j] = ct_load4(dr, wl, drow + 16u * j);
            }
            f32x4m xl[2][2][CIT], xh[2][2][CIT];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < CIT; ++i) {
                        const int ci = ci_base + i * 16 + r;
                        const unsigned base = (xok[a][b] && ci < s.cin) ? (unsigned)(ci * cells * 4) + (unsigned)c0 * 4u : CT_OOB;
                        xl[a][b][i] = ct_load4(xr, lo_ok ? base : CT_OOB, xrow[a][b]);
                        xh[a][b][i] = ct_load4(xr, hi_ok ? base : CT_OOB, xrow[a][b] + 16u);
                    }
            bf16x8m bk[4];
            if constexpr (D16) {
                // window element j sits in word j >> 1 (low half = even j); shifted lanes loaded from element 0: word i = loaded word i - 1.
                // tap kx of cell c0 + e = element 1 + kx + 2e: kx 0 / 2 are the high halves of words e / e + 1, kx 1 / 3 the low halves of
                // words e + 1 / e + 2; a fragment word holds the samples of two consecutive cells
                uint32_t ws[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) ws[i] = shifted ? (i ? ww[i - 1] : 0u) : ww[i];
                const int elast = s.w - 1 - c0;   // the sample of kx = 3 at cell w - 1 is position 2w: it belongs to the next row
                u32x4m f0, f1, f2, f3;
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    f0[pq] = ct_hi_hi(ws[2 * pq], ws[2 * pq + 1]);
                    f1[pq] = ct_lo_lo(ws[2 * pq + 1], ws[2 * pq + 2]);
                    f2[pq] = ct_hi_hi(ws[2 * pq + 1], ws[2 * pq + 2]);
                    const uint32_t m3 = elast == 2 * pq ? 0xFFFF0000u : (elast == 2 * pq + 1 ? 0x0000FFFFu : 0xFFFFFFFFu);
                    f3[pq] = ct_lo_lo(ws[2 * pq + 2], ws[2 * pq + 3]) & m3;
                }
                bk[0] = __builtin_bit_cast(bf16x8m, f0); bk[1] = __builtin_bit_cast(bf16x8m, f1);
                bk[2] = __builtin_bit_cast(bf16x8m, f2); bk[3] = __builtin_bit_cast(bf16x8m, f3);
            } else {
            float win[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                const float plain = wv[j >> 2][j & 3];
                const float early = j >= 2 ? wv[(j - 2) >> 2][(j - 2) & 3] : 0.f;   // shifted lanes: element j of the window = loaded j - 2
                win[j] = shifted ? early : plain;
            }
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = win[1 + kx + 2 * e];
                    if (kx == 3) v = (c0 + e == s.w - 1) ? 0.f : v;   // position 2w belongs to the next row
                    bk[kx][e] = (__bf16)v;
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < CIT; ++i) {
                        bf16x8m av;
                        if (in_norm) {   // x is the raw tensor in front of a BatchNorm3d + ReLU (see ct_fwd_mfma_kernel); loads outside the tensor stay zero
                            const int ci = ci_base + i * 16 + r;
                            const bool cv = xok[a][b] && ci < s.cin;
                            const float sc = cv ? in_norm[ci] : 0.f, sh = cv ? in_norm[s.cin + ci] : 0.f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                xl[a][b][i][e] = (cv && lo_ok) ? fmaxf(fmaf(xl[a][b][i][e], sc, sh), 0.f) : 0.f;
                                xh[a][b][i][e] = (cv && hi_ok) ? fmaxf(fmaf(xh[a][b][i][e], sc, sh), 0.f) : 0.f;
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { av[e] = (__bf16)xl[a][b][i][e]; av[4 + e] = (__bf16)xh[a][b][i][e]; }
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx)
                            acc[a][b][kx][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bk[kx], acc[a][b][kx][i], 0, 0, 0);
                    }
        }
    }
    for (int wv = 0; wv < 4; ++wv) {
        if (wid == wv) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < CIT; ++i)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) {
                                float *slot = &red[((((a * 2 + b) * 4 + k) * CIT + i) * 4) + reg][lane];
                                *slot = wv ? *slot + acc[a][b][k][i][reg] : acc[a][b][k][i][reg];
                            }
        }
        __syncthreads();
    }
    float *dst = partial + (int64_t)chunk * s.cin * s.cout * 64;
    for (int e = t; e < FR * 4 * 64; e += 256) {
        const int ln = e % 64, f4 = e / 64, reg = f4 % 4, f = f4 / 4;
        const int i = f % CIT, k = (f / CIT) % 4, b = (f / (CIT * 4)) % 2, a = f / (CIT * 8);
        const int ci = ci_base + i * 16 + 4 * (ln >> 4) + reg, oc = bt * 16 + (ln & 15);   // C/D layout: row = 4*(lane>>4)+reg, col = lane&15
        const int kz = ct_k(pz, a), ky = ct_k(py, b);
        if (ci < s.cin && oc < s.cout) dst[(((int64_t)ci * s.cout + oc) * 16 + kz * 4 + ky) * 4 + k] = red[f4][ln];
    }
}

// narrow-output variant (cout <= 4, the 16 -> 3 layer; w % 8 == 0): the 16 MFMA columns carry (co, kx) pairs instead of 16 output
// channels (which would leave 13 of 16 columns and lanes idle): column n = co*4 + kx needs dout[co][2c-1+kx] for the lane's 8 cells
// = every second float of one 16-float window (four 16-byte loads), one MFMA per (kz,ky) row and 32-cell step, and one wave keeps
// all 16 (kz,ky) accumulators.  The 16 window loads of four (kz,ky) rows are issued together, branch-free (rows / columns / lanes
// outside the tensor get an out-of-range buffer offset): the first version waited for memory once per (kz,ky) row with two
// waves per SIMD (3.5 ms at [4,16,10,376,376]).
template <int CIT, typename TD = float, typename TXI = float>
__global__ __launch_bounds__(256) void ct_wgrad_narrow_kernel(const TXI *__restrict__ x, const TD *__restrict__ dout, CtDims s, int rows_per_block,
                                                              int n_chunks, float *__restrict__ partial, const float *__restrict__ in_norm = nullptr) {
    constexpr int FR = 16 * CIT;
    constexpr unsigned XS = sizeof(TXI);   // r06: bf16-stored x (the 16-channel z): the lane's 8 cells are ONE 16-byte load
    constexpr bool D16 = sizeof(TD) == 2;   // bf16 dout: the 16-element window is eight dwords (two 16-byte loads instead of four)
    constexpr unsigned ES = sizeof(TD);
    __shared__ float red[FR * 4][64];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int co = r >> 2, kx = r & 3;
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    const int64_t cells = (int64_t)s.d * s.h * s.w, oplane = (int64_t)od * oh * ow;
    const int64_t total_rows = (int64_t)s.n * s.d * s.h;
    // XCD-aware: an XCD works on a contiguous run of row chunks - the chunk one z plane further re-reads half of this chunk's dout rows
    const int chunk = xcd_tile(blockIdx.x, gridDim.x);
    if (chunk >= n_chunks) return;   // grid padding
    const int64_t r0 = (int64_t)chunk * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < total_rows ? r0 + rows_per_block : total_rows;
    f32x4m acc[16][CIT];
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int a = 0; a < CIT; ++a) acc[g][a] = f32x4m{0.f, 0.f, 0.f, 0.f};
    const int steps = (s.w + 31) / 32;
    // the lane's scale / shift of the folded batch norm, read once (inside the step loop the two dword loads sat in front of every x load's use)
    float nsc_l[CIT], nsh_l[CIT];
#pragma unroll
    for (int ai = 0; ai < CIT; ++ai) {
        const int ci = ai * 16 + r;
        nsc_l[ai] = (in_norm && ci < s.cin) ? in_norm[ci] : 0.f;
        nsh_l[ai] = (in_norm && ci < s.cin) ? in_norm[s.cin + ci] : 0.f;
    }
    const unsigned xbytes = (unsigned)(s.cin * cells * XS), dbytes = (unsigned)(s.cout * oplane * ES);
    // window of the lane: floats [base, base + 16) of the dout row with base = 2*c0 - 2 + 2*((kx + 1) >> 1); sample e = element
    // 2e + (kx odd ? 0 : 1), i.e. position 2*(c0 + e) - 1 + kx
    const int wshift = 2 * ((kx + 1) >> 1) - 2;
    const bool odd = kx & 1;
    const unsigned dlane = co < s.cout ? (unsigned)(co * oplane * ES) : CT_OOB;
    const uint32_t half_sel = odd ? 0x05040100u : 0x07060302u;   // bf16 dout: sample e = the low (odd kx) / high (even kx) half of window word e
    for (int64_t row = r0 + wid; row < r1; row += 4) {
        const int hy = (int)(row % s.h), hz = (int)((row / s.h) % s.d), n = (int)(row / ((int64_t)s.h * s.d));
        const __amdgpu_buffer_rsrc_t xr = ct_rsrc(x + (int64_t)n * s.cin * cells, xbytes);
        const __amdgpu_buffer_rsrc_t dr = ct_rsrc(dout + (int64_t)n * s.cout * oplane, dbytes);
        const unsigned xrow = (unsigned)(((int64_t)hz * s.h + hy) * s.w * XS);
        for (int st = 0; st < steps; ++st) {
            const int c0 = st * 32 + 8 * q;
            const bool cok = c0 < s.w;   // w % 8 == 0: the lane's 8 cells are all inside or all outside
            bf16x8m a[CIT];
#pragma unroll
            for (int ai = 0; ai < CIT; ++ai) {
                const int ci = ai * 16 + r;
                const unsigned voff = (cok && ci < s.cin) ? (unsigned)(ci * cells * XS) + xrow + (unsigned)c0 * XS : CT_OOB;
                f32x4m lo, hi;
                if constexpr (XS == 2) {
                    const u32x4m u = ct_load4u(xr, voff, 0);
                    if (!in_norm) {   // (wave-uniform) the stored bf16 values are the operand
                        a[ai] = __builtin_bit_cast(bf16x8m, u);
                        continue;
                    }
                    {   // widen, normalise (scalar v_fma_f32: ct_fma_scalar), narrow two per v_cvt_pk_bf16_f32, ReLU on the rounded pair (see ct_fwd_zslide_kernel)
                        typedef float f32x2x __attribute__((ext_vector_type(2)));
                        typedef short s16x2x __attribute__((ext_vector_type(2)));
                        const bool cv = cok && ci < s.cin;
                        const float sc = cv ? nsc_l[ai] : 0.f, sh = cv ? nsh_l[ai] : 0.f;   // (lanes outside the tensor loaded 0 and stay 0)
                        u32x4m o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            f32x2x val = {__uint_as_float(u[e] << 16), __uint_as_float(u[e] & 0xffff0000u)};
                            val = f32x2x{ct_fma_scalar(val[0], sc, sh), ct_fma_scalar(val[1], sc, sh)};
                            const s16x2x r2 = __builtin_elementwise_max(__builtin_bit_cast(s16x2x, ct_cvt_pk_bf16(val[0], val[1])), s16x2x{0, 0});
                            o[e] = __builtin_bit_cast(uint32_t, r2);
                        }
                        a[ai] = __builtin_bit_cast(bf16x8m, o);
                        continue;
                    }
                } else {
                    lo = ct_load4(xr, voff, 0);
                    hi = ct_load4(xr, voff, 16);
                }
                if (in_norm) {   // x is the raw tensor in front of a BatchNorm3d + ReLU (see ct_fwd_mfma_kernel); loads outside the tensor stay zero
                    const bool cv = cok && ci < s.cin;
                    const float sc = cv ? nsc_l[ai] : 0.f, sh = cv ? nsh_l[ai] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = cv ? fmaxf(fmaf(lo[e], sc, sh), 0.f) : 0.f;
                        hi[e] = cv ? fmaxf(fmaf(hi[e], sc, sh), 0.f) : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[ai][e] = (__bf16)lo[e]; a[ai][4 + e] = (__bf16)hi[e]; }
            }
            // the window of the row's first cells with kx = 0 would start at position -2: an offset below the plane start is out
            // of range as a whole for plane 0, so those lanes start at 0 and take their samples two elements earlier
            const bool shifted = 2 * c0 + wshift < 0;
            const unsigned wlane = cok ? dlane + (unsigned)((2 * c0 + wshift + (shifted ? 2 : 0)) * ES) : CT_OOB;
            const bool first_bad = 2 * c0 - 1 + kx < 0, last_bad = 2 * (c0 + 7) - 1 + kx >= ow;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4m wv[4][D16 ? 1 : 4];
                u32x4m wu[4][2];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int z = 2 * hz - 1 + gq, y = 2 * hy - 1 + g4;
                    const bool rok = (unsigned)z < (unsigned)od && (unsigned)y < (unsigned)oh;   // wave-uniform
                    const unsigned soff = rok ? (unsigned)(((int64_t)z * oh + y) * ow * ES) : 0u;
                    const unsigned wl = rok ? wlane : CT_OOB;   // the range check sees the per-lane offset only
                    if constexpr (D16) {
                        wu[g4][0] = ct_load4u(dr, wl, soff);
                        wu[g4][1] = ct_load4u(dr, wl, soff + 16u);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) wv[g4][j] = ct_load4(dr, wl, soff + 16u * j);
                    }
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    bf16x8m b;
                    if constexpr (D16) {
                        // shifted lanes (kx = 0 at the row start) take element 2e - 1 = the high half of word e - 1; sample 0 is then position -1
                        uint32_t wsr[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const uint32_t cur = wu[g4][i >> 2][i & 3], prev = i ? wu[g4][(i - 1) >> 2][(i - 1) & 3] : 0u;
                            wsr[i] = shifted ? prev : cur;
                        }
                        u32x4m f;
#pragma unroll
                        for (int pq = 0; pq < 4; ++pq) f[pq] = __builtin_amdgcn_perm(wsr[2 * pq + 1], wsr[2 * pq], half_sel);
                        f[0] = first_bad ? (f[0] & 0xFFFF0000u) : f[0];   // position -1 of the row belongs to the previous row
                        f[3] = last_bad ? (f[3] & 0x0000FFFFu) : f[3];    // position ow to the next one
                        b = __builtin_bit_cast(bf16x8m, f);
                    } else
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float v = odd ? wv[g4][e >> 1][(2 * e) & 3] : wv[g4][e >> 1][(2 * e + 1) & 3];
                        if (e > 0) v = shifted ? wv[g4][(2 * e - 1) >> 2][(2 * e - 1) & 3] : v;   // (kx = 0 is even: element 2e + 1 - 2)
                        if (e == 0) v = first_bad ? 0.f : v;   // position -1 of the row (c0 = 0, kx = 0) belongs to the previous row
                        if (e == 7) v = last_bad ? 0.f : v;    // position ow (last cells, kx = 3) to the next one
                        b[e] = (__bf16)v;
                    }
#pragma unroll
                    for (int ai = 0; ai < CIT; ++ai)
                        acc[gq * 4 + g4][ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ai], b, acc[gq * 4 + g4][ai], 0, 0, 0);
                }
            }
        }
    }
    for (int wv = 0; wv < 4; ++wv) {
        if (wid == wv) {
#pragma unroll
            for (int g = 0; g < 16; ++g)
#pragma unroll
                for (int a = 0; a < CIT; ++a)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        float *slot = &red[(g * CIT + a) * 4 + reg][lane];
                        *slot = wv ? *slot + acc[g][a][reg] : acc[g][a][reg];
                    }
        }
        __syncthreads();
    }
    float *dst = partial + (int64_t)chunk * s.cin * s.cout * 64;
    for (int e = t; e < FR * 4 * 64; e += 256) {
        const int ln = e % 64, f4 = e / 64, reg = f4 % 4, f = f4 / 4;
        const int a = f % CIT, g = f / CIT;
        const int ci = a * 16 + 4 * (ln >> 4) + reg, n_col = ln & 15;
        const int oc = n_col >> 2, okx = n_col & 3;
        if (ci < s.cin && oc < s.cout) dst[(((int64_t)ci * s.cout + oc) * 16 + g) * 4 + okx] = red[f4][ln];
    }
}

// fold the per-block slabs: 16 outputs x 16 slab lanes per block, fixed-order fold of the lanes through LDS
__global__ __launch_bounds__(256) void ct_slab_reduce_kernel(const float *__restrict__ partial, int n_slabs, int64_t size, float *__restrict__ out) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + o;
    float sum = 0.f;
    if (i < size)
        for (int sidx = sl; sidx < n_slabs; sidx += 16) sum += partial[(int64_t)sidx * size + i];
    red[sl][o] = sum;
    __syncthreads();
    if (sl == 0 && i < size) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k][o];
        out[i] = tot;
    }
}

static bool ct_mfma_ok(int cin, int cout) { return (cin == 32 || cin == 16) && cout >= 1 && cout <= 32; }
static int ct_wgrad_blocks(int64_t rows) { return (int)(rows < 512 ? rows : 512); }
static bool ct_wgrad_is_narrow(int cout, int w) { return cout <= 4 && w % 8 == 0; }
static int ct_wgrad_blocks_for(int64_t rows, int cout, int w) {
    if (ct_wgrad_is_narrow(cout, w)) return (int)(rows < 1024 ? rows : 1024);
    if (w % 4 == 0) return (int)(rows < 128 ? rows : 128);   // output-row-major kernel: grid (chunks, 4 classes x co tiles)
    return ct_wgrad_blocks(rows);
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_convt3d_mfma_supported(int cin, int cout) { return ct_mfma_ok(cin, cout); }

/* packed weight images: forward [8][ksteps][nt][64][8], data gradient [16][ksteps'][nt'][64][8] (staged kernel), then the same
 * shape in the direct kernel's K order (bf16 elements) */
extern "C" size_t s2d_convt3d_mfma_packed_elems(int cin, int cout) {
    if (!ct_mfma_ok(cin, cout)) return 0;
    const int nt_f = (cout + 15) / 16, nt_d = cin / 16, cop = cout <= 8 ? 8 : 32;
    return (size_t)8 * (cin == 32 ? 8 : 4) * nt_f * 512 + (size_t)2 * 16 * (cop == 32 ? 4 : 1) * nt_d * 512;
}

extern "C" int s2d_convt3d_mfma_pack_weights(const float *weight, int cin, int cout, void *packed, s2d_stream_t stream) {
    S2D_CHECK_ARG(weight && packed, "convt3d_mfma_pack: null argument");
    if (!ct_mfma_ok(cin, cout)) {
        set_error("convt3d_mfma: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int nt_f = (cout + 15) / 16, nt_d = cin / 16, cop = cout <= 8 ? 8 : 32;
    const int n_f = 8 * (cin == 32 ? 8 : 4) * nt_f * 512, n_d = 16 * (cop == 32 ? 4 : 1) * nt_d * 512;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ct_pack_fwd_kernel, dim3((n_f + 255) / 256), dim3(256), 0, st, weight, cin, cout, nt_f, (__bf16 *)packed);
    hipLaunchKernelGGL(ct_pack_dgrad_kernel, dim3((n_d + 255) / 256), dim3(256), 0, st, weight, cin, cout, cop, nt_d, 0, (__bf16 *)packed + n_f);
    hipLaunchKernelGGL(ct_pack_dgrad_kernel, dim3((n_d + 255) / 256), dim3(256), 0, st, weight, cin, cout, cop, nt_d, cout <= 4 ? 2 : 1,
                       (__bf16 *)packed + n_f + n_d);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// input rows per forward block: 4 for 16 input channels (3 x 6 staged rows, 37 KB of LDS: 0.69 -> 0.56 ms for the 16 -> 3 layer); 32
// input channels stay at one row - two rows (50 KB, fewer resident workgroups) measured 0.52 ms against 0.35
static int ct_fwd_rows(int cin) { return cin == 32 ? 1 : 4; }

// r06: 16-channel inputs take the z-sliding kernel (a block = (n, 4 rows, x tile) for every z: each plane staged once per block instead of three
// times); S2D_CT_ZSLIDE=0 keeps the per-z blocks for A/B runs
static bool ct_fwd_zslide(int cin) {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("S2D_CT_ZSLIDE");
        on = e ? atoi(e) : 1;
    }
    return on && (cin == 16 || (cin == 32 && on == 2));   // S2D_CT_ZSLIDE=2: the 32-channel layer too (A/B)
}

extern "C" int64_t s2d_convt3d_mfma_stats_tiles(int batch, int cin, int d, int h, int w) {
    const int yr = ct_fwd_rows(cin);
    return (int64_t)batch * (ct_fwd_zslide(cin) ? 1 : d) * ((h + yr - 1) / yr) * ((w + CT_TX - 1) / CT_TX);
}

/* stats_partial (optional, [s2d_convt3d_mfma_stats_tiles][2][cout]): per-block (sum, sum of squares) per output channel of the written
 * output - the statistics pass of the BatchNorm3d that follows (s2d_bn_partials_sum_f32 folds them) */
template <typename TO, typename TXI = float>
static int ct_fwd_launch(const TXI *in, const void *packed, const float *bias, int batch, int cin, int cout, int d, int h, int w, TO *out,
                         float *stats_partial, s2d_stream_t stream, const float *in_norm = nullptr) {
    S2D_CHECK_ARG(in && packed && out && batch > 0 && d > 0 && h > 0 && w > 0, "convt3d_mfma_fwd: bad argument");
    if (!ct_mfma_ok(cin, cout)) return S2D_ERR_UNSUPPORTED;
    CtDims s{batch, d, h, w, cin, cout};
    const int xtiles = (w + CT_TX - 1) / CT_TX;
    const int yr = ct_fwd_rows(cin), hg = (h + yr - 1) / yr;
    const int nt = (cout + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    const __bf16 *wp = (const __bf16 *)packed;
    if constexpr (sizeof(TXI) == 2) {   // bf16-stored input: the z-sliding kernel of the 16 -> (<= 16) layers (callers: s2d_convt3d_mfma_x16_supported)
        if (!(cin == 16 && nt == 1 && w % 8 == 0)) return S2D_ERR_UNSUPPORTED;
        const dim3 zgrid(xcd_grid((int64_t)batch * hg * xtiles)), zblk(256);
        hipLaunchKernelGGL((ct_fwd_zslide_kernel<16, 1, 4, TO, TXI>), zgrid, zblk, 0, st, in, wp, bias, s, xtiles, out, stats_partial, in_norm);
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    } else {
    if (ct_fwd_zslide(cin)) {
        const dim3 zgrid(xcd_grid((int64_t)batch * hg * xtiles)), zblk(256);
        if (cin == 32 && nt == 2) hipLaunchKernelGGL((ct_fwd_zslide_kernel<32, 2, 1, TO>), zgrid, zblk, 0, st, in, wp, bias, s, xtiles, out, stats_partial, in_norm);
        else if (cin == 32) hipLaunchKernelGGL((ct_fwd_zslide_kernel<32, 1, 1, TO>), zgrid, zblk, 0, st, in, wp, bias, s, xtiles, out, stats_partial, in_norm);
        else if (nt == 2) hipLaunchKernelGGL((ct_fwd_zslide_kernel<16, 2, 4, TO>), zgrid, zblk, 0, st, in, wp, bias, s, xtiles, out, stats_partial, in_norm);
        else hipLaunchKernelGGL((ct_fwd_zslide_kernel<16, 1, 4, TO>), zgrid, zblk, 0, st, in, wp, bias, s, xtiles, out, stats_partial, in_norm);
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
    const int64_t blocks = (int64_t)batch * d * hg * xtiles;
    S2D_CHECK_ARG(blocks < 0x7fffffff, "convt3d_mfma_fwd: grid too large");
    const dim3 grid(xcd_grid(blocks)), blk(256);
    // 3 z planes x (yc * yr + 2) input rows live per chunk of yc row groups
    const CtTileMap map = ct_tile_map(d, hg, xtiles, ct_chunk_rows((int64_t)w * cin * 4, yr, 2, 3));
    if (cin == 32 && nt == 2) hipLaunchKernelGGL((ct_fwd_mfma_kernel<32, 2, 1, TO>), grid, blk, 0, st, in, wp, bias, s, xtiles, map, out, stats_partial, in_norm);
    else if (cin == 32) hipLaunchKernelGGL((ct_fwd_mfma_kernel<32, 1, 1, TO>), grid, blk, 0, st, in, wp, bias, s, xtiles, map, out, stats_partial, in_norm);
    else if (nt == 2) hipLaunchKernelGGL((ct_fwd_mfma_kernel<16, 2, 4, TO>), grid, blk, 0, st, in, wp, bias, s, xtiles, map, out, stats_partial, in_norm);
    else hipLaunchKernelGGL((ct_fwd_mfma_kernel<16, 1, 4, TO>), grid, blk, 0, st, in, wp, bias, s, xtiles, map, out, stats_partial, in_norm);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
    }
}

extern "C" int s2d_convt3d_mfma_fwd_stats(const float *in, const void *packed, const float *bias, int batch, int cin, int cout, int d, int h,
                                          int w, float *out, float *stats_partial, s2d_stream_t stream) {
    return ct_fwd_launch<float>(in, packed, bias, batch, cin, cout, d, h, w, out, stats_partial, stream);
}
/* r04: the same forward writing its output as bf16 [B][cout][2d][2h][2w] (read by s2d_pcr_level_*_y16) */
extern "C" int s2d_convt3d_mfma_fwd_stats_y16(const float *in, const void *packed, const float *bias, int batch, int cin, int cout, int d, int h,
                                              int w, void *out_bf16, float *stats_partial, s2d_stream_t stream) {
    return ct_fwd_launch<__bf16>(in, packed, bias, batch, cin, cout, d, h, w, (__bf16 *)out_bf16, stats_partial, stream);
}

/* r04: ... with the BatchNorm3d + ReLU in FRONT of the layer applied while the input is staged: `in` is the raw tensor, in_scale_shift =
 * scale[cin] | shift[cin] (device); the normalised tensor is never materialised (see also s2d_convt3d_mfma_wgrad_d16_norm) */
extern "C" int s2d_convt3d_mfma_fwd_stats_y16_norm(const float *in, const float *in_scale_shift, const void *packed, const float *bias, int batch,
                                                   int cin, int cout, int d, int h, int w, void *out_bf16, float *stats_partial,
                                                   s2d_stream_t stream) {
    S2D_CHECK_ARG(in_scale_shift, "convt3d_mfma_fwd_stats_y16_norm: null scale / shift");
    return ct_fwd_launch<__bf16>(in, packed, bias, batch, cin, cout, d, h, w, (__bf16 *)out_bf16, stats_partial, stream, in_scale_shift);
}

/* r06: ... reading a bf16-STORED raw input (the 16-channel z written in bf16 by s2d_pcr_level_fwd_y16_z16); stats tiles as for the fp32 input
 * with the z-sliding plan (s2d_convt3d_mfma_stats_tiles counts that plan when S2D_CT_ZSLIDE is on - the x16 path requires it) */
extern "C" int s2d_convt3d_mfma_fwd_stats_y16_norm_x16(const void *in_bf16, const float *in_scale_shift, const void *packed, const float *bias,
                                                       int batch, int cin, int cout, int d, int h, int w, void *out_bf16, float *stats_partial,
                                                       s2d_stream_t stream) {
    S2D_CHECK_ARG(in_scale_shift, "convt3d_mfma_fwd_stats_y16_norm_x16: null scale / shift");
    if (!ct_fwd_zslide(cin)) return S2D_ERR_UNSUPPORTED;
    return ct_fwd_launch<__bf16, __bf16>((const __bf16 *)in_bf16, packed, bias, batch, cin, cout, d, h, w, (__bf16 *)out_bf16, stats_partial, stream,
                                         in_scale_shift);
}

extern "C" int s2d_convt3d_mfma_fwd(const float *in, const void *packed, const float *bias, int batch, int cin, int cout, int d, int h,
                                    int w, float *out, s2d_stream_t stream) {
    return s2d_convt3d_mfma_fwd_stats(in, packed, bias, batch, cin, cout, d, h, w, out, nullptr, stream);
}

// the direct (LDS-free) data-gradient kernels cover a layer when one sample's fp32 dout stays under 2 GB
static bool ct_dgrad_is_direct(int cout, int d, int h, int w) { return (int64_t)cout * 8 * d * h * w * 4 < ((int64_t)1 << 31); }

template <typename TD, typename TI = float>
static int ct_dgrad_direct_launch(const TD *dout, const void *packed, int batch, int cin, int cout, int d, int h, int w, TI *din, hipStream_t st) {
    CtDims s{batch, d, h, w, cin, cout};
    const dim3 blk(256);
    const int nt_f = (cout + 15) / 16;
    const __bf16 *wp = (const __bf16 *)packed + (size_t)8 * (cin == 32 ? 8 : 4) * nt_f * 512;
    const bool narrow = cout <= 8;
    constexpr int MT = 4;
    const int tpr = (w + 16 * MT - 1) / (16 * MT);
    const int64_t items = (int64_t)batch * d * h * tpr;
    S2D_CHECK_ARG(items < 0x7fffffff, "convt3d_mfma_dgrad: too many tiles");
    const dim3 g2(xcd_grid(std::min<int64_t>(ceil_div(items, 4), 256 * 16)));
    // live across a z step: 4 z planes x (2 yc + 2) rows of dout, cout planes each
    const CtTileMap map = ct_tile_map(d, h, tpr, ct_chunk_rows((int64_t)2 * w * cout * sizeof(TD), 2, 2, 4));
    const __bf16 *wd = wp + (size_t)16 * (narrow ? 1 : 4) * (cin / 16) * 512;   // the direct kernel's image follows the staged one
    if constexpr (sizeof(TI) == 2) {   // bf16-stored din: the narrow 16-channel layer (the 16 -> 3 up-sampler behind the bf16-stored z)
        if (!(narrow && cout <= 4 && cin == 16 && w % 4 == 0)) return S2D_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((ct_dgrad_direct_kernel<8, 1, MT, true, TD, TI>), g2, blk, 0, st, dout, wd, s, tpr, map, din);
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    } else
    if (narrow && cout <= 4 && cin == 32) hipLaunchKernelGGL((ct_dgrad_direct_kernel<8, 2, MT, true, TD>), g2, blk, 0, st, dout, wd, s, tpr, map, din);
    else if (narrow && cout <= 4) hipLaunchKernelGGL((ct_dgrad_direct_kernel<8, 1, MT, true, TD>), g2, blk, 0, st, dout, wd, s, tpr, map, din);
    else if (narrow && cin == 32) hipLaunchKernelGGL((ct_dgrad_direct_kernel<8, 2, MT, false, TD>), g2, blk, 0, st, dout, wd, s, tpr, map, din);
    else if (narrow) hipLaunchKernelGGL((ct_dgrad_direct_kernel<8, 1, MT, false, TD>), g2, blk, 0, st, dout, wd, s, tpr, map, din);
    else if (cin == 32) hipLaunchKernelGGL((ct_dgrad_direct_kernel<32, 2, MT, false, TD>), g2, blk, 0, st, dout, wd, s, tpr, map, din);
    else hipLaunchKernelGGL((ct_dgrad_direct_kernel<32, 1, MT, false, TD>), g2, blk, 0, st, dout, wd, s, tpr, map, din);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_convt3d_mfma_dgrad(const float *dout, const void *packed, int batch, int cin, int cout, int d, int h, int w,
                                      float *din, s2d_stream_t stream) {
    S2D_CHECK_ARG(dout && packed && din && batch > 0 && d > 0 && h > 0 && w > 0, "convt3d_mfma_dgrad: bad argument");
    if (!ct_mfma_ok(cin, cout)) return S2D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (ct_dgrad_is_direct(cout, d, h, w)) return ct_dgrad_direct_launch<float>(dout, packed, batch, cin, cout, d, h, w, din, st);
    CtDims s{batch, d, h, w, cin, cout};
    const int xtiles = (w + CT_TX - 1) / CT_TX;
    const int64_t blocks = (int64_t)batch * d * h * xtiles;
    S2D_CHECK_ARG(blocks < 0x7fffffff, "convt3d_mfma_dgrad: grid too large");
    const dim3 grid((unsigned)blocks), blk(256);
    const int nt_f = (cout + 15) / 16;
    const __bf16 *wp = (const __bf16 *)packed + (size_t)8 * (cin == 32 ? 8 : 4) * nt_f * 512;
    const bool narrow = cout <= 8;
    if (narrow && cin == 32) hipLaunchKernelGGL((ct_dgrad_mfma_kernel<8, 2, 16>), grid, blk, 0, st, dout, wp, s, xtiles, din);
    else if (narrow) hipLaunchKernelGGL((ct_dgrad_mfma_kernel<8, 1, 16>), grid, blk, 0, st, dout, wp, s, xtiles, din);
    else if (cin == 32) hipLaunchKernelGGL((ct_dgrad_mfma_kernel<32, 2, 2>), grid, blk, 0, st, dout, wp, s, xtiles, din);
    else hipLaunchKernelGGL((ct_dgrad_mfma_kernel<32, 1, 2>), grid, blk, 0, st, dout, wp, s, xtiles, din);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" size_t s2d_convt3d_mfma_wgrad_workspace_bytes(int batch, int cin, int cout, int d, int h, int w) {
    if (batch <= 0 || d <= 0 || h <= 0 || w <= 0 || !ct_mfma_ok(cin, cout)) return 0;
    return align_up((size_t)ct_wgrad_blocks_for((int64_t)batch * d * h, cout, w) * cin * cout * 64 * sizeof(float), 256);
}

// the weight-gradient kernels that read a bf16-stored dout: the narrow one and the output-row-major one
static bool ct_wgrad_has_d16(int cout, int w) { return ct_wgrad_is_narrow(cout, w) || w % 4 == 0; }

template <typename TD, typename TXI = float>
static int ct_wgrad_launch(const TXI *in, const TD *dout, int batch, int cin, int cout, int d, int h, int w, float *dweight, void *ws, hipStream_t st,
                           const float *in_norm = nullptr) {
    CtDims s{batch, d, h, w, cin, cout};
    const int64_t rows = (int64_t)batch * d * h;
    const int bx = ct_wgrad_blocks_for(rows, cout, w);
    const int rpb = (int)ceil_div(rows, bx);
    float *partial = (float *)ws;
    const int cit = cin / 16, cot = (cout + 15) / 16;
    const bool narrow = ct_wgrad_is_narrow(cout, w);
    if (in_norm && !ct_wgrad_has_d16(cout, w)) return S2D_ERR_UNSUPPORTED;   // the input-norm fold lives in the narrow / row-major kernels
    if constexpr (sizeof(TXI) == 2) {   // bf16-stored input: the narrow 16-channel layer only (s2d_convt3d_mfma_x16_supported)
        if (!(narrow && cit == 1)) return S2D_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((ct_wgrad_narrow_kernel<1, TD, TXI>), dim3(xcd_grid(bx)), dim3(256), 0, st, in, dout, s, rpb, bx, partial, in_norm);
    } else
    if (narrow && cit == 1) hipLaunchKernelGGL((ct_wgrad_narrow_kernel<1, TD>), dim3(xcd_grid(bx)), dim3(256), 0, st, in, dout, s, rpb, bx, partial, in_norm);
    else if (narrow) hipLaunchKernelGGL((ct_wgrad_narrow_kernel<2, TD>), dim3(xcd_grid(bx)), dim3(256), 0, st, in, dout, s, rpb, bx, partial, in_norm);
    // both input-channel tiles in one block: dout is read once (measured 0.62 ms against 0.85 ms with one tile per block and two
    // resident waves per SIMD, 32 -> 32 at [4,32,5,188,188])
    else if (w % 4 == 0 && cit == 2) hipLaunchKernelGGL((ct_wgrad_rows_kernel<2, TD>), dim3(xcd_grid((int64_t)bx * 4 * cot)), dim3(256), 0, st, in, dout, s, rpb, cot, 4 * cot, bx, partial, in_norm);
    else if (w % 4 == 0) hipLaunchKernelGGL((ct_wgrad_rows_kernel<1, TD>), dim3(xcd_grid((int64_t)bx * 4 * cot * cit)), dim3(256), 0, st, in, dout, s, rpb, cot, 4 * cot * cit, bx, partial, in_norm);
    else if constexpr (sizeof(TD) == 4) {
        if (cit == 2 && cot == 2) hipLaunchKernelGGL((ct_wgrad_mfma_kernel<2, 2, 2>), dim3(bx, 8), dim3(256), 0, st, in, dout, s, rpb, partial);
        else if (cit == 2) hipLaunchKernelGGL((ct_wgrad_mfma_kernel<2, 1, 4>), dim3(bx, 4), dim3(256), 0, st, in, dout, s, rpb, partial);
        else if (cot == 2) hipLaunchKernelGGL((ct_wgrad_mfma_kernel<1, 2, 4>), dim3(bx, 4), dim3(256), 0, st, in, dout, s, rpb, partial);
        else hipLaunchKernelGGL((ct_wgrad_mfma_kernel<1, 1, 8>), dim3(bx, 2), dim3(256), 0, st, in, dout, s, rpb, partial);
    } else {
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t size = (int64_t)cin * cout * 64;
    hipLaunchKernelGGL(ct_slab_reduce_kernel, dim3((unsigned)ceil_div(size, 16)), dim3(256), 0, st, partial, bx, size, dweight);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_convt3d_mfma_wgrad(const float *in, const float *dout, int batch, int cin, int cout, int d, int h, int w,
                                      float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(in && dout && dweight && batch > 0 && d > 0 && h > 0 && w > 0, "convt3d_mfma_wgrad: bad argument");
    if (!ct_mfma_ok(cin, cout)) return S2D_ERR_UNSUPPORTED;
    const size_t need = s2d_convt3d_mfma_wgrad_workspace_bytes(batch, cin, cout, d, h, w);
    if (!ws || ws_bytes < need) {
        set_error("convt3d_mfma_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
        return S2D_ERR_WORKSPACE;
    }
    return ct_wgrad_launch<float>(in, dout, batch, cin, cout, d, h, w, dweight, ws, (hipStream_t)stream);
}

/* ---- bf16-stored output gradient (r04): the same data / weight gradients from a dout the producer wrote in bf16 (the kernels round it to
 * bf16 anyway: identical results, half the bytes on the 0.5-0.7 GB tensor).  _supported: the layer shapes the bf16-reading kernels cover. */
extern "C" int s2d_convt3d_mfma_d16_supported(int cin, int cout, int d, int h, int w) {
    return ct_mfma_ok(cin, cout) && d > 0 && h > 0 && w > 0 && ct_dgrad_is_direct(cout, d, h, w) && ct_wgrad_has_d16(cout, w);
}
/* the input-norm fold (s2d_convt3d_mfma_fwd_stats_y16_norm / _wgrad_d16_norm) pays on the narrow layers only: measured on 32 -> 32 at
 * [4,32,5,188,188] the output-row-major weight gradient went 0.48 -> 0.77 ms with the normalisation in its load path (it saves a 27 us
 * apply there); 16 -> 3 at [4,16,10,376,376]: forward 0.53 -> 0.56, weight gradient 0.45 -> 0.47 ms against a 0.12 ms apply pass */
extern "C" int s2d_convt3d_mfma_norm_supported(int cin, int cout, int d, int h, int w) {
    return s2d_convt3d_mfma_d16_supported(cin, cout, d, h, w) && ct_wgrad_is_narrow(cout, w);
}
extern "C" int s2d_convt3d_mfma_dgrad_d16(const void *dout_bf16, const void *packed, int batch, int cin, int cout, int d, int h, int w, float *din,
                                          s2d_stream_t stream) {
    S2D_CHECK_ARG(dout_bf16 && packed && din && batch > 0, "convt3d_mfma_dgrad_d16: bad argument");
    if (!s2d_convt3d_mfma_d16_supported(cin, cout, d, h, w)) return S2D_ERR_UNSUPPORTED;
    return ct_dgrad_direct_launch<__bf16>((const __bf16 *)dout_bf16, packed, batch, cin, cout, d, h, w, din, (hipStream_t)stream);
}
/* r06: ... writing the input gradient in bf16 as well (x16: the layer's input is the bf16-stored z, see s2d_convt3d_mfma_x16_supported) */
extern "C" int s2d_convt3d_mfma_x16_supported(int cin, int cout, int d, int h, int w) {
    return s2d_convt3d_mfma_norm_supported(cin, cout, d, h, w) && cin == 16 && cout <= 4 && w % 8 == 0 && ct_fwd_zslide(cin);
}
extern "C" int s2d_convt3d_mfma_dgrad_d16_x16(const void *dout_bf16, const void *packed, int batch, int cin, int cout, int d, int h, int w,
                                              void *din_bf16, s2d_stream_t stream) {
    S2D_CHECK_ARG(dout_bf16 && packed && din_bf16 && batch > 0, "convt3d_mfma_dgrad_d16_x16: bad argument");
    if (!s2d_convt3d_mfma_x16_supported(cin, cout, d, h, w)) return S2D_ERR_UNSUPPORTED;
    return ct_dgrad_direct_launch<__bf16, __bf16>((const __bf16 *)dout_bf16, packed, batch, cin, cout, d, h, w, (__bf16 *)din_bf16, (hipStream_t)stream);
}
extern "C" int s2d_convt3d_mfma_wgrad_d16(const float *in, const void *dout_bf16, int batch, int cin, int cout, int d, int h, int w, float *dweight,
                                          void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(in && dout_bf16 && dweight && batch > 0, "convt3d_mfma_wgrad_d16: bad argument");
    if (!s2d_convt3d_mfma_d16_supported(cin, cout, d, h, w)) return S2D_ERR_UNSUPPORTED;
    const size_t need = s2d_convt3d_mfma_wgrad_workspace_bytes(batch, cin, cout, d, h, w);
    if (!ws || ws_bytes < need) {
        set_error("convt3d_mfma_wgrad_d16: workspace too small (%zu < %zu)", ws_bytes, need);
        return S2D_ERR_WORKSPACE;
    }
    return ct_wgrad_launch<__bf16>(in, (const __bf16 *)dout_bf16, batch, cin, cout, d, h, w, dweight, ws, (hipStream_t)stream);
}
/* ... with the BatchNorm3d + ReLU in front of the layer applied to `in` on load (see s2d_convt3d_mfma_fwd_stats_y16_norm) */
extern "C" int s2d_convt3d_mfma_wgrad_d16_norm(const float *in, const float *in_scale_shift, const void *dout_bf16, int batch, int cin, int cout, int d,
                                               int h, int w, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(in && in_scale_shift && dout_bf16 && dweight && batch > 0, "convt3d_mfma_wgrad_d16_norm: bad argument");
    if (!s2d_convt3d_mfma_norm_supported(cin, cout, d, h, w)) return S2D_ERR_UNSUPPORTED;
    const size_t need = s2d_convt3d_mfma_wgrad_workspace_bytes(batch, cin, cout, d, h, w);
    if (!ws || ws_bytes < need) {
        set_error("convt3d_mfma_wgrad_d16_norm: workspace too small (%zu < %zu)", ws_bytes, need);
        return S2D_ERR_WORKSPACE;
    }
    return ct_wgrad_launch<__bf16>(in, (const __bf16 *)dout_bf16, batch, cin, cout, d, h, w, dweight, ws, (hipStream_t)stream, in_scale_shift);
}

/* r06: ... with `in` stored in bf16 (the 16-channel z, see s2d_convt3d_mfma_x16_supported) */
extern "C" int s2d_convt3d_mfma_wgrad_d16_norm_x16(const void *in_bf16, const float *in_scale_shift, const void *dout_bf16, int batch, int cin, int cout,
                                                   int d, int h, int w, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(in_bf16 && in_scale_shift && dout_bf16 && dweight && batch > 0, "convt3d_mfma_wgrad_d16_norm_x16: bad argument");
    if (!s2d_convt3d_mfma_x16_supported(cin, cout, d, h, w)) return S2D_ERR_UNSUPPORTED;
    const size_t need = s2d_convt3d_mfma_wgrad_workspace_bytes(batch, cin, cout, d, h, w);
    if (!ws || ws_bytes < need) {
        set_error("convt3d_mfma_wgrad_d16_norm_x16: workspace too small (%zu < %zu)", ws_bytes, need);
        return S2D_ERR_WORKSPACE;
    }
    return ct_wgrad_launch<__bf16, __bf16>((const __bf16 *)in_bf16, (const __bf16 *)dout_bf16, batch, cin, cout, d, h, w, dweight, ws,
                                           (hipStream_t)stream, in_scale_shift);
}

// ---- 1x1x1 Conv3d weight gradient (planar fp32 tensors) ------------------------------------------------------------
// dW[co][ci] = sum_{n,p} dy[n][co][p] * x[n][ci][p], db[co] = sum dy: skinny GEMMs with a reduction axis of 1e6..5e7 positions
// (hipBLASLt ran them at 1.0-1.8 ms each, r02 baseline).  Streaming reduction: grid (position chunks, co tiles of PW_CT, ci tiles
// of PW_CIT); a thread accumulates PW_CT x PW_CIT products over float4 position quads, the block folds its lanes by shuffles +
// LDS in a fixed order and writes partial[chunk][co][ci]; pw_wgrad_reduce folds the chunks.
namespace s2d {
constexpr int PW_CT = 4, PW_CIT = 32;

__global__ __launch_bounds__(256) void pw_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, int64_t p4, int batch, int cin,
                                                       int cout, int64_t quads_per_block, float *__restrict__ partial) {
    __shared__ float red[4][PW_CT * (PW_CIT + 1)];
    const int co0 = blockIdx.y * PW_CT, ci0 = blockIdx.z * PW_CIT;
    float acc[PW_CT][PW_CIT + 1];   // column PW_CIT: sum of dy (bias gradient), written by the ci0 == 0 blocks
#pragma unroll
    for (int c = 0; c < PW_CT; ++c)
#pragma unroll
        for (int i = 0; i <= PW_CIT; ++i) acc[c][i] = 0.f;
    const int64_t q0 = (int64_t)blockIdx.x * quads_per_block;
    const int64_t q1 = q0 + quads_per_block < p4 ? q0 + quads_per_block : p4;
    for (int n = 0; n < batch; ++n) {
        const float4 *xb = reinterpret_cast<const float4 *>(x) + (int64_t)n * cin * p4;
        const float4 *db = reinterpret_cast<const float4 *>(dy) + (int64_t)n * cout * p4;
        for (int64_t qd = q0 + threadIdx.x; qd < q1; qd += 256) {
            float4 g[PW_CT];
#pragma unroll
            for (int c = 0; c < PW_CT; ++c) {
                g[c] = co0 + c < cout ? db[(int64_t)(co0 + c) * p4 + qd] : float4{0.f, 0.f, 0.f, 0.f};
                acc[c][PW_CIT] += (g[c].x + g[c].y) + (g[c].z + g[c].w);
            }
#pragma unroll
            for (int i = 0; i < PW_CIT; ++i) {
                if (ci0 + i < cin) {
                    const float4 v = xb[(int64_t)(ci0 + i) * p4 + qd];
#pragma unroll
                    for (int c = 0; c < PW_CT; ++c)
                        acc[c][i] += (g[c].x * v.x + g[c].y * v.y) + (g[c].z * v.z + g[c].w * v.w);
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < PW_CT; ++c)
#pragma unroll
        for (int i = 0; i <= PW_CIT; ++i) {
            float s = acc[c][i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) red[wid][c * (PW_CIT + 1) + i] = s;
        }
    __syncthreads();
    float *dst = partial + (int64_t)blockIdx.x * cout * (cin + 1);
    for (int e = threadIdx.x; e < PW_CT * (PW_CIT + 1); e += 256) {
        const int c = e / (PW_CIT + 1), i = e % (PW_CIT + 1);
        const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (co0 + c >= cout) continue;
        if (i == PW_CIT) {
            if (ci0 == 0) dst[(int64_t)(co0 + c) * (cin + 1) + cin] = v;
        } else if (ci0 + i < cin) {
            dst[(int64_t)(co0 + c) * (cin + 1) + ci0 + i] = v;
        }
    }
}

// 16 outputs x 16 chunk lanes per block, fixed-order fold of the lanes through LDS
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float *__restrict__ partial, int chunks, int cin, int cout, float *__restrict__ dw,
                                                              float *__restrict__ db) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    const int total = cout * (cin + 1);
    float s = 0.f;
    if (i < total)
        for (int k = sl; k < chunks; k += 16) s += partial[(int64_t)k * total + i];
    red[sl][o] = s;
    __syncthreads();
    if (sl != 0 || i >= total) return;
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[k][o];
    const int co = i / (cin + 1), ci = i % (cin + 1);
    if (ci < cin) dw[(int64_t)co * cin + ci] = tot;
    else if (db) db[co] = tot;
}

// bf16-operand variant on the matrix cores (the bf16 compute mode): M = output channels (MT tiles of 16), N = input channels (NT
// tiles), K = positions, 32 per MFMA.  Both operands are read ONCE (the VALU kernel above re-reads x per tile of 4 output
// channels: 2.9 GB for the 32 -> 16 layer at [4,32,10,376,376], 0.76 ms).  A lane's 8 K-elements are the two 4-float chunks at
// p0 + 4q and p0 + 16 + 4q of its plane (the same position permutation on both operands, so the products pair up correctly):
// every load instruction reads 64 contiguous bytes per plane.  db = the VALU sum of the lane's own dy values.
template <int MT, int NT, typename TX = float, typename TG = float>
__global__ __launch_bounds__(256) void pw_wgrad_mfma_kernel(const TX *__restrict__ x, const TG *__restrict__ dy, const float *__restrict__ bnp,
                                                            int64_t positions, int batch, int cin, int cout, int steps_per_block,
                                                            float *__restrict__ partial) {
    __shared__ float red[MT * NT * 4 + MT][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    // optional prologue on x: relu(x*scale + shift) per input channel (x = the raw input of a batch norm + ReLU whose output the conv read)
    float psc[NT], psh[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        psc[n] = bnp ? bnp[n * 16 + r] : 1.f;
        psh[n] = bnp ? bnp[cin + n * 16 + r] : 0.f;
    }
    f32x4m acc[MT][NT];
    float bsum[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        bsum[m] = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4m{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t steps = (positions + 31) / 32;
    const int64_t s0 = (int64_t)blockIdx.x * steps_per_block;
    const int64_t s1 = s0 + steps_per_block < steps ? s0 + steps_per_block : steps;
    const unsigned plane = (unsigned)(positions * sizeof(TG)), xplane = (unsigned)(positions * sizeof(TX));   // (TG = __bf16, r06: the bf16-stored dz)
    for (int b = 0; b < batch; ++b) {
        const __amdgpu_buffer_rsrc_t xr = ct_rsrc(x + (int64_t)b * cin * positions, (unsigned)cin * xplane);
        const __amdgpu_buffer_rsrc_t yr = ct_rsrc(dy + (int64_t)b * cout * positions, (unsigned)cout * plane);
        for (int64_t st = s0 + wid; st < s1; st += 4) {
            const int64_t p0 = st * 32 + 4 * q;
            const bool ok0 = p0 < positions, ok1 = p0 + 16 < positions;   // positions % 4 == 0: chunks are all-in or all-out
            bf16x8m av[MT], bv[NT];
            f32x4m lo[MT + NT], hi[MT + NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int o = m * 16 + r;
                if constexpr (sizeof(TG) == 4) {
                    const unsigned base = o < cout ? (unsigned)o * plane + (unsigned)p0 * 4u : CT_OOB;
                    lo[m] = ct_load4(yr, ok0 ? base : CT_OOB, 0);
                    hi[m] = ct_load4(yr, ok1 ? base : CT_OOB, 64);
                } else {
                    const unsigned base = o < cout ? (unsigned)o * plane + (unsigned)p0 * 2u : CT_OOB;
                    const uint64_t l2 = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(yr, ok0 ? base : CT_OOB, 0, 0));
                    const uint64_t h2 = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(yr, ok1 ? base : CT_OOB, 32, 0));
                    const uint32_t la = (uint32_t)l2, lb = (uint32_t)(l2 >> 32), ha = (uint32_t)h2, hb = (uint32_t)(h2 >> 32);
                    lo[m] = f32x4m{__builtin_bit_cast(float, la << 16), __builtin_bit_cast(float, la & 0xFFFF0000u),
                                   __builtin_bit_cast(float, lb << 16), __builtin_bit_cast(float, lb & 0xFFFF0000u)};
                    hi[m] = f32x4m{__builtin_bit_cast(float, ha << 16), __builtin_bit_cast(float, ha & 0xFFFF0000u),
                                   __builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xFFFF0000u)};
                }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if constexpr (sizeof(TX) == 4) {
                    const unsigned base = (unsigned)(n * 16 + r) * plane + (unsigned)p0 * 4u;
                    lo[MT + n] = ct_load4(xr, ok0 ? base : CT_OOB, 0);
                    hi[MT + n] = ct_load4(xr, ok1 ? base : CT_OOB, 64);
                } else {   // bf16 planes: the lane's two 4-element chunks are 8-byte loads
                    const unsigned base = (unsigned)(n * 16 + r) * xplane + (unsigned)p0 * 2u;
                    // (the 8 bytes as ONE 64-bit integer: element access on the float-pair view of a pair of bf16 pairs returned element 0 twice)
                    const uint64_t l2 = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(xr, ok0 ? base : CT_OOB, 0, 0));
                    const uint64_t h2 = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(xr, ok1 ? base : CT_OOB, 32, 0));
                    const uint32_t la = (uint32_t)l2, lb = (uint32_t)(l2 >> 32), ha = (uint32_t)h2, hb = (uint32_t)(h2 >> 32);
                    lo[MT + n] = f32x4m{__builtin_bit_cast(float, la << 16), __builtin_bit_cast(float, la & 0xFFFF0000u),
                                        __builtin_bit_cast(float, lb << 16), __builtin_bit_cast(float, lb & 0xFFFF0000u)};
                    hi[MT + n] = f32x4m{__builtin_bit_cast(float, ha << 16), __builtin_bit_cast(float, ha & 0xFFFF0000u),
                                        __builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xFFFF0000u)};
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { av[m][e] = (__bf16)lo[m][e]; av[m][4 + e] = (__bf16)hi[m][e]; }
                bsum[m] += ((lo[m][0] + lo[m][1]) + (lo[m][2] + lo[m][3])) + ((hi[m][0] + hi[m][1]) + (hi[m][2] + hi[m][3]));
            }
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a0 = lo[MT + n][e], a1 = hi[MT + n][e];
                    if (bnp) {   // out-of-range positions read 0 and must stay 0: their dy is 0 too, so the product vanishes either way
                        a0 = fmaxf(fmaf(a0, psc[n], psh[n]), 0.f);
                        a1 = fmaxf(fmaf(a1, psc[n], psh[n]), 0.f);
                    }
                    bv[n][e] = (__bf16)a0;
                    bv[n][4 + e] = (__bf16)a1;
                }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
    }
    // fold the 4 waves in a fixed order, then write this block's slab partial[block][co][cin + 1]
    for (int wv = 0; wv < 4; ++wv) {
        if (wid == wv) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        float *slot = &red[(m * NT + n) * 4 + reg][lane];
                        *slot = wv ? *slot + acc[m][n][reg] : acc[m][n][reg];
                    }
                float *bs = &red[MT * NT * 4 + m][lane];
                *bs = wv ? *bs + bsum[m] : bsum[m];
            }
        }
        __syncthreads();
    }
    float *dst = partial + (int64_t)blockIdx.x * cout * (cin + 1);
    for (int e = threadIdx.x; e < MT * NT * 4 * 64; e += 256) {
        const int ln = e & 63, f4 = e >> 6, reg = f4 & 3, tile = f4 >> 2;
        const int n = tile % NT, m = tile / NT;
        const int o = m * 16 + 4 * (ln >> 4) + reg, c = n * 16 + (ln & 15);   // C/D layout: row = 4*(lane>>4)+reg, col = lane&15
        if (o < cout && c < cin) dst[(int64_t)o * (cin + 1) + c] = red[f4][ln];
    }
    for (int e = threadIdx.x; e < MT * 16; e += 256) {   // bias gradient: the 4 q-lanes of row r hold disjoint positions
        const int m = e >> 4, rr = e & 15, o = m * 16 + rr;
        if (o < cout) {
            const float *bs = red[MT * NT * 4 + m];
            dst[(int64_t)o * (cin + 1) + cin] = (bs[rr] + bs[16 + rr]) + (bs[32 + rr] + bs[48 + rr]);
        }
    }
}

constexpr int PW_CHUNKS = 512;
}  // namespace s2d

extern "C" size_t s2d_pointwise_conv_wgrad_workspace_bytes(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    return s2d::align_up((size_t)s2d::PW_CHUNKS * cout * (cin + 1) * sizeof(float), 256);
}

extern "C" int s2d_pointwise_conv_wgrad_f32(const float *in, const float *dout, int batch, int cin, int cout, int64_t positions,
                                            float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(in && dout && dweight && batch > 0 && cin > 0 && cout > 0 && positions > 0, "pointwise_conv_wgrad: bad argument");
    if (positions % 4) {
        s2d::set_error("pointwise_conv_wgrad: the position count must be a multiple of 4 (%lld)", (long long)positions);
        return S2D_ERR_UNSUPPORTED;
    }
    const size_t need = s2d_pointwise_conv_wgrad_workspace_bytes(cin, cout);
    if (!ws || ws_bytes < need) {
        s2d::set_error("pointwise_conv_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
        return S2D_ERR_WORKSPACE;
    }
    const int64_t p4 = positions / 4;
    const int chunks = (int)std::min<int64_t>(s2d::PW_CHUNKS, s2d::ceil_div(p4, 256));
    const int64_t qpb = s2d::ceil_div(p4, chunks);
    hipStream_t st = (hipStream_t)stream;
    float *partial = (float *)ws;
    hipLaunchKernelGGL(s2d::pw_wgrad_kernel, dim3(chunks, (unsigned)s2d::ceil_div(cout, s2d::PW_CT), (unsigned)s2d::ceil_div(cin, s2d::PW_CIT)),
                       dim3(256), 0, st, in, dout, p4, batch, cin, cout, qpb, partial);
    hipLaunchKernelGGL(s2d::pw_wgrad_reduce_kernel, dim3((unsigned)s2d::ceil_div((int64_t)cout * (cin + 1), 16)), dim3(256), 0, st, partial, chunks,
                       cin, cout, dweight, dbias);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_pointwise_conv_wgrad_norm_bf16(const float *in, const float *in_scale_shift, const float *dout, int batch, int cin, int cout,
                                                  int64_t positions, float *dweight, float *dbias, void *ws, size_t ws_bytes,
                                                  s2d_stream_t stream);

extern "C" int s2d_pointwise_conv_wgrad_bf16_supported(int cin, int cout, int64_t positions) {
    return ((cin == 128 && cout <= 32) || (cin == 32 && cout <= 16)) && cout > 0 && positions > 0 && positions % 4 == 0 &&
           (int64_t)cin * positions * 4 < ((int64_t)1 << 31) && (int64_t)cout * positions * 4 < ((int64_t)1 << 31);
}

/* same contract as s2d_pointwise_conv_wgrad_f32 (workspace included) with the operands rounded to bf16 on the matrix cores */
extern "C" int s2d_pointwise_conv_wgrad_bf16(const float *in, const float *dout, int batch, int cin, int cout, int64_t positions,
                                             float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    return s2d_pointwise_conv_wgrad_norm_bf16(in, nullptr, dout, batch, cin, cout, positions, dweight, dbias, ws, ws_bytes, stream);
}

/* ... with x = relu(in*scale + shift) applied on the fly: in_scale_shift (device, 2*cin) = scale[cin] | shift[cin], or NULL */
template <typename TX, typename TG = float>
static int pw_wgrad_norm_launch(const TX *in, const float *in_scale_shift, const TG *dout, int batch, int cin, int cout, int64_t positions,
                                float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(in && dout && dweight && batch > 0, "pointwise_conv_wgrad_bf16: bad argument");
    if (!s2d_pointwise_conv_wgrad_bf16_supported(cin, cout, positions)) {
        s2d::set_error("pointwise_conv_wgrad_bf16: unsupported %d -> %d over %lld positions", cin, cout, (long long)positions);
        return S2D_ERR_UNSUPPORTED;
    }
    const size_t need = s2d_pointwise_conv_wgrad_workspace_bytes(cin, cout);
    if (!ws || ws_bytes < need) {
        s2d::set_error("pointwise_conv_wgrad_bf16: workspace too small (%zu < %zu)", ws_bytes, need);
        return S2D_ERR_WORKSPACE;
    }
    const int64_t steps = (positions + 31) / 32;
    const int chunks = (int)std::min<int64_t>(s2d::PW_CHUNKS, s2d::ceil_div(steps, 16));
    const int spb = (int)s2d::ceil_div(steps, chunks);
    hipStream_t st = (hipStream_t)stream;
    float *partial = (float *)ws;
    if constexpr (sizeof(TG) == 2) {   // bf16-stored dout: the 32 -> 16 conv behind the bf16-stored z
        if (cin != 32) return S2D_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((s2d::pw_wgrad_mfma_kernel<1, 2, TX, TG>), dim3(chunks), dim3(256), 0, st, in, dout, in_scale_shift, positions, batch, cin, cout, spb,
                           partial);
    } else
    if (cin == 128)
        hipLaunchKernelGGL((s2d::pw_wgrad_mfma_kernel<2, 8, TX>), dim3(chunks), dim3(256), 0, st, in, dout, in_scale_shift, positions, batch, cin, cout, spb,
                           partial);
    else
        hipLaunchKernelGGL((s2d::pw_wgrad_mfma_kernel<1, 2, TX>), dim3(chunks), dim3(256), 0, st, in, dout, in_scale_shift, positions, batch, cin, cout, spb,
                           partial);
    hipLaunchKernelGGL(s2d::pw_wgrad_reduce_kernel, dim3((unsigned)s2d::ceil_div((int64_t)cout * (cin + 1), 16)), dim3(256), 0, st, partial, chunks,
                       cin, cout, dweight, dbias);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_pointwise_conv_wgrad_norm_bf16(const float *in, const float *in_scale_shift, const float *dout, int batch, int cin, int cout,
                                                  int64_t positions, float *dweight, float *dbias, void *ws, size_t ws_bytes,
                                                  s2d_stream_t stream) {
    return pw_wgrad_norm_launch<float>(in, in_scale_shift, dout, batch, cin, cout, positions, dweight, dbias, ws, ws_bytes, stream);
}
/* r04: the same with the input operand stored as bf16 [B][cin][positions] (the bf16 up-sampler output) */
extern "C" int s2d_pointwise_conv_wgrad_norm_x16(const void *in_bf16, const float *in_scale_shift, const float *dout, int batch, int cin, int cout,
                                                 int64_t positions, float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    return pw_wgrad_norm_launch<__bf16>((const __bf16 *)in_bf16, in_scale_shift, dout, batch, cin, cout, positions, dweight, dbias, ws, ws_bytes, stream);
}
/* r06: input AND output gradient stored as bf16 (x = the bf16 up-sampler output, dout = the bf16-stored dz; cin = 32) */
extern "C" int s2d_pointwise_conv_wgrad_norm_x16_d16(const void *in_bf16, const float *in_scale_shift, const void *dout_bf16, int batch, int cin, int cout,
                                                     int64_t positions, float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    return pw_wgrad_norm_launch<__bf16, __bf16>((const __bf16 *)in_bf16, in_scale_shift, (const __bf16 *)dout_bf16, batch, cin, cout, positions, dweight,
                                                dbias, ws, ws_bytes, stream);
}
