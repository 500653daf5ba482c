// Weight gradient of a bias-free Linear over a very long row-major matrix, exact fp32 on the matrix cores:
//   dW[co][ci] = sum over rows r of dY[r][co] * X[r][ci],      X [rows][ci], dY [rows][co] fp32 row-major, rows ~ 7e5
// (the PFN layers of the pillar reader, /root/reference/det3d/models/readers/pillar_encoder.py:41-56: nn.Linear(10, 32) and
// nn.Linear(64, 64) over P * 20 point slots; the library GEMM behind torch.mm takes 0.8 / 1.3 ms for these two skinny products).
// One v_mfma_f32_16x16x4_f32 contracts 4 rows: lane l supplies dY[r0 + (l >> 4)][co0 + (l & 15)] and X[r0 + (l >> 4)][ci0 + (l & 15)]
// (64-byte row pieces, every byte of both matrices is read exactly once); a wave keeps the whole [CO][CI] product in registers, a
// workgroup of 4 waves walks a contiguous range of row quads, folds its waves through LDS and writes one partial; a second launch
// sums the partials in a fixed order (deterministic).  Columns beyond ci / co are masked, so ci = 10 needs no padded copy.
#include "s2d_common.h"

namespace s2d {

typedef float f32x4r __attribute__((ext_vector_type(4)));

template <int COT, int CIT>   // 16-column tiles of dY and X
__global__ __launch_bounds__(256) void rows_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, int64_t rows, int ci, int co,
                                                         int64_t quads_per_block, float *__restrict__ partial) {
    __shared__ float red[4][COT * 16][CIT * 16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, k = lane >> 4;
    const int64_t q0 = (int64_t)blockIdx.x * quads_per_block;
    const int64_t q1 = min((rows + 3) / 4, q0 + quads_per_block);
    f32x4r acc[COT][CIT];
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < CIT; ++j) acc[i][j] = f32x4r{0.f, 0.f, 0.f, 0.f};
    bool a_ok[COT], b_ok[CIT];
#pragma unroll
    for (int i = 0; i < COT; ++i) a_ok[i] = 16 * i + c < co;
#pragma unroll
    for (int j = 0; j < CIT; ++j) b_ok[j] = 16 * j + c < ci;
    for (int64_t q = q0 + wid; q < q1; q += 4) {
        const int64_t r = 4 * q + k;
        const bool r_ok = r < rows;
        float a[COT], b[CIT];
#pragma unroll
        for (int i = 0; i < COT; ++i) a[i] = (r_ok && a_ok[i]) ? dy[r * co + 16 * i + c] : 0.f;
#pragma unroll
        for (int j = 0; j < CIT; ++j) b[j] = (r_ok && b_ok[j]) ? x[r * ci + 16 * j + c] : 0.f;
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
            for (int j = 0; j < CIT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    // C/D layout: row (co) = 4 * (lane >> 4) + reg, col (ci) = lane & 15
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < CIT; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) red[wid][16 * i + 4 * k + reg][16 * j + c] = acc[i][j][reg];
    __syncthreads();
    float *dst = partial + (int64_t)blockIdx.x * co * ci;
    for (int e = threadIdx.x; e < co * ci; e += 256) {
        const int o = e / ci, i = e - o * ci;
        dst[e] = (red[0][o][i] + red[1][o][i]) + (red[2][o][i] + red[3][o][i]);
    }
}

__global__ __launch_bounds__(256) void rows_wgrad_reduce_kernel(const float *__restrict__ partial, int blocks, int n, float *__restrict__ dw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += partial[(int64_t)b * n + e];
    dw[e] = s;
}

static int rows_wgrad_blocks(int64_t rows) {
    const int64_t quads = (rows + 3) / 4;
    return (int)(quads < 512 ? (quads > 0 ? quads : 1) : 512);
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_rows_wgrad_supported(int ci, int co) { return ci >= 1 && co >= 1 && ci <= 64 && co <= 64; }

extern "C" size_t s2d_rows_wgrad_workspace_bytes(int64_t rows, int ci, int co) {
    if (!s2d_rows_wgrad_supported(ci, co) || rows <= 0) return 0;
    return (size_t)rows_wgrad_blocks(rows) * ci * co * sizeof(float);
}

extern "C" int s2d_rows_wgrad_f32(const float *x, const float *dy, int64_t rows, int ci, int co, float *dweight, void *ws, size_t ws_bytes,
                                  s2d_stream_t stream) {
    S2D_CHECK_ARG(x && dy && dweight && rows > 0, "rows_wgrad: bad argument");
    if (!s2d_rows_wgrad_supported(ci, co)) {
        set_error("rows_wgrad: unsupported widths %d x %d (<= 64 each)", ci, co);
        return S2D_ERR_UNSUPPORTED;
    }
    const int blocks = rows_wgrad_blocks(rows);
    if (!ws || ws_bytes < (size_t)blocks * ci * co * sizeof(float)) {
        set_error("rows_wgrad: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t qpb = ceil_div((rows + 3) / 4, (int64_t)blocks);
    float *partial = (float *)ws;
    const int cot = (co + 15) / 16, cit = (ci + 15) / 16;
#define S2D_RW(A_, B_) hipLaunchKernelGGL((rows_wgrad_kernel<A_, B_>), dim3(blocks), dim3(256), 0, st, x, dy, rows, ci, co, qpb, partial)
    switch (cot * 8 + cit) {
        case 1 * 8 + 1: S2D_RW(1, 1); break;
        case 2 * 8 + 1: S2D_RW(2, 1); break;
        case 4 * 8 + 1: S2D_RW(4, 1); break;
        case 2 * 8 + 2: S2D_RW(2, 2); break;
        case 4 * 8 + 4: S2D_RW(4, 4); break;
        default:   // round up to the nearest instantiated shape
            if (cit == 1) S2D_RW(4, 1); else S2D_RW(4, 4);
    }
#undef S2D_RW
    S2D_LAUNCH_CHECK();
    hipLaunchKernelGGL(rows_wgrad_reduce_kernel, dim3((unsigned)ceil_div(co * ci, 256)), dim3(256), 0, st, (const float *)partial, blocks, co * ci,
                       dweight);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
