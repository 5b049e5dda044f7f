// contrastive.hip -- pieces of the image-text contrastive losses (ConVIRT / InfoNCE / GLoRIA-global) around the
// [B,B] similarity GEMM (which runs on the tuned bf16 MFMA path with alpha = 1/tau):
//   vm_rownorm_cast      x fp32 [R,D] -> x/max(|x|,eps) as bf16 (+ the norms), or a plain cast (InfoNCE)
//   vm_lse_rows/cols_f32 log-sum-exp of every row / column of S fp32 [R,C]
//   vm_contrastive_grad  G = g_row_i softmax_row(S)_ij + g_col_j softmax_col(S)_ij - [i==j](g_row_i+g_col_i)  -> bf16
// ref: vilmedic/blocks/losses/selfsup/ConVIRTLoss.py:12-31, InfoNCELoss.py:11-19, GLoRIALoss.py:54-75.
#include "common.h"

__global__ __launch_bounds__(256) void rownorm_cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, float* __restrict__ norms,
                                                           int rows, int D, int normalize, float eps) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const float* xr = x + (int64_t)row * D;
        float ss = 0.f;
        for (int c = lane; c < D; c += 64) ss += xr[c] * xr[c];
        const float nrm = sqrtf(wave_sum(ss));
        const float inv = normalize ? 1.0f / fmaxf(nrm, eps) : 1.0f;
        for (int c = lane; c < D; c += 64) out[(int64_t)row * D + c] = f32_to_bf16(xr[c] * inv);
        if (lane == 0 && norms) norms[row] = nrm;
    }
}
extern "C" int vm_rownorm_cast(const float* x, void* out_bf16, float* norms, int rows, int D, int normalize, float eps, void* stream) {
    VM_REQUIRE(x && out_bf16 && rows > 0 && D > 0, "vm_rownorm_cast: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 6.0 * rows * (double)D, s);
    int blocks = (rows + 3) / 4; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rownorm_cast_kernel, dim3(blocks), dim3(256), 0, s, x, (bf16_t*)out_bf16, norms, rows, D, normalize, eps);
    return vm_check_launch("vm_rownorm_cast");
}

__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ S, int64_t ld, float* __restrict__ lse, float* __restrict__ diag,
                                                       int rows, int cols, int diag_offset) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const float* r = S + (int64_t)row * ld;
        float mx = -INFINITY;
        for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, r[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < cols; c += 64) se += expf(r[c] - mx);
        se = wave_sum(se);
        if (lane == 0) {
            lse[row] = mx + logf(se);
            if (diag) diag[row] = r[row + diag_offset];
        }
    }
}
// one block = 64 columns x 4 row-groups, online (max,sum) per thread, merged through LDS
__global__ __launch_bounds__(256) void lse_cols_kernel(const float* __restrict__ S, int64_t ld, float* __restrict__ lse, int rows, int cols) {
    __shared__ float sm[4][64], sl[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float m = -INFINITY, l = 0.f;
    if (c < cols) {
        for (int r = ty; r < rows; r += 4) {
            const float v = S[(int64_t)r * ld + c];
            const float mn = fmaxf(m, v);
            l = l * expf(m - mn) + expf(v - mn);
            m = mn;
        }
    }
    sm[ty][tx] = m; sl[ty][tx] = l;
    __syncthreads();
    if (ty == 0 && c < cols) {
        float M = sm[0][tx];
        for (int k = 1; k < 4; ++k) M = fmaxf(M, sm[k][tx]);
        float L = 0.f;
        for (int k = 0; k < 4; ++k) L += sl[k][tx] * expf(sm[k][tx] - M);
        lse[c] = M + logf(L);
    }
}
extern "C" int vm_lse_rows_f32(const float* S, int64_t ld, float* lse, float* diag, int rows, int cols, int diag_offset, void* stream) {
    VM_REQUIRE(S && lse && rows > 0 && cols > 0 && ld >= cols, "vm_lse_rows_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 8.0 * rows * (double)cols, s);
    int blocks = (rows + 3) / 4; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lse_rows_kernel, dim3(blocks), dim3(256), 0, s, S, ld, lse, diag, rows, cols, diag_offset);
    return vm_check_launch("vm_lse_rows_f32");
}
extern "C" int vm_lse_cols_f32(const float* S, int64_t ld, float* lse, int rows, int cols, void* stream) {
    VM_REQUIRE(S && lse && rows > 0 && cols > 0 && ld >= cols, "vm_lse_cols_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 4.0 * rows * (double)cols, s);
    hipLaunchKernelGGL(lse_cols_kernel, dim3((cols + 63) / 64), dim3(256), 0, s, S, ld, lse, rows, cols);
    return vm_check_launch("vm_lse_cols_f32");
}

__global__ __launch_bounds__(256) void contrastive_grad_kernel(const float* __restrict__ S, int64_t ld, const float* __restrict__ lse_r,
                                                               const float* __restrict__ lse_c, const float* __restrict__ g_r,
                                                               const float* __restrict__ g_c, bf16_t* __restrict__ G, int64_t ldg,
                                                               int rows, int cols, int diag_offset) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        const float s = S[(int64_t)r * ld + c];
        float g = g_r[r] * expf(s - lse_r[r]) + g_c[c] * expf(s - lse_c[c]);
        if (c == r + diag_offset) g -= g_r[r] + g_c[c];
        G[(int64_t)r * ldg + c] = f32_to_bf16(g);
    }
}
extern "C" int vm_contrastive_grad(const float* S, int64_t ld, const float* lse_rows, const float* lse_cols, const float* g_rows,
                                   const float* g_cols, void* G_bf16, int64_t ldg, int rows, int cols, int diag_offset, void* stream) {
    VM_REQUIRE(S && lse_rows && lse_cols && g_rows && g_cols && G_bf16 && rows > 0 && cols > 0, "vm_contrastive_grad: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 6.0 * rows * (double)cols, s);
    int64_t blocks = ((int64_t)rows * cols + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(contrastive_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, S, ld, lse_rows, lse_cols, g_rows, g_cols,
                       (bf16_t*)G_bf16, ldg, rows, cols, diag_offset);
    return vm_check_launch("vm_contrastive_grad");
}
