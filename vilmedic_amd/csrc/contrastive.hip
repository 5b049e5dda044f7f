// contrastive.hip -- the image-text contrastive losses (ConVIRT / InfoNCE / GLoRIA-global) on the [B,B] similarity S = a_hat b_hat^T / tau:
//   vm_rownorm_cast      x fp32 [R,D] -> x/max(|x|,eps) as bf16 (+ the norms), or a plain cast (InfoNCE)
//   vm_contrastive_fwd   row / column log-sum-exp and the diagonal of S, tile by tile on the MFMA (S never reaches HBM)
//   vm_contrastive_bwd   G = g_row_i softmax_row(S)_ij + g_col_j softmax_col(S)_ij - [i==j](g_row_i+g_col_i)  -> bf16, from recomputed tiles
// (round 1 materialised S in fp32 and made three scalar passes over it: 0.314 vs 0.147 ms of kernel time at B = 2048, 1.45 vs 0.80 ms
// at B = 8192 -- profiles/r02_l_contrastive_bench.txt -- that path and its kernels are gone.)
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



// ------------------------------------------------------------------ fused similarity tiles: S is never written to HBM
// One workgroup owns a 128 x 128 tile of S = A_hat B_hat^T * inv_tau (bf16 MFMA, fp32 accumulate; operands are the normalised
// embeddings, 2 x R x D bf16 = 6 MB at R = 2048, D = 768: L2 / MALL resident, so they go global -> registers -> LDS with a
// one-tile software pipeline), keeps the tile in LDS as fp32 and turns it into
//   MODE 0 (forward):  per-row and per-column (max, sum exp) PARTIALS over the tile + the paired-diagonal entries; a second tiny
//                      kernel merges the partials of a row / column across tiles into the log-sum-exps (online-softmax merge);
//   MODE 1 (backward): G = g_row_i softmax_row(S)_ij + g_col_j softmax_col(S)_ij - [j == i + off] (g_row_i + g_col_j) as bf16,
//                      from a RECOMPUTED tile (one more pass of the 2 R C D product instead of a 4 R C byte read of a stored S).
// Versus the unfused path (S fp32 written once and read by three scalar kernels): at R = C = 2048 the 16.8 MB matrix and its
// three re-reads disappear; HBM traffic of the loss is the 6 MB of embeddings plus G (8 MB, bf16) for the two gradient GEMMs.
typedef __bf16 c_bf16x8_t __attribute__((ext_vector_type(8)));
#define CT 128
#define CT_KS 72          // LDS row stride in elements for a [128][64] operand slab (144 B: conflict-spreading pad)
#define CT_CS 132         // fp32 row stride of the staged S tile
#define CT_LDS (2 * 2 * CT * CT_KS * 2)      // two stages x two operands = 73728 B >= 128 * 132 * 4 = 67584 B
struct ContrArgs {
    const bf16_t* A; const bf16_t* B; int R, C, D, diag_offset, tiles_m, tiles_n; float inv_tau;
    float* row_part; float* col_part; float* diag;                                   // MODE 0 outputs
    const float* lse_r; const float* lse_c; const float* g_r; const float* g_c; bf16_t* G; int64_t ldg;   // MODE 1
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void contrastive_tile_kernel(const ContrArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x - tm * p.tiles_n;
    const int m0 = tm * CT, n0 = tn * CT;
    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    uint4 ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            const int ga = m0 + row, gb = n0 + row, gk = k0 + ch * 8;
            ra[i] = (ga < p.R && gk < p.D) ? *reinterpret_cast<const uint4*>(p.A + (int64_t)ga * p.D + gk) : make_uint4(0, 0, 0, 0);
            rb[i] = (gb < p.C && gk < p.D) ? *reinterpret_cast<const uint4*>(p.B + (int64_t)gb * p.D + gk) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store = [&](char* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            *reinterpret_cast<uint4*>(st + row * (CT_KS * 2) + ch * 16) = ra[i];
            *reinterpret_cast<uint4*>(st + CT * CT_KS * 2 + row * (CT_KS * 2) + ch * 16) = rb[i];
        }
    };
    const int ktiles = (p.D + 63) / 64;
    load(0);
    store(smem);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const char* sa = smem + (kt & 1) * (2 * CT * CT_KS * 2);
        const char* sb = sa + CT * CT_KS * 2;
        if (kt + 1 < ktiles) load((kt + 1) * 64);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            c_bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const c_bf16x8_t*>(sa + (wm * 64 + i * 16 + (lane & 15)) * (CT_KS * 2) + (kk * 32 + (lane >> 4) * 8) * 2);
                fb[i] = *reinterpret_cast<const c_bf16x8_t*>(sb + (wn * 64 + i * 16 + (lane & 15)) * (CT_KS * 2) + (kk * 32 + (lane >> 4) * 8) * 2);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < ktiles) store(smem + ((kt + 1) & 1) * (2 * CT * CT_KS * 2));
        __syncthreads();
    }
    // stage S tile (fp32, scaled): C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg
    float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = wn * 64 + j * 16 + (lane & 15), row = wm * 64 + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[(row + r) * CT_CS + col] = acc[i][j][r] * p.inv_tau;
        }
    __syncthreads();
    const int rows = min(CT, p.R - m0), cols = min(CT, p.C - n0);
    if (MODE == 0) {
        if (tid < CT) {                        // threads 0..127: one row each
            const int r = tid;
            if (r < rows) {
                float mx = -INFINITY;
                for (int j = 0; j < cols; ++j) mx = fmaxf(mx, cs[r * CT_CS + j]);
                float se = 0.f;
                for (int j = 0; j < cols; ++j) se += expf(cs[r * CT_CS + j] - mx);
                float* o = p.row_part + ((int64_t)tn * p.R + m0 + r) * 2;
                o[0] = mx; o[1] = se;
                const int dj = m0 + r + p.diag_offset - n0;             // column of the paired entry inside this tile
                if (p.diag && dj >= 0 && dj < cols) p.diag[m0 + r] = cs[r * CT_CS + dj];
            }
        } else {                               // threads 128..255: one column each
            const int c = tid - CT;
            if (c < cols) {
                float mx = -INFINITY;
                for (int i = 0; i < rows; ++i) mx = fmaxf(mx, cs[i * CT_CS + c]);
                float se = 0.f;
                for (int i = 0; i < rows; ++i) se += expf(cs[i * CT_CS + c] - mx);
                float* o = p.col_part + ((int64_t)tm * p.C + n0 + c) * 2;
                o[0] = mx; o[1] = se;
            }
        }
    } else {
        for (int it = 0; it < 8; ++it) {       // 8 consecutive columns of a row per thread: 16-B bf16 stores, whole 256-B rows per 16 lanes
            const int id = tid + 256 * it, r = id >> 4, c0 = (id & 15) * 8;
            if (r >= rows || c0 >= cols) continue;
            const int gr = m0 + r;
            const float gri = p.g_r[gr], lri = p.lse_r[gr];
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gc = n0 + c0 + j;
                float g = 0.f;
                if (c0 + j < cols) {
                    const float sv = cs[r * CT_CS + c0 + j];
                    const float gcj = p.g_c[gc];
                    g = gri * expf(sv - lri) + gcj * expf(sv - p.lse_c[gc]);
                    if (gc == gr + p.diag_offset) g -= gri + gcj;
                }
                v[j] = g;
            }
            bf16_t* o = p.G + (int64_t)gr * p.ldg + n0 + c0;
            if (c0 + 8 <= cols) *reinterpret_cast<uint4*>(o) = pack8(v);
            else for (int j = 0; j < cols - c0; ++j) o[j] = f32_to_bf16(v[j]);
        }
    }
}

// merge the per-tile (max, sum exp) partials of every row and every column: lse = M + log(sum_t s_t exp(m_t - M))
__global__ __launch_bounds__(256) void contrastive_merge_kernel(const float* __restrict__ row_part, const float* __restrict__ col_part,
                                                                float* __restrict__ lse_r, float* __restrict__ lse_c, int R, int C,
                                                                int tiles_m, int tiles_n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < R) {
        float M = -INFINITY;
        for (int t = 0; t < tiles_n; ++t) M = fmaxf(M, row_part[((int64_t)t * R + i) * 2]);
        float L = 0.f;
        for (int t = 0; t < tiles_n; ++t) L += row_part[((int64_t)t * R + i) * 2 + 1] * expf(row_part[((int64_t)t * R + i) * 2] - M);
        lse_r[i] = M + logf(L);
    } else if (i - R < C) {
        const int j = i - R;
        float M = -INFINITY;
        for (int t = 0; t < tiles_m; ++t) M = fmaxf(M, col_part[((int64_t)t * C + j) * 2]);
        float L = 0.f;
        for (int t = 0; t < tiles_m; ++t) L += col_part[((int64_t)t * C + j) * 2 + 1] * expf(col_part[((int64_t)t * C + j) * 2] - M);
        lse_c[j] = M + logf(L);
    }
}

static int contr_common(const char* fn, const void* a, const void* b, int R, int C, int D) {
    VM_REQUIRE(a && b && R > 0 && C > 0 && D > 0 && (D % 8) == 0, "%s: bad arguments (R=%d C=%d D=%d, D must be a multiple of 8)", fn, R, C, D);
    VM_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0, "%s: operands must be 16-byte aligned", fn);
    return VM_OK;
}
static void contr_attr() {
    static bool set = false;
    if (set) return;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&contrastive_tile_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, CT_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&contrastive_tile_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, CT_LDS);
    set = true;
}

extern "C" size_t vm_contrastive_ws(int R, int C) {
    const size_t tm = (R + CT - 1) / CT, tn = (C + CT - 1) / CT;
    return (tn * (size_t)R + tm * (size_t)C) * 2 * sizeof(float);
}

extern "C" int vm_contrastive_fwd(const void* a_hat, const void* b_hat, int R, int C, int D, float inv_tau, int diag_offset,
                                  float* lse_rows, float* lse_cols, float* diag, void* ws, size_t ws_bytes, void* stream) {
    int rc = contr_common("vm_contrastive_fwd", a_hat, b_hat, R, C, D);
    if (rc) return rc;
    VM_REQUIRE(lse_rows && lse_cols && ws && ws_bytes >= vm_contrastive_ws(R, C), "vm_contrastive_fwd: outputs / workspace (%zu bytes needed)", vm_contrastive_ws(R, C));
    ContrArgs p = {};
    p.A = (const bf16_t*)a_hat; p.B = (const bf16_t*)b_hat; p.R = R; p.C = C; p.D = D; p.diag_offset = diag_offset; p.inv_tau = inv_tau;
    p.tiles_m = (R + CT - 1) / CT; p.tiles_n = (C + CT - 1) / CT;
    p.row_part = (float*)ws; p.col_part = p.row_part + (size_t)p.tiles_n * R * 2; p.diag = diag;
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 2.0 * R * (double)C * D, s, "contrastive_fwd_R%d_C%d_D%d", R, C, D);
    contr_attr();
    hipLaunchKernelGGL(contrastive_tile_kernel<0>, dim3(p.tiles_m * p.tiles_n), dim3(256), CT_LDS, s, p);
    hipLaunchKernelGGL(contrastive_merge_kernel, dim3((R + C + 255) / 256), dim3(256), 0, s, p.row_part, p.col_part, lse_rows, lse_cols, R, C,
                       p.tiles_m, p.tiles_n);
    return vm_check_launch("vm_contrastive_fwd");
}

extern "C" int vm_contrastive_bwd(const void* a_hat, const void* b_hat, int R, int C, int D, float inv_tau, int diag_offset,
                                  const float* lse_rows, const float* lse_cols, const float* g_rows, const float* g_cols,
                                  void* G_bf16, int64_t ldg, void* stream) {
    int rc = contr_common("vm_contrastive_bwd", a_hat, b_hat, R, C, D);
    if (rc) return rc;
    VM_REQUIRE(lse_rows && lse_cols && g_rows && g_cols && G_bf16 && ldg >= C && (ldg % 8) == 0, "vm_contrastive_bwd: bad arguments");
    ContrArgs p = {};
    p.A = (const bf16_t*)a_hat; p.B = (const bf16_t*)b_hat; p.R = R; p.C = C; p.D = D; p.diag_offset = diag_offset; p.inv_tau = inv_tau;
    p.tiles_m = (R + CT - 1) / CT; p.tiles_n = (C + CT - 1) / CT;
    p.lse_r = lse_rows; p.lse_c = lse_cols; p.g_r = g_rows; p.g_c = g_cols; p.G = (bf16_t*)G_bf16; p.ldg = ldg;
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 2.0 * R * (double)C * D, s, "contrastive_bwd_R%d_C%d_D%d", R, C, D);
    contr_attr();
    hipLaunchKernelGGL(contrastive_tile_kernel<1>, dim3(p.tiles_m * p.tiles_n), dim3(256), CT_LDS, s, p);
    return vm_check_launch("vm_contrastive_bwd");
}
