// gemm_skinny.hip -- C[M,N] = epi(A[M,K] . B[N,K]^T) for the decode step's M = batch x beams <= 128 rows.
//
// The decode step is a chain of ~100 such GEMMs per token; with a 128x128 tile each of them puts 6..24 workgroups on a
// 256-CU chip and runs at per-kernel latency.  Here a workgroup owns 16 output columns for ALL rows, its 8 waves split the
// contraction eight ways (each wave: its K-slice of all rows x 16 columns, operands straight from global/L2 into MFMA
// fragments -- no LDS staging: A is re-read by every workgroup from L2, B is read exactly once overall), the partial
// accumulators are summed through LDS, and the epilogue (alpha, bias, erf-GELU, residual, bf16 / fp32 store) is applied
// once.  N/16 workgroups of short K-loops instead of N/128 long ones: 48..1908 workgroups for the decoder's shapes.
// MFMA operands are swapped (D^T = W_frag x X_frag) so a lane ends up with 4 consecutive output columns of one row.
#include "common.h"
#include "gemm_args.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#define SK_NW 8        // waves per workgroup: the contraction is split 8 ways
#define SK_D 4         // register ring: operand fragments of 4 k-steps in flight per wave

// sum of the SK_NW waves' partial fragments (in wave order) + epilogue; wave w finishes fragments w, w + SK_NW, ...
template <int MF>
__device__ __forceinline__ void skinny_finish(const GemmArgs& p, const float* red, int wave, int lane, int m0, int n0) {
    const int g = lane >> 4, c = lane & 15;
    const vm_gemm_epilogue& e = p.e;
    const float alpha = e.alpha_dev ? e.alpha * (*e.alpha_dev) : e.alpha;
    const int gn = n0 + 4 * g;
    for (int i = wave; i < MF; i += SK_NW) {                              // wave w finishes fragments w, w + SK_NW, ...
        const int gm = m0 + 16 * i + c;
        float4_t s = reinterpret_cast<const float4_t*>(red)[(0 * MF + i) * 64 + lane];
#pragma unroll
        for (int w = 1; w < SK_NW; ++w) {
            const float4_t t = reinterpret_cast<const float4_t*>(red)[(w * MF + i) * 64 + lane];
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        if (gm >= p.M || gn >= p.N) continue;
        float v[4] = {s[0] * alpha, s[1] * alpha, s[2] * alpha, s[3] * alpha};
        const int nvalid = min(4, p.N - gn);
        if (e.bias) for (int r = 0; r < nvalid; ++r) v[r] += e.bias[gn + r];
        if (e.act == 1) for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
        if (e.residual) {
            const bf16_t* rp = reinterpret_cast<const bf16_t*>(e.residual) + (int64_t)gm * e.ldr + gn;
            for (int r = 0; r < nvalid; ++r) v[r] += bf16_to_f32(rp[r]);
        }
        const int64_t off = (int64_t)gm * p.ldc + gn;
        if (e.out_dtype == VM_BF16) {
            bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + off;
            if (nvalid == 4 && (p.ldc & 3) == 0) {
                uint2 u; u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(cp) = u;
            } else for (int r = 0; r < nvalid; ++r) cp[r] = f32_to_bf16(v[r]);
        } else {
            float* cp = reinterpret_cast<float*>(p.C) + off;
            if (e.accumulate) for (int r = 0; r < nvalid; ++r) cp[r] += v[r];
            else if (nvalid == 4 && (p.ldc & 3) == 0) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int r = 0; r < nvalid; ++r) cp[r] = v[r];
        }
    }
}

template <int MF>   // MF = number of 16-row fragments (M <= 16 * MF)
__global__ __launch_bounds__(SK_NW * 64) void gemm_skinny_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // [SK_NW waves][MF][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MF);
    const int ksteps = p.K >> 5;                                          // 32-deep MFMA steps
    const int per = (ksteps + SK_NW - 1) / SK_NW;
    const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
    const bf16_t* brow = p.B + (int64_t)min(n0 + c, p.N - 1) * p.ldb + g * 8;   // W row n0+c (clamped), this lane's k-octet
    const bf16_t* arow[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) arow[i] = p.A + (int64_t)min(m0 + 16 * i + c, p.M - 1) * p.lda + g * 8;
    float4_t acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    // The kernel is pure latency (a 768 x 768 weight is 1.2 MB: nothing to stream): with one k-step requested ahead, a wave paid one
    // memory round trip per step (6 in a row at K = 768).  Now 8 waves split K and each keeps SK_D steps' fragments in flight in a
    // register ring, so K = 768 (3 steps per wave) costs ONE round trip and K = 3072 (12 steps) three.
    bf16x8_t bq[SK_D], aq[SK_D][MF];
    auto load = [&](int ks, bf16x8_t& b_, bf16x8_t (&a_)[MF]) {
        b_ = *reinterpret_cast<const bf16x8_t*>(brow + ks * 32);
#pragma unroll
        for (int i = 0; i < MF; ++i) a_[i] = *reinterpret_cast<const bf16x8_t*>(arow[i] + ks * 32);
    };
#pragma unroll
    for (int d = 0; d < SK_D; ++d) if (ks0 + d < ks1) load(ks0 + d, bq[d], aq[d]);
    for (int ks = ks0; ks < ks1; ks += SK_D) {
#pragma unroll
        for (int d = 0; d < SK_D; ++d) {
            if (ks + d < ks1) {
#pragma unroll
                for (int i = 0; i < MF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[d], aq[d][i], acc[i], 0, 0, 0);
                if (ks + d + SK_D < ks1) load(ks + d + SK_D, bq[d], aq[d]);
            }
        }
    }
    // D^T layout: lane (c, g) of fragment i holds row m = m0 + 16 i + c, columns n0 + 4 g .. + 3
    float4_t* mine = reinterpret_cast<float4_t*>(red) + (wave * MF) * 64 + lane;
#pragma unroll
    for (int i = 0; i < MF; ++i) mine[i * 64] = acc[i];
    __syncthreads();
    skinny_finish<MF>(p, red, wave, lane, m0, n0);
}

// The wide-output form (LM head: N = 30522 columns of K = 768): 1908 column blocks, and with one workgroup per block the 64-row operand
// (98 KB) is re-read from L2 1908 times (the weights once).  Here a workgroup walks column blocks blockIdx.x, + gridDim.x, ... with its
// waves' row fragments RESIDENT in registers (K <= 768: 3 k-steps x 4 fragments per wave), the next block's weight fragments requested
// before the current block's reduction, and the LDS partials double-buffered (one barrier per block).  Same K split, same reduction
// order: bit-identical to gemm_skinny_kernel.
#define SKC_STEPS 3
__global__ __launch_bounds__(SK_NW * 64) void gemm_skinny_cols_kernel(const GemmArgs p) {
    constexpr int MF = 4;
    extern __shared__ __attribute__((aligned(16))) float red[];          // 2 x [SK_NW waves][MF][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * (16 * MF);
    const int ksteps = p.K >> 5, per = (ksteps + SK_NW - 1) / SK_NW;
    const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
    const int col_blocks = (p.N + 15) >> 4;
    bf16x8_t aq[SKC_STEPS][MF], bq[SKC_STEPS], bn[SKC_STEPS];
#pragma unroll
    for (int d = 0; d < SKC_STEPS; ++d)
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const bf16_t* arow = p.A + (int64_t)min(m0 + 16 * i + c, p.M - 1) * p.lda + g * 8;
            if (ks0 + d < ks1) aq[d][i] = *reinterpret_cast<const bf16x8_t*>(arow + (ks0 + d) * 32);
        }
    auto load_b = [&](int cb, bf16x8_t (&b_)[SKC_STEPS]) {
        const bf16_t* brow = p.B + (int64_t)min(cb * 16 + c, p.N - 1) * p.ldb + g * 8;
#pragma unroll
        for (int d = 0; d < SKC_STEPS; ++d) if (ks0 + d < ks1) b_[d] = *reinterpret_cast<const bf16x8_t*>(brow + (ks0 + d) * 32);
    };
    int cb = blockIdx.x, it = 0;
    if (cb < col_blocks) load_b(cb, bq);
    for (; cb < col_blocks; cb += gridDim.x, ++it) {
        if (cb + (int)gridDim.x < col_blocks) load_b(cb + gridDim.x, bn);
        float4_t acc[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < SKC_STEPS; ++d)
            if (ks0 + d < ks1) {
#pragma unroll
                for (int i = 0; i < MF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[d], aq[d][i], acc[i], 0, 0, 0);
            }
        float* buf = red + (it & 1) * (SK_NW * MF * 64 * 4);
        float4_t* mine = reinterpret_cast<float4_t*>(buf) + (wave * MF) * 64 + lane;
#pragma unroll
        for (int i = 0; i < MF; ++i) mine[i * 64] = acc[i];
        __syncthreads();           // (a wave reaches the next block's barrier only after it finished reading this buffer: two buffers suffice)
        skinny_finish<MF>(p, buf, wave, lane, m0, cb * 16);
#pragma unroll
        for (int d = 0; d < SKC_STEPS; ++d) bq[d] = bn[d];
    }
}

template <int MF>
static int launch_skinny(const GemmArgs& a, hipStream_t s) {
    const size_t lds = (size_t)SK_NW * MF * 64 * sizeof(float4_t);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<MF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_skinny_kernel<MF>), dim3((a.N + 15) / 16, (a.M + 16 * MF - 1) / (16 * MF)), dim3(SK_NW * 64), lds, s, a);
    return vm_check_launch("vm_gemm_bf16(skinny)");
}

// Rows per workgroup: the tallest block of 16-row fragments that still leaves one workgroup per CU -- (N / 16) x ceil(M / (16 MF)) >=
// 256 -- so that a 768-column projection of 64 rows runs on 192 CUs (MF = 1) instead of 48 (MF = 4); the weights are re-read from L2 by
// the row blocks (a 768 x 768 matrix is 1.2 MB).
int vm_skinny_rows_per_wg(int M, int N, int max_mf) {
    const int col_blocks = (N + 15) / 16, mf_all = (M + 15) / 16;
    int mf = 1;
    for (int cand = 2; cand <= max_mf; cand *= 2)
        if (cand / 2 < mf_all && (int64_t)col_blocks * ((mf_all + cand - 1) / cand) >= 256) mf = cand;
    return mf;
}

// eligibility is decided by the caller (gemm.hip): row-major A and B (K contiguous), K % 32 == 0, M <= 256, no split-K,
// no z side output / gelu' multiply / dropout
int vm_gemm_skinny_dispatch(const GemmArgs& a, hipStream_t s) {
    if (a.N >= 4096 && a.K <= 32 * SK_NW * SKC_STEPS) {          // wide output, short contraction: resident row fragments
        const int row_blocks = (a.M + 63) / 64, col_blocks = (a.N + 15) / 16;
        int gx = 512 / row_blocks; if (gx > col_blocks) gx = col_blocks; if (gx < 1) gx = 1;
        const size_t lds = (size_t)2 * SK_NW * 4 * 64 * sizeof(float4_t);
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_skinny_cols_kernel, dim3(gx, row_blocks), dim3(SK_NW * 64), lds, s, a);
        return vm_check_launch("vm_gemm_bf16(skinny, column walk)");
    }
    switch (vm_skinny_rows_per_wg(a.M, a.N, 8)) {
        case 1: return launch_skinny<1>(a, s);
        case 2: return launch_skinny<2>(a, s);
        case 4: return launch_skinny<4>(a, s);
        default: return launch_skinny<8>(a, s);
    }
}
