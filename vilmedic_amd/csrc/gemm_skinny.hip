// gemm_skinny.hip -- C[M,N] = epi(A[M,K] . B[N,K]^T) for the decode step's M = batch x beams <= 128 rows.
//
// The decode step is a chain of ~100 such GEMMs per token; with a 128x128 tile each of them puts 6..24 workgroups on a
// 256-CU chip and runs at per-kernel latency.  Here a workgroup owns 16 output columns for ALL rows, its 4 waves split the
// contraction four ways (each wave: its K-quarter of all rows x 16 columns, operands straight from global/L2 into MFMA
// fragments -- no LDS staging: A is re-read by every workgroup from L2, B is read exactly once overall), the partial
// accumulators are summed through LDS, and the epilogue (alpha, bias, erf-GELU, residual, bf16 / fp32 store) is applied
// once.  N/16 workgroups of short K-loops instead of N/128 long ones: 48..1908 workgroups for the decoder's shapes.
// MFMA operands are swapped (D^T = W_frag x X_frag) so a lane ends up with 4 consecutive output columns of one row.
#include "common.h"
#include "gemm_args.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int MF>   // MF = number of 16-row fragments (M <= 16 * MF)
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // [4 waves][MF][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16;
    const int ksteps = p.K >> 5;                                          // 32-deep MFMA steps
    const int per = (ksteps + 3) >> 2;
    const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
    const bf16_t* brow = p.B + (int64_t)min(n0 + c, p.N - 1) * p.ldb + g * 8;   // W row n0+c (clamped), this lane's k-octet
    const bf16_t* arow[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) arow[i] = p.A + (int64_t)min(16 * i + c, p.M - 1) * p.lda + g * 8;
    float4_t acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    // software pipeline: the fragments of step ks+1 are requested before the MFMAs of step ks
    bf16x8_t bf, af[MF], bn, an[MF];
    auto load = [&](int ks, bf16x8_t& b_, bf16x8_t (&a_)[MF]) {
        b_ = *reinterpret_cast<const bf16x8_t*>(brow + ks * 32);
#pragma unroll
        for (int i = 0; i < MF; ++i) a_[i] = *reinterpret_cast<const bf16x8_t*>(arow[i] + ks * 32);
    };
    if (ks0 < ks1) load(ks0, bf, af);
    for (int ks = ks0; ks < ks1; ++ks) {
        if (ks + 1 < ks1) load(ks + 1, bn, an);
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af[i], acc[i], 0, 0, 0);
        bf = bn;
#pragma unroll
        for (int i = 0; i < MF; ++i) af[i] = an[i];
    }
    // D^T layout: lane (c, g) of fragment i holds row m = 16 i + c, columns n0 + 4 g .. + 3
    float4_t* mine = reinterpret_cast<float4_t*>(red) + (wave * MF) * 64 + lane;
#pragma unroll
    for (int i = 0; i < MF; ++i) mine[i * 64] = acc[i];
    __syncthreads();
    const vm_gemm_epilogue& e = p.e;
    const float alpha = e.alpha_dev ? e.alpha * (*e.alpha_dev) : e.alpha;
    const int gn = n0 + 4 * g;
    for (int i = wave; i < MF; i += 4) {                                  // wave w finishes fragments w, w+4, ...
        const int gm = 16 * i + c;
        float4_t s = reinterpret_cast<const float4_t*>(red)[(0 * MF + i) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
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

template <int MF>
static int launch_skinny(const GemmArgs& a, hipStream_t s) {
    const size_t lds = (size_t)4 * MF * 64 * sizeof(float4_t);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<MF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_skinny_kernel<MF>), dim3((a.N + 15) / 16), dim3(256), lds, s, a);
    return vm_check_launch("vm_gemm_bf16(skinny)");
}

// eligibility is decided by the caller (gemm.hip): row-major A and B (K contiguous), K % 32 == 0, M <= 128, no split-K,
// no z side output / gelu' multiply / dropout
int vm_gemm_skinny_dispatch(const GemmArgs& a, hipStream_t s) {
    const int mf = (a.M + 15) / 16;
    if (mf <= 1) return launch_skinny<1>(a, s);
    if (mf <= 2) return launch_skinny<2>(a, s);
    if (mf <= 4) return launch_skinny<4>(a, s);
    return launch_skinny<8>(a, s);       // M <= 128 (measured: at M = 256 the 128x128-tile kernel is faster, 3.0 vs 3.3 ms per beam-4 step)
}
