// decode_gemm.hip -- the projections of ONE decode step (M = batch x beams <= 256 rows), bf16 and exact-fp32, with everything that used
// to be a launch of its own around them folded in (round 3: a decoder layer is 8 launches instead of 12):
//   * LayerNorm on load: the post-LN sum s of the previous sub-layer is the A operand; every workgroup reads its whole row block of s
//     anyway (it owns 16 output columns for a block of rows), so it computes the rows' mean / rstd itself (two passes, fp32), normalises
//     the MFMA fragments in registers, and the workgroups of column block 0 also write x = LN(s) -- the residual of the sub-layer that
//     follows.  No LayerNorm kernel, no extra pass over s.
//   * two destinations: the fused Q|K|V projection writes Q to its own buffer and K|V of the new token straight into the cache row
//     (columns >= split_n go to c2 with their own leading dimension) -- one launch instead of two.
//   * bias and residual of the fragments a wave will finish are requested BEFORE the contraction (the kernel is pure latency: one more
//     dependent memory round trip after the LDS reduction was ~1 us of a ~7 us launch).
// Structure as gemm_skinny.hip / decode_f32.hip: a workgroup owns 16 output columns for a block of 16 MF rows, its 8 waves split the
// contraction eight ways with operands straight from global / L2 into MFMA fragments, partial accumulators meet in LDS in a fixed order
// (deterministic).  bf16: v_mfma_f32_16x16x32_bf16; fp32: v_mfma_f32_16x16x4_f32 (exact f32 products == an fmaf chain).
// Replaces, per decode step, hf:models/bert_generation/modeling_bert_generation.py:60-106,181-231,264-358 as reached from
// ref:vilmedic/blocks/huggingface/decoder/evaluation.py:73-78 and ref:vilmedic/blocks/rl/SCST.py:112-174.
#include "common.h"

typedef __bf16 dg_bf16x8_t __attribute__((ext_vector_type(8)));
#define DG_NW 8

struct DgArgs {
    const void* A; const void* W; void* C; void* c2; const float* bias; const void* residual;
    const float* ln_gamma; const float* ln_beta; void* ln_out;
    int64_t lda, ldw, ldc, ldc2, ldr, ln_out_ld;
    int M, N, K, act, split_n; float ln_eps;
};

template <bool F32> struct DgT;
template <> struct DgT<false> {          // bf16: a k-step is 32 wide, lane group g holds k = 32 ks + 8 g .. + 7 (16 B)
    typedef bf16_t elem; typedef dg_bf16x8_t frag;
    static constexpr int STEP = 32, PER_LANE = 8, LN_STEPS = 4;       // LN mode: K <= 8 waves x 4 steps x 32 = 1024
    static __device__ __forceinline__ void unpack(const frag& f, float* v) { unpack8(__builtin_bit_cast(uint4, f), v); }
    static __device__ __forceinline__ frag pack(const float* v) { return __builtin_bit_cast(frag, pack8(v)); }
    static __device__ __forceinline__ void mma(const frag& w, const frag& a, float4_t& acc) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc, 0, 0, 0); }
    static __device__ __forceinline__ float ld1(const elem* p) { return bf16_to_f32(*p); }
};
template <> struct DgT<true> {           // fp32: a k-step is 16 wide (4 MFMAs), lane group g holds k = 16 ks + 4 g .. + 3 (16 B)
    typedef float elem; typedef float4 frag;
    static constexpr int STEP = 16, PER_LANE = 4, LN_STEPS = 8;
    static __device__ __forceinline__ void unpack(const frag& f, float* v) { v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
    static __device__ __forceinline__ frag pack(const float* v) { return make_float4(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ void mma(const frag& w, const frag& a, float4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a.w, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float ld1(const elem* p) { return *p; }
};
__device__ __forceinline__ float dg_gelu(float x, bool f32) { return f32 ? 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)) : gelu_f(x); }
__device__ __forceinline__ float dg_quarters_sum(float v) {      // over lanes l, l^16, l^32, l^48 (the 4 k-groups of a row)
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// LN = false: operands of D k-steps in flight per wave (register ring); LN = true: the wave's whole K slice of A resident (LN_STEPS)
template <bool F32, int MF, bool LN>
__global__ __launch_bounds__(DG_NW * 64) void decode_gemm_kernel(const DgArgs p) {
    typedef DgT<F32> T;
    typedef typename T::elem elem;
    typedef typename T::frag frag;
    extern __shared__ __attribute__((aligned(16))) float red[];          // [DG_NW][MF][64 lanes][4] accumulators; LN: [DG_NW][16 MF] row partials first
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MF);
    const int ksteps = p.K / T::STEP;
    const int per = (ksteps + DG_NW - 1) / DG_NW;
    const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
    const elem* wrow = reinterpret_cast<const elem*>(p.W) + (int64_t)min(n0 + c, p.N - 1) * p.ldw + g * T::PER_LANE;
    const elem* arow[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) arow[i] = reinterpret_cast<const elem*>(p.A) + (int64_t)min(m0 + 16 * i + c, p.M - 1) * p.lda + g * T::PER_LANE;
    // ---- epilogue operands of the fragments this wave finishes (i = wave, wave + 8), requested now
    constexpr int NFIN = (MF + DG_NW - 1) / DG_NW;
    const int gn = n0 + 4 * g;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    float res4[NFIN][4];
    if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[r] = p.bias[min(gn + r, p.N - 1)];
    }
#pragma unroll
    for (int f = 0; f < NFIN; ++f) {
        const int i = wave + DG_NW * f, gm = min(m0 + 16 * i + c, p.M - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            res4[f][r] = (p.residual && i < MF) ? T::ld1(reinterpret_cast<const elem*>(p.residual) + (int64_t)gm * p.ldr + min(gn + r, p.N - 1)) : 0.f;
    }
    float4_t acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    auto ldw = [&](int ks) { return *reinterpret_cast<const frag*>(wrow + ks * T::STEP); };
    auto lda = [&](int i, int ks) { return *reinterpret_cast<const frag*>(arow[i] + ks * T::STEP); };
    if constexpr (LN) {
        constexpr int S = T::LN_STEPS;
        frag af[S][MF], wf[S];
#pragma unroll
        for (int d = 0; d < S; ++d) {
            if (ks0 + d < ks1) {
                wf[d] = ldw(ks0 + d);
#pragma unroll
                for (int i = 0; i < MF; ++i) af[d][i] = lda(i, ks0 + d);
            }
        }
        // gamma / beta of the wave's K slice, requested together with the operands (one memory round trip for everything the kernel reads)
        float gam[S][T::PER_LANE], bet[S][T::PER_LANE];
#pragma unroll
        for (int d = 0; d < S; ++d) {
            if (ks0 + d < ks1) {
                const int k = (ks0 + d) * T::STEP + g * T::PER_LANE;
#pragma unroll
                for (int e = 0; e < T::PER_LANE; e += 4) {
                    const float4 gv = *reinterpret_cast<const float4*>(p.ln_gamma + k + e), bv = *reinterpret_cast<const float4*>(p.ln_beta + k + e);
                    gam[d][e] = gv.x; gam[d][e + 1] = gv.y; gam[d][e + 2] = gv.z; gam[d][e + 3] = gv.w;
                    bet[d][e] = bv.x; bet[d][e + 1] = bv.y; bet[d][e + 2] = bv.z; bet[d][e + 3] = bv.w;
                }
            }
        }
        // ---- LayerNorm statistics of the block's rows: pass 1 mean, pass 2 centred variance (both over all 8 waves through LDS)
        float mean[MF], rstd[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) { mean[i] = 0.f; rstd[i] = 0.f; }
        const float invk = 1.0f / (float)p.K;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < S; ++d) {
                    if (ks0 + d < ks1) {
                        float v[T::PER_LANE];
                        T::unpack(af[d][i], v);
#pragma unroll
                        for (int e = 0; e < T::PER_LANE; ++e) { const float t = pass ? v[e] - mean[i] : v[e]; s += pass ? t * t : t; }
                    }
                }
                s = dg_quarters_sum(s);
                if (g == 0) red[wave * (16 * MF) + 16 * i + c] = s;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < DG_NW; ++w) t += red[w * (16 * MF) + 16 * i + c];
                if (pass == 0) mean[i] = t * invk;
                else rstd[i] = rsqrtf(t * invk + p.ln_eps);
            }
            __syncthreads();
        }
        // ---- normalise the fragments in registers (bf16 mode rounds them like the LayerNorm kernel's bf16 output did), write x = LN(s)
#pragma unroll
        for (int d = 0; d < S; ++d) {
            if (ks0 + d < ks1) {
                const int k = (ks0 + d) * T::STEP + g * T::PER_LANE;
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    float v[T::PER_LANE];
                    T::unpack(af[d][i], v);
#pragma unroll
                    for (int e = 0; e < T::PER_LANE; ++e) v[e] = (v[e] - mean[i]) * rstd[i] * gam[d][e] + bet[d][e];
                    af[d][i] = T::pack(v);
                    const int gm = m0 + 16 * i + c;
                    if (p.ln_out && blockIdx.x == 0 && gm < p.M)
                        *reinterpret_cast<frag*>(reinterpret_cast<elem*>(p.ln_out) + (int64_t)gm * p.ln_out_ld + k) = af[d][i];
                }
            }
        }
#pragma unroll
        for (int d = 0; d < S; ++d) {
            if (ks0 + d < ks1) {
#pragma unroll
                for (int i = 0; i < MF; ++i) T::mma(wf[d], af[d][i], acc[i]);
            }
        }
    } else {
        constexpr int D = F32 ? (MF <= 4 ? 4 : MF <= 8 ? 2 : 1) : 4;
        frag wq[D], aq[D][MF];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (ks0 + d < ks1) {
                wq[d] = ldw(ks0 + d);
#pragma unroll
                for (int i = 0; i < MF; ++i) aq[d][i] = lda(i, ks0 + d);
            }
        }
        for (int ks = ks0; ks < ks1; ks += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (ks + d < ks1) {
#pragma unroll
                    for (int i = 0; i < MF; ++i) T::mma(wq[d], aq[d][i], acc[i]);
                    if (ks + d + D < ks1) {
                        wq[d] = ldw(ks + d + D);
#pragma unroll
                        for (int i = 0; i < MF; ++i) aq[d][i] = lda(i, ks + d + D);
                    }
                }
            }
        }
    }
    // D^T layout: lane (c, g) of fragment i holds row m0 + 16 i + c, columns n0 + 4 g .. + 3
    float4_t* mine = reinterpret_cast<float4_t*>(red) + (wave * MF) * 64 + lane;
#pragma unroll
    for (int i = 0; i < MF; ++i) mine[i * 64] = acc[i];
    __syncthreads();
#pragma unroll
    for (int f = 0; f < NFIN; ++f) {
        const int i = wave + DG_NW * f;
        if (i >= MF) continue;
        const int gm = m0 + 16 * i + c;
        float4_t s = reinterpret_cast<const float4_t*>(red)[(0 * MF + i) * 64 + lane];
#pragma unroll
        for (int w = 1; w < DG_NW; ++w) {                                 // fixed order: deterministic
            const float4_t t = reinterpret_cast<const float4_t*>(red)[(w * MF + i) * 64 + lane];
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        if (gm >= p.M || gn >= p.N) continue;
        float v[4] = {s[0] + bias4[0], s[1] + bias4[1], s[2] + bias4[2], s[3] + bias4[3]};
        if (p.act == 1) for (int r = 0; r < 4; ++r) v[r] = dg_gelu(v[r], F32);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += res4[f][r];
        const int nvalid = min(4, p.N - gn);
        const bool second = p.c2 && gn >= p.split_n;
        elem* cp = second ? reinterpret_cast<elem*>(p.c2) + (int64_t)gm * p.ldc2 + (gn - p.split_n) : reinterpret_cast<elem*>(p.C) + (int64_t)gm * p.ldc + gn;
        const bool vec = nvalid == 4 && (((second ? p.ldc2 : p.ldc) & 3) == 0);
        if constexpr (F32) {
            if (vec) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int r = 0; r < nvalid; ++r) cp[r] = v[r];
        } else {
            if (vec) { uint2 u; u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]); *reinterpret_cast<uint2*>(cp) = u; }
            else for (int r = 0; r < nvalid; ++r) cp[r] = f32_to_bf16(v[r]);
        }
    }
}

// The wide-output form (the fp32 LM head: N = 30522 columns of K = 768; gemm_skinny.hip has the bf16 one): a workgroup walks column
// blocks blockIdx.x, + gridDim.x, ... with the 64-row operand RESIDENT in its waves' registers (6 k-steps x 4 fragments per wave at
// K = 768) instead of being re-read from L2 by each of 1908 workgroups; next block's weight fragments requested before the current
// block's reduction; LDS partials double-buffered (one barrier per block).  Plain + bias only.  K split and reduction order are
// decode_gemm_kernel's: bit-identical.
#define DGC_STEPS 6
template <bool F32>
__global__ __launch_bounds__(DG_NW * 64) void decode_gemm_cols_kernel(const DgArgs p) {
    typedef DgT<F32> T;
    typedef typename T::elem elem;
    typedef typename T::frag frag;
    constexpr int MF = 4;
    extern __shared__ __attribute__((aligned(16))) float red[];          // 2 x [DG_NW][MF][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * (16 * MF);
    const int ksteps = p.K / T::STEP, per = (ksteps + DG_NW - 1) / DG_NW;
    const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
    const int col_blocks = (p.N + 15) >> 4;
    frag aq[DGC_STEPS][MF], wq[DGC_STEPS], wn[DGC_STEPS];
#pragma unroll
    for (int d = 0; d < DGC_STEPS; ++d)
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const elem* arow = reinterpret_cast<const elem*>(p.A) + (int64_t)min(m0 + 16 * i + c, p.M - 1) * p.lda + g * T::PER_LANE;
            if (ks0 + d < ks1) aq[d][i] = *reinterpret_cast<const frag*>(arow + (ks0 + d) * T::STEP);
        }
    auto load_w = [&](int cb, frag (&w_)[DGC_STEPS]) {
        const elem* wrow = reinterpret_cast<const elem*>(p.W) + (int64_t)min(cb * 16 + c, p.N - 1) * p.ldw + g * T::PER_LANE;
#pragma unroll
        for (int d = 0; d < DGC_STEPS; ++d) if (ks0 + d < ks1) w_[d] = *reinterpret_cast<const frag*>(wrow + (ks0 + d) * T::STEP);
    };
    int cb = blockIdx.x, it = 0;
    if (cb < col_blocks) load_w(cb, wq);
    for (; cb < col_blocks; cb += gridDim.x, ++it) {
        if (cb + (int)gridDim.x < col_blocks) load_w(cb + gridDim.x, wn);
        const int gn = cb * 16 + 4 * g;
        float bias4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && wave < MF) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[r] = p.bias[min(gn + r, p.N - 1)];
        }
        float4_t acc[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < DGC_STEPS; ++d)
            if (ks0 + d < ks1) {
#pragma unroll
                for (int i = 0; i < MF; ++i) T::mma(wq[d], aq[d][i], acc[i]);
            }
        float* buf = red + (it & 1) * (DG_NW * MF * 64 * 4);
        float4_t* mine = reinterpret_cast<float4_t*>(buf) + (wave * MF) * 64 + lane;
#pragma unroll
        for (int i = 0; i < MF; ++i) mine[i * 64] = acc[i];
        __syncthreads();           // (a wave reaches the next block's barrier only after it finished reading this buffer: two buffers suffice)
        if (wave < MF) {
            const int i = wave, gm = m0 + 16 * i + c;
            float4_t sres = reinterpret_cast<const float4_t*>(buf)[(0 * MF + i) * 64 + lane];
#pragma unroll
            for (int w = 1; w < DG_NW; ++w) {
                const float4_t t = reinterpret_cast<const float4_t*>(buf)[(w * MF + i) * 64 + lane];
                sres[0] += t[0]; sres[1] += t[1]; sres[2] += t[2]; sres[3] += t[3];
            }
            if (gm < p.M && gn < p.N) {
                const float v[4] = {sres[0] + bias4[0], sres[1] + bias4[1], sres[2] + bias4[2], sres[3] + bias4[3]};
                const int nvalid = min(4, p.N - gn);
                elem* cp = reinterpret_cast<elem*>(p.C) + (int64_t)gm * p.ldc + gn;
                const bool vec = nvalid == 4 && ((p.ldc & 3) == 0);
                if constexpr (F32) {
                    if (vec) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                    else for (int r = 0; r < nvalid; ++r) cp[r] = v[r];
                } else {
                    if (vec) { uint2 u; u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]); *reinterpret_cast<uint2*>(cp) = u; }
                    else for (int r = 0; r < nvalid; ++r) cp[r] = f32_to_bf16(v[r]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < DGC_STEPS; ++d) wq[d] = wn[d];
    }
}

int vm_skinny_rows_per_wg(int M, int N, int max_mf);      // gemm_skinny.hip

template <bool F32, int MF, bool LN>
static int dg_launch(const DgArgs& a, hipStream_t s) {
    size_t lds = (size_t)DG_NW * MF * 64 * sizeof(float4_t);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_gemm_kernel<F32, MF, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((decode_gemm_kernel<F32, MF, LN>), dim3((a.N + 15) / 16, (a.M + 16 * MF - 1) / (16 * MF)), dim3(DG_NW * 64), lds, s, a);
    return vm_check_launch("vm_decode_gemm");
}
template <bool F32, bool LN>
static int dg_dispatch(const DgArgs& a, int mf, hipStream_t s) {
    switch (mf) {
        case 1: return dg_launch<F32, 1, LN>(a, s);
        case 2: return dg_launch<F32, 2, LN>(a, s);
        case 4:
            if constexpr (LN) return dg_launch<F32, 2, LN>(a, s);
            else return dg_launch<F32, 4, LN>(a, s);
        default:
            if constexpr (LN) return dg_launch<F32, 2, LN>(a, s);       // (LN keeps the whole K slice of every fragment + gamma / beta resident)
            else return dg_launch<F32, 8, LN>(a, s);
    }
}

extern "C" int vm_decode_gemm(const vm_decode_gemm_args* x, void* stream) {
    VM_REQUIRE(x && x->A && x->W && x->C, "vm_decode_gemm: null pointer");
    VM_REQUIRE(x->dtype == VM_BF16 || x->dtype == VM_F32, "vm_decode_gemm: dtype must be VM_BF16 or VM_F32");
    const bool f32 = x->dtype == VM_F32;
    const int step = f32 ? 16 : 32, al = f32 ? 4 : 8;
    VM_REQUIRE(x->M > 0 && x->M <= 256 && x->N > 0 && x->K > 0 && (x->K % step) == 0, "vm_decode_gemm: M <= 256 rows, K a multiple of %d (M=%d N=%d K=%d)", step,
               x->M, x->N, x->K);
    VM_REQUIRE((x->lda % al) == 0 && (x->ldw % al) == 0 && ((uintptr_t)x->A % 16) == 0 && ((uintptr_t)x->W % 16) == 0 && ((uintptr_t)x->C % 8) == 0,
               "vm_decode_gemm: leading dimensions must be multiples of %d elements and A / W 16-byte aligned", al);
    VM_REQUIRE(x->act == 0 || x->act == 1, "vm_decode_gemm: act must be 0 or 1 (erf-GELU)");
    VM_REQUIRE(!x->c2 || (x->split_n > 0 && x->split_n < x->N && (x->split_n % 4) == 0 && ((uintptr_t)x->c2 % 8) == 0), "vm_decode_gemm: bad second destination");
    const bool ln = x->ln_gamma != nullptr;
    VM_REQUIRE(!ln || (x->ln_beta && x->K <= 1024 && ((uintptr_t)x->ln_gamma % 16) == 0 && ((uintptr_t)x->ln_beta % 16) == 0 && (!x->ln_out || ((x->ln_out_ld % al) == 0 && ((uintptr_t)x->ln_out % 16) == 0))),
               "vm_decode_gemm: LayerNorm-on-load needs beta, K <= 1024 and an aligned ln_out");
    DgArgs a = {x->A, x->W, x->C, x->c2, x->bias, x->residual, x->ln_gamma, x->ln_beta, x->ln_out,
                x->lda, x->ldw, x->ldc, x->ldc2, x->ldr, x->ln_out_ld, x->M, x->N, x->K, x->act, x->split_n, x->ln_eps};
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_DECODE, 2.0 * x->M * (double)x->N * x->K, s, "dg_%s_M%d_N%d_K%d_ln%d", f32 ? "f32" : "bf16", x->M, x->N, x->K, (int)ln);
    // rows per workgroup: at least one workgroup per CU (gemm_skinny.hip), at most 64 rows -- measured at 256 rows (beam 4) on the step's
    // shapes: 64-row blocks 14.9 / 15.3 / 17.8 us (N = 2304 / 3072 / 768 x K = 3072) against 18.4 / 18.8 / 28.3 us for 128-row blocks and
    // 15.0 / 17.5 / 20.3 us for 32-row blocks; fp32 likewise (26.0 / 26.8 / 32.6 vs 33.3 / 33.9 / 56.6 us)
    if (f32 && !ln && !x->residual && !x->act && !x->c2 && x->N >= 4096 && x->K / step <= DG_NW * DGC_STEPS) {      // wide output: column walk
        const int row_blocks = (x->M + 63) / 64, col_blocks = (x->N + 15) / 16;
        int gx = 512 / row_blocks; if (gx > col_blocks) gx = col_blocks; if (gx < 1) gx = 1;
        const size_t lds = (size_t)2 * DG_NW * 4 * 64 * sizeof(float4_t);
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_gemm_cols_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        hipLaunchKernelGGL((decode_gemm_cols_kernel<true>), dim3(gx, row_blocks), dim3(DG_NW * 64), lds, s, a);
        return vm_check_launch("vm_decode_gemm(column walk)");
    }
    int mf = vm_skinny_rows_per_wg(x->M, x->N, ln ? 2 : 4);
    if (!ln && mf == 2 && x->M > 128 && ((x->N + 15) / 16) * ((x->M + 63) / 64) >= 192) mf = 4;
    if (f32) return ln ? dg_dispatch<true, true>(a, mf, s) : dg_dispatch<true, false>(a, mf, s);
    return ln ? dg_dispatch<false, true>(a, mf, s) : dg_dispatch<false, false>(a, mf, s);
}
