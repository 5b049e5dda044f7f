// decode_f32.hip -- the fp32 decode step: every kernel of vilmedic_amd.generation's step in full fp32, so that greedy / beam
// token indices are those of the reference's fp32 CPU path (north_star: "bit-exact token indices for greedy decode").
//
// The bf16 step rounds every activation to 8 mantissa bits; a near-tie between the two best logits (a random-weight decoder
// has many) can then flip an arg-max.  Here nothing is rounded below fp32: fp32 master weights straight from the parameter
// arena (no shadows), fp32 activations and KV caches, matrix products on the exact f32-input MFMA
// (v_mfma_f32_16x16x4_f32 == an fmaf chain, MI355X_MICROARCH.md "Matrix cores"), fp32 softmax / LayerNorm / erf-GELU.
// What still differs from the CPU reference is the summation order only (~1e-6 relative), which token selection survives
// unless the top-2 logits tie within that noise.
//
// Replaces (in fp32 mode) hf:models/bert_generation/modeling_bert_generation.py:45-231,264-358,394-426,590-610 as reached
// from ref:vilmedic/blocks/huggingface/decoder/evaluation.py:73-78 (generate) -- same call sites as the bf16 step.
#include "common.h"
#include "gemm_args.h"

// ------------------------------------------------------------------ C[M,N] = epi(A[M,K] . W[N,K]^T), all fp32
// One workgroup owns 16 output columns for a block of 16*MF rows; its 8 waves split the contraction eight ways (operands go
// straight from global / L2 into MFMA fragments, W is streamed exactly once per row block), partial accumulators meet in LDS.
// A lane loads 4 consecutive k of its row (16 B): MFMA sub-step j contracts k = k0 + 4g + j, g = lane / 16 -- any bijection
// of the contraction index works as long as both operands use the same one.
// Operands are swapped (D^T = W_frag x A_frag) so a lane ends with 4 consecutive output columns of one row.
struct GemmF32Args {
    const float* A; const float* W; float* C; const float* bias; const float* residual;
    int64_t lda, ldw, ldc, ldr;
    int M, N, K, act;
};

__device__ __forceinline__ float gelu_erf_f32(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

#define F32_NW 8       // waves per workgroup (the contraction is split 8 ways)
// k-steps of operand fragments in flight per wave (register ring, see gemm_skinny.hip): 4 for <= 64 rows, fewer for the taller blocks
// (16 fragments x 4 steps would be 256 registers of operands alone)
#define F32_D (MF <= 4 ? 4 : MF <= 8 ? 2 : 1)

template <int MF>
__global__ __launch_bounds__(F32_NW * 64) void gemm_f32_kernel(const GemmF32Args p) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // [F32_NW waves][MF][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MF);
    const int ksteps = p.K >> 4;                                          // 16-deep steps (4 MFMAs each)
    const int per = (ksteps + F32_NW - 1) / F32_NW;
    const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
    const float* wrow = p.W + (int64_t)min(n0 + c, p.N - 1) * p.ldw + g * 4;
    const float* arow[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) arow[i] = p.A + (int64_t)min(m0 + 16 * i + c, p.M - 1) * p.lda + g * 4;
    float4_t acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    constexpr int D = F32_D;
    float4 wq[D], aq[D][MF];
    auto load = [&](int ks, float4& w_, float4 (&a_)[MF]) {
        w_ = *reinterpret_cast<const float4*>(wrow + ks * 16);
#pragma unroll
        for (int i = 0; i < MF; ++i) a_[i] = *reinterpret_cast<const float4*>(arow[i] + ks * 16);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) if (ks0 + d < ks1) load(ks0 + d, wq[d], aq[d]);
    for (int ks = ks0; ks < ks1; ks += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (ks + d < ks1) {
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[d].x, aq[d][i].x, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[d].y, aq[d][i].y, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[d].z, aq[d][i].z, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[d].w, aq[d][i].w, acc[i], 0, 0, 0);
                }
                if (ks + d + D < ks1) load(ks + d + D, wq[d], aq[d]);
            }
        }
    }
    // D^T layout: lane (c, g) of fragment i holds row m0 + 16 i + c, columns n0 + 4 g .. + 3
    float4_t* mine = reinterpret_cast<float4_t*>(red) + (wave * MF) * 64 + lane;
#pragma unroll
    for (int i = 0; i < MF; ++i) mine[i * 64] = acc[i];
    __syncthreads();
    const int gn = n0 + 4 * g;
    for (int i = wave; i < MF; i += F32_NW) {
        const int gm = m0 + 16 * i + c;
        float4_t s = reinterpret_cast<const float4_t*>(red)[(0 * MF + i) * 64 + lane];
#pragma unroll
        for (int w = 1; w < F32_NW; ++w) {                                 // fixed order: deterministic
            const float4_t t = reinterpret_cast<const float4_t*>(red)[(w * MF + i) * 64 + lane];
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        if (gm >= p.M || gn >= p.N) continue;
        float v[4] = {s[0], s[1], s[2], s[3]};
        const int nvalid = min(4, p.N - gn);
        if (p.bias) for (int r = 0; r < nvalid; ++r) v[r] += p.bias[gn + r];
        if (p.act == 1) for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f32(v[r]);
        if (p.residual) for (int r = 0; r < nvalid; ++r) v[r] += p.residual[(int64_t)gm * p.ldr + gn + r];
        float* cp = p.C + (int64_t)gm * p.ldc + gn;
        if (nvalid == 4 && (p.ldc & 3) == 0) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int r = 0; r < nvalid; ++r) cp[r] = v[r];
    }
}

template <int MF>
static int launch_gemm_f32(const GemmF32Args& a, hipStream_t s) {
    const size_t lds = (size_t)F32_NW * MF * 64 * sizeof(float4_t);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<MF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<MF>), dim3((a.N + 15) / 16, (a.M + 16 * MF - 1) / (16 * MF)), dim3(F32_NW * 64), lds, s, a);
    return vm_check_launch("vm_gemm_f32");
}

extern "C" int vm_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int M, int N, int K,
                           const float* bias, int act, const float* residual, int64_t ldr, void* stream) {
    VM_REQUIRE(A && W && C, "vm_gemm_f32: null pointer");
    VM_REQUIRE(M > 0 && N > 0 && K > 0 && (K % 16) == 0, "vm_gemm_f32: K must be a positive multiple of 16 (M=%d N=%d K=%d)", M, N, K);
    VM_REQUIRE((lda % 4) == 0 && (ldw % 4) == 0, "vm_gemm_f32: lda / ldw must be multiples of 4");
    VM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0, "vm_gemm_f32: pointers must be 16-byte aligned");
    VM_REQUIRE(act == 0 || act == 1, "vm_gemm_f32: act must be 0 or 1 (erf-GELU)");
    GemmF32Args a = {A, W, C, bias, residual, lda, ldw, ldc, ldr, M, N, K, act};
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_DECODE, 2.0 * M * (double)N * K, s, "f32_M%d_N%d_K%d", M, N, K);
    switch (vm_skinny_rows_per_wg(M, N, 16)) {         // rows per workgroup: at least one workgroup per CU (gemm_skinny.hip)
        case 1: return launch_gemm_f32<1>(a, s);
        case 2: return launch_gemm_f32<2>(a, s);
        case 4: return launch_gemm_f32<4>(a, s);
        case 8: return launch_gemm_f32<8>(a, s);
        default: return launch_gemm_f32<16>(a, s);
    }
}

// ------------------------------------------------------------------ LayerNorm, fp32 in / out (one wave per row)
__global__ __launch_bounds__(256) void ln_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * cols;
    float* yr = y + (int64_t)row * cols;
    if (cols <= 64 * 16) {           // the row stays in registers: ONE global read instead of three dependent passes (same sums, same order)
        float v[16], gm[16], bt[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < cols;
            v[i] = ok ? xr[c] : 0.f; gm[i] = ok ? gamma[c] : 0.f; bt[i] = ok ? beta[c] : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (lane + 64 * i < cols) s += v[i];
        const float mu = wave_sum(s) / (float)cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (lane + 64 * i < cols) { const float d = v[i] - mu; q += d * d; }
        const float rs = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
        for (int i = 0; i < 16; ++i) if (lane + 64 * i < cols) yr[lane + 64 * i] = (v[i] - mu) * rs * gm[i] + bt[i];
        return;
    }
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += xr[c];
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
    for (int c = lane; c < cols; c += 64) { const float d = xr[c] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
    for (int c = lane; c < cols; c += 64) yr[c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
}

extern "C" int vm_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int rows, int cols, float eps, void* stream) {
    VM_REQUIRE(x && gamma && beta && y && rows > 0 && cols > 0, "vm_layernorm_f32: bad arguments");
    hipLaunchKernelGGL(ln_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, cols, eps);
    return vm_check_launch("vm_layernorm_f32");
}

// ------------------------------------------------------------------ embeddings, fp32 out: word[ids] + pos[past_len + t]
__global__ __launch_bounds__(256) void embedding_f32_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                                            float* __restrict__ out, int L, int D, int past_len, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / D; const int d = (int)(i - row * D);
        const int t = (int)(row % L);
        out[i] = word[ids[row] * (int64_t)D + d] + pos[(int64_t)(past_len + t) * D + d];
    }
}

extern "C" int vm_embedding_fwd_f32(const int64_t* ids, const float* word, const float* pos, float* out, int B, int L, int D, int past_len, void* stream) {
    VM_REQUIRE(ids && word && pos && out && B > 0 && L > 0 && D > 0, "vm_embedding_fwd_f32: bad arguments");
    const int64_t total = (int64_t)B * L * D;
    int64_t blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embedding_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ids, word, pos, out, L, D, past_len, total);
    return vm_check_launch("vm_embedding_fwd_f32");
}

// ------------------------------------------------------------------ attention of the decode step, fp32
// One wave per (query row, head).  Query row r belongs to key/value batch r / q_per_kv (cross-attention: the beams of a
// sample share its projected image features; self-attention: q_per_kv = 1).  Key j of the row is K row
//   kv_index ? kv_index[r * kv_index_ld + j]  :  (r / q_per_kv) * Lk + j
// (the beam-search cache indirection of vm_attention_fwd).  key_mask [n_kv_batches, Lk] (1 = attend) or NULL; a masked key
// gets the additive float-min of hf:modeling_attn_mask_utils (so a fully masked row degrades to uniform, as in HF).
// Scores live in LDS (Lk <= 4096); lanes split the keys for QK^T / softmax and the head dim for PV.
struct AttnF32Args {
    const float* q; const float* k; const float* v; float* o; const uint8_t* key_mask; const int32_t* kv_index;
    int64_t ldq, ldk, ldv, ldo, kv_index_ld;
    int rows, H, Lk, dh, q_per_kv; float scale;
};

// Round 3: K and V reach the wave through LDS in 64-key tiles -- every lane requests dh / 4 consecutive 16-B pieces of the tile
// (coalesced: dh / 4 lanes per key row, all requests of a tile in flight together, the next tile requested before the current one is
// used) -- instead of one lane walking a whole key row (QK^T: 16 dependent-stride loads per key per lane) and 8 keys per memory round
// trip in P.V (25 round trips for the 197 image keys): 27 us -> the traffic's own time per launch.  The arithmetic and its ORDER are
// unchanged (a lane accumulates one key's dot product over d = 0 .. dh - 1, one output column over the keys in order), so the
// results are bit-identical to the round-2 kernel.
#define AF_TS 64                      // keys per tile
template <int N4>                     // N4 = dh / 4 pieces per key row
__device__ __forceinline__ void af32_request(const float* base, int64_t ld, const int* ro, int j0, int Lk, int lane, float4 (&r)[N4]) {
#pragma unroll
    for (int u = 0; u < N4; ++u) {
        const int f = lane + 64 * u, kk = f / N4, c4 = f - kk * N4;
        const int j = j0 + kk;
        r[u] = j < Lk ? *reinterpret_cast<const float4*>(base + (int64_t)ro[j] * ld + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int N4>
__device__ __forceinline__ void af32_stage(float* tile, int lane, const float4 (&r)[N4]) {
    constexpr int TS = 4 * N4 + 4;    // padded row: lane l reads row l at 16-B steps -> 4 l (mod 64) words apart: conflict-free
#pragma unroll
    for (int u = 0; u < N4; ++u) {
        const int f = lane + 64 * u, kk = f / N4, c4 = f - kk * N4;
        *reinterpret_cast<float4*>(tile + kk * TS + 4 * c4) = r[u];
    }
}

template <int N4>
__global__ __launch_bounds__(64) void attn_decode_f32_kernel(const AttnF32Args p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [64][dh + 4] tile | [Lk8] scores -> probabilities | [Lk8] key rows | [dh] query
    constexpr int DH = 4 * N4, TS = DH + 4;
    const int lane = threadIdx.x;
    const int r = blockIdx.x / p.H, h = blockIdx.x - r * p.H;
    const int kvb = r / p.q_per_kv;
    const int Lk8 = (p.Lk + 7) & ~7;
    float* tile = sm;
    float* sc = sm + AF_TS * TS;
    int* ro = reinterpret_cast<int*>(sc + Lk8);
    float* qs = sc + 2 * Lk8;
    const float* qr = p.q + (int64_t)r * p.ldq + h * DH;
    for (int d = lane; d < DH; d += 64) qs[d] = qr[d];
    for (int j = lane; j < Lk8; j += 64) ro[j] = j >= p.Lk ? 0 : (p.kv_index ? p.kv_index[(int64_t)r * p.kv_index_ld + j] : kvb * p.Lk + j);
    __syncthreads();
    const int ntiles = (p.Lk + AF_TS - 1) / AF_TS;
    float4 cur[N4], nxt[N4];
    // ---- scores
    const float* kb = p.k + h * DH;
    af32_request<N4>(kb, p.ldk, ro, 0, p.Lk, lane, cur);
    float mx = -INFINITY;
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) af32_request<N4>(kb, p.ldk, ro, (t + 1) * AF_TS, p.Lk, lane, nxt);
        af32_stage<N4>(tile, lane, cur);
        __syncthreads();
        const int j = t * AF_TS + lane;
        if (j < Lk8) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const float4 kk = *reinterpret_cast<const float4*>(tile + lane * TS + d);
                s = fmaf(qs[d], kk.x, s); s = fmaf(qs[d + 1], kk.y, s); s = fmaf(qs[d + 2], kk.z, s); s = fmaf(qs[d + 3], kk.w, s);
            }
            s *= p.scale;
            if (j >= p.Lk) s = -INFINITY;
            else if (p.key_mask && !p.key_mask[(int64_t)kvb * p.Lk + j]) s += -3.4028234663852886e38f;
            sc[j] = s;
            mx = fmaxf(mx, s);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < N4; ++u) cur[u] = nxt[u];
    }
    mx = wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < Lk8; j += 64) { const float e = j < p.Lk ? expf(sc[j] - mx) : 0.f; sc[j] = e; se += e; }
    se = wave_sum(se);
    __syncthreads();
    const float inv = 1.0f / se;
    // ---- P.V: lane d (and d + 64 for wider heads) accumulates its output column over the keys in order
    const float* vb = p.v + h * DH;
    constexpr int ND = (DH + 63) / 64;
    float acc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) acc[i] = 0.f;
    af32_request<N4>(vb, p.ldv, ro, 0, p.Lk, lane, cur);
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) af32_request<N4>(vb, p.ldv, ro, (t + 1) * AF_TS, p.Lk, lane, nxt);
        af32_stage<N4>(tile, lane, cur);
        __syncthreads();
        const int nk = min(AF_TS, Lk8 - t * AF_TS);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d = lane + 64 * i;
            if (d < DH) {
                float a = acc[i];
                for (int kk = 0; kk < nk; ++kk) a = fmaf(sc[t * AF_TS + kk], tile[kk * TS + d], a);
                acc[i] = a;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < N4; ++u) cur[u] = nxt[u];
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int d = lane + 64 * i;
        if (d < DH) p.o[(int64_t)r * p.ldo + h * DH + d] = acc[i] * inv;
    }
}
template <int N4>
static void af32_launch(const AttnF32Args& a, hipStream_t s) {
    const size_t lds = (size_t)(AF_TS * (4 * N4 + 4) + 2 * ((a.Lk + 7) & ~7) + 4 * N4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_decode_f32_kernel<N4>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_decode_f32_kernel<N4>, dim3((unsigned)(a.rows * a.H)), dim3(64), lds, s, a);
}
extern "C" int vm_attention_decode_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                       float* o, int64_t ldo, const uint8_t* key_mask, const int32_t* kv_row_index, int64_t kv_index_ld,
                                       int rows, int H, int Lk, int dh, int q_per_kv, float scale, void* stream) {
    VM_REQUIRE(q && k && v && o, "vm_attention_decode_f32: null pointer");
    VM_REQUIRE(rows > 0 && H > 0 && Lk > 0 && Lk <= 4096 && dh > 0 && (dh % 4) == 0 && q_per_kv > 0, "vm_attention_decode_f32: bad shape (rows=%d H=%d Lk=%d dh=%d)", rows, H, Lk, dh);
    VM_REQUIRE((ldk % 4) == 0 && ((uintptr_t)k % 16) == 0, "vm_attention_decode_f32: K rows must be 16-byte aligned");
    AttnF32Args a = {q, k, v, o, key_mask, kv_row_index, ldq, ldk, ldv, ldo, kv_index_ld, rows, H, Lk, dh, q_per_kv, scale};
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_DECODE, 4.0 * rows * H * (double)Lk * dh, s, "attn_f32_r%d_H%d_Lk%d", rows, H, Lk);
    switch (dh) {
        case 32: af32_launch<8>(a, s); break;
        case 64: af32_launch<16>(a, s); break;
        case 96: af32_launch<24>(a, s); break;
        case 128: af32_launch<32>(a, s); break;
        default: vm_set_error("vm_attention_decode_f32: head dim %d (32, 64, 96 or 128)", dh); return VM_EUNSUPPORTED;
    }
    return vm_check_launch("vm_attention_decode_f32");
}
