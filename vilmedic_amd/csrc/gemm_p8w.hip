// gemm_p8w.hip -- grouped weight gradients on the wide-tile ping-pong structure of gemm_p8.hip:
//     dW_i[n_out, k_in] (+)= alpha_i * dY_i[rows, n_out]^T . X_i[rows, k_in]     and     db_i[n_out] += alpha_i * colsum(dY_i)
// for up to VM_GEMM_MAX_GROUP problems per launch (what autograd computes for nn.Linear weight / bias in the reference's backward:
// hf:models/bert_generation/modeling_bert_generation.py:104-106,264-291; hf:models/vit/modeling_vit.py:192-251).
//
// The contraction runs over the ROWS of the batch (8192 / 12608 here: 128 - 197 K-tiles), so a launch is almost all main loop, and the main
// loop of the 256 x 256 tile measured 6.6 TFLOP/s per CU against 3.6 for the 128 x 128 tiles of gemm_grouped_kernel (profiles/r04_*):
//   * one 8-wave workgroup per CU, two wave groups (rows 0-127 / 128-255 of the tile) in ping-pong on the SIMDs they share: while one
//     group issues its MFMAs the other reads its fragments (ds_read_b64_tr_b16: both operands are contraction-major) and stages;
//   * LDS-DMA half-tiles [64 k][128], BOTH operands two K-tiles ahead (A ring of three K-tiles, B ring of two), one counted
//     s_waitcnt vmcnt(8) per K-tile; the epilogue stages through the idle A ring (see gemm_p8.hip for the hazards);
//   * a workgroup owns its output tile (no split of the contraction: the launches are sized to fill the chip by grouping the linears of
//     two transformer layers), so accumulation into dW needs no atomics.  ``accumulate == 0`` (the caller knows this is the first
//     contribution to dW since the gradients were zeroed -- ops.param_grads tracks it) stores; otherwise the epilogue reads, adds and
//     writes, which on gfx950 costs one store round trip per 32-row chunk (loads and stores share one in-order counter);
//   * the bias gradient is one extra MFMA per A fragment against an all-ones operand in the workgroups of tile column 0, each of the four
//     waves of a group taking two of the group's eight fragments.
#include <type_traits>
#include "common.h"
#include "gemm_args.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ int pw_swz1(int krow) { return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3); }

__device__ __forceinline__ void pw_glds16(const bf16_t* sbase, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds)) : "memory", "m0");
}

__device__ __forceinline__ int pw_xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

#define PW_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)

constexpr int PW_HALF = 128 * 128;                 // bytes of a half-tile: [64 k][128] bf16
constexpr int PW_KT = 2 * PW_HALF;                 // one operand's K-tile: lo and hi half
constexpr int PW_A0 = 0, PW_NA = 3;                // A ring: THREE K-tiles (both operands are requested two K-tiles ahead: a weight-gradient
constexpr int PW_B0 = PW_NA * PW_KT, PW_NB = 2;    //   launch streams its operands from HBM / MALL, and one K-tile of lead -- 1.7 us -- did not cover
constexpr int PW_LDS = PW_B0 + PW_NB * PW_KT;      //   that latency); B ring: two (its registers free the slot after the first phase).  160 KiB.
constexpr int PW_EPI = 32 * 256 * 4;               // the epilogue stages through the (then idle) A ring

template <int NPH>
__global__ __launch_bounds__(512, 2) void gemm_p8w_kernel(const P8wArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, g = lane >> 4, c = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const int total = ga.tile_start[P8W_MAX_GROUP];

    // fragment (rbase + 16 i .. + 15) x (32 kk .. + 31) of a [64 k][128] half-tile, chunk ^= swz1(k) (gemm_fast.hip layout 1)
    const int j4 = c >> 2, s1 = (j4 << 1) | ((g & 1) << 3), sub1 = (c & 1) * 8, h1 = (c & 3) >> 1;
    auto read1 = [&](const char* tile, int rbase, int i, int kk) -> bf16x8_t {
        const int krow = kk * 32 + 8 * g + j4;
        const int lc = (rbase >> 3) + 2 * i + h1;
        const int off = ((lc ^ s1) << 4) + sub1;
        v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(tile + krow * 256 + off));
        v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(tile + (krow + 4) * 256 + off));
        short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };
    const int b_rb = (wn & 1) * 64;

    uint32_t offA[2][2], offB[2][2];
    const bf16_t* pA = nullptr; const bf16_t* pB = nullptr;
    int64_t stepA = 0, stepB = 0;

    auto decode = [&](int bid, int& gi, int& tm, int& tn) {
        gi = 0;
#pragma unroll
        for (int i = 1; i < P8W_MAX_GROUP; ++i) if (i < ga.n && bid >= ga.tile_start[i]) gi = i;
        const int t = bid - ga.tile_start[gi];
        const int tnn = ga.g[gi].tiles_n;
        tm = t / tnn; tn = t - tm * tnn;
    };
    auto tile_setup = [&](int gi, int m0, int n0) {
        const P8wProblem& p = ga.g[gi];
        const int lda = (int)p.lda, ldb = (int)p.ldb;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = wave + 8 * i;
                const int krow = 4 * q + (lane >> 4);
                const int lc = (lane & 15) ^ pw_swz1(krow);
                int col = h * 128 + lc * 8;
                offA[h][i] = (uint32_t)(krow * lda * 2 + (m0 + col < p.M ? col : 0) * 2);
                offB[h][i] = (uint32_t)(krow * ldb * 2 + (n0 + col < p.N ? col : 0) * 2);
            }
        pA = p.A + m0; pB = p.B + n0;
        stepA = (int64_t)64 * p.lda; stepB = (int64_t)64 * p.ldb;
    };
    auto stageA = [&](int slot, int h) {
        const uint32_t dst = lds0 + PW_A0 + slot * PW_KT + h * PW_HALF;
#pragma unroll
        for (int i = 0; i < 2; ++i) pw_glds16(pA, offA[h][i], dst + (wave + 8 * i) * 1024);
    };
    auto stageB = [&](int slot, int h) {
        const uint32_t dst = lds0 + PW_B0 + slot * PW_KT + h * PW_HALF;
#pragma unroll
        for (int i = 0; i < 2; ++i) pw_glds16(pB, offB[h][i], dst + (wave + 8 * i) * 1024);
    };
    auto prologue = [&](int nk) {                          // K-tiles 0 and 1 of both operands; the first wait leaves the second in flight
        stageA(0, 0); stageA(0, 1); pA += stepA;
        stageB(0, 0); stageB(0, 1); pB += stepB;
        if (nk > 1) { stageA(1, 0); stageA(1, 1); pA += stepA; stageB(1, 0); stageB(1, 1); pB += stepB; }
    };

    for (int work = (int)blockIdx.x; work < total; work += (int)gridDim.x) {
        int gi, tm, tn;
        decode(pw_xcd_remap(work, total), gi, tm, tn);
        const P8wProblem& p = ga.g[gi];
        const int m0 = tm * 256, n0 = tn * 256;
        const int nk = p.ktiles;
        const bool has_alpha = p.alpha_dev != nullptr;
        const bool bias_wg = p.bias_grad != nullptr && tn == 0;

        tile_setup(gi, m0, n0);
        prologue(nk);

        float4_t acc[4][8];
        float4_t accb[2] = {(float4_t){0.f, 0.f, 0.f, 0.f}, (float4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = (float4_t){0.f, 0.f, 0.f, 0.f};
        if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PW_BARRIER();

        bf16x8_t fa[2][4], fb[2][4];
        const short8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
        const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

        // one K-tile t; sa = t % 3 (its A slot), H2: there is a K-tile t + 2 to request; BG: this workgroup also sums the rows of A
        auto ktile = [&](int t, int sa, auto h2_c, auto bg_c) {
            constexpr bool H2 = decltype(h2_c)::value, BG = decltype(bg_c)::value;
            const int s = t & 1;
            const int sa2 = sa == 0 ? 2 : sa - 1;           // (t + 2) % 3: the slot K-tile t - 1 was read from
            const char* As = smem + PW_A0 + sa * PW_KT + wm * PW_HALF;
            const char* Bs = smem + PW_B0 + s * PW_KT + (wn >> 1) * PW_HALF;
            auto mfma_rows = [&](auto i0_c, auto j_lo, auto j_hi) {       // A fragments i0..i0+3 (in fa) x B fragments [j_lo, j_hi)
                constexpr int i0 = decltype(i0_c)::value, JL = decltype(j_lo)::value, JH = decltype(j_hi)::value;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = JL; j < JH; ++j) acc[j][i0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[j][i0 + i], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            };
            auto mfma_bias = [&](int half) {      // waves 0,1 own fragments 0..3 (first register sub-tile), waves 2,3 fragments 4..7
                if constexpr (BG) {
                    if ((wn >> 1) == half) {
                        if (wn & 1) {
#pragma unroll
                            for (int kk = 0; kk < 2; ++kk) { accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[kk][2], accb[0], 0, 0, 0); accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[kk][3], accb[1], 0, 0, 0); }
                        } else {
#pragma unroll
                            for (int kk = 0; kk < 2; ++kk) { accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[kk][0], accb[0], 0, 0, 0); accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[kk][1], accb[1], 0, 0, 0); }
                        }
                    }
                }
            };
            using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
            if constexpr (NPH == 2) {
                // ---- phase 0: all of B + A fragments 0..3; A(t+1) -> other slot
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[kk][j] = read1(Bs, b_rb, j, kk);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[kk][i] = read1(As, 0, i, kk);
                if constexpr (H2) { stageA(sa2, 0); stageA(sa2, 1); pA += stepA; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PW_BARRIER();
                mfma_rows(I0{}, I0{}, I4{});
                mfma_bias(0);
                PW_BARRIER();
                // ---- phase 1: A fragments 4..7; B(t+2) -> this slot; K-tile t+1 landed
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[kk][i] = read1(As, 0, 4 + i, kk);
                if constexpr (H2) { stageB(s, 0); stageB(s, 1); pB += stepB; asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }      // K-tile t + 2 (8 pieces) may be in flight
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PW_BARRIER();
                mfma_rows(I4{}, I0{}, I4{});
                mfma_bias(1);
                PW_BARRIER();
            } else {
                // ---- four phases of 16 MFMAs: (A 0..3, B 0..1), (A 0..3, B 2..3), (A 4..7, B 2..3), (A 4..7, B 0..1)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[kk][j] = read1(Bs, b_rb, j, kk);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[kk][i] = read1(As, 0, i, kk);
                if constexpr (H2) stageA(sa2, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PW_BARRIER();
                mfma_rows(I0{}, I0{}, I2{});
                mfma_bias(0);
                PW_BARRIER();
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 2; j < 4; ++j) fb[kk][j] = read1(Bs, b_rb, j, kk);
                if constexpr (H2) { stageA(sa2, 1); pA += stepA; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PW_BARRIER();
                mfma_rows(I0{}, I2{}, I4{});
                PW_BARRIER();
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[kk][i] = read1(As, 0, 4 + i, kk);
                if constexpr (H2) stageB(s, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PW_BARRIER();
                mfma_rows(I4{}, I2{}, I4{});
                mfma_bias(1);
                PW_BARRIER();
                if constexpr (H2) { stageB(s, 1); pB += stepB; asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PW_BARRIER();
                mfma_rows(I4{}, I0{}, I2{});
                PW_BARRIER();
            }
        };
        auto kloop = [&](auto bg_c) {
            int t = 0, sa = 0;
            for (; t + 2 < nk; ++t) { ktile(t, sa, std::true_type{}, bg_c); sa = sa == 2 ? 0 : sa + 1; }
            for (; t < nk; ++t) { ktile(t, sa, std::false_type{}, bg_c); sa = sa == 2 ? 0 : sa + 1; }
        };
        if (wm == 1) PW_BARRIER();                         // group 1 runs one barrier behind group 0
        if (bias_wg) kloop(std::true_type{}); else kloop(std::false_type{});
        if (wm == 0) PW_BARRIER();

        // ---- epilogue: 8 chunks of 32 rows (fragment row i of both groups) through the fp32 stage; whole 1-KiB rows to HBM
        float alpha = has_alpha ? *p.alpha_dev : 1.0f;
        asm volatile("" : "+v"(alpha));
        float* cs = reinterpret_cast<float*>(smem + PW_A0);          // the A ring is idle: nothing is in flight behind the last K-tile's vmcnt(0)
        int tid_e = tid, c_e = c, g_e = g;
        asm volatile("" : "+v"(tid_e), "+v"(c_e), "+v"(g_e));
        const int q = tid_e & 31, row_t = tid_e >> 5;
        const int gn = n0 + q * 8;
        const int nvalid = min(8, p.N - gn);
        const bool rmw = p.accumulate != 0;
        float* Cf = reinterpret_cast<float*>(p.C);
        if (bias_wg && g_e == 0) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int gm = m0 + wm * 128 + (2 * wn + f) * 16 + c_e;
                if (gm < p.M) {
                    if (p.accumulate != 0) p.bias_grad[gm] += accb[f][1] * alpha;      // single owner of the row: plain read-modify-write
                    else p.bias_grad[gm] = accb[f][1] * alpha;
                }
            }
        }
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PW_BARRIER();
            {
                const int r = wm * 16 + c_e;
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) {
                    if (ii != i) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int chunk = (wn * 16 + j * 4 + g_e) ^ (r & 7);
                        *reinterpret_cast<float4*>(cs + r * 256 + chunk * 4) =
                            make_float4(acc[j][ii][0] * alpha, acc[j][ii][1] * alpha, acc[j][ii][2] * alpha, acc[j][ii][3] * alpha);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PW_BARRIER();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = it * 16 + row_t;
                const int gm = m0 + it * 128 + i * 16 + row_t;
                if (gm >= p.M || nvalid <= 0) continue;
                float* cp = Cf + (int64_t)gm * p.ldc + gn;
                const float4 lo = *reinterpret_cast<const float4*>(cs + r * 256 + (((2 * q) ^ (r & 7)) << 2));
                const float4 hi = *reinterpret_cast<const float4*>(cs + r * 256 + (((2 * q + 1) ^ (r & 7)) << 2));
                if (nvalid == 8) {
                    if (rmw) {
                        const float4 o0 = *reinterpret_cast<float4*>(cp), o1 = *reinterpret_cast<float4*>(cp + 4);
                        *reinterpret_cast<float4*>(cp) = make_float4(o0.x + lo.x, o0.y + lo.y, o0.z + lo.z, o0.w + lo.w);
                        *reinterpret_cast<float4*>(cp + 4) = make_float4(o1.x + hi.x, o1.y + hi.y, o1.z + hi.z, o1.w + hi.w);
                    } else {
                        *reinterpret_cast<float4*>(cp) = lo;
                        *reinterpret_cast<float4*>(cp + 4) = hi;
                    }
                } else {
                    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    for (int r2 = 0; r2 < nvalid; ++r2) cp[r2] = rmw ? cp[r2] + v[r2] : v[r2];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PW_BARRIER();                                      // the stage and the ring are free for the next tile
    }
}

template <int NPH>
int launch_p8w(const P8wArgs& ga, int total, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8w_kernel<NPH>), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
        attr_set = true;
    }
    const int grid = total < 256 ? total : 256;
    hipLaunchKernelGGL((gemm_p8w_kernel<NPH>), dim3(grid), dim3(512), PW_LDS, s, ga);
    return vm_check_launch("vm_wgrad_grouped(p8w)");
}

}  // namespace

int vm_wgrad_p8w_launch(const P8wArgs& ga, int phases, hipStream_t s) {
    const int total = ga.tile_start[P8W_MAX_GROUP];
    return phases == 4 ? launch_p8w<4>(ga, total, s) : launch_p8w<2>(ga, total, s);
}
