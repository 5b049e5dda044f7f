// gemm_p8.hip -- the wide-output bf16 MFMA GEMM: one 8-wave workgroup per CU, (32 MF) x 256 tile, two wave groups in ping-pong.
//
// Why a second tuned kernel beside gemm_fast.hip (4 waves of 64 x 64, two workgroups per CU): on the step's K = 768 products the
// 128 / 160 x 128 tiles spend 30 % of a launch in prologue + epilogue with every CU in lockstep, their fill traffic per flop is
// 1.8x that of a 256 x 256 tile, and a wave's LDS-read latency is only covered by whatever the OTHER workgroup happens to do.
// Here (MI355X guide, "256^2 8-phase" structure, re-derived for this library's operand layouts and epilogues):
//   * 8 waves = 2 (M) x 4 (N); a wave owns (16 MF) x 64 outputs (MF <= 8 A fragments x 4 B fragments = 128 accumulator registers);
//     waves w and w + 4 (the two M groups) share a SIMD;
//   * a K-tile (64 deep) is four phases; a phase = [LDS reads of one register sub-tile + this phase's share of the LDS-DMA] s_barrier
//     [16 MFMAs] s_barrier.  Group 1 runs one barrier behind group 0, so on every SIMD one wave issues MFMAs while its partner reads
//     and stages: the matrix pipe never waits for LDS latency and the reads never compete with the owner's own MFMAs;
//   * operands arrive by global_load_lds_dwordx4 (inline asm, SGPR base + 32-bit lane offset) as four 16-KB half-tiles per K-tile,
//     each re-staged right after its last read: A one K-tile ahead, B two K-tiles ahead; ONE counted s_waitcnt vmcnt(4) per K-tile;
//   * the epilogue is staged through a separate 32-KB LDS region (fp32, one 16-row fragment row of both groups per chunk), so the
//     persistent form (PERSIST) requests the next tile's first K-tiles BEFORE the epilogue and finds them landed after it.
// Epilogue arithmetic (bias, GELU / z side output, gelu'(z), dropout, residual, bf16 / fp32 / slab output) is the same element-wise
// code as gemm_fast.hip's, in the same order: results are bit-identical to the 128-row kernels.
// Reference call sites: hf:models/bert_generation/modeling_bert_generation.py:104-106 (Q|K|V), :264-291 (MLP), :590-598 (LM head),
// hf:models/vit/modeling_vit.py:241-252 via ref:vilmedic/blocks/huggingface/decoder/decoder_model.py:39-49, ref:vilmedic/blocks/vision/visual_encoder.py:57,181.
#include <type_traits>
#include <utility>
#include "common.h"
#include "gemm_args.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ int p8_swz1(int krow) { return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3); }

// LDS-DMA of 16 B per lane: global address = SGPR base + 32-bit lane offset; LDS address = M0 + 16 * lane.  Invisible to hipcc's
// waitcnt pass on purpose: the main loop orders reads behind the DMA with its own counted s_waitcnt vmcnt + s_barrier.
__device__ __forceinline__ void glds16_s(const bf16_t* sbase, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds)) : "memory", "m0");
}

__device__ __forceinline__ void p8_st16(void* p, uint4 v) {
    const uint4_t vv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(vv, reinterpret_cast<uint4_t*>(p));
}

__device__ __forceinline__ int p8_xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// raw barrier (no fence, no vmcnt drain: LDS-DMA and global stores stay in flight across it), opaque to both the IR and the machine scheduler
#define P8_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)

template <class F, int... Is>
__device__ __forceinline__ void p8_static_for_impl(F& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void p8_static_for(F&& f) { p8_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int P8_BN = 256;
constexpr int P8_EPI_BYTES = 32 * P8_BN * 4;      // one 16-row fragment row of both wave groups, fp32

template <int LA, int LB, int MF, int NPH, int EPI, int OPS>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MFA = (MF + 1) / 2, MFB = MF / 2;      // A fragments of the two register sub-tiles
    constexpr int WR = MF * 16;                          // rows of a wave group
    constexpr int BM = 2 * WR;
    constexpr int A_HALF = WR * 128, B_HALF = 128 * 128; // bytes of a half-tile (64 k deep)
    constexpr int SLOT = 2 * A_HALF + 2 * B_HALF;
    constexpr int NQA = A_HALF / 1024;                   // DMA instructions (1 KiB) per A half-tile; B: 16
    static_assert(LA == 0 || MF == 8, "k-major A needs the 128-row half-tile");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, g = lane >> 4, c = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const int tiles = p.tiles_m * p.tiles_n;
    const int total = tiles * ((p.ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split);

    // ---- per-lane pieces that do not depend on the tile
    // fragment read offsets (same image and swizzles as gemm_fast.hip: layout 0 [rows][64 k] chunk ^= row & 7 for ds_read_b128;
    // layout 1 [64 k][128] chunk ^= swz1(k) for ds_read_b64_tr_b16)
    const int j4 = c >> 2, s1 = (j4 << 1) | ((g & 1) << 3), sub1 = (c & 1) * 8, h1 = (c & 3) >> 1;
    auto read0 = [&](const char* tile, int rbase, int i, int kk) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(tile + (rbase + i * 16 + c) * 128 + (((kk * 4 + g) ^ (c & 7)) << 4));
    };
    auto read1 = [&](const char* tile, int rbase, int i, int kk) -> bf16x8_t {
        const int krow = kk * 32 + 8 * g + j4;
        const int lc = (rbase >> 3) + 2 * i + h1;
        const int off = ((lc ^ s1) << 4) + sub1;
        v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(tile + krow * 256 + off));
        v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(tile + (krow + 4) * 256 + off));
        short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };
    const int b_rb = (wn & 1) * 64;                       // this wave's 64 columns inside its 128-column B half

    int work = (int)blockIdx.x;                           // persistent: block b walks tiles b, b + grid, ... (same XCD every round)
    bool staged = false;                                  // PERSIST: the prologue DMA of this tile was issued before the previous epilogue
    int young = 0;                                        // EPI 1: store instructions of the previous epilogue issued behind that DMA (-1: unknown)
    // DMA state of the tile being staged (set by tile_setup)
    uint32_t offA[2][2], offB[2][2];                      // [half][instruction] byte offsets from the SGPR bases
    const bf16_t* pA = nullptr; const bf16_t* pB = nullptr;   // SGPR bases of the next K-tile to request
    int64_t stepA = LA == 0 ? 64 : (int64_t)64 * p.lda, stepB = LB == 0 ? 64 : (int64_t)64 * p.ldb;

    auto decode = [&](int bid, int& tm, int& tn, int& split) {
        split = bid / tiles;
        const int t = bid - split * tiles;
        const int gw = p.group_w, per_group = p.tiles_m * gw;
        const int grp = t / per_group, rr = t - grp * per_group;
        const int w = min(gw, p.tiles_n - grp * gw);
        tm = rr / w; tn = grp * gw + (rr - tm * w);
    };
    auto tile_setup = [&](int m0, int n0, int kt_begin) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = wave + 8 * i;
                if (LA == 0) {
                    const int row = 8 * q + (lane >> 3);
                    const int lc = (lane & 7) ^ (row & 7);
                    const int gr = min(m0 + h * WR + row, p.M - 1) - m0;
                    offA[h][i] = (uint32_t)(gr * (int)p.lda * 2 + lc * 16);
                } else {
                    const int krow = 4 * q + (lane >> 4);
                    const int lc = (lane & 15) ^ p8_swz1(krow);
                    int col = h * 128 + lc * 8;
                    if (m0 + col >= p.M) col = 0;
                    offA[h][i] = (uint32_t)(krow * (int)p.lda * 2 + col * 2);
                }
                if (LB == 0) {
                    const int row = 8 * q + (lane >> 3);
                    const int lc = (lane & 7) ^ (row & 7);
                    const int gr = min(n0 + h * 128 + row, p.N - 1) - n0;
                    offB[h][i] = (uint32_t)(gr * (int)p.ldb * 2 + lc * 16);
                } else {
                    const int krow = 4 * q + (lane >> 4);
                    const int lc = (lane & 15) ^ p8_swz1(krow);
                    int col = h * 128 + lc * 8;
                    if (n0 + col >= p.N) col = 0;
                    offB[h][i] = (uint32_t)(krow * (int)p.ldb * 2 + col * 2);
                }
            }
        pA = p.A + (LA == 0 ? (int64_t)m0 * p.lda : (int64_t)m0) + kt_begin * stepA;
        pB = p.B + (LB == 0 ? (int64_t)n0 * p.ldb : (int64_t)n0) + kt_begin * stepB;
    };
    auto stageA = [&](int slot, int h) {                  // this wave's share of A half-tile h of K-tile pA -> LDS slot
        const uint32_t dst = lds0 + slot * SLOT + h * A_HALF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = wave + 8 * i;
            if (NQA == 16 || q < NQA) glds16_s(pA, offA[h][i], dst + q * 1024);
        }
    };
    auto stageB = [&](int slot, int h) {
        const uint32_t dst = lds0 + slot * SLOT + 2 * A_HALF + h * B_HALF;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_s(pB, offB[h][i], dst + (wave + 8 * i) * 1024);
    };
    auto prologue = [&](int nk) {                          // A(0), B(0) -> slot 0; B(1) -> slot 1
        stageA(0, 0); stageA(0, 1); pA += stepA;
        stageB(0, 0); stageB(0, 1); pB += stepB;
        if (nk > 1) { stageB(1, 0); stageB(1, 1); pB += stepB; }
    };

    while (work < total) {
        const int bid = p8_xcd_remap(work, total);
        int tm, tn, split;
        decode(bid, tm, tn, split);
        const int m0 = tm * BM, n0 = tn * P8_BN;
        const int kt_begin = split * p.ktiles_per_split;
        int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);
        if (p.dbg == 2) kt_end = kt_begin + 1;
        const int nk = kt_end - kt_begin;
        if (nk <= 0) { work += (int)gridDim.x; continue; }

        if (!staged) { tile_setup(m0, n0, kt_begin); prologue(nk); young = 0; }
        // A(0), B(0) landed; B(1) (4 requests) and -- EPI 1 -- the previous tile's stores (issued behind this tile's prologue DMA, counted exactly on
        // whole tiles) may stay in flight: the memory counter retires in order, so "at most 4 + young outstanding" means the older DMA is done
        if (nk > 1 && young == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (EPI == 1 && nk > 1 && young == 2 * MF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + 2 * MF) : "memory");
        else if (EPI == 1 && nk > 1 && young == 4 * MF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + 4 * MF) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        P8_BARRIER();

        float4_t acc[4][MF];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MF; ++i) acc[j][i] = (float4_t){0.f, 0.f, 0.f, 0.f};
        bf16x8_t fa[2][MFA], fb[2][4];

        auto ktile = [&](int t, auto ha_c, auto hb_c) {
            constexpr bool HA = decltype(ha_c)::value, HB = decltype(hb_c)::value;
            const int s = t & 1;
            const char* As = smem + s * SLOT + wm * A_HALF;
            const char* Bs = smem + s * SLOT + 2 * A_HALF + (wn >> 1) * B_HALF;
            // ---- phase 0: A sub-tile 0 and B columns 0..31; A_lo(t+1) -> other slot
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[kk][j] = LB == 0 ? read0(Bs, b_rb, j, kk) : read1(Bs, b_rb, j, kk);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFA; ++i) fa[kk][i] = LA == 0 ? read0(As, 0, i, kk) : read1(As, 0, i, kk);
            if constexpr (HA) stageA(s ^ 1, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            P8_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFA; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            P8_BARRIER();
            // ---- phase 1: B columns 32..63; A_hi(t+1)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 2; j < 4; ++j) fb[kk][j] = LB == 0 ? read0(Bs, b_rb, j, kk) : read1(Bs, b_rb, j, kk);
            if constexpr (HA) { stageA(s ^ 1, 1); pA += stepA; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            P8_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFA; ++i)
#pragma unroll
                    for (int j = 2; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            P8_BARRIER();
            // ---- phase 2: A sub-tile 1; the B halves of THIS slot are dead (both register sub-tiles hold them): B_lo(t+2)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFB; ++i) fa[kk][i] = LA == 0 ? read0(As, 0, MFA + i, kk) : read1(As, 0, MFA + i, kk);
            if constexpr (HB) stageB(s, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            P8_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFB; ++i)
#pragma unroll
                    for (int j = 2; j < 4; ++j) acc[j][MFA + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[j][MFA + i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            P8_BARRIER();
            // ---- phase 3: no reads; B_hi(t+2); everything of K-tile t+1 must have landed: only B(t+2) (4 instructions) may be in flight
            if constexpr (HB) { stageB(s, 1); pB += stepB; asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            P8_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFB; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j][MFA + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[j][MFA + i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            P8_BARRIER();
        };

        // NPH == 2: the same K-tile as TWO phases of 32 MFMAs (A sub-tile x all four B fragments): half the barriers per K-tile; a LOAD
        // interval (16 / 8 LDS reads + 4 DMA pieces) still fits beside the partner's 32 MFMAs
        auto ktile2 = [&](int t, auto ha_c, auto hb_c) {
            constexpr bool HA = decltype(ha_c)::value, HB = decltype(hb_c)::value;
            const int s = t & 1;
            const char* As = smem + s * SLOT + wm * A_HALF;
            const char* Bs = smem + s * SLOT + 2 * A_HALF + (wn >> 1) * B_HALF;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[kk][j] = LB == 0 ? read0(Bs, b_rb, j, kk) : read1(Bs, b_rb, j, kk);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFA; ++i) fa[kk][i] = LA == 0 ? read0(As, 0, i, kk) : read1(As, 0, i, kk);
            if constexpr (HA) { stageA(s ^ 1, 0); stageA(s ^ 1, 1); pA += stepA; }      // A of this slot's partner: last read in phase 1 of K-tile t-1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            P8_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            P8_BARRIER();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFB; ++i) fa[kk][i] = LA == 0 ? read0(As, 0, MFA + i, kk) : read1(As, 0, MFA + i, kk);
            // B of THIS slot is dead (fb holds it): B(t+2); then everything of K-tile t+1 must have landed -- only B(t+2) may be in flight
            if constexpr (HB) { stageB(s, 0); stageB(s, 1); pB += stepB; asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            P8_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MFB; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][MFA + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[j][MFA + i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            P8_BARRIER();
        };
        auto ktile_any = [&](int t, auto ha_c, auto hb_c) { if constexpr (NPH == 2) ktile2(t, ha_c, hb_c); else ktile(t, ha_c, hb_c); };

        if (wm == 1) P8_BARRIER();                         // group 1 runs one barrier behind group 0
        int t = 0;
        for (; t + 2 < nk; ++t) ktile_any(t, std::true_type{}, std::true_type{});
        if (t + 1 < nk) { ktile_any(t, std::true_type{}, std::false_type{}); ++t; }
        ktile_any(t, std::false_type{}, std::false_type{});
        if (wm == 0) P8_BARRIER();

        // ---- next tile's first K-tiles: the ring is idle during the epilogue.  Issued from INSIDE the epilogue, behind its bias / operand
        // loads: a compiler-generated wait in front of it (it counts neither the DMA nor their age) would otherwise wait for the DMA itself.
        const int next = work + (int)gridDim.x;
        staged = false;
        auto stage_next = [&]() {
            if (next < total && p.dbg != 2) {
                int tm2, tn2, sp2;
                decode(p8_xcd_remap(next, total), tm2, tn2, sp2);
                const int kb2 = sp2 * p.ktiles_per_split;
                const int nk2 = min(p.ktiles, kb2 + p.ktiles_per_split) - kb2;
                if (nk2 > 0) { tile_setup(tm2 * BM, tn2 * P8_BN, kb2); prologue(nk2); staged = true; }
            }
        };

        if constexpr (EPI == 1) {
        // ---- wave-private epilogue [r6].  Every wave stages ITS OWN 16 x 64 fragment row (4 KB of the 32-KB region) as fp32 and reads it back as
        // whole 128-B output rows: no workgroup barrier anywhere in the epilogue (the staged form above needs 2 MF of them and moves one fragment
        // row of all eight waves per step), MF fully unrolled passes of [4 ds_write_b128, 4 ds_read_b128, the element-wise chain on 16 values per
        // lane, 2 (4 with the z side output) 16-B stores per lane = 8 rows x 128 B per instruction].  The next tile's prologue DMA goes out FIRST
        // (the ring is idle); the stores are younger than it and are left in flight across the tile boundary (see the wait at the loop head).
        // Per-element operands (gelu'(z) input or residual) arrive in two halves: the first is waited for before the DMA is issued, the second is
        // requested behind the DMA and waited for with a counted vmcnt that leaves the first half's stores in flight.
        // The arithmetic and its order are those of the staged epilogue: bit-identical results.
            const vm_gemm_epilogue& e = p.e;
            float alpha = e.alpha_dev ? e.alpha * (*e.alpha_dev) : e.alpha;
            int lane_e = lane;                                 // opaque copy: nothing derived from it is hoisted above (and kept live through) the main loop
            asm volatile("" : "+v"(lane_e));
            const int c_e = lane_e & 15, g_e = lane_e >> 4, q8 = lane_e & 7, r8 = lane_e >> 3;
            float* cs = reinterpret_cast<float*>(smem + 2 * SLOT) + wave * 1024;      // [16 rows][64 fp32], 16-B chunk ^= row
            const int gn = n0 + wn * 64 + q8 * 8;
            const bool col_ok = gn < p.N;
            const int nvalid = min(8, p.N - gn);
            const bool vec = nvalid == 8;
            const int gn_safe = col_ok ? gn : 0;
            const int mw = m0 + wm * WR;
            const bool whole = m0 + BM <= p.M && n0 + P8_BN <= p.N;
            const bf16_t* zsrc = p.slabs ? nullptr : reinterpret_cast<const bf16_t*>(e.mul_gelu_z);
            const bf16_t* rsrc = p.slabs ? nullptr : reinterpret_cast<const bf16_t*>(e.residual);
            const bf16_t* osrc = zsrc ? zsrc : rsrc;           // ONE pre-loaded operand (a launch with both reads the residual in place)
            const int64_t ldo = zsrc ? p.ldc : e.ldr;
            const bool aux = e.aux_out != nullptr && !p.slabs;
            auto ldg16 = [](const void* src, uint4_t& dst) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory"); };
            constexpr int HP = (MF + 1) / 2;                   // passes per operand half
            constexpr int NO = OPS ? HP : 1;
            uint4_t bq[2] = {(uint4_t){0u, 0u, 0u, 0u}, (uint4_t){0u, 0u, 0u, 0u}};
            uint4_t oq[NO][2];                                  // ONE register set: the second half is loaded into it once the first is consumed
#pragma unroll
            for (int pp = 0; pp < NO; ++pp)
#pragma unroll
                for (int it = 0; it < 2; ++it) oq[pp][it] = (uint4_t){0u, 0u, 0u, 0u};
            auto issue_ops = [&](auto h_c) {
                constexpr int H = decltype(h_c)::value;
                if constexpr (OPS != 0) {
#pragma unroll
                    for (int pp = 0; pp < HP; ++pp) {
                        if (H * HP + pp >= MF) continue;
#pragma unroll
                        for (int it = 0; it < 2; ++it) {
                            const int gm = min(mw + (H * HP + pp) * 16 + it * 8 + r8, p.M - 1);
                            ldg16(osrc + (int64_t)gm * ldo + gn_safe, oq[pp][it]);
                        }
                    }
                }
            };
            const bool bias_on = e.bias != nullptr && !p.slabs;
            if (bias_on) { ldg16(e.bias + gn_safe, bq[0]); ldg16(e.bias + (vec ? gn + 4 : gn_safe), bq[1]); }
            issue_ops(std::integral_constant<int, 0>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(bq[0]), "+v"(bq[1]));
            if constexpr (OPS != 0) {
#pragma unroll
                for (int pp = 0; pp < HP; ++pp) asm volatile("" : "+v"(oq[pp][0]), "+v"(oq[pp][1]));
            }
            stage_next();
            float bias8[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { bias8[r] = __uint_as_float(bq[0][r]); bias8[4 + r] = vec ? __uint_as_float(bq[1][r]) : 0.f; }
            if (bias_on && !vec) {
#pragma unroll
                for (int r = 0; r < 8; ++r) bias8[r] = 0.f;
                if (col_ok) for (int r = 0; r < nvalid; ++r) bias8[r] = e.bias[gn + r];
            }
            const DropKey dkey = drop_key(eff_seed(e.dropout_seed, e.dropout_seed_dev));
            auto pass = [&](const int I, const float4_t (&t)[4], const uint4_t (&oqi)[2]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int chunk = (j * 4 + g_e) ^ c_e;
                    *reinterpret_cast<float4*>(cs + c_e * 64 + chunk * 4) = make_float4(t[j][0] * alpha, t[j][1] * alpha, t[j][2] * alpha, t[j][3] * alpha);
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int r = it * 8 + r8;
                    const int gm = mw + I * 16 + r;
                    float v[8];
                    {
                        const float4 lo = *reinterpret_cast<const float4*>(cs + r * 64 + (((2 * q8) ^ r) << 2));
                        const float4 hi = *reinterpret_cast<const float4*>(cs + r * 64 + (((2 * q8 + 1) ^ r) << 2));
                        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                    }
                    if (gm >= p.M || !col_ok) continue;
                    if (p.dbg == 3) { if (v[0] == 12345.678f) reinterpret_cast<float*>(p.C)[0] = v[0]; continue; }      // timing experiment: everything but the stores
                    const int64_t off = (int64_t)gm * p.ldc + gn;
                    if (p.slabs) {
                        float* sp = p.slabs + (int64_t)split * p.M * p.ldc + off;
                        if (vec) {
                            *reinterpret_cast<float4*>(sp) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(sp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        } else for (int r2 = 0; r2 < nvalid; ++r2) sp[r2] = v[r2];
                        continue;
                    }
#pragma unroll
                    for (int r2 = 0; r2 < 8; ++r2) v[r2] += bias8[r2];
                    if (aux) {
                        bf16_t* z = reinterpret_cast<bf16_t*>(e.aux_out) + off;
                        if (vec) p8_st16(z, pack8(v));
                        else for (int r2 = 0; r2 < nvalid; ++r2) z[r2] = f32_to_bf16(v[r2]);
                    }
                    if (e.act == 1) {
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) v[r2] = gelu_f(v[r2]);
                    }
                    const uint4 oqv = make_uint4(oqi[it][0], oqi[it][1], oqi[it][2], oqi[it][3]);
                    if (zsrc) {
                        float zf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (vec) unpack8(oqv, zf);
                        else for (int r2 = 0; r2 < nvalid; ++r2) zf[r2] = bf16_to_f32(zsrc[off + r2]);
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) v[r2] *= gelu_grad_f(zf[r2]);
                    }
                    if (e.dropout_p > 0.f) {
                        bool keep[8];
                        dropout_keep_n<8>(dkey, (uint64_t)gm * (uint64_t)p.N + (uint64_t)gn, p.drop_thresh, keep);
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) v[r2] = keep[r2] ? v[r2] * p.drop_scale : 0.f;
                    }
                    if (rsrc) {
                        float rf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (vec && !zsrc) unpack8(oqv, rf);
                        else if (vec) unpack8(*reinterpret_cast<const uint4*>(rsrc + (int64_t)gm * e.ldr + gn), rf);
                        else for (int r2 = 0; r2 < nvalid; ++r2) rf[r2] = bf16_to_f32(rsrc[(int64_t)gm * e.ldr + gn + r2]);
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) v[r2] += rf[r2];
                    }
                    if (e.out_dtype == VM_BF16) {
                        bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + off;
                        if (vec) p8_st16(cp, pack8(v));
                        else for (int r2 = 0; r2 < nvalid; ++r2) cp[r2] = f32_to_bf16(v[r2]);
                    } else {
                        float* cp = reinterpret_cast<float*>(p.C) + off;
                        if (e.accumulate) {
                            if (vec) {
                                const float4 o0 = *reinterpret_cast<float4*>(cp), o1 = *reinterpret_cast<float4*>(cp + 4);
                                *reinterpret_cast<float4*>(cp) = make_float4(o0.x + v[0], o0.y + v[1], o0.z + v[2], o0.w + v[3]);
                                *reinterpret_cast<float4*>(cp + 4) = make_float4(o1.x + v[4], o1.y + v[5], o1.z + v[6], o1.w + v[7]);
                            } else for (int r2 = 0; r2 < nvalid; ++r2) cp[r2] += v[r2];
                        } else if (vec) {
                            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        } else for (int r2 = 0; r2 < nvalid; ++r2) cp[r2] = v[r2];
                    }
                }
            };
            const bool counted = whole && e.out_dtype == VM_BF16 && !p.slabs && p.dbg == 0;      // every store of the passes is issued by every wave: exact counts
            if (p.dbg == 1) { young = -1; work = next; continue; }      // timing experiment: no epilogue passes at all
            // ONE copy of the element-wise code: the passes are a runtime loop, the accumulator row and the operand pair of pass i are picked by a
            // wave-uniform branch chain.  (Fully unrolled -- MF x 2 copies of every epilogue variant, ~60 KB of code walked by eight free-running
            // waves -- the passes measured 20 us per tile against 2.6 us with the stores compiled out: instruction-cache misses, not stores.)
#pragma unroll 1
            for (int i = 0; i < MF; ++i) {
                if constexpr (OPS != 0 && MF > HP) {
                    if (i == HP) {                                   // second operand half into the same registers (one exposed round trip per tile; two
                        issue_ops(std::integral_constant<int, 1>{}); // register sets next to the 128 accumulators spill)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int pp = 0; pp < HP; ++pp) asm volatile("" : "+v"(oq[pp][0]), "+v"(oq[pp][1]));
                    }
                }
                float4_t t[4];
                uint4_t o2[2] = {(uint4_t){0u, 0u, 0u, 0u}, (uint4_t){0u, 0u, 0u, 0u}};
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = acc[j][0];
                p8_static_for<MF>([&](auto i_c) {
                    constexpr int I = decltype(i_c)::value;
                    if (I > 0 && i == I) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[j] = acc[j][I];
                    }
                    if constexpr (OPS != 0) {
                        if (i == I) { o2[0] = oq[I % HP][0]; o2[1] = oq[I % HP][1]; }
                    }
                });
                pass(i, t, o2);
            }
            young = counted ? (aux ? 4 * MF : 2 * MF) : -1;
        } else {
        // ---- epilogue: MF chunks of 32 rows (fragment row i of both groups) through the fp32 stage.
        // gfx950 has ONE counter for loads and stores: any wait for a load issued inside the chunk loop also waits for every older store, i.e.
        // a store round trip per chunk (measured: 8 us per tile).  So everything the loop consumes from HBM -- the bias, and ALL of this
        // thread's gelu'(z) / residual operands (2 x 16 B per chunk) -- is requested and waited for BEFORE the first store; the loop itself
        // then only issues stores and never waits on the memory counter.
        {   // (no data-dependent skip of the epilogue here: anything assigned under a divergent condition -- the next tile's DMA base pointers
            //  are -- becomes a VGPR value)
            const vm_gemm_epilogue& e = p.e;
            float alpha = e.alpha_dev ? e.alpha * (*e.alpha_dev) : e.alpha;
            float* cs = reinterpret_cast<float*>(smem + 2 * SLOT);
            // lane coordinates re-derived behind an opaque copy: everything the epilogue computes from them would otherwise be hoisted out of
            // the TILE loop and live (spilled) through the main loop
            int tid_e = tid, c_e = c, g_e = g;
            asm volatile("" : "+v"(tid_e), "+v"(c_e), "+v"(g_e));
            const int q = tid_e & 31, row_t = tid_e >> 5;      // 8 columns 8q..8q+7 of stage rows row_t and 16 + row_t
            const int gn = n0 + q * 8;
            const bool col_ok = gn < p.N;
            const int nvalid = min(8, p.N - gn);
            const bool vec = nvalid == 8;
            float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (e.bias && col_ok) {
                if (vec) {
                    const float4 b0 = *reinterpret_cast<const float4*>(e.bias + gn), b1 = *reinterpret_cast<const float4*>(e.bias + gn + 4);
                    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
                } else for (int r = 0; r < nvalid; ++r) bias8[r] = e.bias[gn + r];
            }
            const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(e.mul_gelu_z);
            const bf16_t* rsrc = reinterpret_cast<const bf16_t*>(e.residual);
            // the pre-loaded operand: gelu'(z) when present, else the residual (a launch with both keeps the residual on in-loop loads)
            const bf16_t* osrc = zsrc ? zsrc : rsrc;
            const int64_t ldo = zsrc ? p.ldc : e.ldr;
            constexpr int CG = 2;                                    // chunks per operand group (2 x 16 B per chunk and thread stay in registers)
            uint4_t opq[CG][2];
#pragma unroll
            for (int ii = 0; ii < CG; ++ii)
#pragma unroll
                for (int it = 0; it < 2; ++it) opq[ii][it] = (uint4_t){0u, 0u, 0u, 0u};
            // uses the compiler can see: its waits for these loads land HERE, not at their first use inside the loop
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(bias8[r]));
            asm volatile("" : "+v"(alpha));
            const DropKey dkey = drop_key(eff_seed(e.dropout_seed, e.dropout_seed_dev));
#pragma unroll 1
            for (int i0 = 0; i0 < MF; i0 += CG) {
            if (osrc && vec) {                                       // (the second group's wait also covers the first group's stores: one
#pragma unroll                                                       //  store round trip per tile, and only for launches with an operand)
                for (int ii = 0; ii < CG; ++ii)
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int gm = min(m0 + it * WR + (i0 + ii) * 16 + row_t, p.M - 1);
                        opq[ii][it] = *reinterpret_cast<const uint4_t*>(osrc + (int64_t)gm * ldo + gn);
                    }
#pragma unroll
                for (int ii = 0; ii < CG; ++ii)
#pragma unroll
                    for (int it = 0; it < 2; ++it) asm volatile("" : "+v"(opq[ii][it]));
            }
            if (i0 == 0) stage_next();
#pragma unroll 1
            for (int i = i0; i < min(i0 + CG, MF); ++i) {            // runtime loop: one copy of the element-wise code
                uint4_t oq[2] = {opq[0][0], opq[0][1]};
#pragma unroll
                for (int ii = 1; ii < CG; ++ii)
                    if (ii == i - i0) { oq[0] = opq[ii][0]; oq[1] = opq[ii][1]; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous chunk's stage reads are done (raw barriers: the
                P8_BARRIER();                                        // stores of earlier chunks stay in flight)
                {
                    const int r = wm * 16 + c_e;
#pragma unroll
                    for (int ii = 0; ii < MF; ++ii) {               // wave-uniform branch chain: the accumulator index stays a constant
                        if (ii != i) continue;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int chunk = (wn * 16 + j * 4 + g_e) ^ (r & 7);
                            *reinterpret_cast<float4*>(cs + r * P8_BN + chunk * 4) =
                                make_float4(acc[j][ii][0] * alpha, acc[j][ii][1] * alpha, acc[j][ii][2] * alpha, acc[j][ii][3] * alpha);
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                P8_BARRIER();
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int r = it * 16 + row_t;
                    const int gm = m0 + it * WR + i * 16 + row_t;
                    if (gm >= p.M || !col_ok) continue;
                    const int64_t off = (int64_t)gm * p.ldc + gn;
                    float v[8];
                    {
                        const float4 lo = *reinterpret_cast<const float4*>(cs + r * P8_BN + (((2 * q) ^ (r & 7)) << 2));
                        const float4 hi = *reinterpret_cast<const float4*>(cs + r * P8_BN + (((2 * q + 1) ^ (r & 7)) << 2));
                        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                    }
                    if (p.slabs) {
                        float* sp = p.slabs + (int64_t)split * p.M * p.ldc + off;
                        if (vec) {
                            *reinterpret_cast<float4*>(sp) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(sp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        } else for (int r2 = 0; r2 < nvalid; ++r2) sp[r2] = v[r2];
                        continue;
                    }
#pragma unroll
                    for (int r2 = 0; r2 < 8; ++r2) v[r2] += bias8[r2];
                    if (e.aux_out) {
                        bf16_t* z = reinterpret_cast<bf16_t*>(e.aux_out) + off;
                        if (vec) p8_st16(z, pack8(v));
                        else for (int r2 = 0; r2 < nvalid; ++r2) z[r2] = f32_to_bf16(v[r2]);
                    }
                    if (e.act == 1) {
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) v[r2] = gelu_f(v[r2]);
                    }
                    const uint4 oqv = make_uint4(oq[it][0], oq[it][1], oq[it][2], oq[it][3]);
                    if (zsrc) {
                        float zf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (vec) unpack8(oqv, zf);
                        else for (int r2 = 0; r2 < nvalid; ++r2) zf[r2] = bf16_to_f32(zsrc[off + r2]);
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) v[r2] *= gelu_grad_f(zf[r2]);
                    }
                    if (e.dropout_p > 0.f) {
                        bool keep[8];
                        dropout_keep_n<8>(dkey, (uint64_t)gm * (uint64_t)p.N + (uint64_t)gn, p.drop_thresh, keep);
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) v[r2] = keep[r2] ? v[r2] * p.drop_scale : 0.f;
                    }
                    if (rsrc) {
                        float rf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (vec && !zsrc) unpack8(oqv, rf);
                        else if (vec) unpack8(*reinterpret_cast<const uint4*>(rsrc + (int64_t)gm * e.ldr + gn), rf);
                        else for (int r2 = 0; r2 < nvalid; ++r2) rf[r2] = bf16_to_f32(rsrc[(int64_t)gm * e.ldr + gn + r2]);
#pragma unroll
                        for (int r2 = 0; r2 < 8; ++r2) v[r2] += rf[r2];
                    }
                    if (e.out_dtype == VM_BF16) {
                        bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + off;
                        if (vec) p8_st16(cp, pack8(v));
                        else for (int r2 = 0; r2 < nvalid; ++r2) cp[r2] = f32_to_bf16(v[r2]);
                    } else {
                        float* cp = reinterpret_cast<float*>(p.C) + off;
                        if (e.accumulate) {
                            if (vec) {
                                const float4 o0 = *reinterpret_cast<float4*>(cp), o1 = *reinterpret_cast<float4*>(cp + 4);
                                *reinterpret_cast<float4*>(cp) = make_float4(o0.x + v[0], o0.y + v[1], o0.z + v[2], o0.w + v[3]);
                                *reinterpret_cast<float4*>(cp + 4) = make_float4(o1.x + v[4], o1.y + v[5], o1.z + v[6], o1.w + v[7]);
                            } else for (int r2 = 0; r2 < nvalid; ++r2) cp[r2] += v[r2];
                        } else if (vec) {
                            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        } else for (int r2 = 0; r2 < nvalid; ++r2) cp[r2] = v[r2];
                    }
                }
            }
            }   // operand groups
        }
        }   // EPI
        work = next;
    }
}

template <int LA, int LB, int MF, int NPH, int EPI, int OPS>
int launch_p8(const GemmArgs& a, int total, hipStream_t s) {
    constexpr int LDS = 2 * (2 * MF * 16 * 128 + 2 * 128 * 128) + P8_EPI_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_kernel<LA, LB, MF, NPH, EPI, OPS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int grid = total < 256 ? total : 256;          // one workgroup per CU walks its tiles (block b: tiles b, b + 256, ...)
    hipLaunchKernelGGL((gemm_p8_kernel<LA, LB, MF, NPH, EPI, OPS>), dim3(grid), dim3(512), LDS, s, a);
    return vm_check_launch("vm_gemm_bf16(p8)");
}

template <int LA, int LB, int NPH, int EPI, int OPS>
int launch_p8_mf(const GemmArgs& a, int mf, int total, hipStream_t s) {
    switch (mf) {
        case 5: return launch_p8<LA, LB, 5, NPH, EPI, OPS>(a, total, s);
        case 6: return launch_p8<LA, LB, 6, NPH, EPI, OPS>(a, total, s);
        case 7: return launch_p8<LA, LB, 7, NPH, EPI, OPS>(a, total, s);
        default: return launch_p8<LA, LB, 8, NPH, EPI, OPS>(a, total, s);
    }
}

}  // namespace

// tile = (32 mf) x 256, mf in 5..8; phases = 2 or 4 barrier pairs per K-tile
int vm_gemm_p8_dispatch(const GemmArgs& a, int a_layout, int b_layout, int mf, int phases, int epi, int total, hipStream_t s) {
    if (epi == 1) {              // wave-private epilogue (two barrier pairs per K-tile only); OPS: a per-element operand (gelu'(z) input or residual) is pre-loaded
        const bool ops = !a.slabs && (a.e.mul_gelu_z || a.e.residual);
        if (a_layout == 0 && b_layout == 0) return ops ? launch_p8_mf<0, 0, 2, 1, 1>(a, mf, total, s) : launch_p8_mf<0, 0, 2, 1, 0>(a, mf, total, s);
        if (a_layout == 0 && b_layout == 1) return ops ? launch_p8_mf<0, 1, 2, 1, 1>(a, mf, total, s) : launch_p8_mf<0, 1, 2, 1, 0>(a, mf, total, s);
    }
    if (a_layout == 0 && b_layout == 0) return phases == 4 ? launch_p8_mf<0, 0, 4, 0, 1>(a, mf, total, s) : launch_p8_mf<0, 0, 2, 0, 1>(a, mf, total, s);
    if (a_layout == 0 && b_layout == 1) return phases == 4 ? launch_p8_mf<0, 1, 4, 0, 1>(a, mf, total, s) : launch_p8_mf<0, 1, 2, 0, 1>(a, mf, total, s);
    vm_set_error("vm_gemm_bf16(p8): layout (a=%d, b=%d) has no wide-tile kernel", a_layout, b_layout);
    return VM_EUNSUPPORTED;
}
