// gemm_fast.hip -- the tuned bf16 MFMA GEMM path (K % 64 == 0): LDS-DMA staging + XOR-swizzled LDS + register epilogue.
//
// Differences from the generic kernel in gemm.hip (kept as the any-shape fallback):
//   * operand tiles go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass); the LDS image
//     is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and undone on the read
//     (layout 0: 16-B chunk ^= row&7 for ds_read_b128; layout 1: chunk ^= ((k&3)<<1 | ((k>>3)&1)<<3) which makes the
//     16 lanes x 2 groups of a ds_read_b64_tr_b16 hit 16 distinct 16-B slots);
//   * the next K-tile's DMA is issued before the MFMA phase of the current one; one vmcnt(0)+barrier per K-tile;
//   * MFMA operands are swapped (D^T = B_frag x A_frag) so each lane ends up with 4 CONSECUTIVE output columns of one
//     row; the tile is then staged through LDS (fp32, chunk-swizzled, conflict-free 16-B writes) so that the epilogue
//     touches HBM only in whole 256-B rows.
// Out-of-range rows/columns are clamped to valid addresses (they only feed outputs that are never stored); the
// contraction dim has no tail by construction (K % 64 == 0), so no zero-fill is needed.
#include <type_traits>
#include "common.h"
#include "gemm_args.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));

#define FBK 64
// template parameters: WM x WN = waves along M x N, each wave owns a 64x64 sub-tile (tile = 64 WM x 64 WN):
//                        2x2 -> 128x128, 256 threads, 2 workgroups/CU;  4x2 -> 256x128, 512 threads;  4x4 -> 256x256, 1024 threads
//                      STAGES = LDS ring depth (2: one tile in flight, vmcnt(0)+barrier per K-tile;
//                                               3: two tiles in flight, COUNTED vmcnt + raw s_barrier so the DMA of
//                                                  tiles kt+1, kt+2 stays in flight across the barrier)

__device__ __forceinline__ int swz1(int krow) { return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3); }

template <int LAYOUT, int ROWS, int BKT>   // ROWS = tile extent along the non-contraction dim (128 or 256); BKT = k extent (32 / 64)
__device__ __forceinline__ const bf16_t* stage_src(const bf16_t* base, int64_t ld, int row0, int nrows, int q, int lane) {
    if (LAYOUT == 0 && BKT == 64) {          // tile [ROWS][64 k]: one DMA instruction = 8 rows x 128 B
        const int row = 8 * q + (lane >> 3);
        const int lc = (lane & 7) ^ (row & 7);
        const int gr = min(row0 + row, nrows - 1);
        return base + (int64_t)gr * ld + lc * 8;
    } else if (LAYOUT == 0) {                // tile [ROWS][32 k]: one DMA instruction = 16 rows x 64 B; 4 rows share a
        const int row = 16 * q + (lane >> 2);   // 256-B bank period, so the 16-B chunk is XORed with (row>>2)&3
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        const int gr = min(row0 + row, nrows - 1);
        return base + (int64_t)gr * ld + lc * 8;
    } else {                                 // tile [64 k][ROWS]: one DMA instruction = 1 KiB = 1024/(2*ROWS) k-rows
        constexpr int LPR = ROWS / 8;        // lanes (16-B chunks) per k-row: 16 or 32
        const int krow = (64 / LPR) * q + lane / LPR;
        const int lc = (lane % LPR) ^ swz1(krow);     // swizzle the low 4 chunk bits (256-B bank period)
        int col = row0 + lc * 8;
        if (col >= nrows) col = row0;
        return base + (int64_t)krow * ld + col;
    }
}

// 16-B output store, non-temporal: the bf16 outputs are streamed once and read by a LATER kernel; not allocating them
// in L2 leaves the operand panels resident (rocprofv3 FETCH_SIZE of a 12608x2304x768 launch: 54.0 -> 45.1 MB, +1.7 % rate).
__device__ __forceinline__ void st16(void* p, uint4 v) {
    const uint4_t vv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(vv, reinterpret_cast<uint4_t*>(p));
}

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Same DMA as glds16 but hidden from the compiler.  With the builtin, hipcc (ROCm 7.2) puts "s_waitcnt vmcnt(0)" in front of the first
// ds_read_b64_tr_b16 that follows a DMA issue (the transpose-read builtin carries no memory operand, so the waitcnt pass assumes it
// may alias the pending LDS write): the tile just requested is waited for at once and the pipeline is serial for every strided
// operand.  The inline-asm form is invisible to that pass; ordering is then entirely by the explicit s_waitcnt vmcnt + s_barrier of
// the main loop (MI355X_MICROARCH.md, "Two waves per SIMD" item 7: nothing else orders a ds_read behind an LDS-DMA anyway).
__device__ __forceinline__ void glds16_asm(const bf16_t* src, char* lds_dst) {
    const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst;      // wave-uniform
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory", "m0");
}

//                      PIPE = 1 (2-stage, k-tile 64 only): the fragments of BOTH 32-wide k-halves of a tile are requested right after
//                             the barrier and the MFMAs follow behind a scheduling barrier, so one LDS latency is exposed per
//                             K-tile instead of one per dependent read group of the compiler's own schedule
//                      PIPE = 4: cross-tile register pipeline with the barrier in the middle of the K-tile (see the main loop)
//                      BG   = 1: the workgroups of tile column 0 also produce the bias gradient sum_k A(m, k) (one extra MFMA per A
//                             fragment against an all-ones B fragment) -- the grouped weight-gradient launch
// ``bid`` = logical work item (XCD-remapped by the caller): split * tiles + tile
template <int LA, int LB, int WM, int WN, int STAGES, int BKT, int MF, int PIPE, int BG>
__device__ __forceinline__ void gemm_fast_body(const GemmArgs& p, const int bid, char* smem) {
    constexpr int WR = MF * 16;                     // rows per wave (MF 16-row A fragments; 4 or 5)
    constexpr int FBM = WM * WR, FBN = WN * 64, NW = WM * WN, NT = NW * 64;
    constexpr int F_OPER_A = FBM * BKT * 2, F_OPER_B = FBN * BKT * 2, F_STAGE = F_OPER_A + F_OPER_B;
    constexpr int NA_I = F_OPER_A / 1024 / NW, NB_I = F_OPER_B / 1024 / NW;   // A- / B-tile DMA instructions (1 KiB each) per wave
    static_assert(NA_I * NW * 1024 == F_OPER_A && NB_I * NW * 1024 == F_OPER_B, "tile must split evenly over the waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, g = lane >> 4, c = lane & 15;
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = bid / tiles;
    const int t = bid - split * tiles;
    // tile order: column groups of group_w tile columns, row-major inside a group.  An XCD runs a contiguous range of
    // ids (remap above), i.e. a (rows x group_w) rectangle: its B working set is group_w column tiles, not all of N.
    int tm, tn;
    {
        const int gw = p.group_w, per_group = p.tiles_m * gw;
        const int grp = t / per_group, rr = t - grp * per_group;
        const int w = min(gw, p.tiles_n - grp * gw);
        tm = rr / w; tn = grp * gw + (rr - tm * w);
    }
    const int m0 = tm * FBM, n0 = tn * FBN;
    constexpr int KSC = FBK / BKT;              // the host counts K-tiles of 64; a 32-wide k-tile kernel runs two per host tile
    const int kt_begin = split * p.ktiles_per_split * KSC;
    int kt_end = min(p.ktiles, split * p.ktiles_per_split + p.ktiles_per_split) * KSC;
    if (kt_begin >= kt_end) return;
    if (p.dbg == 2) kt_end = kt_begin + 1;

    // ---- per-thread DMA sources (4 instructions per operand per wave), advanced by one K-tile per iteration
    const bf16_t* srcA[NA_I];
    const bf16_t* srcB[NB_I];
    const int64_t stepA = LA == 0 ? BKT : (int64_t)BKT * p.lda;
    const int64_t stepB = LB == 0 ? BKT : (int64_t)BKT * p.ldb;
#pragma unroll
    for (int i = 0; i < NA_I; ++i) srcA[i] = stage_src<LA, FBM, BKT>(p.A, p.lda, m0, p.M, wave * NA_I + i, lane) + kt_begin * stepA;
#pragma unroll
    for (int i = 0; i < NB_I; ++i) srcB[i] = stage_src<LB, FBN, BKT>(p.B, p.ldb, n0, p.N, wave * NB_I + i, lane) + kt_begin * stepB;
    constexpr bool ASM_DMA = PIPE == 4 && (LA != 0 || LB != 0);
    auto stage = [&](int buf) {
        char* da = smem + buf * F_STAGE;
#pragma unroll
        for (int i = 0; i < NA_I; ++i) {
            if constexpr (ASM_DMA) glds16_asm(srcA[i], da + (wave * NA_I + i) * 1024); else glds16(srcA[i], da + (wave * NA_I + i) * 1024);
            srcA[i] += stepA;
        }
#pragma unroll
        for (int i = 0; i < NB_I; ++i) {
            if constexpr (ASM_DMA) glds16_asm(srcB[i], da + F_OPER_A + (wave * NB_I + i) * 1024); else glds16(srcB[i], da + F_OPER_A + (wave * NB_I + i) * 1024);
            srcB[i] += stepB;
        }
    };

    // ---- per-lane fragment read offsets
    // layout 0: addr = (rbase + i*16 + c)*128 + (((kk*4+g) ^ (c&7)) * 16)
    // layout 1: addr = krow*256 + ((lc ^ s)*16) + sub,  krow = kk*32 + 8g + (c>>2) (+4), lc = (rbase>>3) + 2i + ((c&3)>>1)
    const int a_rb = wm * WR, b_rb = wn * 64;
    const int j4 = c >> 2, s1 = (j4 << 1) | ((g & 1) << 3), sub1 = (c & 1) * 8, h1 = (c & 3) >> 1;

    auto read_frag0 = [&](const char* tile, int rbase, int i, int kk) -> bf16x8_t {
        if (BKT == 64) return *reinterpret_cast<const bf16x8_t*>(tile + (rbase + i * 16 + c) * 128 + (((kk * 4 + g) ^ (c & 7)) << 4));
        return *reinterpret_cast<const bf16x8_t*>(tile + (rbase + i * 16 + c) * 64 + ((g ^ ((c >> 2) & 3)) << 4));
    };
    auto read_frag1 = [&](const char* tile, int rbase, int i, int kk, int row_bytes) -> bf16x8_t {
        const int krow = kk * 32 + 8 * g + j4;
        const int lc = (rbase >> 3) + 2 * i + h1;
        const int off = ((lc ^ s1) << 4) + sub1;
        const __attribute__((address_space(3))) v4s* p0 = (const __attribute__((address_space(3))) v4s*)(tile + krow * row_bytes + off);
        const __attribute__((address_space(3))) v4s* p1 = (const __attribute__((address_space(3))) v4s*)(tile + (krow + 4) * row_bytes + off);
        v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p0);
        v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p1);
        short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };

    float4_t acc[4][MF];   // acc[j][i]: n-fragment j, m-fragment i  (D^T layout: lane -> row m = c, 4 consecutive n = 4g..4g+3)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[j][i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    float4_t accb[BG ? MF : 1];                     // bias gradient (BG): lane (c, g) ends with rowsum_k A(a_rb + 16 i + c, k) in all 4 slots
#pragma unroll
    for (int i = 0; i < (BG ? MF : 1); ++i) accb[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    const bool bias_wg = BG && p.bias_grad != nullptr && tn == 0 && split == 0;

    // ---- epilogue geometry, and the epilogue's global operands.  Every thread owns a fixed group of 8 columns (q) and rows
    // tid / QPR + it * (NT / QPR) of each staged pass: the per-column operand (bias) is loaded once, the per-element operands (residual,
    // gelu'(z) input) of ALL of the tile's items by issue_operands() -- called in front of the LAST K-tile's MFMAs (round 4), so that their
    // round trip runs under the main loop's tail instead of in front of the epilogue: with every workgroup of a round finishing at the same
    // time these reads were an exposed burst (26 us of the 102 us FC2-dgrad launch, and it was not the gelu' arithmetic: storing the
    // derivative in the forward pass and multiplying it in as is bought 5 us).
    const vm_gemm_epilogue& e = p.e;
    constexpr int RING_BYTES = STAGES * F_STAGE;
    constexpr int LDS_BYTES = RING_BYTES >= WR * FBN * 4 ? RING_BYTES : WR * FBN * 4;   // at least one wave-row of fp32 staging
    constexpr int RPP_RAW = LDS_BYTES / (FBN * 4);
    constexpr int RPP = RPP_RAW >= FBM ? FBM : (RPP_RAW / WR) * WR;   // rows staged per pass (multiple of a wave's WR rows)
    static_assert(FBM % RPP == 0 && (RPP * (FBN / 8)) % NT == 0, "epilogue pass geometry");
    constexpr int NPASS = FBM / RPP;
    constexpr int QPR = FBN / 8;                                     // 8-column items per row
    constexpr int ITEMS = RPP * QPR / NT;
    static_assert(NT % QPR == 0, "column group must be thread-invariant");
    constexpr int RSTEP = NT / QPR;
    const int q = tid % QPR, row_t = tid / QPR;
    const int gn = n0 + q * 8;
    const bool col_ok = gn < p.N;
    const int nvalid = min(8, p.N - gn);
    const bool vec = nvalid == 8;
    const bf16_t* zsrc = p.slabs ? nullptr : reinterpret_cast<const bf16_t*>(e.mul_gelu_z);
    const bf16_t* rsrc = p.slabs ? nullptr : reinterpret_cast<const bf16_t*>(e.residual);
    // The loads are inline asm: written as C++ loads, hipcc merged them with their first use and sank them behind the main loop (ISA: the
    // eight global_load_dwordx4 sat between the last MFMA and the staging barrier).  Invisible to the compiler's waitcnt pass, they are
    // completed by the explicit s_waitcnt vmcnt(0) at the head of the epilogue.
    uint4_t bq[2] = {(uint4_t){0u, 0u, 0u, 0u}, (uint4_t){0u, 0u, 0u, 0u}};
    // ONE per-element operand array: gelu'(z) input and residual never meet in a launch of the step (if they do, the residual is read in place)
    uint4_t oq[NPASS][ITEMS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) oq[ps][it] = (uint4_t){0u, 0u, 0u, 0u};
    auto ldg16 = [](const void* src, uint4_t& dst) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src)); };
    const bool bias_vec = e.bias && col_ok && !p.slabs && vec;
    const bool late_ops = p.dbg == 4;          // VM_GEMM_DEBUG=4: operands requested behind the main loop, as before round 4 (A/B)
    const bf16_t* osrc = zsrc ? zsrc : rsrc;
    const int64_t ldo = zsrc ? p.ldc : e.ldr;
    auto issue_operands = [&]() {
        if (bias_vec) { ldg16(e.bias + gn, bq[0]); ldg16(e.bias + gn + 4, bq[1]); }
        if (osrc && vec) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const int gm = min(m0 + ps * RPP + row_t + it * RSTEP, p.M - 1);
                    ldg16(osrc + (int64_t)gm * ldo + gn, oq[ps][it]);
                }
        }
    };

    auto compute = [&](int buf) {
        const char* sa = smem + buf * F_STAGE;
        const char* sb = sa + F_OPER_A;
        if constexpr (PIPE == 1 && BKT == 64) {
            // software-pipelined K-tile: the first 32-wide k-half's fragments are requested up front, the second half's
            // reads are slotted one per MFMA behind the first MFMAs of the first half (sched_group_barrier: 0x100 = LDS read, 0x008 = MFMA),
            // so only the first reads' latency is exposed per K-tile
            bf16x8_t fa[2][MF], fb[2][4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fb[kk][j] = LB == 0 ? read_frag0(sb, b_rb, j, kk) : read_frag1(sb, b_rb, j, kk, FBN * 2);
                    if (j < MF) fa[kk][j] = LA == 0 ? read_frag0(sa, a_rb, j, kk) : read_frag1(sa, a_rb, j, kk, FBM * 2);
                }
#pragma unroll
                for (int i = 4; i < MF; ++i) fa[kk][i] = LA == 0 ? read_frag0(sa, a_rb, i, kk) : read_frag1(sa, a_rb, i, kk, FBM * 2);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int d = 0; d < 4 + MF - 1; ++d)          // anti-diagonal order: the first MFMAs need only the first reads
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = d - j;
                        if (i >= 0 && i < MF) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], acc[j][i], 0, 0, 0);
                    }
            constexpr int NRD = (LA == 0 ? MF : 2 * MF) + (LB == 0 ? 4 : 8);      // LDS read instructions per k-half
            constexpr int NMF = 4 * MF;
            static_assert(NRD <= NMF, "one read slot per first-half MFMA");
            __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
            for (int n = 0; n < NRD; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NMF - NRD, 0);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < BKT / 32; ++kk) {
            bf16x8_t fa[MF], fb[4];
#pragma unroll
            for (int i = 0; i < MF; ++i) fa[i] = LA == 0 ? read_frag0(sa, a_rb, i, kk) : read_frag1(sa, a_rb, i, kk, FBM * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = LB == 0 ? read_frag0(sb, b_rb, j, kk) : read_frag1(sb, b_rb, j, kk, FBN * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MF; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
        }
    };
    if constexpr (PIPE == 4) {
        // Cross-tile register pipeline.  Two fragment sets live in registers: set kk holds the kk-th 32-wide k-half of a K-tile.
        // Per K-tile t (LDS slot t & 1):
        //   phase A:  MFMAs of set 0 (tile t), with the LDS reads of set 1 (tile t) slotted behind them one per MFMA;
        //   s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier in the MIDDLE of the tile: tile t+1 (requested a whole K-tile ago) has
        //             landed for every wave, and every wave has drained its reads of slot t & 1 -> that slot is free;
        //   phase C:  tile t+2 is requested into the freed slot, and the MFMAs of set 1 (tile t) run with the reads of set 0
        //             of tile t+1 slotted behind them.
        // So no LDS latency is exposed behind the barrier (the MFMAs that follow it have their operands in registers), every
        // DMA has a full K-tile of MFMA time to land, and there is ONE barrier per K-tile.  The DMA pieces go out as a burst right
        // behind the barrier (slotting them one per MFMA behind the fragment reads measured equal: profiles/r02_b_gemm_pipe4_asmdma_ab.txt).
        static_assert(STAGES == 2 && BKT == 64, "cross-tile register pipeline: 2 LDS slots of 64-wide k-tiles");
        constexpr int NLD = NA_I + NB_I, RA = LA == 0 ? 1 : 2, RB = LB == 0 ? 1 : 2;
        constexpr int NRD = MF * RA + 4 * RB;                       // LDS read instructions per k-half
        static_assert(NRD <= 8 * MF, "at most two read slots per MFMA");
        const int nk = kt_end - kt_begin;
        bf16x8_t fa[2][MF], fb[2][4];
        auto read_half = [&](const char* sa, auto kk_c) {
            constexpr int KK = decltype(kk_c)::value;
            const char* sb = sa + F_OPER_A;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                fb[KK][j] = LB == 0 ? read_frag0(sb, b_rb, j, KK) : read_frag1(sb, b_rb, j, KK, FBN * 2);
                if (j < MF) fa[KK][j] = LA == 0 ? read_frag0(sa, a_rb, j, KK) : read_frag1(sa, a_rb, j, KK, FBM * 2);
            }
#pragma unroll
            for (int i = 4; i < MF; ++i) fa[KK][i] = LA == 0 ? read_frag0(sa, a_rb, i, KK) : read_frag1(sa, a_rb, i, KK, FBM * 2);
        };
        auto mid_tile_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);       // MFMAs are register-only: without this the scheduler sinks phase A's tail below
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the wait and exposes the last reads' latency
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        // WB = true: this workgroup also accumulates accb[i] = sum_k A(row, k) (bias gradient of a weight-gradient GEMM)
        auto main_loop = [&](auto wb_c) {
            constexpr bool WB = decltype(wb_c)::value;
            constexpr int NMF = 4 * MF + (WB ? MF : 0);
            const short8_t ones_s = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
            const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);
            auto mfma_half = [&](auto kk_c) {
                constexpr int KK = decltype(kk_c)::value;
#pragma unroll
                for (int i = 0; i < MF; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[KK][j], fa[KK][i], acc[j][i], 0, 0, 0);
                    if constexpr (WB) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[KK][i], accb[i], 0, 0, 0);
                }
            };
            auto slot_reads = [&]() {                               // NRD reads behind the first MFMAs, then the remaining MFMAs
                if constexpr (NRD <= NMF) {
#pragma unroll
                    for (int n = 0; n < NRD; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
                } else {                                            // (the 64-row tile with a transposed operand: 10 reads for 8 MFMAs) two reads per slot
                    constexpr int PAIRS = NRD / 2;                  // (constant arguments only: PAIRS slots of two reads, one odd read, the rest)
#pragma unroll
                    for (int n = 0; n < PAIRS; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    if constexpr (NRD & 1) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NMF - PAIRS - (NRD & 1), 0);
                }
            };
            stage(0);
            if (nk > 1) {
                stage(1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            read_half(smem, K0{});
            int t = 0;
            for (; t + 2 < nk; ++t) {                               // steady state: there is a tile t+2 to request
                const int slot = t & 1;
                const char* cur = smem + slot * F_STAGE;
                const char* nxt = smem + (slot ^ 1) * F_STAGE;
                read_half(cur, K1{});
                mfma_half(K0{});
                slot_reads();
                mid_tile_barrier();
                stage(slot);                                        // DMA pieces as one burst
                __builtin_amdgcn_sched_barrier(0);
                read_half(nxt, K0{});
                mfma_half(K1{});
                slot_reads();
            }
            if (t + 1 < nk) {                                       // second-to-last tile: nothing left to request
                const char* cur = smem + (t & 1) * F_STAGE;
                const char* nxt = smem + ((t & 1) ^ 1) * F_STAGE;
                read_half(cur, K1{});
                mfma_half(K0{});
                slot_reads();
                mid_tile_barrier();
                if (!late_ops) issue_operands();                    // (behind the loop's last vmcnt wait: 1.5 K-tiles of MFMAs to land under)
                __builtin_amdgcn_sched_barrier(0);
                read_half(nxt, K0{});
                mfma_half(K1{});
                slot_reads();
                ++t;
            }
            if (nk == 1 && !late_ops) issue_operands();
            {                                                       // last tile
                const char* cur = smem + (t & 1) * F_STAGE;
                read_half(cur, K1{});
                mfma_half(K0{});
                slot_reads();
                mfma_half(K1{});
            }
        };
        if constexpr (BG != 0) {
            if (bias_wg) main_loop(std::true_type{});
            else main_loop(std::false_type{});
        } else {
            main_loop(std::false_type{});
        }
    } else if (STAGES == 2) {
        stage(0);
        for (int kt = kt_begin; kt + 1 < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            __syncthreads();                   // (compiler adds vmcnt(0)): tile kt landed for every wave; buf^1 is free
            stage(buf ^ 1);
            compute(buf);
        }
        __syncthreads();                       // last K-tile (peeled): nothing left to request, the epilogue's operands go out instead
        if (!late_ops) issue_operands();
        compute((kt_end - 1 - kt_begin) & 1);
    } else {
        // STAGES-deep ring, STAGES-1 tiles in flight.  Each wave issues NLD = NA_I + NB_I DMA instructions per tile, so
        // "tile kt has landed" == at most NLD * (tiles issued after kt) of this wave's loads are still outstanding.
        constexpr int NLD = NA_I + NB_I, D = STAGES - 1;
        const int nk = kt_end - kt_begin;
#pragma unroll
        for (int t = 0; t < D; ++t) if (t < nk) stage(t);
        int buf = 0, nbuf = D;                                // nbuf = ring slot of the next tile to issue
        for (int it = 0; it < nk; ++it) {
            const int ahead = min(nk - 1 - it, D - 1);       // tiles issued after tile `it` at this point
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // every wave's part of tile `it` landed; ring slot (it-1)%STAGES is drained
            if (it + D < nk) stage(nbuf);
            compute(buf);
            buf = buf == STAGES - 1 ? 0 : buf + 1;
            nbuf = nbuf == STAGES - 1 ? 0 : nbuf + 1;
        }
        if (!late_ops) issue_operands();
    }
    if (late_ops) issue_operands();

    // ---- epilogue.  The accumulators (lane: row m = c, 4 consecutive columns) are staged through LDS as fp32 with a
    // 16-B-chunk XOR swizzle (conflict-free ds_write_b128), then every thread handles 8 consecutive columns of a row so
    // that each wave store instruction covers 4 FULL 256-B bf16 rows (512-B fp32 rows): whole-line HBM writes, and the
    // residual / gelu'(z) operands are read with the same coalesced pattern.  (Storing straight from the MFMA layout
    // -- 8-B pieces scattered over 16 rows per instruction -- measured 2.4x slower end to end on K = 768 GEMMs.)
    if (p.dbg == 1 && acc[0][0][0] != 12345.678f) return;
    const float alpha = e.alpha_dev ? e.alpha * (*e.alpha_dev) : e.alpha;
    if constexpr (BG != 0) {
        if (bias_wg && wn == 0 && g == 0) {         // one owner per output row (tile column 0, no split): plain read-modify-write
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int row = m0 + wm * WR + i * 16 + c;
                if (row < p.M) p.bias_grad[row] += accb[i][0] * alpha;
            }
        }
    }
    float* cs = reinterpret_cast<float*>(smem);     // [RPP][FBN] fp32, 16-B chunks XOR-swizzled by row
    auto load8 = [&](int row, int q, float* v) {     // 8 consecutive columns 8q..8q+7 of staged row
        const float4 lo = *reinterpret_cast<const float4*>(cs + row * FBN + (((2 * q) ^ (row & 7)) << 2));
        const float4 hi = *reinterpret_cast<const float4*>(cs + row * FBN + (((2 * q + 1) ^ (row & 7)) << 2));
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    };
    const DropKey dkey = drop_key(eff_seed(e.dropout_seed, e.dropout_seed_dev));
    // gfx950 has ONE in-order counter for loads and stores: a wait for ANY load issued after the first pass's stores also waits for those stores
    // (measured on the 256-row kernel of gemm_p8.hip: 8 us per tile of store round trips).  The operand loads of BOTH passes (the 160-row tile
    // has two) were therefore issued up front (issue_operands, in front of the last K-tile's MFMAs) and are -- with the bias -- complete before
    // the first store; the pass loop is unrolled, and its barriers are raw s_barrier + lgkmcnt(0) (a __syncthreads() in front of pending
    // loads waits vmcnt(0)): pass 2 no longer starts with the drain of pass 1's stores.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) asm volatile("" : "+v"(oq[ps][it]));
    asm volatile("" : "+v"(bq[0]));
    asm volatile("" : "+v"(bq[1]));
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias_vec) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { bias8[r] = __uint_as_float(bq[0][r]); bias8[4 + r] = __uint_as_float(bq[1][r]); }
    } else if (e.bias && col_ok && !p.slabs) {
        for (int r = 0; r < nvalid; ++r) bias8[r] = e.bias[gn + r];
    }
    auto epi_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
    epi_barrier();                                  // operand buffers / previous pass fully consumed by every wave
    if (wm * WR >= ps * RPP && wm * WR < (ps + 1) * RPP) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int row = wm * WR + i * 16 + c - ps * RPP;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int chunk = (wn * 16 + j * 4 + g) ^ (row & 7);
                *reinterpret_cast<float4*>(cs + row * FBN + chunk * 4) =
                    make_float4(acc[j][i][0] * alpha, acc[j][i][1] * alpha, acc[j][i][2] * alpha, acc[j][i][3] * alpha);
            }
        }
    }
    epi_barrier();
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int row = row_t + it * RSTEP;
        const int gm = m0 + ps * RPP + row;
        if (gm >= p.M || !col_ok) continue;
        const int64_t off = (int64_t)gm * p.ldc + gn;
        float v[8];
        load8(row, q, v);
        if (p.dbg == 3) { if (v[0] == 12345.678f) reinterpret_cast<float*>(p.C)[0] = v[0]; continue; }      // timing experiment: everything but the stores
        if (p.slabs) {     // split-K: plain partial-slab store, reduced by splitk_reduce_kernel (deterministic, no atomics)
            float* sp = p.slabs + (int64_t)split * p.M * p.ldc + off;
            if (vec) {
                *reinterpret_cast<float4*>(sp) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(sp + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else for (int r = 0; r < nvalid; ++r) sp[r] = v[r];
            continue;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += bias8[r];
        if (e.aux_out) {     // pre-activation side output z (bf16), same coalesced pattern
            bf16_t* z = reinterpret_cast<bf16_t*>(e.aux_out) + off;
            if (vec) st16(z, pack8(v));
            else for (int r = 0; r < nvalid; ++r) z[r] = f32_to_bf16(v[r]);
        }
        if (e.act == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = gelu_f(v[r]);
        }
        if (zsrc) {
            float zf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (vec) unpack8(make_uint4(oq[ps][it][0], oq[ps][it][1], oq[ps][it][2], oq[ps][it][3]), zf);
            else for (int r = 0; r < nvalid; ++r) zf[r] = bf16_to_f32(zsrc[off + r]);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= gelu_grad_f(zf[r]);
        }
        if (e.dropout_p > 0.f) {
            bool keep[8];
            dropout_keep_n<8>(dkey, (uint64_t)gm * (uint64_t)p.N + (uint64_t)gn, p.drop_thresh, keep);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = keep[r] ? v[r] * p.drop_scale : 0.f;
        }
        if (rsrc) {
            float rf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (vec && !zsrc) unpack8(make_uint4(oq[ps][it][0], oq[ps][it][1], oq[ps][it][2], oq[ps][it][3]), rf);
            else if (vec) unpack8(*reinterpret_cast<const uint4*>(rsrc + (int64_t)gm * e.ldr + gn), rf);
            else for (int r = 0; r < nvalid; ++r) rf[r] = bf16_to_f32(rsrc[(int64_t)gm * e.ldr + gn + r]);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += rf[r];
        }
        if (e.out_dtype == VM_BF16) {
            bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + off;
            if (vec) st16(cp, pack8(v));
            else for (int r = 0; r < nvalid; ++r) cp[r] = f32_to_bf16(v[r]);
        } else {
            float* cp = reinterpret_cast<float*>(p.C) + off;
            if (e.accumulate) {     // this thread owns the elements (no split): plain read-modify-write
                if (vec) {
                    const float4 o0 = *reinterpret_cast<float4*>(cp), o1 = *reinterpret_cast<float4*>(cp + 4);
                    *reinterpret_cast<float4*>(cp) = make_float4(o0.x + v[0], o0.y + v[1], o0.z + v[2], o0.w + v[3]);
                    *reinterpret_cast<float4*>(cp + 4) = make_float4(o1.x + v[4], o1.y + v[5], o1.z + v[6], o1.w + v[7]);
                } else for (int r = 0; r < nvalid; ++r) cp[r] += v[r];
            } else if (vec) {
                *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else for (int r = 0; r < nvalid; ++r) cp[r] = v[r];
        }
    }
    }   // passes
}

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {       // hardware places block b on XCD b % 8; give each XCD a contiguous range
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

template <int LA, int LB, int WM, int WN, int STAGES, int BKT, int MF, int PIPE>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4) ? 2 : 1) void gemm_fast_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_fast_body<LA, LB, WM, WN, STAGES, BKT, MF, PIPE, 0>(p, xcd_remap(blockIdx.x, gridDim.x), smem);
}

// Several independent GEMMs in ONE launch (the weight gradients of a layer): block -> (problem, tile) through the prefix sums
// of the per-problem tile counts.  Every problem fills whole tiles of the chip, so none of them needs split-K (no fp32 slabs,
// no reduce kernel), and the workgroups of tile column 0 also produce the bias gradient (no column-sum kernel).
template <int LA, int LB, int WM, int WN, int STAGES, int BKT, int MF, int PIPE, int BG>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4) ? 2 : 1) void gemm_grouped_kernel(const GemmGroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int gi = 0;
#pragma unroll
    for (int i = 1; i < VM_GEMM_MAX_GROUP; ++i) if (i < ga.n && bid >= ga.tile_start[i]) gi = i;
    gemm_fast_body<LA, LB, WM, WN, STAGES, BKT, MF, PIPE, BG>(ga.g[gi], bid - ga.tile_start[gi], smem);
}

template <int LA, int LB, int WM, int WN, int STAGES, int BKT, int MF, int PIPE>
static int launch_fast(const GemmArgs& a, int nblocks, hipStream_t s) {
    constexpr int RING = STAGES * (WM * MF * 16 + WN * 64) * BKT * 2, STAGE_MIN = MF * 16 * WN * 64 * 4;
    constexpr int LDS = RING >= STAGE_MIN ? RING : STAGE_MIN;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fast_kernel<LA, LB, WM, WN, STAGES, BKT, MF, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_fast_kernel<LA, LB, WM, WN, STAGES, BKT, MF, PIPE>), dim3(nblocks), dim3(WM * WN * 64), LDS, s, a);
    return vm_check_launch("vm_gemm_bf16(fast)");
}

// Two independent GEMMs of DIFFERENT operand layouts in one launch (the two gradient products of the similarity losses, csrc/contrastive.hip:
// dA = G B^ and dB = G^T A^): a workgroup runs one of the two main loops (uniform branch on the block id), 128 x 128 tiles, PIPE 4.
template <int LA0, int LB0, int LA1, int LB1>
__global__ __launch_bounds__(256, 2) void gemm_pair_kernel(const GemmArgs a0, const GemmArgs a1, const int tiles0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    if (bid < tiles0) gemm_fast_body<LA0, LB0, 2, 2, 2, 64, 4, 4, 0>(a0, bid, smem);
    else gemm_fast_body<LA1, LB1, 2, 2, 2, 64, 4, 4, 0>(a1, bid - tiles0, smem);
}
// a0: A row-major x B k-major, a1: A k-major x B k-major; both with K % 64 == 0, tiles_m / tiles_n counted in 128 x 128 tiles, no split
int vm_gemm_pair_launch(const GemmArgs& a0, const GemmArgs& a1, hipStream_t s) {
    constexpr int LDS = 2 * (128 + 128) * 64 * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pair_kernel<0, 1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int t0 = a0.tiles_m * a0.tiles_n, t1 = a1.tiles_m * a1.tiles_n;
    hipLaunchKernelGGL((gemm_pair_kernel<0, 1, 1, 1>), dim3(t0 + t1), dim3(256), LDS, s, a0, a1, t0);
    return vm_check_launch("vm_gemm_pair");
}

// C[m, 0:N] (+)= sum_s slabs[s][m, 0:N]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, int M, int N, int64_t ldc,
                                                            int nsplit, int accumulate) {
    const int nq = (N + 3) >> 2;
    const int64_t total = (int64_t)M * nq;
    const int64_t slab = (int64_t)M * ldc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / nq), n = (int)(i - (int64_t)m * nq) * 4;
        const int64_t off = (int64_t)m * ldc + n;
        if (n + 4 <= N) {
            float4 a = accumulate ? *reinterpret_cast<const float4*>(C + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s = 0; s < nsplit; ++s) {
                const float4 b = *reinterpret_cast<const float4*>(slabs + s * slab + off);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            *reinterpret_cast<float4*>(C + off) = a;
        } else {
            for (int r = 0; r < N - n; ++r) {
                float a = accumulate ? C[off + r] : 0.f;
                for (int s = 0; s < nsplit; ++s) a += slabs[s * slab + off + r];
                C[off + r] = a;
            }
        }
    }
}

int vm_gemm_splitk_reduce(const GemmArgs& a, int nsplit, hipStream_t s) {
    const int64_t total = (int64_t)a.M * ((a.N + 3) / 4);
    int64_t blocks = (total + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.slabs, (float*)a.C, a.M, a.N, a.ldc, nsplit, a.e.accumulate);
    return vm_check_launch("vm_gemm_bf16(split-k reduce)");
}

// Tile variants (the host's cost model picks 0 or 4 per launch; VM_GEMM_VARIANT forces one):
//   0: 128 x 128, 4 waves of 64 x 64, two workgroups per CU      4: 160 x 128 (5 A fragments per wave), row-major A only
//   1: 256 x 128, 8 waves, 3-stage ring (huge square problems)
// Main loop per operand layout: row-major x row-major runs the software-pipelined K-tile (PIPE 1); every launch with a strided
// operand (dgrad, wgrad) runs the cross-tile register pipeline with the inline-asm LDS-DMA (PIPE 4, see glds16_asm).
// Measured and dropped -- round 1: 128 x 128 with k-tile 32 x 4-stage ring, 256 x 128 with 128 x 64 per wave, DMA pieces issued from
// between the MFMAs (profiles/r01_e_gemm_pipe_ab.txt); round 2: k-tile 32 x 3-slot ring with register prefetch 0.72-0.97x, 256 x 128 /
// 128 x 64 per wave on that ring 0.51-0.98x, s_setprio around the MFMA stream 0.83-1.0x (profiles/r02_a_gemm_candidates_ab.txt); PIPE 4 for
// row-major operands 0.87-1.05x, DMA pieces slotted one per MFMA = burst (profiles/r02_b_gemm_pipe4_asmdma_ab.txt); late start of the
// second workgroup of a CU 0.83-1.0x (profiles/r02_c_gemm_stagger_and_epilogue_breakdown.txt); the 16-wave 256 x 256 tile (45 spills);
// a resident grid of 512 workgroups walking the tiles instead of one workgroup per tile 0.87-1.05x (profiles/r02_e_gemm_persist_ab.txt);
// a register-direct epilogue (v_permlane16_swap to 8 consecutive columns per lane, no LDS staging, no barriers) 0.88-1.06x and the 8-wave
// 256 x 256 tile (128 x 64 per wave, one workgroup per CU) 0.80-1.02x (profiles/r02_d_gemm_register_epilogue_ab.txt); forced column-group
// widths of the tile order 0.85-1.03x (profiles/r02_k_gemm_groupw_ab.txt).  None of that code is kept.
int vm_gemm_fast_dispatch(const GemmArgs& a0, int a_layout, int b_layout, int nblocks, int variant, hipStream_t s) {
    if (variant == 1) {
        if (a_layout == 0 && b_layout == 0) return launch_fast<0, 0, 4, 2, 3, 64, 4, 0>(a0, nblocks, s);
        if (a_layout == 0 && b_layout == 1) return launch_fast<0, 1, 4, 2, 3, 64, 4, 0>(a0, nblocks, s);
        if (a_layout == 1 && b_layout == 0) return launch_fast<1, 0, 4, 2, 3, 64, 4, 0>(a0, nblocks, s);
        return launch_fast<1, 1, 4, 2, 3, 64, 4, 0>(a0, nblocks, s);
    }
    if (variant == 4 && a_layout == 0)
        return b_layout == 0 ? launch_fast<0, 0, 2, 2, 2, 64, 5, 1>(a0, nblocks, s) : launch_fast<0, 1, 2, 2, 2, 64, 5, 4>(a0, nblocks, s);
    if (variant == 5 && a_layout == 0)       // 64 x 128 (two A fragments per wave): three workgroups per CU for problems of less than one round of 128-row tiles
        return b_layout == 0 ? launch_fast<0, 0, 2, 2, 2, 64, 2, 1>(a0, nblocks, s) : launch_fast<0, 1, 2, 2, 2, 64, 2, 4>(a0, nblocks, s);
    if (a_layout == 0 && b_layout == 0) return launch_fast<0, 0, 2, 2, 2, 64, 4, 1>(a0, nblocks, s);
    if (a_layout == 0 && b_layout == 1) return launch_fast<0, 1, 2, 2, 2, 64, 4, 4>(a0, nblocks, s);
    if (a_layout == 1 && b_layout == 0) return launch_fast<1, 0, 2, 2, 2, 64, 4, 4>(a0, nblocks, s);
    return launch_fast<1, 1, 2, 2, 2, 64, 4, 4>(a0, nblocks, s);
}

// the tile a variant runs for these layouts (strided A falls back to the 128-row tile)
void vm_gemm_variant_tile(int variant, int a_layout, int* bm, int* bn) {
    if (variant == 1) { *bm = 256; *bn = 128; return; }
    *bm = (variant == 4 && a_layout == 0) ? 160 : (variant == 5 && a_layout == 0) ? 64 : 128;
    *bn = 128;
}


template <int LA, int LB, int PIPE, int BG>
static int launch_grouped(const GemmGroupArgs& ga, int nblocks, hipStream_t s) {
    constexpr int LDS = 2 * (128 + 128) * 64 * 2;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_grouped_kernel<LA, LB, 2, 2, 2, 64, 4, PIPE, BG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_grouped_kernel<LA, LB, 2, 2, 2, 64, 4, PIPE, BG>), dim3(nblocks), dim3(256), LDS, s, ga);
    return vm_check_launch("vm_gemm_grouped");
}

// 128 x 128 tiles.  TN (both operands contraction-major): the weight-gradient form, with the bias-gradient MFMAs compiled in (BG);
// NT / NN: independent problems of one layout in one launch (vm_gemm_grouped: the per-image products of the GLoRIA local loss).
int vm_gemm_grouped_launch(const GemmGroupArgs& ga, int nblocks, int a_layout, int b_layout, hipStream_t s) {
    if (a_layout == 1 && b_layout == 1) return launch_grouped<1, 1, 4, 1>(ga, nblocks, s);
    if (a_layout == 0 && b_layout == 0) return launch_grouped<0, 0, 1, 0>(ga, nblocks, s);
    if (a_layout == 0 && b_layout == 1) return launch_grouped<0, 1, 4, 0>(ga, nblocks, s);
    vm_set_error("vm_gemm_grouped: layout (a=%d, b=%d) has no grouped kernel", a_layout, b_layout);
    return VM_EUNSUPPORTED;
}
