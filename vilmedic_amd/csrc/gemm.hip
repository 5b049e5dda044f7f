// gemm.hip -- bf16 MFMA GEMM for gfx950 with fused epilogues.
//
//   C[M,N] = epi( sum_k A(m,k) * B(n,k) ),  fp32 accumulate.
//
// One 256-thread workgroup (4 waves, 2x2) per 128x128 output tile, BK = 64, v_mfma_f32_16x16x32_bf16,
// 4x4 fragments per wave.  Operands are staged HBM -> VGPR -> LDS (16-B loads, issue-early /
// write-late so HBM latency hides under the MFMA phase), double-buffered, one barrier per K-tile.
// Either operand may be stored with the contraction dim contiguous (layout 0, read with ds_read_b128)
// or strided (layout 1: the LDS image keeps the global row-major order and the MFMA fragment is
// gathered with the gfx950 transpose read ds_read_b64_tr_b16), so forward (NT), dgrad (NN) and
// wgrad (TN) all run on the same kernel without transposed copies of weights or activations.
// The accumulator tile is staged through LDS as fp32 so the epilogue (bias, erf-GELU, gelu' multiply,
// dropout, residual add, bf16/fp32 store, fp32 atomic accumulate for split-K) runs on whole 16-B
// row segments with coalesced HBM traffic.
#include "common.h"
#include "gemm_args.h"
#include <cstdlib>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#define BM 128
#define BN 128
#define BK 64
#define LDS_K_STRIDE 72    // elements per row for layout-0 tiles  [128][72]  (144 B rows, conflict-spreading pad)
#define LDS_M_STRIDE 136   // elements per row for layout-1 tiles  [64][136]  (272 B rows)
#define OPER_BYTES 18432   // max(128*72*2, 64*136*2)
#define STAGE_BYTES (2 * OPER_BYTES)
#define C_STRIDE 132       // fp32 elements per row of the staged accumulator tile
#define GEMM_LDS_BYTES (2 * STAGE_BYTES)   // 73728 >= 128*132*4 = 67584


// predicated 16-B load: out-of-range lanes re-read the (always valid) matrix base and zero the result,
// which keeps a plain global_load instead of a pointer-select + flat_load
__device__ __forceinline__ uint4 ldg16(const bf16_t* p, const bf16_t* safe, bool ok) {
    uint4 v = *reinterpret_cast<const uint4*>(ok ? p : safe);
    if (!ok) v = make_uint4(0, 0, 0, 0);
    return v;
}

// ---- HBM -> registers for one operand tile (4 x 16 B per thread)
template <int LAYOUT>
__device__ __forceinline__ void load_tile(uint4 (&r)[4], const bf16_t* base, int64_t ld, int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i;
        if (LAYOUT == 0) {          // [rows][K]: 8 chunks per 64-wide k-slab
            const int row = id >> 3, c = id & 7;
            const int gr = row0 + row, gk = k0 + c * 8;
            r[i] = ldg16(base + (int64_t)gr * ld + gk, base, gr < nrows && gk < K);
        } else {                    // [K][rows]: 16 chunks per 128-wide row-slab
            const int kr = id >> 4, c = id & 15;
            const int gk = k0 + kr, gr = row0 + c * 8;
            r[i] = ldg16(base + (int64_t)gk * ld + gr, base, gk < K && gr < nrows);
        }
    }
}

template <int LAYOUT>
__device__ __forceinline__ void store_tile(const uint4 (&r)[4], char* lds, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i;
        if (LAYOUT == 0) {
            const int row = id >> 3, c = id & 7;
            *reinterpret_cast<uint4*>(lds + row * (LDS_K_STRIDE * 2) + c * 16) = r[i];
        } else {
            const int kr = id >> 4, c = id & 15;
            *reinterpret_cast<uint4*>(lds + kr * (LDS_M_STRIDE * 2) + c * 16) = r[i];
        }
    }
}

// ---- LDS -> MFMA fragment: lane l holds 8 consecutive k for row (l&15), k-group (l>>4)
template <int LAYOUT>
__device__ __forceinline__ bf16x8_t read_frag(const char* lds, int row_base, int kk, int lane) {
    if (LAYOUT == 0) {
        const int row = row_base + (lane & 15);
        const int kel = kk * 32 + (lane >> 4) * 8;
        return *reinterpret_cast<const bf16x8_t*>(lds + row * (LDS_K_STRIDE * 2) + kel * 2);
    } else {
        // transpose read: within a 16-lane group, lane i supplies the address of 4 contiguous
        // elements (row-dim) at k = kb + (i>>2); lane c receives the 4 k-values of column c.
        const int k = kk * 32 + (lane >> 4) * 8 + ((lane & 15) >> 2);
        const int col = row_base + (lane & 3) * 4;
        typedef short v4s __attribute__((ext_vector_type(4)));
        const __attribute__((address_space(3))) v4s* p0 =
            (const __attribute__((address_space(3))) v4s*)(lds + k * (LDS_M_STRIDE * 2) + col * 2);
        const __attribute__((address_space(3))) v4s* p1 =
            (const __attribute__((address_space(3))) v4s*)(lds + (k + 4) * (LDS_M_STRIDE * 2) + col * 2);
        v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p0);
        v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p1);
        short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    }
}

template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: hardware places block b on XCD b%8; give each XCD a contiguous run of tiles
    const int nwg = gridDim.x;
    int bid;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = bid / tiles;
    const int t = bid - split * tiles;
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);

    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4];
    if (kt_begin < kt_end) {
        load_tile<LA>(ra, p.A, p.lda, m0, p.M, kt_begin * BK, p.K, tid);
        load_tile<LB>(rb, p.B, p.ldb, n0, p.N, kt_begin * BK, p.K, tid);
        store_tile<LA>(ra, smem, tid);
        store_tile<LB>(rb, smem + OPER_BYTES, tid);
    }
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        const char* sa = smem + buf * STAGE_BYTES;
        const char* sb = sa + OPER_BYTES;
        const bool more = kt + 1 < kt_end;
        if (more) {   // issue next tile's HBM loads before the MFMA phase
            load_tile<LA>(ra, p.A, p.lda, m0, p.M, (kt + 1) * BK, p.K, tid);
            load_tile<LB>(rb, p.B, p.ldb, n0, p.N, (kt + 1) * BK, p.K, tid);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag<LA>(sa, wm * 64 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<LB>(sb, wn * 64 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) {   // write-late into the other buffer (last read two iterations ago, fenced by the barrier)
            char* da = smem + (buf ^ 1) * STAGE_BYTES;
            store_tile<LA>(ra, da, tid);
            store_tile<LB>(rb, da + OPER_BYTES, tid);
        }
        __syncthreads();
    }

    // ---- stage the fp32 accumulator tile through LDS:  C/D layout col = lane&15, row = (lane>>4)*4 + reg
    float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = wn * 64 + j * 16 + (lane & 15);
            const int row = wm * 64 + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[(row + r) * C_STRIDE + col] = acc[i][j][r];
        }
    __syncthreads();

    // ---- coalesced epilogue: each thread owns 8 consecutive columns of a row, 8 such segments
    const vm_gemm_epilogue& e = p.e;
    const float alpha = e.alpha_dev ? e.alpha * (*e.alpha_dev) : e.alpha;
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int id = tid + 256 * it;
        const int row = id >> 4, cc = (id & 15) * 8;
        const int gm = m0 + row, gn = n0 + cc;
        if (gm >= p.M || gn >= p.N) continue;
        float v[8];
        {
            const float4 lo = *reinterpret_cast<const float4*>(cs + row * C_STRIDE + cc);
            const float4 hi = *reinterpret_cast<const float4*>(cs + row * C_STRIDE + cc + 4);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        }
        const int nvalid = min(8, p.N - gn);
        const int64_t off = (int64_t)gm * p.ldc + gn;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= alpha;
        if (e.bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += e.bias[gn + j];
        }
        if (e.aux_out) {
            bf16_t* z = reinterpret_cast<bf16_t*>(e.aux_out) + off;
            if (nvalid == 8) *reinterpret_cast<uint4*>(z) = pack8(v);
            else for (int j = 0; j < nvalid; ++j) z[j] = f32_to_bf16(v[j]);
        }
        if (e.act == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
        }
        if (e.mul_gelu_z) {
            const bf16_t* z = reinterpret_cast<const bf16_t*>(e.mul_gelu_z) + off;
            float zf[8];
            if (nvalid == 8) unpack8(*reinterpret_cast<const uint4*>(z), zf);
            else for (int j = 0; j < 8; ++j) zf[j] = j < nvalid ? bf16_to_f32(z[j]) : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(zf[j]);
        }
        if (e.dropout_p > 0.f) {
            const uint64_t idx = (uint64_t)gm * (uint64_t)p.N + (uint64_t)gn;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dropout_keep(drop_key(eff_seed(e.dropout_seed, e.dropout_seed_dev)), idx + j, p.drop_thresh) ? v[j] * p.drop_scale : 0.f;
        }
        if (e.residual) {
            const bf16_t* rp = reinterpret_cast<const bf16_t*>(e.residual) + (int64_t)gm * e.ldr + gn;
            float rf[8];
            if (nvalid == 8) unpack8(*reinterpret_cast<const uint4*>(rp), rf);
            else for (int j = 0; j < 8; ++j) rf[j] = j < nvalid ? bf16_to_f32(rp[j]) : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rf[j];
        }
        if (e.out_dtype == VM_BF16) {
            bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + off;
            if (nvalid == 8) *reinterpret_cast<uint4*>(c) = pack8(v);
            else for (int j = 0; j < nvalid; ++j) c[j] = f32_to_bf16(v[j]);
        } else {
            float* c = reinterpret_cast<float*>(p.C) + off;
            if (e.accumulate) {
                for (int j = 0; j < nvalid; ++j) atomicAdd(c + j, v[j]);
            } else if (nvalid == 8) {
                *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                for (int j = 0; j < nvalid; ++j) c[j] = v[j];
            }
        }
    }
}

template <int LA, int LB>
static int launch(const GemmArgs& a, int nblocks, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<LA, LB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<LA, LB>), dim3(nblocks), dim3(256), GEMM_LDS_BYTES, s, a);
    return vm_check_launch("vm_gemm_bf16");
}

extern "C" int vm_gemm_bf16(const void* A, int64_t lda, int a_layout, const void* B, int64_t ldb, int b_layout,
                            void* C, int64_t ldc, int M, int N, int K, const vm_gemm_epilogue* epi, void* stream) {
    VM_REQUIRE(A && B && C && epi, "vm_gemm_bf16: null pointer");
    VM_REQUIRE(M > 0 && N > 0 && K > 0, "vm_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    VM_REQUIRE((lda % 8) == 0 && (ldb % 8) == 0, "vm_gemm_bf16: lda/ldb must be multiples of 8 (got %lld, %lld)", (long long)lda, (long long)ldb);
    VM_REQUIRE(epi->out_dtype == VM_F32 ? (ldc % 4) == 0 : (ldc % 8) == 0, "vm_gemm_bf16: ldc alignment");
    if (a_layout == 0 || b_layout == 0) VM_REQUIRE((K % 8) == 0, "vm_gemm_bf16: K must be a multiple of 8 for K-contiguous operands (K=%d)", K);
    VM_REQUIRE(epi->split_k >= 1, "vm_gemm_bf16: split_k must be >= 1");
    if (epi->split_k > 1) VM_REQUIRE(epi->out_dtype == VM_F32 && !epi->bias && !epi->act && !epi->aux_out && !epi->mul_gelu_z && !epi->residual && epi->dropout_p == 0.f,
                                     "vm_gemm_bf16: split_k needs fp32 output and a plain (alpha-only) epilogue");
    if (epi->accumulate) VM_REQUIRE(epi->out_dtype == VM_F32, "vm_gemm_bf16: accumulate needs fp32 output");
    if (epi->residual) VM_REQUIRE((epi->ldr % 8) == 0, "vm_gemm_bf16: ldr alignment");
    VM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0, "vm_gemm_bf16: pointers must be 16-byte aligned");

    GemmArgs a;
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    a.ktiles = (K + BK - 1) / BK;
    int split = epi->split_k;
    if (split > a.ktiles) split = a.ktiles;
    a.ktiles_per_split = (a.ktiles + split - 1) / split;
    split = (a.ktiles + a.ktiles_per_split - 1) / a.ktiles_per_split;
    a.e = *epi;
    a.drop_thresh = dropout_thresh16(epi->dropout_p);
    a.drop_scale = epi->dropout_p > 0.f ? 1.0f / (1.0f - epi->dropout_p) : 1.0f;
    const int nblocks = a.tiles_m * a.tiles_n * split;
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_GEMM, 2.0 * (double)M * (double)N * (double)K, s, "M%d_N%d_K%d_l%d%d_sk%d_b%d_a%d_z%d_g%d_d%d_r%d_f%d", M, N, K, a_layout, b_layout,
                     epi->split_k, epi->bias != nullptr, epi->act, epi->aux_out != nullptr, epi->mul_gelu_z != nullptr, epi->dropout_p > 0.f, epi->residual != nullptr, epi->out_dtype == VM_F32);
    const bool fast_ok = (K % 64) == 0 && (ldc % 8) == 0 && (!epi->bias || ((uintptr_t)epi->bias % 16) == 0) &&
                         (!epi->residual || (epi->ldr % 8) == 0) && !vm_env().gemm_generic;
    a.slabs = nullptr;
    a.bias_grad = nullptr;
    a.dbg = vm_env().gemm_debug;
    // decode-step shapes: few rows -> one workgroup per 16 output columns, K split over its waves (gemm_skinny.hip)
    if (M <= 256 && a_layout == 0 && b_layout == 0 && (K % 32) == 0 && split == 1 && !epi->aux_out && !epi->mul_gelu_z &&
        epi->dropout_p == 0.f && (!epi->residual || (epi->ldr % 4) == 0) && !vm_env().gemm_no_skinny && !vm_env().gemm_generic && a.dbg == 0 &&
        vm_env().gemm_variant < 0)
        return vm_gemm_skinny_dispatch(a, s);
    if (fast_ok) {
        if (split > 1) {
            const size_t need = (size_t)split * (size_t)M * (size_t)ldc * sizeof(float);
            VM_REQUIRE(epi->workspace && epi->workspace_bytes >= need, "vm_gemm_bf16: split_k=%d needs a %zu-byte workspace", split, need);
            a.slabs = (float*)epi->workspace;
        }
        // tile variant: 256x128 3-stage ring when it still yields enough workgroups (1 per CU), else 128x128 2-stage
        int variant = 0;
        const int force = vm_env().gemm_variant;
        // wide-tile kernel (gemm_p8.hip): (32 MF) x 256 tiles, one 8-wave workgroup per CU; 8 = one tile per workgroup, 9 = persistent
        // (cost model: only very wide row-major outputs with many rounds of 256 x 256 tiles -- the LM head, 8192 x 30522 x 768: 15 rounds,
        //  measured 383-421 us against 427-438 us on the 128-row kernels; everywhere else the lockstep prologue / epilogue bursts of one
        //  workgroup per CU cost more than the faster main loop gains, profiles/r04_a_gemm_p8_probe.txt)
        // (whole 256-row tiles only: the cross K|V projection of all layers, 12608 x 18432 x 768, runs 192-row tiles there and measured
        //  410-436 us against 397 us on the 160-row kernel, profiles/r04_g_shape_table.txt vs r03_b_kernel_shape_breakdown.txt)
        const bool p8_auto = force < 0 && a_layout == 0 && b_layout == 0 && split == 1 && N >= 16384 && M >= 4096 && (M % 256) == 0 && K <= 1024 &&
                             (int64_t)((M + 255) / 256) * ((N + 255) / 256) >= 12 * 256;
        if (((force == 8 || force == 9 || force == 10) && a_layout == 0) || p8_auto) {       // 8: four barrier pairs per K-tile, 9 / auto: two
            int mf = vm_env().gemm_p8_mf;
            if (mf < 5 || mf > 8) {      // rounds of the 256 CUs x rows per tile, ties to the larger tile
                int64_t best = -1;
                for (int m = 8; m >= 5; --m) {
                    if (p8_auto && (M % (32 * m)) != 0) continue;   // the automatic choice never takes a ragged last row tile (the gate above admits M % 256 == 0: m = 8 always qualifies)
                    const int64_t t = (int64_t)((M + 32 * m - 1) / (32 * m)) * ((N + 255) / 256) * split;
                    const int64_t cost = (t + 255) / 256 * m;
                    if (best < 0 || cost < best) { best = cost; mf = m; }
                }
            }
            a.tiles_m = (M + 32 * mf - 1) / (32 * mf);
            a.tiles_n = (N + 255) / 256;
            a.group_w = a.tiles_n >= 12 ? 4 : a.tiles_n;
            if (vm_env().gemm_groupw > 0) a.group_w = vm_env().gemm_groupw;
            int rc = vm_gemm_p8_dispatch(a, a_layout, b_layout, mf, force == 8 ? 4 : 2, (force == 10 || p8_auto) ? 1 : 0, a.tiles_m * a.tiles_n * split, s);
            if (rc == VM_OK && split > 1) rc = vm_gemm_splitk_reduce(a, split, s);
            return rc;
        }
        const int tiles_m256 = (M + 255) / 256;
        if (force >= 0) variant = force;
        else if (K >= 4096 && (int64_t)tiles_m256 * a.tiles_n * split >= 1024) variant = 1;   // measured: only huge square-ish problems gain
        else if (a_layout == 0 && K <= 768 && (int64_t)((M + 127) / 128) * a.tiles_n * split <= 384 && (int64_t)((M + 63) / 64) * a.tiles_n * split <= 768)
            variant = 5;    // [r4] less than one round of 128-row tiles and a short K: 64-row tiles, three workgroups per CU (8192 x 768 x 768: 19.3 -> 17.7 us,
                            // its dgrad 16.7 -> 15.7; slower on every larger shape, profiles/r04_i_gemm_64row_tiles_ab.txt)
        else if (a_layout == 0) {
            // 512 workgroup slots (256 CUs x 2 resident workgroups): compare rounds x rows-per-tile of the 128- and the
            // 160-row tile -- e.g. M = 12608, N = 768: 594 tiles = 2 rounds of 128 rows vs 474 tiles = 1 round of 160 rows
            const int64_t slots = 512;
            const int64_t t128 = (int64_t)((M + 127) / 128) * a.tiles_n * split, t160 = (int64_t)((M + 159) / 160) * a.tiles_n * split;
            const int64_t c128 = (t128 + slots - 1) / slots * 128, c160 = (t160 + slots - 1) / slots * 160;
            if (c160 <= c128) variant = 4;   // ties: the larger tile re-reads less of B
        }
        int vbm, vbn;
        vm_gemm_variant_tile(variant, a_layout, &vbm, &vbn);
        a.tiles_m = (M + vbm - 1) / vbm;
        a.tiles_n = (N + vbn - 1) / vbn;
        // column-group width of the tile order: wide outputs (N >= 3072) run in groups of 8 tile columns so that an XCD's
        // B working set stays L2-resident (measured +11 % on the N = 30528 LM head, +4 % at N = 3072, neutral below)
        a.group_w = a.tiles_n >= 24 ? 8 : a.tiles_n;
        if (vm_env().gemm_groupw > 0) a.group_w = vm_env().gemm_groupw;
        const int nb = a.tiles_m * a.tiles_n * split;
        int rc = vm_gemm_fast_dispatch(a, a_layout, b_layout, nb, variant, s);
        if (rc == VM_OK && split > 1) rc = vm_gemm_splitk_reduce(a, split, s);
        return rc;
    }
    if (split > 1) VM_REQUIRE(epi->accumulate, "vm_gemm_bf16: generic-path split_k accumulates atomically and needs accumulate=1");
    if (a_layout == 0 && b_layout == 0) return launch<0, 0>(a, nblocks, s);
    if (a_layout == 0 && b_layout == 1) return launch<0, 1>(a, nblocks, s);
    if (a_layout == 1 && b_layout == 0) return launch<1, 0>(a, nblocks, s);
    if (a_layout == 1 && b_layout == 1) return launch<1, 1>(a, nblocks, s);
    vm_set_error("vm_gemm_bf16: bad layout flags");
    return VM_EINVAL;
}


// ------------------------------------------------------------------ grouped weight gradients
// dW_i[N_i, K_i] += alpha_i * dY_i[M_i, N_i]^T X_i[M_i, K_i]   and   db_i[N_i] += alpha_i * colsum(dY_i)   for i < n, in ONE launch
// per <= VM_GEMM_MAX_GROUP problems.  Replaces, for the weight gradients of one transformer layer, 4-6 split-K GEMM launches +
// their slab-reduce kernels + the column-sum kernels (nn.Linear backward: hf:models/bert_generation/modeling_bert_generation.py
// :104-106,264-291 as differentiated by autograd in the reference).
extern "C" int vm_wgrad_grouped(const vm_wgrad_problem* pr, int n, void* stream) {
    VM_REQUIRE(pr && n > 0, "vm_wgrad_grouped: no problems");
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) {
        const vm_wgrad_problem& q = pr[i];
        VM_REQUIRE(q.dY && q.X && q.dW, "vm_wgrad_grouped: null pointer in problem %d", i);
        VM_REQUIRE(q.rows > 0 && q.n_out > 0 && q.k_in > 0, "vm_wgrad_grouped: empty problem %d", i);
        if ((q.rows % 64) != 0 || (q.ld_dy % 8) != 0 || (q.ld_x % 8) != 0 || (q.ld_dw % 8) != 0 || ((uintptr_t)q.dY % 16) != 0 ||
            ((uintptr_t)q.X % 16) != 0 || ((uintptr_t)q.dW % 16) != 0) {
            vm_set_error("vm_wgrad_grouped: problem %d is not eligible (rows %% 64, leading dims %% 8, 16-byte pointers)", i);
            return VM_EUNSUPPORTED;
        }
    }
    double work = 0;
    for (int i = 0; i < n; ++i) work += 2.0 * pr[i].rows * (double)pr[i].n_out * pr[i].k_in;
    // wide-tile path: every problem in whole 256-column tiles, and enough 256 x 256 tiles in the launch to be worth one workgroup per CU
    bool p8w = vm_env().wgrad_p8 != 0;
    int tiles256 = 0;
    for (int i = 0; i < n && p8w; ++i) {
        p8w = (pr[i].k_in % 256) == 0 && (pr[i].n_out + 7) / 8 * 8 <= pr[i].ld_dy;      // (the last 8-column piece of dY is read whole)
        tiles256 += ((pr[i].n_out + 255) / 256) * (pr[i].k_in / 256);
    }
    p8w = p8w && tiles256 >= vm_env().wgrad_p8_min;
    VmProfScope prof(VM_FAM_GEMM, work, s, p8w ? "wgrad_p8w_n%d_rows%d" : "wgrad_grouped_n%d_rows%d", n, pr[0].rows);
    if (p8w) {
        for (int base = 0; base < n; base += P8W_MAX_GROUP) {
            P8wArgs ga = {};
            ga.n = n - base < P8W_MAX_GROUP ? n - base : P8W_MAX_GROUP;
            int tiles = 0;
            for (int i = 0; i < ga.n; ++i) {
                const vm_wgrad_problem& q = pr[base + i];
                P8wProblem& a = ga.g[i];
                a.A = (const bf16_t*)q.dY; a.B = (const bf16_t*)q.X; a.C = q.dW; a.bias_grad = q.db; a.alpha_dev = q.alpha_dev;
                a.lda = q.ld_dy; a.ldb = q.ld_x; a.ldc = q.ld_dw;
                a.M = q.n_out; a.N = q.k_in; a.ktiles = q.rows / 64; a.tiles_n = q.k_in / 256; a.accumulate = q.overwrite ? 0 : 1;
                ga.tile_start[i] = tiles;
                tiles += ((a.M + 255) / 256) * a.tiles_n;
            }
            for (int i = ga.n; i <= P8W_MAX_GROUP; ++i) ga.tile_start[i] = tiles;
            const int rc = vm_wgrad_p8w_launch(ga, vm_env().wgrad_p8 == 2 ? 2 : 4, s);
            if (rc != VM_OK) return rc;
        }
        return VM_OK;
    }
    for (int base = 0; base < n; base += VM_GEMM_MAX_GROUP) {
        GemmGroupArgs ga = {};
        ga.n = n - base < VM_GEMM_MAX_GROUP ? n - base : VM_GEMM_MAX_GROUP;
        int tiles = 0;
        for (int i = 0; i < ga.n; ++i) {
            const vm_wgrad_problem& q = pr[base + i];
            GemmArgs& a = ga.g[i];
            a.A = (const bf16_t*)q.dY; a.B = (const bf16_t*)q.X; a.C = q.dW;
            a.lda = q.ld_dy; a.ldb = q.ld_x; a.ldc = q.ld_dw;
            a.M = q.n_out; a.N = q.k_in; a.K = q.rows;
            a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 127) / 128;
            a.ktiles = a.K / 64; a.ktiles_per_split = a.ktiles;
            a.group_w = a.tiles_n;
            a.e = vm_gemm_epilogue{};
            a.e.alpha = 1.0f; a.e.alpha_dev = q.alpha_dev; a.e.out_dtype = VM_F32; a.e.accumulate = 1; a.e.split_k = 1;
            a.drop_thresh = 0; a.drop_scale = 1.0f; a.dbg = 0; a.slabs = nullptr; a.bias_grad = q.db;
            ga.tile_start[i] = tiles;
            tiles += a.tiles_m * a.tiles_n;
        }
        for (int i = ga.n; i <= VM_GEMM_MAX_GROUP; ++i) ga.tile_start[i] = tiles;
        const int rc = vm_gemm_grouped_launch(ga, tiles, 1, 1, s);
        if (rc != VM_OK) return rc;
    }
    return VM_OK;
}

// ------------------------------------------------------------------ independent GEMMs of one layout in one launch
// C_i[M_i, N_i] (+)= A_i . B_i with a plain epilogue, up to VM_GEMM_MAX_GROUP problems per launch (block -> problem through the
// prefix sums of the tile counts): 48 per-image products of 72 tiles each (the GLoRIA local loss) fill the chip as 6 launches of
// 576 tiles instead of 48 launches at 28 % occupancy.
extern "C" int vm_gemm_grouped(const vm_gemm_problem* pr, int n, int a_layout, int b_layout, int out_dtype, int accumulate, void* stream) {
    VM_REQUIRE(pr && n > 0, "vm_gemm_grouped: no problems");
    VM_REQUIRE((out_dtype == VM_F32 || out_dtype == VM_BF16) && (a_layout == 0 || a_layout == 1) && (b_layout == 0 || b_layout == 1), "vm_gemm_grouped: bad flags");
    hipStream_t s = (hipStream_t)stream;
    double work = 0;
    for (int i = 0; i < n; ++i) {
        const vm_gemm_problem& q = pr[i];
        VM_REQUIRE(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0, "vm_gemm_grouped: bad problem %d", i);
        if ((q.K % 64) != 0 || (q.lda % 8) != 0 || (q.ldb % 8) != 0 || (q.ldc % 8) != 0 || ((uintptr_t)q.A % 16) != 0 || ((uintptr_t)q.B % 16) != 0 ||
            ((uintptr_t)q.C % 16) != 0 || (a_layout == 1 && b_layout == 0)) {
            vm_set_error("vm_gemm_grouped: problem %d is not eligible (K %% 64, leading dims %% 8, 16-byte pointers, layouts NT / NN / TN)", i);
            return VM_EUNSUPPORTED;
        }
        work += 2.0 * q.M * (double)q.N * q.K;
    }
    VmProfScope prof(VM_FAM_GEMM, work, s, "grouped_l%d%d_n%d_M%d_N%d_K%d", a_layout, b_layout, n, pr[0].M, pr[0].N, pr[0].K);
    for (int base = 0; base < n; base += VM_GEMM_MAX_GROUP) {
        GemmGroupArgs ga = {};
        ga.n = n - base < VM_GEMM_MAX_GROUP ? n - base : VM_GEMM_MAX_GROUP;
        int tiles = 0;
        for (int i = 0; i < ga.n; ++i) {
            const vm_gemm_problem& q = pr[base + i];
            GemmArgs& a = ga.g[i];
            a.A = (const bf16_t*)q.A; a.B = (const bf16_t*)q.B; a.C = q.C;
            a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc;
            a.M = q.M; a.N = q.N; a.K = q.K;
            a.tiles_m = (a.M + 127) / 128; a.tiles_n = (a.N + 127) / 128;
            a.ktiles = a.K / 64; a.ktiles_per_split = a.ktiles;
            a.group_w = a.tiles_n;
            a.e = vm_gemm_epilogue{};
            a.e.alpha = 1.0f; a.e.out_dtype = out_dtype; a.e.accumulate = accumulate ? 1 : 0; a.e.split_k = 1;
            a.drop_thresh = 0; a.drop_scale = 1.0f; a.dbg = 0; a.slabs = nullptr; a.bias_grad = nullptr;
            ga.tile_start[i] = tiles;
            tiles += a.tiles_m * a.tiles_n;
        }
        for (int i = ga.n; i <= VM_GEMM_MAX_GROUP; ++i) ga.tile_start[i] = tiles;
        const int rc = vm_gemm_grouped_launch(ga, tiles, a_layout, b_layout, s);
        if (rc != VM_OK) return rc;
    }
    return VM_OK;
}
