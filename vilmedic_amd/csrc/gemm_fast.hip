// gemm_fast.hip -- the tuned bf16 MFMA GEMM path (K % 64 == 0): LDS-DMA staging + XOR-swizzled LDS + register epilogue.
//
// Differences from the generic kernel in gemm.hip (kept as the any-shape fallback):
//   * operand tiles go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass); the LDS image
//     is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and undone on the read
//     (layout 0: 16-B chunk ^= row&7 for ds_read_b128; layout 1: chunk ^= ((k&3)<<1 | ((k>>3)&1)<<3) which makes the
//     16 lanes x 2 groups of a ds_read_b64_tr_b16 hit 16 distinct 16-B slots);
//   * the next K-tile's DMA is issued before the MFMA phase of the current one; one vmcnt(0)+barrier per K-tile;
//   * MFMA operands are swapped (D^T = B_frag x A_frag) so each lane ends up with 4 CONSECUTIVE output columns of one
//     row: the epilogue runs straight from registers with 8-B bf16 / 16-B fp32 accesses -- no LDS staging of C.
// Out-of-range rows/columns are clamped to valid addresses (they only feed outputs that are never stored); the
// contraction dim has no tail by construction (K % 64 == 0), so no zero-fill is needed.
#include "common.h"
#include "gemm_args.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));

#define FBM 128
#define FBN 128
#define FBK 64
#define F_OPER 16384
#define F_STAGE (2 * F_OPER)
#define F_LDS (2 * F_STAGE)

__device__ __forceinline__ int swz1(int krow) { return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3); }

template <int LAYOUT>
__device__ __forceinline__ const bf16_t* stage_src(const bf16_t* base, int64_t ld, int row0, int nrows, int q, int lane) {
    if (LAYOUT == 0) {                       // tile [128 rows][64 k]: one DMA instruction = 8 rows x 128 B
        const int row = 8 * q + (lane >> 3);
        const int lc = (lane & 7) ^ (row & 7);
        const int gr = min(row0 + row, nrows - 1);
        return base + (int64_t)gr * ld + lc * 8;
    } else {                                 // tile [64 k][128 rows]: one DMA instruction = 4 k-rows x 256 B
        const int krow = 4 * q + (lane >> 4);
        const int lc = (lane & 15) ^ swz1(krow);
        int col = row0 + lc * 8;
        if (col >= nrows) col = row0;
        return base + (int64_t)krow * ld + col;
    }
}

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_fast_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, c = lane & 15;

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
    const int m0 = tm * FBM, n0 = tn * FBN;
    const int kt_begin = split * p.ktiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);
    if (kt_begin >= kt_end) return;

    // ---- per-thread DMA sources (4 instructions per operand per wave), advanced by one K-tile per iteration
    const bf16_t* srcA[4];
    const bf16_t* srcB[4];
    const int64_t stepA = LA == 0 ? FBK : (int64_t)FBK * p.lda;
    const int64_t stepB = LB == 0 ? FBK : (int64_t)FBK * p.ldb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        srcA[i] = stage_src<LA>(p.A, p.lda, m0, p.M, wave * 4 + i, lane) + kt_begin * stepA;
        srcB[i] = stage_src<LB>(p.B, p.ldb, n0, p.N, wave * 4 + i, lane) + kt_begin * stepB;
    }
    auto stage = [&](int buf) {
        char* da = smem + buf * F_STAGE + (wave * 4) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) { glds16(srcA[i], da + i * 1024); srcA[i] += stepA; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { glds16(srcB[i], da + F_OPER + i * 1024); srcB[i] += stepB; }
    };

    // ---- per-lane fragment read offsets
    // layout 0: addr = (rbase + i*16 + c)*128 + (((kk*4+g) ^ (c&7)) * 16)
    // layout 1: addr = krow*256 + ((lc ^ s)*16) + sub,  krow = kk*32 + 8g + (c>>2) (+4), lc = (rbase>>3) + 2i + ((c&3)>>1)
    const int a_rb = wm * 64, b_rb = wn * 64;
    const int j4 = c >> 2, s1 = (j4 << 1) | ((g & 1) << 3), sub1 = (c & 1) * 8, h1 = (c & 3) >> 1;

    auto read_frag0 = [&](const char* tile, int rbase, int i, int kk) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(tile + (rbase + i * 16 + c) * 128 + (((kk * 4 + g) ^ (c & 7)) << 4));
    };
    auto read_frag1 = [&](const char* tile, int rbase, int i, int kk) -> bf16x8_t {
        const int krow = kk * 32 + 8 * g + j4;
        const int lc = (rbase >> 3) + 2 * i + h1;
        const int off = ((lc ^ s1) << 4) + sub1;
        const __attribute__((address_space(3))) v4s* p0 = (const __attribute__((address_space(3))) v4s*)(tile + krow * 256 + off);
        const __attribute__((address_space(3))) v4s* p1 = (const __attribute__((address_space(3))) v4s*)(tile + (krow + 4) * 256 + off);
        v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p0);
        v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p1);
        short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };

    float4_t acc[4][4];   // acc[j][i]: n-fragment j, m-fragment i  (D^T layout: lane -> row m = c, 4 consecutive n = 4g..4g+3)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (float4_t){0.f, 0.f, 0.f, 0.f};

    stage(0);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        __syncthreads();                       // (compiler adds vmcnt(0)): tile kt landed for every wave; buf^1 is free
        if (kt + 1 < kt_end) stage(buf ^ 1);
        const char* sa = smem + buf * F_STAGE;
        const char* sb = sa + F_OPER;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = LA == 0 ? read_frag0(sa, a_rb, i, kk) : read_frag1(sa, a_rb, i, kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = LB == 0 ? read_frag0(sb, b_rb, j, kk) : read_frag1(sb, b_rb, j, kk);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
        }
    }

    // ---- register epilogue
    const vm_gemm_epilogue& e = p.e;
    const float alpha = e.alpha_dev ? e.alpha * (*e.alpha_dev) : e.alpha;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + wm * 64 + i * 16 + c;
        if (gm >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + wn * 64 + j * 16 + g * 4;
            if (gn >= p.N) continue;
            const int nvalid = min(4, p.N - gn);
            const int64_t off = (int64_t)gm * p.ldc + gn;
            float v[4] = {acc[j][i][0] * alpha, acc[j][i][1] * alpha, acc[j][i][2] * alpha, acc[j][i][3] * alpha};
            if (p.slabs) {     // split-K: plain partial-slab store, reduced by splitk_reduce_kernel (deterministic, no atomics)
                float* sp = p.slabs + (int64_t)split * p.M * p.ldc + off;
                if (nvalid == 4) *reinterpret_cast<float4*>(sp) = make_float4(v[0], v[1], v[2], v[3]);
                else for (int r = 0; r < nvalid; ++r) sp[r] = v[r];
                continue;
            }
            if (e.bias) {
                if (nvalid == 4) {
                    const float4 b4 = *reinterpret_cast<const float4*>(e.bias + gn);
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                } else {
                    for (int r = 0; r < nvalid; ++r) v[r] += e.bias[gn + r];
                }
            }
            if (e.aux_out) {
                bf16_t* z = reinterpret_cast<bf16_t*>(e.aux_out) + off;
                if (nvalid == 4) *reinterpret_cast<uint2*>(z) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                else for (int r = 0; r < nvalid; ++r) z[r] = f32_to_bf16(v[r]);
            }
            if (e.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
            }
            if (e.mul_gelu_z) {
                const bf16_t* z = reinterpret_cast<const bf16_t*>(e.mul_gelu_z) + off;
                float zf[4] = {0.f, 0.f, 0.f, 0.f};
                if (nvalid == 4) {
                    const uint2 u = *reinterpret_cast<const uint2*>(z);
                    zf[0] = __uint_as_float(u.x << 16); zf[1] = __uint_as_float(u.x & 0xffff0000u);
                    zf[2] = __uint_as_float(u.y << 16); zf[3] = __uint_as_float(u.y & 0xffff0000u);
                } else {
                    for (int r = 0; r < nvalid; ++r) zf[r] = bf16_to_f32(z[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f(zf[r]);
            }
            if (e.dropout_p > 0.f) {
                const uint64_t idx = (uint64_t)gm * (uint64_t)p.N + (uint64_t)gn;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = dropout_keep(e.dropout_seed, idx + r, p.drop_thresh) ? v[r] * p.drop_scale : 0.f;
            }
            if (e.residual) {
                const bf16_t* rp = reinterpret_cast<const bf16_t*>(e.residual) + (int64_t)gm * e.ldr + gn;
                if (nvalid == 4) {
                    const uint2 u = *reinterpret_cast<const uint2*>(rp);
                    v[0] += __uint_as_float(u.x << 16); v[1] += __uint_as_float(u.x & 0xffff0000u);
                    v[2] += __uint_as_float(u.y << 16); v[3] += __uint_as_float(u.y & 0xffff0000u);
                } else {
                    for (int r = 0; r < nvalid; ++r) v[r] += bf16_to_f32(rp[r]);
                }
            }
            if (e.out_dtype == VM_BF16) {
                bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + off;
                if (nvalid == 4) *reinterpret_cast<uint2*>(cp) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                else for (int r = 0; r < nvalid; ++r) cp[r] = f32_to_bf16(v[r]);
            } else {
                float* cp = reinterpret_cast<float*>(p.C) + off;
                if (e.accumulate) {     // this thread owns the element (no split): plain read-modify-write
                    if (nvalid == 4) {
                        float4 o = *reinterpret_cast<float4*>(cp);
                        *reinterpret_cast<float4*>(cp) = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
                    } else {
                        for (int r = 0; r < nvalid; ++r) cp[r] += v[r];
                    }
                } else if (nvalid == 4) {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    for (int r = 0; r < nvalid; ++r) cp[r] = v[r];
                }
            }
        }
    }
}

template <int LA, int LB>
static int launch_fast(const GemmArgs& a, int nblocks, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fast_kernel<LA, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_fast_kernel<LA, LB>), dim3(nblocks), dim3(256), F_LDS, s, a);
    return vm_check_launch("vm_gemm_bf16(fast)");
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

int vm_gemm_fast_dispatch(const GemmArgs& a, int a_layout, int b_layout, int nblocks, hipStream_t s) {
    if (a_layout == 0 && b_layout == 0) return launch_fast<0, 0>(a, nblocks, s);
    if (a_layout == 0 && b_layout == 1) return launch_fast<0, 1>(a, nblocks, s);
    if (a_layout == 1 && b_layout == 0) return launch_fast<1, 0>(a, nblocks, s);
    return launch_fast<1, 1>(a, nblocks, s);
}
