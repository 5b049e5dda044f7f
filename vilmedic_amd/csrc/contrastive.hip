// contrastive.hip -- the image-text contrastive losses (ConVIRT / InfoNCE / GLoRIA-global) on the [B,B] similarity S = a_hat b_hat^T / tau:
//   vm_rownorm_cast      x fp32 [R,D] -> x/max(|x|,eps) as bf16 (+ the norms), or a plain cast (InfoNCE)
//   vm_contrastive_fwd   row / column log-sum-exp and the diagonal of S, tile by tile on the MFMA (the fp32 tiles are kept in the workspace)
//   vm_contrastive_bwd   G = g_row_i softmax_row(S)_ij + g_col_j softmax_col(S)_ij - [i==j](g_row_i+g_col_i)  -> bf16, from the stored tiles
// (round 1 materialised S in fp32 and made three scalar passes over it: 0.314 vs 0.147 ms of kernel time at B = 2048, 1.45 vs 0.80 ms
// at B = 8192 -- profiles/r02_l_contrastive_bench.txt -- that path and its kernels are gone.)
// ref: vilmedic/blocks/losses/selfsup/ConVIRTLoss.py:12-31, InfoNCELoss.py:11-19, GLoRIALoss.py:54-75.
#include "common.h"
#include "gemm_args.h"

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



// ------------------------------------------------------------------ the similarity loss in six short launches
//   forward   contr_prep_kernel    both embedding matrices in ONE launch: x/max(|x|,eps) -> bf16 (or a plain cast), the norms; zeroes the
//                                  paired-diagonal buffer
//             contr_fwd_kernel     one workgroup per 128 x 128 tile of S = A^ B^^T * inv_tau (bf16 MFMA through an LDS-DMA ring, fp32
//                                  accumulate, the tile staged in LDS as fp32): per-row / per-column (max, sum exp) partials, the
//                                  paired-diagonal entries, and [r4] the tile itself into the workspace for the backward pass
//             contr_merge_kernel   one thread per row / column: log-sum-exp over the per-tile partials, the per-row losses
//   backward  contr_g_rows_kernel  [r4] streaming pass over whole rows of the stored S: G = g_r softmax_row + g_c softmax_col -
//                                  [paired](g_r + g_c) as bf16 and the sums p_i = sum_j G_ij S_ij (complete), q_j = sum_i G_ij S_ij
//                                  (one partial per row block) = the a^.da^ / b^.db^ projections of the normalisation backward
//             gemm_pair_kernel     [r4] dA^ = G B^ / tau and dB^ = G^T A^ / tau as 128 x 128 tiles of the library's LDS-DMA GEMM main loop
//                                  (gemm_fast.hip; R, C multiples of 64 -- else contr_grad_kernel, the 64 x BN register-staged tiles)
//             contr_norm_bwd_kernel [r4] row pass over both results: dx = (dx^ - x^ (x^ . dx^)) / |x|, x^ . dx^ from p / the q partials
// Round 4 (profiles/r04_*_contrastive_bench.txt), B = 2048 / 8192: the G pass recomputed every S tile on the MFMA (32.6 / 269 us) -> reads the
// stored tiles (13.7 / 108 us; the first streaming version read the tile by tile layout -- 512-B row segments at a power-of-two stride -- and
// took 20 us, a 16-wave tile form 31 us); the gradient products on 64 x 96 register-staged tiles with ONE chunk in flight (47.3 / 535 us) ->
// GEMM tiles (30.4 / 276 us; a 2-way K split of the 192 tiles bought 4 us and cost 3 in the row pass: not kept) + the row pass (11.8 / 38 us:
// one wave per row summing 512 column partials 4 B at a time took 17.7 us; 8 rows per workgroup read them 32 B at a time).
// Kernel time forward + backward 121 -> 96 us at B = 2048, 1036 -> 661 us at B = 8192 (0.187 of the MFMA peak).  Measured and dropped: a
// 4-slot LDS-DMA ring with counted waits (three chunks in flight, one workgroup per CU) for the forward tiles, 28.8 vs 25.6 us -- the twelve
// K chunks are ~8 us of that launch, the tail (tile store, the two partial passes) and the launch itself are the rest.
// History (profiles/r03_*): round 2 ran this as six library launches + a dozen torch glue kernels (147 us + ~50 us of kernels, 0.34 ms wall at
// B = 2048; the two gradient GEMMs alone took 61 us because 2048 x 768 outputs are 96 tiles of 128 x 128 on 256 CUs).  Round 3 first built it
// as ONE persistent backward launch pulling tiles from a device queue with agent-scope release / acquire hand-offs between the phases, and a
// forward whose last-arriving workgroups merged the partials: correct, deterministic -- and 161 us, because every hand-off (release fence,
// returned ticket atomics, flag poll, acquire fence) costs microseconds on this chip (MI355X_MICROARCH.md price list: 1.7 us per fence,
// ~1 us per returned atomic) where a dependent kernel boundary costs 1.5 us.  Timing the forward tile kernel with its tail removed:
// 7.7 us for launch + one K chunk, 16.1 us for the twelve chunks, 34.5 us with partial passes + arrival + merge -- 18 us of tail for 8 us of
// math.  The phases are therefore plain dependent launches again; what is kept from the persistent version is the arithmetic (G with the
// projections, the normalisation backward in the GEMM epilogue, 64 x 96 gradient tiles that fill the chip).
typedef __bf16 c_bf16x8_t __attribute__((ext_vector_type(8)));
typedef short c_v4s __attribute__((ext_vector_type(4)));
#define CT 128
#define CT_KS 72          // LDS row stride in elements for a [128][64] operand slab (144 B: conflict-spreading pad)
#define CT_CS 132         // fp32 row stride of the staged S tile
#define CT_LDS (2 * 2 * CT * CT_KS * 2)      // two stages x two operands = 73728 B >= 128 * 132 * 4 = 67584 B
#define CT_RED_OFF (CT * CT_CS * 4)          // 2 KB of cross-wave column partials behind the staged tile
#define CT_LDS_ALL CT_LDS
#define CB_M 64           // phase-B output tile: 64 rows x BN columns (BN = 96 when D % 96 == 0, else 128), K chunks of 64
#define CB_PS 72          // LDS row stride (elements) of the G chunk  [64][64]
#define CB_QS 136         // LDS row stride (elements) of the operand chunk [64][<=128]
#define CB_STAGE (CB_M * CB_PS * 2 + 64 * CB_QS * 2)      // 9216 + 17408 B per stage

struct ContrArgs {
    const bf16_t* A; const bf16_t* B; int R, C, D, diag_offset, tiles_m, tiles_n; float inv_tau;
    // forward
    float* row_part; float* col_part; float* diag; float* lse_r_out; float* lse_c_out; float* loss_r; float* loss_c;
    // backward
    const float* lse_r; const float* lse_c; const float* g_r; const float* g_c; bf16_t* G; int64_t ldg;
    float* p_part; float* q_part;                    // [tiles_n][R], [tiles_m][C]
    int nparts_p, nparts_q;                          // partial sums per row of p_part / per column of q_part
    float* S; int64_t ldS;                           // the scaled similarity, fp32 [R][ldS]: written by the forward tiles, read by the G tiles
    const float* a32; const float* b32; const float* na; const float* nb; float* da; float* db;
    int normalize, bn, items_a, items_da, items_db; float eps;
};

__global__ __launch_bounds__(256) void contr_prep_kernel(const float* __restrict__ a, const float* __restrict__ b, bf16_t* __restrict__ ah,
                                                         bf16_t* __restrict__ bh, float* __restrict__ na, float* __restrict__ nb, int R, int C, int D,
                                                         int normalize, float eps, float* __restrict__ diag) {
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < R; i += gridDim.x * 256) diag[i] = 0.f;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < R + C; row += gridDim.x * 4) {
        const bool first = row < R;
        const int r = first ? row : row - R;
        const float* xr = (first ? a : b) + (int64_t)r * D;
        bf16_t* o = (first ? ah : bh) + (int64_t)r * D;
        float ss = 0.f;
        for (int c = lane * 4; c < D; c += 256) { const float4 v = *reinterpret_cast<const float4*>(xr + c); ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
        const float nrm = sqrtf(wave_sum(ss));
        const float inv = normalize ? 1.0f / fmaxf(nrm, eps) : 1.0f;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            uint2 u; u.x = pack_bf16x2(v.x * inv, v.y * inv); u.y = pack_bf16x2(v.z * inv, v.w * inv);
            *reinterpret_cast<uint2*>(o + c) = u;
        }
        if (lane == 0) (first ? na : nb)[r] = nrm;
    }
}

// acc = A^[m0.., :] B^[n0.., :]^T for one 128 x 128 tile (operands global -> registers -> LDS with a one-chunk software pipeline), then the
// scaled tile staged in LDS as fp32 [128][CT_CS].  Ends with a barrier: every thread may read the whole tile.
// 16-B load through an explicitly GLOBAL pointer: the pinned SGPR copies below lose their address space, and a flat load also counts in
// lgkmcnt -- the waits in front of the MFMAs' LDS reads would then wait for the next chunk's operand prefetch as well
__device__ __forceinline__ uint4 contr_ldg16(const void* q) {
    typedef __attribute__((address_space(1))) const uint4_t gu4;
    const uint4_t v = *(gu4*)(uint64_t)q;      // integer -> global pointer: no generic pointer in between
    return make_uint4(v[0], v[1], v[2], v[3]);
}
struct ContrOperands { const bf16_t* A; const bf16_t* B; int R, C, D; float inv_tau; };
// wave-uniform values that arrive through a by-reference argument block (the noinline phase functions of the backward kernel): pulled
// into SGPRs ONCE -- left in memory, hipcc re-loaded each field in front of every use (flat_load + s_waitcnt vmcnt(0) ahead of each operand
// load of the K loop: four dependent round trips per K chunk, the 0.2 ms backward of the first version)
// (the empty asm makes the SGPR copy an opaque definition: without it the compiler re-materialises the value from memory inside the loops)
__device__ __forceinline__ int contr_uni(int v) { int u = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(u)); return u; }
__device__ __forceinline__ float contr_uni(float v) { return __uint_as_float((uint32_t)contr_uni((int)__float_as_uint(v))); }
template <typename P> __device__ __forceinline__ P* contr_uni(P* q) {
    const uint64_t v = (uint64_t)q;
    const uint32_t lo = (uint32_t)contr_uni((int)(uint32_t)v), hi = (uint32_t)contr_uni((int)(uint32_t)(v >> 32));
    return (P*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ ContrOperands contr_operands(const ContrArgs& a) {
    ContrOperands o;
    o.A = contr_uni(a.A); o.B = contr_uni(a.B); o.R = contr_uni(a.R); o.C = contr_uni(a.C); o.D = contr_uni(a.D); o.inv_tau = contr_uni(a.inv_tau);
    return o;
}
// [r4] D % 64 == 0 (every model's embedding width): the operand chunks go HBM / L2 -> LDS by LDS-DMA into a 2-slot ring of un-padded 128-B rows,
// XOR-swizzled on the DMA source (chunk ^= row & 7, the layout-0 image of gemm_fast.hip), the next chunk requested before the current one's
// MFMAs -- the register-staged pipeline below kept ONE 64-deep chunk in flight per workgroup and paid an exposed L2 round trip per chunk
// (24.6 us for the 6.4 GFLOP of the B = 2048 forward tiles).  Rows past R / C re-read the last valid row: they only feed S entries nobody reads.
__device__ __forceinline__ void contr_dma16(const bf16_t* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void contr_s_tile(const ContrOperands p, char* smem, int m0, int n0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    if ((p.D & 63) == 0) {
        const int wv = __builtin_amdgcn_readfirstlane(wave), g = lane >> 4, c = lane & 15;
        const bf16_t* srcA[4];
        const bf16_t* srcB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (wv * 4 + i) + (lane >> 3), lc = (lane & 7) ^ (row & 7);
            srcA[i] = p.A + (int64_t)min(m0 + row, p.R - 1) * p.D + lc * 8;
            srcB[i] = p.B + (int64_t)min(n0 + row, p.C - 1) * p.D + lc * 8;
        }
        constexpr int SLOT = 2 * CT * 128;                 // A chunk [128][64] + B chunk [128][64], bf16
        auto stage = [&](int buf) {
            char* d = smem + buf * SLOT;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                contr_dma16(srcA[i], d + (wv * 4 + i) * 1024); srcA[i] += 64;
                contr_dma16(srcB[i], d + CT * 128 + (wv * 4 + i) * 1024); srcB[i] += 64;
            }
        };
        const int ktiles = p.D >> 6;
        stage(0);
        for (int kt = 0; kt < ktiles; ++kt) {
            __syncthreads();                               // (the compiler adds vmcnt(0)): chunk kt landed for every wave, the other slot is drained
            if (kt + 1 < ktiles) stage((kt + 1) & 1);
            const char* sa = smem + (kt & 1) * SLOT;
            const char* sb = sa + CT * 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                c_bf16x8_t fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = *reinterpret_cast<const c_bf16x8_t*>(sa + (wm * 64 + i * 16 + c) * 128 + (((kk * 4 + g) ^ (c & 7)) << 4));
                    fb[i] = *reinterpret_cast<const c_bf16x8_t*>(sb + (wn * 64 + i * 16 + c) * 128 + (((kk * 4 + g) ^ (c & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();                                   // every wave is done with the ring: the S tile is staged over it
    } else {
    uint4 ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            const int ga = m0 + row, gb = n0 + row, gk = k0 + ch * 8;
            ra[i] = (ga < p.R && gk < p.D) ? contr_ldg16(p.A + (int64_t)ga * p.D + gk) : make_uint4(0, 0, 0, 0);
            rb[i] = (gb < p.C && gk < p.D) ? contr_ldg16(p.B + (int64_t)gb * p.D + gk) : make_uint4(0, 0, 0, 0);
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
}

// log-sum-exp over the nt per-tile (max, sum exp) partials of row / column x: lse = M + log(sum_t s_t exp(m_t - M)).  Eight independent
// 8-B loads in flight per step: the partials were written by other CUs (L2 misses), a dependent chain of nt loads cost ~0.7 us each.
__device__ __forceinline__ float contr_merge_lse(const float* part, int64_t n, int x, int nt) {
    float M = -INFINITY, L = 0.f;
    for (int t0 = 0; t0 < nt; t0 += 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = (t0 + u < nt) ? *reinterpret_cast<const float2*>(part + ((int64_t)(t0 + u) * n + x) * 2) : make_float2(-INFINITY, 0.f);
        float m2 = M;
#pragma unroll
        for (int u = 0; u < 8; ++u) m2 = fmaxf(m2, v[u].x);
        L *= __expf(M - m2);                      // (0 on the first step: M = -inf, m2 finite)
#pragma unroll
        for (int u = 0; u < 8; ++u) L += v[u].y * __expf(v[u].x - m2);
        M = m2;
    }
    return M + __logf(L);
}
__global__ __launch_bounds__(256, 2) void contr_fwd_kernel(const ContrArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x - tm * p.tiles_n;
    const int m0 = tm * CT, n0 = tn * CT;
    contr_s_tile(contr_operands(p), smem, m0, n0);
    const float* cs = reinterpret_cast<const float*>(smem);
    const int rows = min(CT, p.R - m0), cols = min(CT, p.C - n0);
    {   // [r4] the tile goes to the workspace (fp32, 32 contiguous bytes per lane: whole 512-B row segments per 16 lanes): the backward pass
        // reads it back instead of recomputing it on the MFMA (16 MB at B = 2048: L2 / MALL resident between the two launches)
        const int c0 = (tid & 15) * 8;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = (tid >> 4) + 16 * it;
            if (r < rows && c0 < cols) {
                float* o = p.S + (int64_t)(m0 + r) * p.ldS + n0 + c0;
                const float* src = cs + r * CT_CS + c0;
                if (c0 + 8 <= cols) {
                    *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(src);
                    *reinterpret_cast<float4*>(o + 4) = *reinterpret_cast<const float4*>(src + 4);
                } else for (int j = 0; j < cols - c0; ++j) o[j] = src[j];
            }
        }
    }
    if (tid < CT) {                        // threads 0..127: one row each
        const int r = tid;
        if (r < rows) {
            float mx = -INFINITY;
            for (int j = 0; j < cols; ++j) mx = fmaxf(mx, cs[r * CT_CS + j]);
            float se = 0.f;
            for (int j = 0; j < cols; ++j) se += __expf(cs[r * CT_CS + j] - mx);
            float* o = p.row_part + ((int64_t)tn * p.R + m0 + r) * 2;
            o[0] = mx; o[1] = se;
            const int dj = m0 + r + p.diag_offset - n0;             // column of the paired entry inside this tile
            if (dj >= 0 && dj < cols) p.diag[m0 + r] = cs[r * CT_CS + dj];
        }
    } else {                               // threads 128..255: one column each
        const int c = tid - CT;
        if (c < cols) {
            float mx = -INFINITY;
            for (int i = 0; i < rows; ++i) mx = fmaxf(mx, cs[i * CT_CS + c]);
            float se = 0.f;
            for (int i = 0; i < rows; ++i) se += __expf(cs[i * CT_CS + c] - mx);
            float* o = p.col_part + ((int64_t)tm * p.C + n0 + c) * 2;
            o[0] = mx; o[1] = se;
        }
    }
}

// one thread per row / column of S: lse over the per-tile partials, the losses of the pairs (dependent launch behind contr_fwd_kernel)
__global__ __launch_bounds__(256) void contr_merge_kernel(const ContrArgs p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < p.R) {
        const float lse = contr_merge_lse(p.row_part, p.R, i, p.tiles_n);
        p.lse_r_out[i] = lse;
        p.loss_r[i] = lse - p.diag[i];                            // (diag is 0 for a row without a paired column)
    } else if (i - p.R < p.C) {
        const int x = i - p.R;
        const float lse = contr_merge_lse(p.col_part, p.C, x, p.tiles_m);
        p.lse_c_out[x] = lse;
        const int pr = x - p.diag_offset;                         // the row this column is paired with
        p.loss_c[x] = lse - ((pr >= 0 && pr < p.R) ? p.diag[pr] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward
__device__ __forceinline__ c_v4s contr_lds_tr(const char* q) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c_v4s*)(const __attribute__((address_space(3))) c_v4s*)q);
}
// MFMA operand (16 rows or columns x 32 k) from a chunk stored [k][x] row-major (stride bytes): lane (c = lane & 15, g = lane >> 4) gets
// x = x0 + c, k = k0 + 8 g .. + 7: two transposing reads, each over the 4 x 16 block  rows k0 + 8 g (+ 4) .. + 3,  columns x0 .. x0 + 15
__device__ __forceinline__ c_bf16x8_t contr_frag_t(const char* chunk, int stride, int k0, int x0, int lane) {
    const int g = lane >> 4, c = lane & 15;
    const char* q = chunk + (k0 + 8 * g + (c >> 2)) * stride + (x0 + 4 * (c & 3)) * 2;
    const c_v4s lo = contr_lds_tr(q), hi = contr_lds_tr(q + 4 * stride);
    short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(c_bf16x8_t, v);
}
// sum over the 16 lanes of a DPP row (every lane of the row ends with the total)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); v += dpp_f32<0x140>(v);
    return v;
}
__device__ __forceinline__ float quarters_sum(float v) {      // over lanes l, l^16, l^32, l^48
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// G and the projection sums from the forward's S, as a STREAMING pass over row blocks (round 4; the tile form re-ran the S product on the MFMA,
// 32.6 us at B = 2048).  One workgroup = RB whole rows of S: a lane owns 8 consecutive columns of every row of the block (32 B of S in, 16 B
// of G out per row: whole rows are contiguous kilobytes -- the first version read the stored S tile by tile, 128 row segments of 512 B at a
// power-of-two stride per workgroup, and took 20 us for 24 MB), so
//   p_i = sum_j G_ij S_ij   is complete inside the workgroup (lanes, then waves through LDS: one value per row, no partials), and
//   q_j = sum_i G_ij S_ij   leaves one partial per row block and column, owned by one lane (no cross-lane step): q_part[block][C],
//                            summed by the consumer's waves (contr_norm_bwd_kernel).
template <int RB>
__global__ __launch_bounds__(256) void contr_g_rows_kernel(const ContrArgs p) {
    __shared__ float red[4][RB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = blockIdx.x, m0 = rb * RB;
    const int rows = min(RB, p.R - m0);
    float gr[RB], lr[RB], rs[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int r = min(m0 + i, p.R - 1);
        gr[i] = p.g_r[r]; lr[i] = p.lse_r[r]; rs[i] = 0.f;
    }
    const bool vec4 = (p.C & 3) == 0;
    for (int cb = 0; cb < p.C; cb += 2048) {
        const int c0 = cb + tid * 8;
        if (c0 >= p.C) continue;
        const int nv = min(8, p.C - c0);
        float gc[8], lc[8], colacc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = min(c0 + j, p.C - 1);
            gc[j] = p.g_c[c]; lc[j] = p.lse_c[c]; colacc[j] = 0.f;
        }
        float sv[RB][8];
#pragma unroll
        for (int i = 0; i < RB; ++i) {           // every row's 32 B requested before the first exp
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[i][j] = 0.f;
            if (i < rows) {
                const float* src = p.S + (int64_t)(m0 + i) * p.ldS + c0;
                if (nv == 8) {
                    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
                    sv[i][0] = lo.x; sv[i][1] = lo.y; sv[i][2] = lo.z; sv[i][3] = lo.w; sv[i][4] = hi.x; sv[i][5] = hi.y; sv[i][6] = hi.z; sv[i][7] = hi.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j < nv) sv[i][j] = src[j];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (i >= rows) continue;
            const int grow = m0 + i;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float g = 0.f;
                if (j < nv) {
                    const float x = sv[i][j];
                    g = gr[i] * __expf(x - lr[i]) + gc[j] * __expf(x - lc[j]);
                    if (c0 + j == grow + p.diag_offset) g -= gr[i] + gc[j];
                    rs[i] += g * x;
                    colacc[j] += g * x;
                }
                v[j] = g;
            }
            bf16_t* o = p.G + (int64_t)grow * p.ldg + c0;
            if (nv == 8) *reinterpret_cast<uint4*>(o) = pack8(v);
            else for (int j = 0; j < nv; ++j) o[j] = f32_to_bf16(v[j]);
        }
        float* qo = p.q_part + (int64_t)rb * p.C + c0;
        if (nv == 8 && vec4) {
            *reinterpret_cast<float4*>(qo) = make_float4(colacc[0], colacc[1], colacc[2], colacc[3]);
            *reinterpret_cast<float4*>(qo + 4) = make_float4(colacc[4], colacc[5], colacc[6], colacc[7]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < nv) qo[j] = colacc[j];
        }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const float t = wave_sum(rs[i]);
        if (lane == 0) red[wave][i] = t;
    }
    __syncthreads();
    if (tid < rows) p.p_part[m0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
static int contr_rb(int R) { return R > 2048 ? 16 : 4; }      // rows per workgroup of the G pass: >= 512 workgroups from R = 2048 on

// gradient tiles: out[m0.., n0..] (64 x BN, fp32) = sum_k P[m][k] X^[k][n] / tau, normalisation backward in the epilogue.
//   PT = false (dA): P[m][k] = G[m0 + m][k],  X^ = B^,  K = C;    PT = true (dB): P[m][k] = G[k][m0 + m],  X^ = A^,  K = R
// The product is formed transposed (A operand = X^ chunk read through ds_read_b64_tr_b16, rows = n; B operand = the G chunk, columns = m)
// so that a lane ends with 4 consecutive n of one output row: 16-B stores.
template <int NB, bool PT>
__device__ __noinline__ void contr_grad_tile(const ContrArgs& pa, char* smem, int m0, int n0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave >> 1, wm = wave & 1;
    constexpr int BN = NB * 32;                       // NB 16-column blocks per wave x 2 waves
    struct { int R, C, D; int64_t ldg; float inv_tau; const bf16_t* G; } p;      // in SGPRs (see contr_uni)
    p.R = contr_uni(pa.R); p.C = contr_uni(pa.C); p.D = contr_uni(pa.D); p.ldg = (int64_t)contr_uni((int)pa.ldg);
    p.inv_tau = contr_uni(pa.inv_tau); p.G = contr_uni((const bf16_t*)pa.G);
    const int M = PT ? p.C : p.R, K = PT ? p.R : p.C;
    const bf16_t* X = contr_uni(PT ? pa.A : pa.B);
    float4_t acc[NB][2];
#pragma unroll
    for (int i = 0; i < NB; ++i) { acc[i][0] = (float4_t){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (float4_t){0.f, 0.f, 0.f, 0.f}; }
    uint4 rp[2], rq[NB];
    // G chunk: 64 x 64 elements = 512 16-B pieces (2 per thread); operand chunk: 64 x BN = 8 NB x 64 pieces (NB per thread)
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            const int gr = (PT ? k0 : m0) + row, gc = (PT ? m0 : k0) + ch * 8;     // G[gr][gc .. gc + 7]
            rp[i] = (gr < p.R && gc < p.C) ? contr_ldg16(p.G + (int64_t)gr * p.ldg + gc) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int id = tid + 256 * i, row = id / (BN / 8), ch = id - row * (BN / 8);
            const int gk = k0 + row, gn = n0 + ch * 8;
            rq[i] = (gk < K && gn < p.D) ? contr_ldg16(X + (int64_t)gk * p.D + gn) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store = [&](char* st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            *reinterpret_cast<uint4*>(st + row * (CB_PS * 2) + ch * 16) = rp[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int id = tid + 256 * i, row = id / (BN / 8), ch = id - row * (BN / 8);
            *reinterpret_cast<uint4*>(st + CB_M * CB_PS * 2 + row * (CB_QS * 2) + ch * 16) = rq[i];
        }
    };
    // (the tail of G past the row's C columns is never read as data: pieces are guarded by gc < C, and ldg >= C rounded to 8 with the
    //  phase-A tile writing only columns < C -- the last piece of a row may hold up to 7 stale bf16 values when C % 8 != 0, so zero them)
    const int ktiles = (K + 63) / 64;
    load(0);
    store(smem);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const char* sp = smem + (kt & 1) * CB_STAGE;
        const char* sq = sp + CB_M * CB_PS * 2;
        if (kt + 1 < ktiles) load((kt + 1) * 64);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            c_bf16x8_t fq[NB], fp[2];
#pragma unroll
            for (int i = 0; i < NB; ++i) fq[i] = contr_frag_t(sq, CB_QS * 2, kk * 32, wn * (BN / 2) + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (PT) fp[j] = contr_frag_t(sp, CB_PS * 2, kk * 32, wm * 32 + j * 16, lane);
                else fp[j] = *reinterpret_cast<const c_bf16x8_t*>(sp + (wm * 32 + j * 16 + (lane & 15)) * (CB_PS * 2) + (kk * 32 + (lane >> 4) * 8) * 2);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[i], fp[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < ktiles) store(smem + ((kt + 1) & 1) * CB_STAGE);
        __syncthreads();
    }
    // epilogue: lane holds out[m][n .. n + 3], m = m0 + wm 32 + j 16 + (lane & 15), n = n0 + wn BN/2 + i 16 + (lane >> 4) 4; the plain product
    // (the L2-normalisation backward is contr_norm_bwd_kernel's row pass, for this path and the GEMM-tile path alike)
    float* out = contr_uni(PT ? pa.db : pa.da);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 32 + j * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = n0 + wn * (BN / 2) + i * 16 + (lane >> 4) * 4;
            if (n >= p.D) continue;
            *reinterpret_cast<float4*>(out + (int64_t)m * p.D + n) =
                make_float4(acc[i][j][0] * p.inv_tau, acc[i][j][1] * p.inv_tau, acc[i][j][2] * p.inv_tau, acc[i][j][3] * p.inv_tau);
        }
    }
    __syncthreads();                 // the LDS stages are reused by the next item
}

// items [0, items_da): tiles of dA; [items_da, items_da + items_db): tiles of dB (dependent launch behind contr_g_kernel)
__global__ __launch_bounds__(256, 2) void contr_grad_kernel(const ContrArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nbt = (p.D + p.bn - 1) / p.bn;                 // output column blocks
    const bool isb = (int)blockIdx.x >= p.items_da;
    const int it = (int)blockIdx.x - (isb ? p.items_da : 0);
    const int rb = it / nbt, cb = it - rb * nbt;
    const int m0 = rb * CB_M, n0 = cb * p.bn;
    if (p.bn == 96) { if (isb) contr_grad_tile<3, true>(p, smem, m0, n0); else contr_grad_tile<3, false>(p, smem, m0, n0); }
    else            { if (isb) contr_grad_tile<4, true>(p, smem, m0, n0); else contr_grad_tile<4, false>(p, smem, m0, n0); }
}

// [r4] R and C multiples of 64: the two gradient products run on the library's 128 x 128 LDS-DMA GEMM main loop (vm_gemm_pair_launch, fp32
// output scaled by 1 / tau) -- the 64 x 96 register-staged tiles above kept ONE 64-deep chunk in flight and filled 335 MB of LDS for 13 GFLOP
// (47 us at B = 2048) -- and the L2-normalisation backward is this row pass over the two results, in place:
// dx = (dx^ - x^ (x^ . dx^)) / |x| with x^ . dx^ = the G tiles' partial sums (one wave per row).
#define CN_ROWS 8
__global__ __launch_bounds__(256) void contr_norm_bwd_kernel(const ContrArgs p) {
    __shared__ float sacc[32][CN_ROWS + 1];
    __shared__ float sk[CN_ROWS], sd[CN_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * CN_ROWS;
    // (1) x^ . dx^ of the block's 8 rows: thread (group tq of 32, row c of 8) adds the partials t = tq, tq + 32, ... -- 8 adjacent rows per
    // partial line (one wave per row read 4 B out of every 128-B line: 134 MB of L2 traffic for the 4 MB of column partials at B = 2048)
    if (p.normalize) {
        const int c = tid & (CN_ROWS - 1), tq = tid / CN_ROWS;
        const int row = base + c;
        float acc = 0.f;
        if (row < p.R + p.C) {
            const bool first = row < p.R;
            const int r = first ? row : row - p.R, M = first ? p.R : p.C, nparts = first ? p.nparts_p : p.nparts_q;
            const float* part = (first ? p.p_part : p.q_part) + r;
            for (int t = tq; t < nparts; t += 128) {      // four independent loads per step; fixed order
                const float v0 = part[(int64_t)t * M], v1 = t + 32 < nparts ? part[(int64_t)(t + 32) * M] : 0.f;
                const float v2 = t + 64 < nparts ? part[(int64_t)(t + 64) * M] : 0.f, v3 = t + 96 < nparts ? part[(int64_t)(t + 96) * M] : 0.f;
                acc += (v0 + v1) + (v2 + v3);
            }
        }
        sacc[tq][c] = acc;
        __syncthreads();
        if (tid < CN_ROWS) {
            float proj = 0.f;
#pragma unroll
            for (int t = 0; t < 32; ++t) proj += sacc[t][tid];
            const int row2 = base + tid;
            float k = 0.f, d = 1.f;
            if (row2 < p.R + p.C) {
                const float nm = row2 < p.R ? p.na[row2] : p.nb[row2 - p.R];
                d = fmaxf(nm, p.eps);
                k = nm > p.eps ? proj / d : 0.f;
            }
            sk[tid] = k; sd[tid] = d;
        }
        __syncthreads();
    }
    // (2) the rows themselves, two per wave: dx = (dx^ - x^ (x^ . dx^)) / |x| in place, every load of a row requested before its first use
#pragma unroll
    for (int i = 0; i < CN_ROWS / 4; ++i) {
        const int lr = wave * (CN_ROWS / 4) + i, row = base + lr;
        if (row >= p.R + p.C) continue;
        const bool first = row < p.R;
        const int r = first ? row : row - p.R;
        const float k = p.normalize ? sk[lr] : 0.f, d = p.normalize ? sd[lr] : 1.f;
        const float* x = (first ? p.a32 : p.b32) + (int64_t)r * p.D;
        float* o = (first ? p.da : p.db) + (int64_t)r * p.D;
        for (int c0 = lane * 4; c0 < p.D; c0 += 1024) {
            float4 g[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 256 * u;
                g[u] = make_float4(0.f, 0.f, 0.f, 0.f); xv[u] = g[u];
                if (c < p.D) {
                    g[u] = *reinterpret_cast<const float4*>(o + c);
                    if (p.normalize) xv[u] = *reinterpret_cast<const float4*>(x + c);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 256 * u;
                if (c < p.D) *reinterpret_cast<float4*>(o + c) = make_float4((g[u].x - xv[u].x * k) / d, (g[u].y - xv[u].y * k) / d, (g[u].z - xv[u].z * k) / d, (g[u].w - xv[u].w * k) / d);
            }
        }
    }
}
static GemmArgs contr_gemm_args(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, float alpha) {
    GemmArgs a = {};
    a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.tiles_m = (M + 127) / 128; a.tiles_n = (N + 127) / 128; a.ktiles = K / 64; a.ktiles_per_split = a.ktiles; a.group_w = a.tiles_n;
    a.e.alpha = alpha; a.e.out_dtype = VM_F32; a.e.split_k = 1;
    a.drop_scale = 1.0f;
    return a;
}

static int contr_common(const char* fn, const void* a, const void* b, int R, int C, int D) {
    VM_REQUIRE(a && b && R > 0 && C > 0 && D > 0 && (D % 8) == 0, "%s: bad arguments (R=%d C=%d D=%d, D must be a multiple of 8)", fn, R, C, D);
    VM_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0, "%s: operands must be 16-byte aligned", fn);
    return VM_OK;
}
static void contr_attr() {
    static bool set = false;
    if (set) return;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&contr_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CT_LDS_ALL);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&contr_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CT_LDS_ALL);
    set = true;
}
// workspace layout (bytes, every block 256-B aligned): diag [R] | row_part [tn][R][2] | col_part [tm][C][2] | p_part [tn][R] | q_part [tm][C] |
// G bf16 [R][ldg] | S fp32 [R][ldS] (forward -> backward)
struct ContrWs { size_t diag, row_part, col_part, p_part, q_part, G, S, total; int64_t ldg, ldS; };
static ContrWs contr_ws_layout(int R, int C) {
    const size_t tm = (R + CT - 1) / CT, tn = (C + CT - 1) / CT;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    ContrWs w;
    w.ldg = (C + 7) / 8 * 8;
    w.diag = 0;
    w.row_part = w.diag + up((size_t)R * 4);
    w.col_part = w.row_part + up(tn * (size_t)R * 8);
    w.p_part = w.col_part + up(tm * (size_t)C * 8);
    w.q_part = w.p_part + up(tn * (size_t)R * 4);
    w.G = w.q_part + up((size_t)((R + contr_rb(R) - 1) / contr_rb(R)) * (size_t)C * 4);
    w.ldS = (C + 7) / 8 * 8;
    w.S = w.G + up((size_t)R * w.ldg * 2);
    w.total = w.S + up((size_t)R * w.ldS * 4);
    return w;
}
extern "C" size_t vm_contrastive_ws(int R, int C) { return (R > 0 && C > 0) ? contr_ws_layout(R, C).total : 0; }

static void contr_fill(ContrArgs& p, const void* ah, const void* bh, int R, int C, int D, float inv_tau, int diag_offset, char* ws, const ContrWs& w) {
    p.A = (const bf16_t*)ah; p.B = (const bf16_t*)bh; p.R = R; p.C = C; p.D = D; p.diag_offset = diag_offset; p.inv_tau = inv_tau;
    p.tiles_m = (R + CT - 1) / CT; p.tiles_n = (C + CT - 1) / CT;
    p.diag = (float*)(ws + w.diag);
    p.row_part = (float*)(ws + w.row_part); p.col_part = (float*)(ws + w.col_part);
    p.p_part = (float*)(ws + w.p_part); p.q_part = (float*)(ws + w.q_part);
    p.G = (bf16_t*)(ws + w.G); p.ldg = w.ldg;
    p.S = (float*)(ws + w.S); p.ldS = w.ldS;
}

extern "C" int vm_contrastive_loss_fwd(const float* a, const float* b, int R, int C, int D, int normalize, float eps, float inv_tau, int diag_offset,
                                       void* a_hat, void* b_hat, float* norm_a, float* norm_b, float* lse_rows, float* lse_cols,
                                       float* loss_rows, float* loss_cols, void* ws, size_t ws_bytes, void* stream) {
    int rc = contr_common("vm_contrastive_loss_fwd", a, b, R, C, D);
    if (rc) return rc;
    const ContrWs w = contr_ws_layout(R, C);
    VM_REQUIRE(a_hat && b_hat && norm_a && norm_b && lse_rows && lse_cols && loss_rows && loss_cols && ws && ws_bytes >= w.total &&
               ((uintptr_t)ws % 256) == 0 && ((uintptr_t)a_hat % 16) == 0 && ((uintptr_t)b_hat % 16) == 0,
               "vm_contrastive_loss_fwd: outputs / workspace (%zu bytes, 256-B aligned)", w.total);
    hipStream_t s = (hipStream_t)stream;
    ContrArgs p = {};
    contr_fill(p, a_hat, b_hat, R, C, D, inv_tau, diag_offset, (char*)ws, w);
    p.lse_r_out = lse_rows; p.lse_c_out = lse_cols; p.loss_r = loss_rows; p.loss_c = loss_cols;
    contr_attr();
    {
        VmProfScope prof(VM_FAM_LOSS, 12.0 * (R + C) * (double)D, s, "contrastive_prep_R%d_C%d_D%d", R, C, D);
        int blocks = (R + C + 3) / 4; if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(contr_prep_kernel, dim3(blocks), dim3(256), 0, s, a, b, (bf16_t*)a_hat, (bf16_t*)b_hat, norm_a, norm_b, R, C, D, normalize, eps,
                           p.diag);
    }
    {
        VmProfScope prof(VM_FAM_LOSS, 2.0 * R * (double)C * D, s, "contrastive_fwd_R%d_C%d_D%d", R, C, D);
        hipLaunchKernelGGL(contr_fwd_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), CT_LDS_ALL, s, p);
    }
    {
        VmProfScope prof(VM_FAM_LOSS, 16.0 * ((double)p.tiles_n * R + (double)p.tiles_m * C), s, "contrastive_merge_R%d_C%d", R, C);
        hipLaunchKernelGGL(contr_merge_kernel, dim3((R + C + 255) / 256), dim3(256), 0, s, p);
    }
    return vm_check_launch("vm_contrastive_loss_fwd");
}

extern "C" int vm_contrastive_loss_bwd(const float* a, const float* b, const void* a_hat, const void* b_hat, const float* norm_a, const float* norm_b,
                                       int R, int C, int D, int normalize, float eps, float inv_tau, int diag_offset,
                                       const float* lse_rows, const float* lse_cols, const float* g_rows, const float* g_cols,
                                       float* da, float* db, void* ws, size_t ws_bytes, void* stream) {
    int rc = contr_common("vm_contrastive_loss_bwd", a_hat, b_hat, R, C, D);
    if (rc) return rc;
    const ContrWs w = contr_ws_layout(R, C);
    VM_REQUIRE(a && b && norm_a && norm_b && lse_rows && lse_cols && g_rows && g_cols && da && db && ws && ws_bytes >= w.total &&
               ((uintptr_t)ws % 256) == 0 && ((uintptr_t)da % 16) == 0 && ((uintptr_t)db % 16) == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0,
               "vm_contrastive_loss_bwd: bad arguments (workspace %zu bytes, 256-B aligned)", w.total);
    hipStream_t s = (hipStream_t)stream;
    ContrArgs p = {};
    contr_fill(p, a_hat, b_hat, R, C, D, inv_tau, diag_offset, (char*)ws, w);
    p.lse_r = lse_rows; p.lse_c = lse_cols; p.g_r = g_rows; p.g_c = g_cols;
    p.a32 = a; p.b32 = b; p.na = norm_a; p.nb = norm_b; p.da = da; p.db = db; p.normalize = normalize; p.eps = eps;
    p.bn = (D % 96 == 0) ? 96 : 128;
    const int nbt = (D + p.bn - 1) / p.bn;
    p.items_a = p.tiles_m * p.tiles_n;
    p.items_da = ((R + CB_M - 1) / CB_M) * nbt;
    p.items_db = ((C + CB_M - 1) / CB_M) * nbt;
    contr_attr();
    if (w.ldg != C) hipMemsetAsync(p.G, 0, (size_t)R * w.ldg * 2, s);      // ragged C: the pad columns of G are read as zeros
    const int rb = contr_rb(R);
    p.nparts_p = 1; p.nparts_q = (R + rb - 1) / rb;
    {
        VmProfScope prof(VM_FAM_LOSS, 12.0 * R * (double)C, s, "contrastive_g_R%d_C%d_D%d", R, C, D);
        if (rb == 4) hipLaunchKernelGGL(contr_g_rows_kernel<4>, dim3(p.nparts_q), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(contr_g_rows_kernel<16>, dim3(p.nparts_q), dim3(256), 0, s, p);
    }
    {
        VmProfScope prof(VM_FAM_LOSS, 4.0 * R * (double)C * D, s, "contrastive_grad_R%d_C%d_D%d", R, C, D);
        if ((R % 64) == 0 && (C % 64) == 0 && !vm_env().gemm_generic) {
            // dA^ [R, D] = G [R, C] (row-major) . B^ [C, D] (k-major);  dB^ [C, D] = G^T (G is its k-major form) . A^ [R, D] (k-major)
            const GemmArgs ga = contr_gemm_args(p.G, p.ldg, p.B, D, da, D, R, D, C, inv_tau);
            const GemmArgs gb = contr_gemm_args(p.G, p.ldg, p.A, D, db, D, C, D, R, inv_tau);
            const int rc2 = vm_gemm_pair_launch(ga, gb, s);
            if (rc2) return rc2;
        } else {                                   // ragged sizes: the register-staged 64 x BN tiles (plain products: the normalisation follows)
            hipLaunchKernelGGL(contr_grad_kernel, dim3(p.items_da + p.items_db), dim3(256), CT_LDS_ALL, s, p);
        }
    }
    if (normalize) {
        VmProfScope prof(VM_FAM_LOSS, 12.0 * (R + C) * (double)D, s, "contrastive_norm_R%d_C%d_D%d", R, C, D);
        hipLaunchKernelGGL(contr_norm_bwd_kernel, dim3((R + C + CN_ROWS - 1) / CN_ROWS), dim3(256), 0, s, p);
    }
    return vm_check_launch("vm_contrastive_loss_bwd");
}
