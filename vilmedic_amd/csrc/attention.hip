// attention.hip -- flash-style multi-head attention forward / backward for gfx950 (dh = 64).
//
// One workgroup = 4 waves; each wave OWNS a 16-row fragment (16 queries in fwd / dQ, 16 keys in dK/dV)
// that it keeps in registers as the MFMA B operand, and the other sequence streams through LDS in
// 64-row tiles shared by the 4 waves (K,V tiles for fwd/dQ; Q,dO tiles for dK/dV).
//
// Everything is computed TRANSPOSED ("swapped QK^T"): S^T = K Q^T puts one query per lane column, so
//   * the softmax row reduction is in-register plus two cross-lane shuffles (xor 16, 32);
//   * P^T in the MFMA C/D layout is directly the B operand of O^T = V^T P^T -- no LDS round trip
//     (the k-slot <-> key permutation this implies is applied consistently to the A operand);
//   * V^T / K^T / Q^T / dO^T A-operands are gathered from row-major LDS tiles with the gfx950
//     transpose read ds_read_b64_tr_b16.
// Masks: key_mask (1 = attend) and causal are applied as a -1e30 score (a fully masked row degrades to
// the uniform distribution exactly like HF's additive finfo.min mask); keys >= Lk get -inf.
// stats[b,h,i] = (row max m, row sum l) of the scaled+masked scores, saved for the backward pass.
#include "attention_common.h"
#include <cstdlib>

#define TROWS 64                 // rows per streamed tile
#define TSTRIDE (DH * 2 + 16)    // bytes per LDS tile row (DH bf16 + 16 B pad); DH is a template parameter (32/64/96/128)
#define TILE_BYTES (TROWS * TSTRIDE)
#define NLD (DH / 32)            // 16-B chunks each thread stages per tile (64 rows x DH/8 chunks / 256 threads)
#define NDF (DH / 16)            // 16-wide output fragments along the head dimension
#define NKK (DH / 32)            // 32-deep MFMA k-steps along the head dimension
#ifndef ATTN_MIN_BLOCKS
#define ATTN_MIN_BLOCKS 2      // __launch_bounds__ second argument (waves per SIMD) of the streaming variants
#endif

// A operand from a row-major LDS tile, contraction = the tile's columns (dh): rows rbase+(lane&15)
template <int DH>
__device__ __forceinline__ bf16x8_t lds_frag_rows(const char* tile, int rbase, int kk, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(tile + (rbase + (lane & 15)) * TSTRIDE + (kk * 32 + (lane >> 4) * 8) * 2);
}
// A operand = tile^T: output rows = tile columns cbase+(lane&15); contraction = tile rows.
// k-slot (g,e) of step s maps to tile row 32 s + 4 g + e (e<4) / 32 s + 16 + 4 g + (e-4) (e>=4),
// matching a B operand packed from two C/D fragments {frag 2s, frag 2s+1}.
template <int DH>
__device__ __forceinline__ bf16x8_t lds_frag_tr(const char* tile, int cbase, int s, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int row = 32 * s + 4 * g + (i >> 2);
    const int col = cbase + (i & 3) * 4;
    const __attribute__((address_space(3))) v4s* p0 = (const __attribute__((address_space(3))) v4s*)(tile + row * TSTRIDE + col * 2);
    const __attribute__((address_space(3))) v4s* p1 = (const __attribute__((address_space(3))) v4s*)(tile + (row + 16) * TSTRIDE + col * 2);
    v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p0);
    v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p1);
    short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
// stream one 64 x DH bf16 tile (rows row0.. of a [rows, ld] matrix at column offset already applied)
template <int DH> struct Stage { uint4 r[NLD]; };
template <int DH>
__device__ __forceinline__ Stage<DH> tile_load(const bf16_t* base, int64_t ld, int row0, int nrows, int tid) {
    Stage<DH> st;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int id = tid + 256 * i, row = id / (DH / 8), c = id % (DH / 8);
        st.r[i] = ld16_or_zero(base + (int64_t)(row0 + row) * ld + c * 8, base, row0 + row < nrows);
    }
    return st;
}
// same, rows gathered through an index table (absolute row numbers): decode-time KV cache with beam indirection
template <int DH>
__device__ __forceinline__ Stage<DH> tile_load_indexed(const bf16_t* base0, int64_t ld, const int32_t* idx, int row0, int nrows, int tid) {
    Stage<DH> st;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int id = tid + 256 * i, row = id / (DH / 8), c = id % (DH / 8);
        const bool ok = row0 + row < nrows;
        const int64_t r = ok ? idx[row0 + row] : 0;
        st.r[i] = ld16_or_zero(base0 + r * ld + c * 8, base0, ok);
    }
    return st;
}
template <int DH>
__device__ __forceinline__ void tile_store(const Stage<DH>& st, char* tile, int tid) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int id = tid + 256 * i, row = id / (DH / 8), c = id % (DH / 8);
        *reinterpret_cast<uint4*>(tile + row * TSTRIDE + c * 16) = st.r[i];
    }
}
// store a transposed 64(dh) x 16(rows) accumulator as rows of a [*, ld] bf16 matrix: lane owns row (lane&15)
template <int DH>
__device__ __forceinline__ void store_t_acc(bf16_t* rowptr, const float4_t (&acc)[NDF], float mul, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int f = 0; f < NDF; ++f) {
        uint2 u;
        u.x = pack_bf16x2(acc[f][0] * mul, acc[f][1] * mul);
        u.y = pack_bf16x2(acc[f][2] * mul, acc[f][3] * mul);
        *reinterpret_cast<uint2*>(rowptr + 16 * f + 4 * g) = u;
    }
}

// =============================================================================== forward
// RES = true: every K/V tile of the (b,h) pair is resident in LDS (Lk <= 256): one barrier for the whole kernel, all
// HBM loads issued up front; RES = false: tiles stream through one LDS slot with a register prefetch (any Lk).
template <int DH, bool RES>
__global__ __launch_bounds__(256, RES ? 4 : ATTN_MIN_BLOCKS) void attn_fwd_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NSLOT = RES ? p.nslot_k : 1;
    uint8_t* smask_all = reinterpret_cast<uint8_t*>(smem + NSLOT * 2 * TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int qrow = q0 + c;
    const bool qok = qrow < p.Lq;
    const bf16_t* qp = p.q + (int64_t)(b * p.Lq + qrow) * p.ldq + h * DH;
    bf16x8_t qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = ld_frag_global(qp + kk * 32 + g * 8, p.q, qok);
    const bf16_t* kbase = p.k + (int64_t)b * p.Lk * p.ldk + h * DH;
    const bf16_t* vbase = p.v + (int64_t)b * p.Lk * p.ldv + h * DH;
    float4_t o[NDF];
#pragma unroll
    for (int f = 0; f < NDF; ++f) o[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    int ntiles = (p.Lk + TROWS - 1) / TROWS;
    if (p.causal) ntiles = min(ntiles, min((int)blockIdx.x * 64 + 63, p.Lq - 1) / TROWS + 1);
    Stage<DH> rk, rv;
    const int32_t* kvi = p.kv_index ? p.kv_index + (int64_t)b * p.kv_index_ld : nullptr;
    const bf16_t* k0p = p.k + h * DH;
    const bf16_t* v0p = p.v + h * DH;
    auto load_kv = [&](int t) {
        if (kvi) { rk = tile_load_indexed<DH>(k0p, p.ldk, kvi, t * TROWS, p.Lk, tid); rv = tile_load_indexed<DH>(v0p, p.ldv, kvi, t * TROWS, p.Lk, tid); }
        else { rk = tile_load<DH>(kbase, p.ldk, t * TROWS, p.Lk, tid); rv = tile_load<DH>(vbase, p.ldv, t * TROWS, p.Lk, tid); }
    };
    if (RES) {
        for (int t = 0; t < ntiles; ++t) {
            load_kv(t);
            tile_store<DH>(rk, smem + t * 2 * TILE_BYTES, tid);
            tile_store<DH>(rv, smem + t * 2 * TILE_BYTES + TILE_BYTES, tid);
        }
        {
            const int key = tid;     // 256 threads cover up to 256 keys
            smask_all[tid] = (p.key_mask && key < p.Lk) ? p.key_mask[(int64_t)b * p.Lk + key] : (uint8_t)1;
        }
        __syncthreads();
    } else {
        load_kv(0);
    }
    for (int kt = 0; kt < ntiles; ++kt) {
        char* sk = RES ? smem + kt * 2 * TILE_BYTES : smem;
        char* sv = sk + TILE_BYTES;
        uint8_t* smask = RES ? smask_all + kt * TROWS : smask_all;
        if (!RES) {
            __syncthreads();                   // previous tile fully consumed
            tile_store<DH>(rk, sk, tid);
            tile_store<DH>(rv, sv, tid);
            if (tid < 64) {
                const int key = kt * TROWS + tid;
                smask[tid] = (p.key_mask && key < p.Lk) ? p.key_mask[(int64_t)b * p.Lk + key] : (uint8_t)1;
            }
            __syncthreads();
            if (kt + 1 < ntiles) load_kv(kt + 1);
        }
        float4_t s[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
                s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_rows<DH>(sk, 16 * f, kk, lane), qf[kk], s[f], 0, 0, 0);
        }
        float mt = -INFINITY;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const uint32_t mk = *reinterpret_cast<const uint32_t*>(smask + 16 * f + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * TROWS + 16 * f + 4 * g + r;
                float val = s[f][r] * p.scale;
                const bool keep = ((mk >> (8 * r)) & 0xff) != 0 && !(p.causal && key > qrow);
                val = keep ? val : MASKED_SCORE;
                val = key < p.Lk ? val : -INFINITY;
                s[f][r] = val;
                mt = fmaxf(mt, val);
            }
        }
        mt = col_max(mt);
        const float m_new = fmaxf(m, mt);
        const float alpha = __expf(m - m_new);
        float ls = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __expf(s[f][r] - m_new); s[f][r] = e; ls += e; }
        l = l * alpha + col_sum(ls);
        m = m_new;
#pragma unroll
        for (int f = 0; f < NDF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[f][r] *= alpha;
        if (p.dropout_p > 0.f) {
            const uint64_t base = ((uint64_t)(b * p.H + h) * p.Lq + qrow) * (uint64_t)((p.Lk + 1) & ~1);   // even row pitch
            const DropKey dk_ = drop_key(eff_seed(p.seed, p.seed_dev));
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                bool keep[4];
                dropout_keep_n<4>(dk_, base + (kt * TROWS + 16 * f + 4 * g), p.thresh, keep);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[f][r] = keep[r] ? s[f][r] * p.drop_scale : 0.f;
            }
        }
        bf16x8_t pb[2] = {pack_b_operand(s[0], s[1]), pack_b_operand(s[2], s[3])};
#pragma unroll
        for (int df = 0; df < NDF; ++df)
#pragma unroll
            for (int st = 0; st < 2; ++st)
                o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_tr<DH>(sv, 16 * df, st, lane), pb[st], o[df], 0, 0, 0);
    }
    if (qok) {
        store_t_acc<DH>(p.out + (int64_t)(b * p.Lq + qrow) * p.ldo + h * DH, o, 1.0f / l, lane);
        if (g == 0) {
            float* st = p.stats + ((int64_t)(b * p.H + h) * p.Lq + qrow) * 2;
            st[0] = m; st[1] = l;
        }
    }
}

// =============================================================================== delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnArgs p, int DH) {
    // one 16-lane group per (b,h,row)
    const int64_t gid = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t total = (int64_t)p.B * p.H * p.Lq;
    const int sub = threadIdx.x & 15;
    float acc = 0.f;
    if (gid < total) {
        const int row = (int)(gid % p.Lq); const int bh = (int)(gid / p.Lq); const int h = bh % p.H, b = bh / p.H;
        const bf16_t* op = p.o + (int64_t)(b * p.Lq + row) * p.ldo + h * DH;
        const bf16_t* dp = p.d_o + (int64_t)(b * p.Lq + row) * p.lddo + h * DH;
        for (int e = sub * 4; e < DH; e += 64) {
            const uint2 a = *reinterpret_cast<const uint2*>(op + e);
            const uint2 d = *reinterpret_cast<const uint2*>(dp + e);
            acc += __uint_as_float(a.x << 16) * __uint_as_float(d.x << 16) + __uint_as_float(a.x & 0xffff0000u) * __uint_as_float(d.x & 0xffff0000u)
                 + __uint_as_float(a.y << 16) * __uint_as_float(d.y << 16) + __uint_as_float(a.y & 0xffff0000u) * __uint_as_float(d.y & 0xffff0000u);
        }
    }
    acc += dpp_f32<0xB1>(acc);  acc += dpp_f32<0x4E>(acc);      // sum over the row of 16 lanes: DPP steps of wave_sum (common.h)
    acc += dpp_f32<0x141>(acc); acc += dpp_f32<0x140>(acc);
    if (gid < total && sub == 0) p.delta[gid] = acc;
}

// =============================================================================== dQ  (owner: 16 queries per wave)
template <int DH, bool RES>
__global__ __launch_bounds__(256, RES ? 4 : ATTN_MIN_BLOCKS) void attn_bwd_dq_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NSLOT = RES ? p.nslot_k : 1;
    uint8_t* smask_all = reinterpret_cast<uint8_t*>(smem + NSLOT * 2 * TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int qrow = q0 + c;
    const bool qok = qrow < p.Lq;
    const bf16_t* qp = p.q + (int64_t)(b * p.Lq + qrow) * p.ldq + h * DH;
    const bf16_t* dop = p.d_o + (int64_t)(b * p.Lq + qrow) * p.lddo + h * DH;
    bf16x8_t qf[NKK], dof[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) { qf[kk] = ld_frag_global(qp + kk * 32 + g * 8, p.q, qok); dof[kk] = ld_frag_global(dop + kk * 32 + g * 8, p.d_o, qok); }
    float m = 0.f, inv_l = 0.f, delta = 0.f;
    if (qok) {
        const int64_t si = (int64_t)(b * p.H + h) * p.Lq + qrow;
        m = p.stats[si * 2]; inv_l = 1.0f / p.stats[si * 2 + 1]; delta = p.delta[si];
    }
    const bf16_t* kbase = p.k + (int64_t)b * p.Lk * p.ldk + h * DH;
    const bf16_t* vbase = p.v + (int64_t)b * p.Lk * p.ldv + h * DH;
    float4_t dq[NDF];
#pragma unroll
    for (int f = 0; f < NDF; ++f) dq[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
    int ntiles = (p.Lk + TROWS - 1) / TROWS;
    if (p.causal) ntiles = min(ntiles, min((int)blockIdx.x * 64 + 63, p.Lq - 1) / TROWS + 1);
    Stage<DH> rk, rv;
    auto load_kv = [&](int t) {
        rk = tile_load<DH>(kbase, p.ldk, t * TROWS, p.Lk, tid);
        rv = tile_load<DH>(vbase, p.ldv, t * TROWS, p.Lk, tid);
    };
    if (RES) {
        for (int t = 0; t < ntiles; ++t) {
            load_kv(t);
            tile_store<DH>(rk, smem + t * 2 * TILE_BYTES, tid);
            tile_store<DH>(rv, smem + t * 2 * TILE_BYTES + TILE_BYTES, tid);
        }
        smask_all[tid] = (p.key_mask && tid < p.Lk) ? p.key_mask[(int64_t)b * p.Lk + tid] : (uint8_t)1;
        __syncthreads();
    } else {
        load_kv(0);
    }
    for (int kt = 0; kt < ntiles; ++kt) {
        char* sk = RES ? smem + kt * 2 * TILE_BYTES : smem;
        char* sv = sk + TILE_BYTES;
        uint8_t* smask = RES ? smask_all + kt * TROWS : smask_all;
        if (!RES) {
            __syncthreads();
            tile_store<DH>(rk, sk, tid);
            tile_store<DH>(rv, sv, tid);
            if (tid < 64) {
                const int key = kt * TROWS + tid;
                smask[tid] = (p.key_mask && key < p.Lk) ? p.key_mask[(int64_t)b * p.Lk + key] : (uint8_t)1;
            }
            __syncthreads();
            if (kt + 1 < ntiles) load_kv(kt + 1);
        }
        float4_t s[4], dp[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
            dp[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_rows<DH>(sk, 16 * f, kk, lane), qf[kk], s[f], 0, 0, 0);
                dp[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_rows<DH>(sv, 16 * f, kk, lane), dof[kk], dp[f], 0, 0, 0);
            }
        }
        const uint64_t dbase = ((uint64_t)(b * p.H + h) * p.Lq + qrow) * (uint64_t)((p.Lk + 1) & ~1);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const uint32_t mk = *reinterpret_cast<const uint32_t*>(smask + 16 * f + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * TROWS + 16 * f + 4 * g + r;
                float val = s[f][r] * p.scale;
                const bool keep = ((mk >> (8 * r)) & 0xff) != 0 && !(p.causal && key > qrow);
                val = keep ? val : MASKED_SCORE;
                const float pr = (key < p.Lk && qok) ? __expf(val - m) * inv_l : 0.f;
                float dpv = dp[f][r];
                if (p.dropout_p > 0.f) dpv = dropout_keep(drop_key(eff_seed(p.seed, p.seed_dev)), dbase + key, p.thresh) ? dpv * p.drop_scale : 0.f;
                s[f][r] = pr * (dpv - delta);      // dS^T
            }
        }
        bf16x8_t db[2] = {pack_b_operand(s[0], s[1]), pack_b_operand(s[2], s[3])};
#pragma unroll
        for (int df = 0; df < NDF; ++df)
#pragma unroll
            for (int st = 0; st < 2; ++st)
                dq[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_tr<DH>(sk, 16 * df, st, lane), db[st], dq[df], 0, 0, 0);
    }
    if (qok) store_t_acc<DH>(p.dq + (int64_t)(b * p.Lq + qrow) * p.lddq + h * DH, dq, p.scale, lane);
}

// =============================================================================== dK, dV  (owner: 16 keys per wave)
template <int DH, bool RES>
__global__ __launch_bounds__(256, RES ? 2 : ATTN_MIN_BLOCKS) void attn_bwd_dkv_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NSLOT = RES ? p.nslot_q : 1;
    float* stat_all = reinterpret_cast<float*>(smem + NSLOT * 2 * TILE_BYTES);   // [3][NSLOT*64]: m, 1/l, delta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64 + wave * 16;
    const int key = k0 + c;
    const bool kok = key < p.Lk;
    const bf16_t* kp = p.k + (int64_t)(b * p.Lk + key) * p.ldk + h * DH;
    const bf16_t* vp = p.v + (int64_t)(b * p.Lk + key) * p.ldv + h * DH;
    bf16x8_t kf[NKK], vf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) { kf[kk] = ld_frag_global(kp + kk * 32 + g * 8, p.k, kok); vf[kk] = ld_frag_global(vp + kk * 32 + g * 8, p.v, kok); }
    const bool key_keep = kok && (!p.key_mask || p.key_mask[(int64_t)b * p.Lk + key] != 0);
    const bf16_t* qbase = p.q + (int64_t)b * p.Lq * p.ldq + h * DH;
    const bf16_t* dobase = p.d_o + (int64_t)b * p.Lq * p.lddo + h * DH;
    float4_t dk[NDF], dv[NDF];
#pragma unroll
    for (int f = 0; f < NDF; ++f) { dk[f] = (float4_t){0.f, 0.f, 0.f, 0.f}; dv[f] = (float4_t){0.f, 0.f, 0.f, 0.f}; }
    const int ntiles = (p.Lq + TROWS - 1) / TROWS;
    const int t_begin = p.causal ? (int)blockIdx.x : 0;   // queries < first key of the block see none of its keys
    Stage<DH> rq, rdo;
    auto load_q = [&](int t) {
        rq = tile_load<DH>(qbase, p.ldq, t * TROWS, p.Lq, tid);
        rdo = tile_load<DH>(dobase, p.lddo, t * TROWS, p.Lq, tid);
    };
    auto load_stats = [&](int qr, int slot) {
        float mm = 0.f, il = 0.f, dl = 0.f;
        if (qr < p.Lq) {
            const int64_t si = (int64_t)(b * p.H + h) * p.Lq + qr;
            mm = p.stats[si * 2]; il = 1.0f / p.stats[si * 2 + 1]; dl = p.delta[si];
        }
        stat_all[slot] = mm; stat_all[NSLOT * 64 + slot] = il; stat_all[2 * NSLOT * 64 + slot] = dl;
    };
    if (RES) {
        for (int t = t_begin; t < ntiles; ++t) {
            load_q(t);
            tile_store<DH>(rq, smem + t * 2 * TILE_BYTES, tid);
            tile_store<DH>(rdo, smem + t * 2 * TILE_BYTES + TILE_BYTES, tid);
        }
        if (tid < NSLOT * 64) load_stats(tid, tid);
        __syncthreads();
    } else if (t_begin < ntiles) {
        load_q(t_begin);
    }
    for (int qt = t_begin; qt < ntiles; ++qt) {
        char* sq = RES ? smem + qt * 2 * TILE_BYTES : smem;
        char* sdo = sq + TILE_BYTES;
        float* sm = stat_all + (RES ? qt * TROWS : 0);
        float* sil = sm + NSLOT * 64;
        float* sdl = sm + 2 * NSLOT * 64;
        if (!RES) {
            __syncthreads();
            tile_store<DH>(rq, sq, tid);
            tile_store<DH>(rdo, sdo, tid);
            if (tid < 64) load_stats(qt * TROWS + tid, tid);
            __syncthreads();
            if (qt + 1 < ntiles) load_q(qt + 1);
        }
        float4_t s[4], dp[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
            dp[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_rows<DH>(sq, 16 * f, kk, lane), kf[kk], s[f], 0, 0, 0);
                dp[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_rows<DH>(sdo, 16 * f, kk, lane), vf[kk], dp[f], 0, 0, 0);
            }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float4 m4 = *reinterpret_cast<const float4*>(sm + 16 * f + 4 * g);
            const float4 i4 = *reinterpret_cast<const float4*>(sil + 16 * f + 4 * g);
            const float4 d4 = *reinterpret_cast<const float4*>(sdl + 16 * f + 4 * g);
            const float mr[4] = {m4.x, m4.y, m4.z, m4.w}, ir[4] = {i4.x, i4.y, i4.z, i4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = qt * TROWS + 16 * f + 4 * g + r;
                float val = s[f][r] * p.scale;
                const bool keep = key_keep && !(p.causal && key > qr);
                val = keep ? val : MASKED_SCORE;
                const float pr = (qr < p.Lq && kok) ? __expf(val - mr[r]) * ir[r] : 0.f;
                float dpv = dp[f][r], pd = pr;
                if (p.dropout_p > 0.f) {
                    const bool kp_ = dropout_keep(drop_key(eff_seed(p.seed, p.seed_dev)), ((uint64_t)(b * p.H + h) * p.Lq + qr) * (uint64_t)((p.Lk + 1) & ~1) + key, p.thresh);
                    dpv = kp_ ? dpv * p.drop_scale : 0.f;
                    pd = kp_ ? pr * p.drop_scale : 0.f;
                }
                dp[f][r] = pd;                     // dropped P   -> dV
                s[f][r] = pr * (dpv - dr[r]);      // dS          -> dK
            }
        }
        bf16x8_t pb[2] = {pack_b_operand(dp[0], dp[1]), pack_b_operand(dp[2], dp[3])};
        bf16x8_t sb[2] = {pack_b_operand(s[0], s[1]), pack_b_operand(s[2], s[3])};
#pragma unroll
        for (int df = 0; df < NDF; ++df)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                dv[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_tr<DH>(sdo, 16 * df, st, lane), pb[st], dv[df], 0, 0, 0);
                dk[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag_tr<DH>(sq, 16 * df, st, lane), sb[st], dk[df], 0, 0, 0);
            }
    }
    if (kok) {
        store_t_acc<DH>(p.dk + (int64_t)(b * p.Lk + key) * p.lddk + h * DH, dk, p.scale, lane);
        store_t_acc<DH>(p.dv + (int64_t)(b * p.Lk + key) * p.lddv + h * DH, dv, 1.0f, lane);
    }
}

static size_t attn_lds(int dh, int slots, int extra) { return (size_t)slots * 2 * 64 * (dh * 2 + 16) + extra; }
template <typename K>
static void launch_attn(K kernel, dim3 grid, size_t lds, hipStream_t s, const AttnArgs& a) {
    if (lds > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, a);
}

static int check_common(const char* fn, int B, int H, int Lq, int Lk, int dh, int64_t ld0, int64_t ld1, int64_t ld2, int64_t ld3) {
    VM_REQUIRE(dh == 32 || dh == 64 || dh == 96 || dh == 128, "%s: head dim %d unsupported (32, 64, 96 or 128)", fn, dh);
    VM_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, "%s: empty problem", fn);
    VM_REQUIRE((ld0 % 8) == 0 && (ld1 % 8) == 0 && (ld2 % 8) == 0 && (ld3 % 8) == 0, "%s: leading dims must be multiples of 8", fn);
    return VM_OK;
}

extern "C" int vm_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                void* o, int64_t ldo, float* stats, const uint8_t* key_mask,
                                int B, int H, int Lq, int Lk, int dh, float scale, int causal,
                                float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_dev,
                                const int32_t* kv_row_index, int64_t kv_index_ld, void* stream) {
    VM_REQUIRE(q && k && v && o && stats, "vm_attention_fwd: null pointer");
    int rc = check_common("vm_attention_fwd", B, H, Lq, Lk, dh, ldq, ldk, ldv, ldo);
    if (rc) return rc;
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.stats = stats; a.key_mask = key_mask;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.causal = causal;
    a.kv_index = kv_row_index; a.kv_index_ld = kv_index_ld;
    a.dropout_p = dropout_p; a.seed = dropout_seed; a.seed_dev = dropout_seed_dev; a.thresh = dropout_thresh16(dropout_p);
    a.drop_scale = dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f;
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ATTN, 4.0 * B * H * (double)Lq * Lk * dh, s, "fwd_B%d_H%d_Lq%d_Lk%d_c%d_d%d", B, H, Lq, Lk, causal, dropout_p > 0.f);
    // dh = 64 with a short resident sequence (every ViT-B / BERT-base layer): one workgroup per (b, h), K/V loaded once
    if (dh == 64 && Lk <= 256 && !kv_row_index && !vm_env().attn_tile && !vm_env().attn_stream) return vm_attn_head_fwd(a, s);
    const dim3 grid((Lq + 63) / 64, H, B);
    const bool res = Lk <= 256 && !vm_env().attn_stream;
    a.nslot_k = (Lk + 63) / 64; a.nslot_q = (Lq + 63) / 64;
#define LAUNCH_FWD(D) do { if (res) launch_attn(attn_fwd_kernel<D, true>, grid, attn_lds(D, a.nslot_k, 256), s, a); \
                           else launch_attn(attn_fwd_kernel<D, false>, grid, attn_lds(D, 1, 64), s, a); } while (0)
    switch (dh) { case 32: LAUNCH_FWD(32); break; case 64: LAUNCH_FWD(64); break; case 96: LAUNCH_FWD(96); break; default: LAUNCH_FWD(128); break; }
#undef LAUNCH_FWD
    return vm_check_launch("vm_attention_fwd");
}

extern "C" int vm_attention_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* stats,
                                const uint8_t* key_mask,
                                void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                                int B, int H, int Lq, int Lk, int dh, float scale, int causal,
                                float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_dev, float* ws_delta, void* stream) {
    VM_REQUIRE(q && k && v && o && d_o && stats && dq && dk && dv && ws_delta, "vm_attention_bwd: null pointer");
    int rc = check_common("vm_attention_bwd", B, H, Lq, Lk, dh, ldq, ldk, ldv, ldo);
    if (rc) return rc;
    VM_REQUIRE((lddo % 8) == 0 && (lddq % 8) == 0 && (lddk % 8) == 0 && (lddv % 8) == 0, "vm_attention_bwd: leading dims must be multiples of 8");
    AttnArgs a = {};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.d_o = (const bf16_t*)d_o;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.stats = const_cast<float*>(stats); a.delta = ws_delta; a.key_mask = key_mask;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.causal = causal;
    a.dropout_p = dropout_p; a.seed = dropout_seed; a.seed_dev = dropout_seed_dev; a.thresh = dropout_thresh16(dropout_p);
    a.drop_scale = dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f;
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ATTN, 14.0 * B * H * (double)Lq * Lk * dh, s, "bwd_B%d_H%d_Lq%d_Lk%d_c%d_d%d", B, H, Lq, Lk, causal, dropout_p > 0.f);
    // head-resident kernels: delta = rowsum(dO * O) is computed by the dQ kernel (which holds the rows anyway) and saved for dK/dV
    if (dh == 64 && Lk <= 256 && Lq <= 256 && !vm_env().attn_tile && !vm_env().attn_stream) return vm_attn_head_bwd(a, s);
    const int64_t rows = (int64_t)B * H * Lq;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, a, dh);
    const dim3 gq((Lq + 63) / 64, H, B), gk((Lk + 63) / 64, H, B);
    const bool resk = Lk <= 256 && !vm_env().attn_stream, resq = Lq <= 256 && !vm_env().attn_stream;
    a.nslot_k = (Lk + 63) / 64; a.nslot_q = (Lq + 63) / 64;
#define LAUNCH_BWD(D) do { \
        if (resk) launch_attn(attn_bwd_dq_kernel<D, true>, gq, attn_lds(D, a.nslot_k, 256), s, a); \
        else launch_attn(attn_bwd_dq_kernel<D, false>, gq, attn_lds(D, 1, 64), s, a); \
        if (resq) launch_attn(attn_bwd_dkv_kernel<D, true>, gk, attn_lds(D, a.nslot_q, 3 * 256 * 4), s, a); \
        else launch_attn(attn_bwd_dkv_kernel<D, false>, gk, attn_lds(D, 1, 3 * 64 * 4), s, a); } while (0)
    switch (dh) { case 32: LAUNCH_BWD(32); break; case 64: LAUNCH_BWD(64); break; case 96: LAUNCH_BWD(96); break; default: LAUNCH_BWD(128); break; }
#undef LAUNCH_BWD
    return vm_check_launch("vm_attention_bwd");
}
