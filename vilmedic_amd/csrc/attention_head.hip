// attention_head.hip -- head-resident attention kernels for gfx950: dh = 64 and the resident sequence <= 256 rows
// (every ViT-B/16, BERT-base self- and cross-attention of the RRG hot path).
//
// One workgroup = one (batch, head) pair (x a chunk of 256 owner rows).  The "other" sequence of the pair (K,V for
// forward / dQ; Q,dO for dK,dV) is loaded from HBM exactly ONCE per head -- by LDS-DMA, consumed tile by tile as it lands (see the
// staging helpers below) -- into un-padded 128-B LDS rows, and the 4 waves then loop over 16-row owner groups
// (13 groups for L = 197: waves get 4/3/3/3).  The math of a group is the transposed flash scheme of attention.hip
// (S^T = K Q^T, P^T feeds the next MFMA as the B operand, transposed A operands via ds_read_b64_tr_b16).
//
// LDS layout: row r, 16-B chunk c lives at r*128 + ((c ^ hswz(r)) << 4) with hswz(r) = ((r >> 1) & 3) << 1.
//   * ds_read_b128 row fragments: one LDS cycle serves 8 rows at chunk x and the 8 other rows of the 16-row fragment
//     at chunk x+1; row parity picks the 128-B half of the 256-B bank line, (r>>1)&3 the chunk PAIR -> 16 distinct slots;
//   * ds_read_b64_tr_b16 fragments: one cycle serves 8 consecutive rows x 32 B (a chunk pair each) -> 16 distinct slots.
// Per-lane fragment offsets are kernel constants (the swizzle of row 16 F + c does not depend on F), control flow is
// wave-uniform (fragment counts live in SGPRs), and fragments that need no key mask / causal / ragged-tail handling
// take a one-multiply-per-score path.  Rows past L repeat the last valid row.  With L = 197 a workgroup needs ~52 KiB, so THREE
// workgroups (12 waves) share a CU and the 768 (b,h) pairs of a B=64 x 12-head layer are all resident at once.
#include "attention_common.h"

#define HB 128                                     // bytes per LDS row (64 bf16)
#define HCHUNK 256                                 // owner rows per workgroup
__device__ __forceinline__ int hswz(int row) { return ((row >> 1) & 3) << 1; }

// ---- staging by LDS-DMA, consumed progressively (round 3)
// Round 2 staged the resident sequence through registers (16 uint4 per thread) and started the math behind ONE barrier: with 768 (b, h)
// pairs = exactly one round of three workgroups per CU, every workgroup of the chip sat in that load phase at the same time and nothing
// covered it (25-35 % of a forward launch).  Now each wave requests its share of the rows with global_load_lds_dwordx4 (1 KiB = 8 rows
// per wave-instruction, no VGPR round trip), TILE-MAJOR -- the K and V (resp. Q and dO) rows of 64-row tile 0 first, then tile 1, ... --
// and the first owner group of every wave starts on tile 0 as soon as ITS requests have landed (s_waitcnt vmcnt(4 x later full tiles) +
// s_barrier) while tiles 1.. are still in flight: the load phase overlaps the first group's math instead of preceding it.
// The swizzle is applied to the DMA SOURCE: lane l writes LDS bytes [16 l, 16 l + 16) of the block, i.e. row l >> 3, slot l & 7, which
// must hold chunk slot ^ hswz(row).  DMA cannot zero-fill: rows past the valid length re-read the last valid row (finite values that
// only ever meet P = dS = 0: keys >= Lk score -inf, queries >= Lq carry nm = -inf).
// The DMA is issued by inline asm (invisible to hipcc's waitcnt pass, see gemm_fast.hip glds16_asm), so every compiler-visible global
// load that precedes the first MFMA is pinned into registers BEFORE the burst (head_pin) -- otherwise the compiler's own
// "s_waitcnt vmcnt(0)" in front of their first use would wait for the whole burst.
__device__ __forceinline__ void hdma16(const bf16_t* src, char* lds_dst) {
    const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst;      // wave-uniform
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory", "m0");
}
// 8-row block jb of both matrices is requested by wave jb & 3: per wave and per full 64-row tile exactly 4 requests (a, b, a, b), in tile order
__device__ __forceinline__ void head_dma2(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nvalid, int nalloc,
                                          char* sa, char* sb, int wave, int lane) {
    const int rl = lane >> 3, ch = (lane & 7) ^ hswz(rl);          // hswz(8 jb + rl) = hswz(rl)
    for (int jb = wave; jb * 8 < nalloc; jb += 4) {
        const int row = min(jb * 8 + rl, nvalid - 1);
        hdma16(a + (int64_t)row * lda + ch * 8, sa + jb * 1024);
        hdma16(b + (int64_t)row * ldb + ch * 8, sb + jb * 1024);
    }
}
// the wave's requests for tile kt have landed (and, behind the barrier, every wave's).  ``precise``: nothing but the DMA burst has been
// issued to vector memory since the pin, so "all but the 4 x (later full tiles) youngest" is exact; afterwards (stores and prefetches may
// be in flight, and stores are not ordered with loads in the counter) the wait is for everything.
__device__ __forceinline__ void head_wait_tile(int nfull, int kt, bool precise) {
    const int later = (precise && kt < nfull) ? nfull - 1 - kt : 0;
    if (later >= 3) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// every wave arrives at exactly ``ntiles`` barriers per kernel, wherever it first touches a tile
struct HeadStage {
    int nfull, ntiles, waited; bool precise;
    __device__ __forceinline__ void need(int kt) { while (waited <= kt) { head_wait_tile(nfull, waited, precise); ++waited; } }
    __device__ __forceinline__ void finish() { need(ntiles - 1); }
};
__device__ __forceinline__ void head_pin(bf16x8_t& v) {
    uint4_t u = __builtin_bit_cast(uint4_t, v);
    asm volatile("" : "+v"(u));
    v = __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ void head_pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void head_pin(int& v) { asm volatile("" : "+v"(v)); }
// Per-lane LDS byte offsets, constant for the whole kernel (row r = 16 F + (lane&15): (r>>1)&3 does not depend on F):
//   row[kk]  ds_read_b128 A fragment of 16-row fragment F at k-step kk:     tile + F*2048 + row[kk]
//   tr[df]   ds_read_b64_tr_b16 of rows 16 F + 4g + e, columns 16 df + ..:   tile + F*2048 + tr[df]  (second half: F+1)
struct HLane { int row[2]; int tr[4]; };
__device__ __forceinline__ HLane hlane_offsets(int lane) {
    const int g = lane >> 4, c = lane & 15;
    HLane L;
    const int hs = ((c >> 1) & 3) << 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) L.row[kk] = c * HB + (((kk * 4 + g) ^ hs) << 4);
    const int ri = 4 * g + (c >> 2), hst = ((ri >> 1) & 3) << 1, yi = (c & 3) >> 1, sub = (c & 1) * 8;
#pragma unroll
    for (int df = 0; df < 4; ++df) L.tr[df] = ri * HB + (((2 * df + yi) ^ hst) << 4) + sub;
    return L;
}
__device__ __forceinline__ bf16x8_t lds_b128(const char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ v4s lds_tr(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(const __attribute__((address_space(3))) v4s*)p);
}
__device__ __forceinline__ bf16x8_t join_tr(v4s lo, v4s hi) {
    short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void hstore_t_acc(bf16_t* rowptr, const float4_t (&acc)[4], float mul, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        uint2 u;
        u.x = pack_bf16x2(acc[f][0] * mul, acc[f][1] * mul);
        u.y = pack_bf16x2(acc[f][2] * mul, acc[f][3] * mul);
        *reinterpret_cast<uint2*>(rowptr + 16 * f + 4 * g) = u;
    }
}
#define LOG2E_F 1.4426950408889634f
__device__ __forceinline__ float max4(const float4_t& v) { return fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])); }
// A operand = (64-row tile)^T for k-step st: rows 32 st + {4g+e, 16+4g+e} of the tile, output rows = columns 16 df + c
__device__ __forceinline__ bf16x8_t tr_pair(const char* tile, int st, int off) {
    return join_tr(lds_tr(tile + (2 * st) * 2048 + off), lds_tr(tile + (2 * st + 1) * 2048 + off));
}

// scaled + masked score of one 16-key fragment; `plain` (wave-uniform) = no key mask, not on the causal diagonal, no
// keys past Lk in this fragment: the common case costs one multiply per element.
__device__ __forceinline__ void mask_scores(float4_t& s, bool plain, const uint8_t* smask, int key0, int qrow, const AttnArgs& p) {
    if (plain) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] *= p.scale;
    } else {
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(smask + key0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + r;
            float val = s[r] * p.scale;
            const bool keep = ((mk >> (8 * r)) & 0xff) != 0 && !(p.causal && key > qrow);
            val = keep ? val : MASKED_SCORE;
            s[r] = key < p.Lk ? val : -INFINITY;
        }
    }
}

// =============================================================================== forward
// K/V rows are allocated up to Lk rounded to 16 (zero-filled past Lk), so every fragment a group attends to is in range.
// NG = owner groups (16 queries each) a wave works on at once; the code below is written over NG and instantiated with NG = 1.
// The kernel is latency-, not throughput-bound (rocprofv3 PMC, profiles/r02_f_attention_pmc.txt: waves wait 48 % of their cycles,
// VALU busy 23 % per wave, 23 VALU instructions per MFMA): the chain  S = K Q^T (MFMA) -> max / exp / sum (VALU, cross-lane) -> P V
// (MFMA)  is serial per group.  NG = 2 (two independent groups interleaved in one basic block, 166 VGPRs, no spills) was measured on
// the three production shapes and gave nothing (ViT 48.9 vs 46.4 us, cross 28.8 vs 28.5 us, same file): the 3 resident waves per
// SIMD already overlap as much as the LDS-read / cross-lane latency allows, so the instantiation and its switch were removed.
template <int NG>
__global__ __launch_bounds__(256, 3) void attn_head_fwd_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nalloc = p.ralloc_k;
    char* sk = smem;
    char* sv = smem + nalloc * HB;
    uint8_t* smask = reinterpret_cast<uint8_t*>(sv + nalloc * HB);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y, qc0 = blockIdx.x * HCHUNK;
    const int ngroups = (min(HCHUNK, p.Lq - qc0) + 15) / 16;
    const int kfr_all = (p.Lk + 15) / 16;
    const bool has_mask = p.key_mask != nullptr;
    const bool ragged = (p.Lk & 15) != 0;
    const HLane L = hlane_offsets(lane);
    const DropKey dkey = drop_key(eff_seed(p.seed, p.seed_dev));
    const uint32_t lk_even = (uint32_t)((p.Lk + 1) & ~1);
    const float c2 = p.scale * LOG2E_F;
    auto load_q = [&](int grp, bf16x8_t (&qf)[2]) {
        const int qrow = qc0 + grp * 16 + c;
        const bf16_t* qp = p.q + (int64_t)(b * p.Lq + qrow) * p.ldq + h * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[kk] = ld_frag_global(qp + kk * 32 + g * 8, p.q, qrow < p.Lq && grp < ngroups);
    };
    // (1) what this wave reads from global memory before its first MFMA -- the first group's queries, the key mask -- pinned in registers
    bf16x8_t qn[NG][2];
#pragma unroll
    for (int i = 0; i < NG; ++i) load_q(wave + 4 * i, qn[i]);
    smask[tid] = (p.key_mask && tid < p.Lk) ? p.key_mask[(int64_t)b * p.Lk + tid] : (uint8_t)1;
#pragma unroll
    for (int i = 0; i < NG; ++i) { head_pin(qn[i][0]); head_pin(qn[i][1]); }
    // (2) K / V -> LDS by DMA, tile-major; (3) tile 0 (and with the barrier the mask bytes of every thread)
    head_dma2(p.k + (int64_t)b * p.Lk * p.ldk + h * 64, p.ldk, p.v + (int64_t)b * p.Lk * p.ldv + h * 64, p.ldv, p.Lk, nalloc, sk, sv, wave, lane);
    HeadStage stage = {nalloc >> 6, (nalloc + 63) >> 6, 0, true};
    stage.need(0);
    // bit F of unmasked = every key of 16-key fragment F is attendable (key mask byte != 0; bytes past Lk are 1): decided at
    // run time, so a mask that is all ones (real images, unpadded reports) costs nothing
    uint32_t unmasked = 0xffffu;
    if (has_mask) {
        unmasked = 0;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const uint64_t ok = __ballot(smask[q4 * 64 + lane] != 0);
#pragma unroll
            for (int f = 0; f < 4; ++f) unmasked |= (((ok >> (16 * f)) & 0xffffull) == 0xffffull ? 1u : 0u) << (4 * q4 + f);
        }
    }
    for (int grp0 = wave; grp0 < ngroups; grp0 += 4 * NG) {
        bf16x8_t qf[NG][2];
        float4_t o[NG][4];
        float m[NG], l[NG];
        int qrow[NG], diag[NG], nfr[NG];
        uint32_t dbase[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int q0 = qc0 + (grp0 + 4 * i) * 16;
            qrow[i] = q0 + c;
            qf[i][0] = qn[i][0]; qf[i][1] = qn[i][1];
            if (!stage.precise) load_q(grp0 + 4 * NG + 4 * i, qn[i]);      // prefetch the next iteration's queries behind this iteration's math
#pragma unroll
            for (int f = 0; f < 4; ++f) o[i][f] = (float4_t){0.f, 0.f, 0.f, 0.f};
            m[i] = -INFINITY; l[i] = 0.f;
            diag[i] = q0 >> 4;                                                          // fragment holding the causal diagonal
            nfr[i] = grp0 + 4 * i >= ngroups ? 0 : (p.causal ? min(kfr_all, diag[i] + 1) : kfr_all);   // 16-key fragments the group attends to
            dbase[i] = (uint32_t)(((uint64_t)(b * p.H + h) * p.Lq + qrow[i]) * lk_even >> 1);         // pair index of key 0
        }
        int nfr_max = nfr[0];
#pragma unroll
        for (int i = 1; i < NG; ++i) nfr_max = max(nfr_max, nfr[i]);
        for (int kt = 0; kt * 4 < nfr_max; ++kt) {
            stage.need(kt);
            const char* skt = sk + kt * 8192;
            const char* svt = sv + kt * 8192;
            // "clean" tile (the common case): 4 full fragments, no key mask, not touching a causal diagonal or the ragged tail -- for
            // EVERY group of the iteration.  Scores stay unscaled; exp(s*scale - m) is one fma + one v_exp_f32 (base 2) per element.
            bool clean = ((unmasked >> (4 * kt)) & 0xfu) == 0xfu && !(ragged && 4 * kt + 3 >= kfr_all - 1);
#pragma unroll
            for (int i = 0; i < NG; ++i) clean = clean && nfr[i] - 4 * kt >= 4 && !(p.causal && 4 * kt + 3 >= diag[i]);
            if (clean) {
                float4_t s[NG][4];
#pragma unroll
                for (int i = 0; i < NG; ++i)
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        s[i][f] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            s[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(skt + f * 2048 + L.row[kk]), qf[i][kk], s[i][f], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const float mt = col_max(fmaxf(fmaxf(max4(s[i][0]), max4(s[i][1])), fmaxf(max4(s[i][2]), max4(s[i][3])))) * p.scale;
                    const float m_new = fmaxf(m[i], mt);
                    const float alpha = __expf(m[i] - m_new);
                    const float nm = -m_new * LOG2E_F;
                    float ls = 0.f;
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(s[i][f][r], c2, nm)); s[i][f][r] = e; ls += e; }
                    l[i] = l[i] * alpha + col_sum(ls);
                    m[i] = m_new;
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[i][f][r] *= alpha;
                    if (p.dropout_p > 0.f) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            const uint32_t pr0 = dbase[i] + (uint32_t)(8 * (4 * kt + f) + 2 * g);
                            const uint32_t h0 = drop_hash(dkey, pr0), h1 = drop_hash(dkey, pr0 + 1);
                            s[i][f][0] = (h0 & 0xffffu) >= p.thresh ? s[i][f][0] * p.drop_scale : 0.f;
                            s[i][f][1] = (h0 >> 16) >= p.thresh ? s[i][f][1] * p.drop_scale : 0.f;
                            s[i][f][2] = (h1 & 0xffffu) >= p.thresh ? s[i][f][2] * p.drop_scale : 0.f;
                            s[i][f][3] = (h1 >> 16) >= p.thresh ? s[i][f][3] * p.drop_scale : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int df = 0; df < 4; ++df) {
                        const bf16x8_t vt = tr_pair(svt, st, L.tr[df]);        // one transposed V fragment feeds every group
#pragma unroll
                        for (int i = 0; i < NG; ++i) {
                            const bf16x8_t pb = st == 0 ? pack_b_operand(s[i][0], s[i][1]) : pack_b_operand(s[i][2], s[i][3]);
                            o[i][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pb, o[i][df], 0, 0, 0);
                        }
                    }
                continue;
            }
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                if (4 * kt >= nfr[i]) continue;
                const int nf = min(4, nfr[i] - 4 * kt);
                float4_t s[4];
                float mt = -INFINITY;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if (f < nf) {
                        const int F = 4 * kt + f;
                        s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(skt + f * 2048 + L.row[kk]), qf[i][kk], s[f], 0, 0, 0);
                        const bool plain = ((unmasked >> F) & 1u) && !(p.causal && F == diag[i]) && !(ragged && F == kfr_all - 1);
                        mask_scores(s[f], plain, smask, 16 * F + 4 * g, qrow[i], p);
                        mt = fmaxf(fmaxf(mt, fmaxf(s[f][0], s[f][1])), fmaxf(s[f][2], s[f][3]));
                    } else {
                        s[f] = (float4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    }
                }
                mt = col_max(mt);
                const float m_new = fmaxf(m[i], mt);
                const float alpha = __expf(m[i] - m_new);
                float ls = 0.f;
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = __expf(s[f][r] - m_new); s[f][r] = e; ls += e; }
                l[i] = l[i] * alpha + col_sum(ls);
                m[i] = m_new;
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[i][f][r] *= alpha;
                if (p.dropout_p > 0.f) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        if (f < nf) {
                            const uint32_t pr0 = dbase[i] + (uint32_t)(8 * (4 * kt + f) + 2 * g);
                            const uint32_t h0 = drop_hash(dkey, pr0), h1 = drop_hash(dkey, pr0 + 1);
                            s[f][0] = (h0 & 0xffffu) >= p.thresh ? s[f][0] * p.drop_scale : 0.f;
                            s[f][1] = (h0 >> 16) >= p.thresh ? s[f][1] * p.drop_scale : 0.f;
                            s[f][2] = (h1 & 0xffffu) >= p.thresh ? s[f][2] * p.drop_scale : 0.f;
                            s[f][3] = (h1 >> 16) >= p.thresh ? s[f][3] * p.drop_scale : 0.f;
                        }
                    }
                }
                const bf16x8_t pb[2] = {pack_b_operand(s[0], s[1]), pack_b_operand(s[2], s[3])};
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    if (2 * st < nf) {
                        const bool hi_ok = 2 * st + 1 < nf;            // the odd fragment may lie past the allocation
#pragma unroll
                        for (int df = 0; df < 4; ++df) {
                            const v4s lo = lds_tr(svt + (2 * st) * 2048 + L.tr[df]);
                            v4s hi = (v4s){0, 0, 0, 0};
                            if (hi_ok) hi = lds_tr(svt + (2 * st + 1) * 2048 + L.tr[df]);
                            o[i][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_tr(lo, hi), pb[st], o[i][df], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (stage.precise) {             // the first group ran beside the DMA burst: its prefetch comes now, waits are "everything" from here on
            stage.precise = false;
#pragma unroll
            for (int i = 0; i < NG; ++i) load_q(grp0 + 4 * NG + 4 * i, qn[i]);
        }
        // [r4] the next group's queries (requested a whole group of math ago) are made complete HERE, before this group's stores: loads and
        // stores retire through one in-order counter, and the compiler's own wait for them sat at the top of the next iteration, behind the
        // stores -- vmcnt(0): every group began with the round trip of the previous group's stores (ISA: "LD LD W(1) W(0)" at the loop head)
#pragma unroll
        for (int i = 0; i < NG; ++i) { head_pin(qn[i][0]); head_pin(qn[i][1]); }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            if (grp0 + 4 * i < ngroups && qrow[i] < p.Lq) {
                hstore_t_acc(p.out + (int64_t)(b * p.Lq + qrow[i]) * p.ldo + h * 64, o[i], 1.0f / l[i], lane);
                if (g == 0) {
                    float* st = p.stats + ((int64_t)(b * p.H + h) * p.Lq + qrow[i]) * 2;
                    st[0] = m[i]; st[1] = l[i];
                }
            }
        }
    }
    stage.precise = false;
    stage.finish();
}

// =============================================================================== dQ  (owner: 16 queries per group)
__global__ __launch_bounds__(256, 3) void attn_head_dq_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nalloc = p.ralloc_k;
    char* sk = smem;
    char* sv = smem + nalloc * HB;
    uint8_t* smask = reinterpret_cast<uint8_t*>(sv + nalloc * HB);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y, qc0 = blockIdx.x * HCHUNK;
    const int ngroups = (min(HCHUNK, p.Lq - qc0) + 15) / 16;
    const int kfr_all = (p.Lk + 15) / 16;
    const bool has_mask = p.key_mask != nullptr;
    const bool ragged = (p.Lk & 15) != 0;
    const HLane L = hlane_offsets(lane);
    const DropKey dkey = drop_key(eff_seed(p.seed, p.seed_dev));
    const uint32_t lk_even = (uint32_t)((p.Lk + 1) & ~1);
    const float c2 = p.scale * LOG2E_F;
    struct Own { bf16x8_t qf[2], dof[2], of[2]; float m, inv_l; };      // of: the forward output rows (for delta = rowsum(dO * O)); inv_l holds l until the group starts
    auto load_own = [&](int grp, Own& w) {
        const int qrow = qc0 + grp * 16 + c;
        const bool ok = qrow < p.Lq && grp < ngroups;
        const bf16_t* qp = p.q + (int64_t)(b * p.Lq + qrow) * p.ldq + h * 64;
        const bf16_t* dop = p.d_o + (int64_t)(b * p.Lq + qrow) * p.lddo + h * 64;
        const bf16_t* op = p.o + (int64_t)(b * p.Lq + qrow) * p.ldo + h * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            w.qf[kk] = ld_frag_global(qp + kk * 32 + g * 8, p.q, ok);
            w.dof[kk] = ld_frag_global(dop + kk * 32 + g * 8, p.d_o, ok);
            w.of[kk] = ld_frag_global(op + kk * 32 + g * 8, p.o, ok);
        }
        // predicated loads, and NO arithmetic on the loaded values here: anything that consumes them makes hipcc wait for the loads at once
        float mm = 0.f, ll = 0.f;
        if (ok) {
            const float2 st = *reinterpret_cast<const float2*>(p.stats + ((int64_t)(b * p.H + h) * p.Lq + qrow) * 2);
            mm = st.x; ll = st.y;
        }
        w.m = mm; w.inv_l = ll;                  // l, inverted by the consumer
    };
    Own nx;
    load_own(wave, nx);
    smask[tid] = (p.key_mask && tid < p.Lk) ? p.key_mask[(int64_t)b * p.Lk + tid] : (uint8_t)1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) { head_pin(nx.qf[kk]); head_pin(nx.dof[kk]); head_pin(nx.of[kk]); }
    head_pin(nx.m); head_pin(nx.inv_l);
    head_dma2(p.k + (int64_t)b * p.Lk * p.ldk + h * 64, p.ldk, p.v + (int64_t)b * p.Lk * p.ldv + h * 64, p.ldv, p.Lk, nalloc, sk, sv, wave, lane);
    HeadStage stage = {nalloc >> 6, (nalloc + 63) >> 6, 0, true};
    stage.need(0);
    uint32_t unmasked = 0xffffu;              // see attn_head_fwd_kernel
    if (has_mask) {
        unmasked = 0;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const uint64_t ok = __ballot(smask[q4 * 64 + lane] != 0);
#pragma unroll
            for (int f = 0; f < 4; ++f) unmasked |= (((ok >> (16 * f)) & 0xffffull) == 0xffffull ? 1u : 0u) << (4 * q4 + f);
        }
    }
    for (int grp = wave; grp < ngroups; grp += 4) {
        const int q0 = qc0 + grp * 16, qrow = q0 + c;
        const bool qok = qrow < p.Lq;
        Own w = nx;
        w.inv_l = w.inv_l > 0.f ? 1.0f / w.inv_l : 0.f;      // (load_own leaves l there; 0 for dead queries)
        if (!stage.precise) load_own(grp + 4, nx);
        const float nm = fmaf(-w.m, LOG2E_F, __builtin_amdgcn_logf(w.inv_l));   // log2 of exp(-m)/l; -inf for dead queries
        // delta = rowsum(dO * O) of the lane's query: 16 of the 64 columns per lane, summed over the 4 lane groups; saved for dK/dV
        float delta = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float a8[8], b8[8];
            unpack8(__builtin_bit_cast(uint4, w.dof[kk]), a8);
            unpack8(__builtin_bit_cast(uint4, w.of[kk]), b8);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta = fmaf(a8[e], b8[e], delta);
        }
        delta = col_sum(delta);          // (stored after the key loop: no vector-memory traffic besides the DMA burst while the waits are precise)
        float4_t dq[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) dq[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
        const int diag = q0 >> 4;
        const int nfr = p.causal ? min(kfr_all, diag + 1) : kfr_all;
        const uint32_t dbase = (uint32_t)(((uint64_t)(b * p.H + h) * p.Lq + qrow) * lk_even >> 1);
        for (int kt = 0; kt * 4 < nfr; ++kt) {
            stage.need(kt);
            const int nf = min(4, nfr - 4 * kt);
            const char* skt = sk + kt * 8192;
            const char* svt = sv + kt * 8192;
            if (nf == 4 && ((unmasked >> (4 * kt)) & 0xfu) == 0xfu && !(p.causal && 4 * kt + 3 >= diag) && !(ragged && 4 * kt + 3 >= kfr_all - 1)) {   // clean tile
                float4_t s[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
                    float4_t dp = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(skt + f * 2048 + L.row[kk]), w.qf[kk], s[f], 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(svt + f * 2048 + L.row[kk]), w.dof[kk], dp, 0, 0, 0);
                    }
                    if (p.dropout_p > 0.f) {
                        const uint32_t pr0 = dbase + (uint32_t)(8 * (4 * kt + f) + 2 * g);
                        const uint32_t h0 = drop_hash(dkey, pr0), h1 = drop_hash(dkey, pr0 + 1);
                        dp[0] = (h0 & 0xffffu) >= p.thresh ? dp[0] * p.drop_scale : 0.f;
                        dp[1] = (h0 >> 16) >= p.thresh ? dp[1] * p.drop_scale : 0.f;
                        dp[2] = (h1 & 0xffffu) >= p.thresh ? dp[2] * p.drop_scale : 0.f;
                        dp[3] = (h1 >> 16) >= p.thresh ? dp[3] * p.drop_scale : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        s[f][r] = __builtin_amdgcn_exp2f(fmaf(s[f][r], c2, nm)) * (dp[r] - delta);      // P * (dP - delta)
                }
                const bf16x8_t db[2] = {pack_b_operand(s[0], s[1]), pack_b_operand(s[2], s[3])};
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int df = 0; df < 4; ++df)
                        dq[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(skt, st, L.tr[df]), db[st], dq[df], 0, 0, 0);
                continue;
            }
            float4_t s[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
                if (f < nf) {
                    const int F = 4 * kt + f;
                    float4_t dp = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(skt + f * 2048 + L.row[kk]), w.qf[kk], s[f], 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(svt + f * 2048 + L.row[kk]), w.dof[kk], dp, 0, 0, 0);
                    }
                    const bool plain = ((unmasked >> F) & 1u) && !(p.causal && F == diag) && !(ragged && F == kfr_all - 1);
                    mask_scores(s[f], plain, smask, 16 * F + 4 * g, qrow, p);
                    if (p.dropout_p > 0.f) {
                        const uint32_t pr0 = dbase + (uint32_t)(8 * F + 2 * g);
                        const uint32_t h0 = drop_hash(dkey, pr0), h1 = drop_hash(dkey, pr0 + 1);
                        dp[0] = (h0 & 0xffffu) >= p.thresh ? dp[0] * p.drop_scale : 0.f;
                        dp[1] = (h0 >> 16) >= p.thresh ? dp[1] * p.drop_scale : 0.f;
                        dp[2] = (h1 & 0xffffu) >= p.thresh ? dp[2] * p.drop_scale : 0.f;
                        dp[3] = (h1 >> 16) >= p.thresh ? dp[3] * p.drop_scale : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = __expf(s[f][r] - w.m) * w.inv_l;      // 0 for keys >= Lk (score -inf) and dead queries (inv_l 0)
                        s[f][r] = pr * (dp[r] - delta);                      // dS^T
                    }
                }
            }
            const bf16x8_t db[2] = {pack_b_operand(s[0], s[1]), pack_b_operand(s[2], s[3])};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (2 * st < nf) {
                    const bool hi_ok = 2 * st + 1 < nf;
#pragma unroll
                    for (int df = 0; df < 4; ++df) {
                        const v4s lo = lds_tr(skt + (2 * st) * 2048 + L.tr[df]);
                        v4s hi = (v4s){0, 0, 0, 0};
                        if (hi_ok) hi = lds_tr(skt + (2 * st + 1) * 2048 + L.tr[df]);
                        dq[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_tr(lo, hi), db[st], dq[df], 0, 0, 0);
                    }
                }
            }
        }
        if (stage.precise) { stage.precise = false; load_own(grp + 4, nx); }      // the first group ran beside the DMA burst
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { head_pin(nx.qf[kk]); head_pin(nx.dof[kk]); head_pin(nx.of[kk]); }      // complete before the stores (see the forward kernel)
        head_pin(nx.m); head_pin(nx.inv_l);
        if (qok && g == 0) p.delta[(int64_t)(b * p.H + h) * p.Lq + qrow] = delta;
        if (qok) hstore_t_acc(p.dq + (int64_t)(b * p.Lq + qrow) * p.lddq + h * 64, dq, p.scale, lane);
    }
    stage.precise = false;
    stage.finish();
}

// =============================================================================== dK, dV  (owner: 16 keys per group)
// Q/dO rows are allocated up to Lq rounded to 8 only (whole DMA blocks; with the 3 stats per query this is what still lets three
// workgroups share a CU at Lq = 197: 3 x 53.6 KB): the last, partial 16-row fragment uses clamped per-lane addresses, and the
// transposed reads of its missing rows are zeroed in registers (they meet P = dS = 0, and 0 x garbage must stay 0).
__global__ __launch_bounds__(256, 3) void attn_head_dkv_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nalloc = p.ralloc_q;
    char* sq = smem;
    char* sdo = smem + nalloc * HB;
    float* sm = reinterpret_cast<float*>(sdo + nalloc * HB);      // [3][nalloc]: m, nm = log2(exp(-m) / l), delta
    float* snm = sm + nalloc;
    float* sdl = snm + nalloc;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y, kc0 = blockIdx.x * HCHUNK;
    if (tid < nalloc) {
        float mm = 0.f, nm = -INFINITY, dl = 0.f;
        if (tid < p.Lq) {
            const int64_t si = (int64_t)(b * p.H + h) * p.Lq + tid;
            mm = p.stats[si * 2]; dl = p.delta[si];
            nm = fmaf(-mm, LOG2E_F, -__builtin_amdgcn_logf(p.stats[si * 2 + 1]));
        }
        sm[tid] = mm; snm[tid] = nm; sdl[tid] = dl;      // rows >= Lq: nm = -inf  ->  P = dS = 0
    }
    const int ngroups = (min(HCHUNK, p.Lk - kc0) + 15) / 16;
    const int qfr_all = (p.Lq + 15) / 16;
    const int F_part = (nalloc & 15) ? (nalloc >> 4) : -1;         // the partial fragment, if any
    const bool has_mask = p.key_mask != nullptr;
    const HLane L = hlane_offsets(lane);
    // absolute per-lane offsets for the partial fragment F_part, recomputed where used (rare path; keeps 6 VGPRs free)
    const bool part_dead = F_part >= 0 && 16 * F_part + 4 * g >= nalloc;       // this lane's 4 transposed rows lie past the allocation
    auto part_row = [&](int kk) {
        const int row = min(16 * F_part + c, nalloc - 1);
        return row * HB + (((kk * 4 + g) ^ hswz(row)) << 4);
    };
    auto part_tr = [&](int df) {
        const int rt = min(16 * F_part + 4 * g + (c >> 2), nalloc - 1);
        return rt * HB + (((2 * df + ((c & 3) >> 1)) ^ hswz(rt)) << 4) + (c & 1) * 8;
    };
    const DropKey dkey = drop_key(eff_seed(p.seed, p.seed_dev));
    const uint32_t lk_even = (uint32_t)((p.Lk + 1) & ~1);
    const float c2 = p.scale * LOG2E_F;
    struct Own { bf16x8_t kf[2], vf[2]; bool keep; };
    auto load_own = [&](int grp, Own& w) {
        const int key = kc0 + grp * 16 + c;
        const bool ok = key < p.Lk && grp < ngroups;
        const bf16_t* kp = p.k + (int64_t)(b * p.Lk + key) * p.ldk + h * 64;
        const bf16_t* vp = p.v + (int64_t)(b * p.Lk + key) * p.ldv + h * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { w.kf[kk] = ld_frag_global(kp + kk * 32 + g * 8, p.k, ok); w.vf[kk] = ld_frag_global(vp + kk * 32 + g * 8, p.v, ok); }
        uint8_t mk = 1;
        if (ok && p.key_mask) mk = p.key_mask[(int64_t)b * p.Lk + key];
        w.keep = mk != 0;
    };
    // one half (16 tile rows = fragment F) of a transposed A operand
    auto tr_half = [&](const char* tile, int F, int df) -> v4s {
        if (F >= qfr_all) return (v4s){0, 0, 0, 0};
        if (F == F_part) {
            v4s v = lds_tr(tile + part_tr(df));
            if (part_dead) v = (v4s){0, 0, 0, 0};
            return v;
        }
        return lds_tr(tile + F * 2048 + L.tr[df]);
    };
    // the first group's keys / values and the per-query statistics come from global memory before the DMA burst (pinned), then Q / dO
    // tile-major by DMA; the barrier of tile 0 also publishes the statistics
    Own w0;
    load_own(wave, w0);
    {
        int kp = w0.keep ? 1 : 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { head_pin(w0.kf[kk]); head_pin(w0.vf[kk]); }
        head_pin(kp);
        w0.keep = kp != 0;
    }
    head_dma2(p.q + (int64_t)b * p.Lq * p.ldq + h * 64, p.ldq, p.d_o + (int64_t)b * p.Lq * p.lddo + h * 64, p.lddo, p.Lq, nalloc, sq, sdo, wave, lane);
    HeadStage stage = {nalloc >> 6, (nalloc + 63) >> 6, 0, true};
    stage.need(0);
    for (int grp = wave; grp < ngroups; grp += 4) {
        const int k0 = kc0 + grp * 16, key = k0 + c;
        const bool kok = key < p.Lk;
        // (no prefetch behind the math here: the kernel is at its VGPR budget for 3 waves/SIMD.  The next group's keys / values are requested
        // at the END of a group, when only dK / dV are live, and completed before the group's stores -- requested behind the stores, their wait
        // also covered the round trip of those eight stores)
        const Own w = w0;
        float4_t dk[4], dv[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) { dk[f] = (float4_t){0.f, 0.f, 0.f, 0.f}; dv[f] = (float4_t){0.f, 0.f, 0.f, 0.f}; }
        const int fr_begin = p.causal ? (k0 >> 4) : 0;      // query fragments before the group's first key see none of its keys
        const bool group_full = k0 + 16 <= p.Lk;
        const bool group_unmasked = !has_mask || __all(w.keep);         // run-time: an all-ones mask costs nothing
        for (int qt = fr_begin >> 2; qt * 4 < qfr_all; ++qt) {
            stage.need(qt);
            const int f_lo = max(0, fr_begin - 4 * qt), nf = min(4, qfr_all - 4 * qt);
            if (f_lo == 0 && nf == 4 && group_full && group_unmasked && !(F_part >= 0 && 4 * qt + 3 >= F_part) && !(p.causal && fr_begin >= 4 * qt)) {
                // clean tile: 4 full query fragments, every key of the group live, no mask / diagonal / partial rows
                const char* sqt = sq + qt * 8192;
                const char* sdt = sdo + qt * 8192;
                float4_t s[4], dp[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
                    dp[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(sqt + f * 2048 + L.row[kk]), w.kf[kk], s[f], 0, 0, 0);
                        dp[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(sdt + f * 2048 + L.row[kk]), w.vf[kk], dp[f], 0, 0, 0);
                    }
                    const int si = 64 * qt + 16 * f + 4 * g;
                    const float4 n4 = *reinterpret_cast<const float4*>(snm + si);
                    const float4 d4 = *reinterpret_cast<const float4*>(sdl + si);
                    const float nr[4] = {n4.x, n4.y, n4.z, n4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = __builtin_amdgcn_exp2f(fmaf(s[f][r], c2, nr[r]));
                        float dpv = dp[f][r], pd = pr;
                        if (p.dropout_p > 0.f) {
                            const int qr = si + r;
                            const uint32_t pair = (uint32_t)((((uint64_t)(b * p.H + h) * p.Lq + qr) * lk_even + (uint32_t)key) >> 1);
                            const uint32_t hh = drop_hash(dkey, pair);
                            const bool kp_ = ((key & 1) ? (hh >> 16) : (hh & 0xffffu)) >= p.thresh;
                            dpv = kp_ ? dpv * p.drop_scale : 0.f;
                            pd = kp_ ? pr * p.drop_scale : 0.f;
                        }
                        dp[f][r] = pd;
                        s[f][r] = pr * (dpv - dr[r]);
                    }
                }
                const bf16x8_t pb[2] = {pack_b_operand(dp[0], dp[1]), pack_b_operand(dp[2], dp[3])};
                const bf16x8_t sb[2] = {pack_b_operand(s[0], s[1]), pack_b_operand(s[2], s[3])};
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int df = 0; df < 4; ++df) {
                        dv[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(sdt, st, L.tr[df]), pb[st], dv[df], 0, 0, 0);
                        dk[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(sqt, st, L.tr[df]), sb[st], dk[df], 0, 0, 0);
                    }
                continue;
            }
            float4_t s[4], dp[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                s[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
                dp[f] = (float4_t){0.f, 0.f, 0.f, 0.f};
                if (f >= f_lo && f < nf) {
                    const int F = 4 * qt + f;
                    const bool part = F == F_part;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const int off = part ? part_row(kk) : F * 2048 + L.row[kk];
                        s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(sq + off), w.kf[kk], s[f], 0, 0, 0);
                        dp[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(sdo + off), w.vf[kk], dp[f], 0, 0, 0);
                    }
                    const int si = min(16 * F + 4 * g, nalloc - 4);
                    const float4 m4 = *reinterpret_cast<const float4*>(sm + si);
                    const float4 n4 = *reinterpret_cast<const float4*>(snm + si);
                    const float4 d4 = *reinterpret_cast<const float4*>(sdl + si);
                    const float mr[4] = {m4.x, m4.y, m4.z, m4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
                    float nr[4] = {n4.x, n4.y, n4.z, n4.w};
                    if (part && part_dead) { nr[0] = -INFINITY; nr[1] = -INFINITY; nr[2] = -INFINITY; nr[3] = -INFINITY; }   // clamped stats index: dead rows
                    const bool diag = p.causal && F == fr_begin;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qr = 16 * F + 4 * g + r;
                        float val = s[f][r] * p.scale;
                        const bool keep = w.keep && !(diag && key > qr);
                        val = keep ? val : MASKED_SCORE;
                        // exp(val - m) / l = 2^((val - m) log2e + (nm + m log2e)); live rows have val <= m, dead rows nm = -inf
                        const float pr = kok ? __builtin_amdgcn_exp2f(fmaf(fminf(val - mr[r], 0.f), LOG2E_F, fmaf(mr[r], LOG2E_F, nr[r]))) : 0.f;
                        float dpv = dp[f][r], pd = pr;
                        if (p.dropout_p > 0.f) {
                            const uint32_t pair = (uint32_t)((((uint64_t)(b * p.H + h) * p.Lq + qr) * lk_even + (uint32_t)key) >> 1);
                            const uint32_t hh = drop_hash(dkey, pair);
                            const bool kp_ = ((key & 1) ? (hh >> 16) : (hh & 0xffffu)) >= p.thresh;
                            dpv = kp_ ? dpv * p.drop_scale : 0.f;
                            pd = kp_ ? pr * p.drop_scale : 0.f;
                        }
                        dp[f][r] = pd;                     // dropped P   -> dV
                        s[f][r] = pr * (dpv - dr[r]);      // dS          -> dK
                    }
                }
            }
            const bf16x8_t pb[2] = {pack_b_operand(dp[0], dp[1]), pack_b_operand(dp[2], dp[3])};
            const bf16x8_t sb[2] = {pack_b_operand(s[0], s[1]), pack_b_operand(s[2], s[3])};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (2 * st + 1 >= f_lo && 2 * st < nf) {
                    const int F0 = 4 * qt + 2 * st;
#pragma unroll
                    for (int df = 0; df < 4; ++df) {
                        dv[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_tr(tr_half(sdo, F0, df), tr_half(sdo, F0 + 1, df)), pb[st], dv[df], 0, 0, 0);
                        dk[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_tr(tr_half(sq, F0, df), tr_half(sq, F0 + 1, df)), sb[st], dk[df], 0, 0, 0);
                    }
                }
            }
        }
        stage.precise = false;
        load_own(grp + 4, w0);
        {
            int kp = w0.keep ? 1 : 0;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) { head_pin(w0.kf[kk]); head_pin(w0.vf[kk]); }
            head_pin(kp);
            w0.keep = kp != 0;
        }
        if (kok) {
            hstore_t_acc(p.dk + (int64_t)(b * p.Lk + key) * p.lddk + h * 64, dk, p.scale, lane);
            hstore_t_acc(p.dv + (int64_t)(b * p.Lk + key) * p.lddv + h * 64, dv, 1.0f, lane);
        }
    }
    stage.precise = false;
    stage.finish();
}

// =============================================================================== host
template <typename K>
static void launch_head(K kernel, dim3 grid, size_t lds, hipStream_t s, const AttnArgs& a) {
    // (the three kernels of this file have the same C++ type, so this template is ONE function and a single "done" flag would cover only
    // the first kernel launched: the kernels already prepared are kept by address.  Matters for L in (240, 256]: > 64 KiB of LDS)
    static const void* ready[8];
    static int nready = 0;
    const void* fn = reinterpret_cast<const void*>(kernel);
    bool seen = false;
    for (int i = 0; i < nready; ++i) seen = seen || ready[i] == fn;
    if (!seen) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * HB + 4096);
        if (nready < 8) ready[nready++] = fn;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, a);
}

int vm_attn_head_fwd(const AttnArgs& a0, hipStream_t s) {
    AttnArgs a = a0;
    a.ralloc_k = (a.Lk + 15) / 16 * 16;
    const dim3 grid((a.Lq + HCHUNK - 1) / HCHUNK, a.H, a.B);
    const size_t lds = (size_t)2 * a.ralloc_k * HB + 256;
    // two owner groups in flight per wave where every group of a pair sees the same key tiles (non-causal) and a wave has >= 2 groups
    launch_head(attn_head_fwd_kernel<1>, grid, lds, s, a);
    return vm_check_launch("vm_attention_fwd(head)");
}

int vm_attn_head_bwd(const AttnArgs& a0, hipStream_t s) {
    AttnArgs a = a0;
    a.ralloc_k = (a.Lk + 15) / 16 * 16;
    a.ralloc_q = (a.Lq + 7) / 8 * 8;          // whole 8-row DMA blocks
    launch_head(attn_head_dq_kernel, dim3((a.Lq + HCHUNK - 1) / HCHUNK, a.H, a.B), (size_t)2 * a.ralloc_k * HB + 256, s, a);
    launch_head(attn_head_dkv_kernel, dim3((a.Lk + HCHUNK - 1) / HCHUNK, a.H, a.B), (size_t)2 * a.ralloc_q * HB + 12 * a.ralloc_q, s, a);
    return vm_check_launch("vm_attention_bwd(head)");
}
