// attention_common.h -- argument block and small device helpers shared by attention.hip (tile-streaming kernels, any
// supported head dim / length) and attention_head.hip (head-resident kernels, dh = 64, L <= 256).
#pragma once
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));

#define MASKED_SCORE (-1e30f)

struct AttnArgs {
    const bf16_t *q, *k, *v, *o, *d_o;
    bf16_t *out, *dq, *dk, *dv;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    float* stats;            // [B,H,Lq,2]
    float* delta;            // [B,H,Lq]
    const uint8_t* key_mask; // [B,Lk] or null
    const int32_t* kv_index; // fwd only: [B, kv_index_ld] absolute K/V row of key j of batch b (KV-cache indirection), or null
    int64_t kv_index_ld;
    int B, H, Lq, Lk;
    int nslot_k, nslot_q;    // resident variants: LDS tile slots actually allocated (ceil(L/64))
    int ralloc_k, ralloc_q;  // head-resident variants: LDS rows allocated for K/V resp. Q/dO (L rounded up to 4)
    float scale; int causal;
    float dropout_p; uint64_t seed; const uint64_t* seed_dev; uint32_t thresh; float drop_scale;
};

// 16 B of a row, or zeros for a lane whose row does not exist.  A PREDICATED load (exec-masked: dead lanes keep the zeros), not "load from a safe
// address, then select": the select consumed the loaded value at once, so hipcc waited for every such load right behind its issue -- the
// next-group prefetches of the head kernels never overlapped anything (ISA of round 3: "LD LD W(1) W(0)" at the head of the group loop).
__device__ __forceinline__ bf16x8_t ld_frag_global(const bf16_t* p, const bf16_t* safe, bool ok) {
    (void)safe;
    uint4_t v = {0u, 0u, 0u, 0u};
    if (ok) v = *reinterpret_cast<const uint4_t*>(p);
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t pack_b_operand(const float4_t& a, const float4_t& b) {
    uint4 u;
    u.x = pack_bf16x2(a[0], a[1]); u.y = pack_bf16x2(a[2], a[3]);
    u.z = pack_bf16x2(b[0], b[1]); u.w = pack_bf16x2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8_t, u);
}
// reductions over the 4 lanes (c, g = 0..3) that share an MFMA output column: lane ^ 16 and lane ^ 32.  v_permlane16_swap /
// v_permlane32_swap of a register WITH ITSELF leave {the even rows | lower half} replicated in one result and {the odd rows | upper
// half} in the other, so one swap + one VALU op is an xor-16 (xor-32) butterfly step -- no ds_bpermute round trip through the LDS
// pipeline on the serial  S -> max -> exp -> P V  chain of the attention kernels.
__device__ __forceinline__ float col_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float col_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// predicated 16-B load: out-of-range lanes re-read a valid address (``safe``) and zero the result, so the
// compiler keeps a plain global_load (a select between the pointer and a stack zero becomes a flat load)
__device__ __forceinline__ uint4 ld16_or_zero(const bf16_t* p, const bf16_t* safe, bool ok) {
    uint4 v = *reinterpret_cast<const uint4*>(ok ? p : safe);
    if (!ok) v = make_uint4(0, 0, 0, 0);
    return v;
}

int vm_attn_head_fwd(const AttnArgs& a, hipStream_t s);
int vm_attn_head_bwd(const AttnArgs& a, hipStream_t s);
