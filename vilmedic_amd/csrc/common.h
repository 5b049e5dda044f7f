// common.h -- shared device helpers for libvmhip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vmhip.h"

typedef uint16_t bf16_t;  // raw bf16 storage

typedef __attribute__((ext_vector_type(8))) short short8_t;   // 8 bf16 = MFMA A/B fragment (16x16x32)
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;   // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) uint32_t uint4_t;

#define VM_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN preserved (matches torch .to(bfloat16))
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-GELU and its derivative (hf:activations.py "gelu" -> nn.functional.gelu, the exact erf form).
// erf is evaluated with Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of every consumer):
// one v_rcp + one v_exp + 5 FMAs instead of libm's erff; the same exponential exp(-z^2/2) also yields the normal pdf
// that the derivative needs.
struct GeluParts { float cdf, pdf; };
__device__ __forceinline__ GeluParts gelu_parts(float z) {
    const float x = fabsf(z) * 0.70710678118654752f;
    const float t = __frcp_rn(1.0f + 0.3275911f * x);
    const float e = __expf(-x * x);                                  // = exp(-z^2 / 2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;                           // erf(|z|/sqrt2)
    GeluParts r;
    r.cdf = 0.5f * (1.0f + copysignf(erf_abs, z));
    r.pdf = 0.3989422804014327f * e;
    return r;
}
__device__ __forceinline__ float gelu_f(float x) { return x * gelu_parts(x).cdf; }
__device__ __forceinline__ float gelu_grad_f(float x) { const GeluParts g = gelu_parts(x); return g.cdf + x * g.pdf; }

// counter-based RNG for dropout: keep iff u24(seed, idx) >= p * 2^24.  Stateless so that the
// backward pass regenerates the mask instead of storing it.
// (32-bit murmur3 finaliser over the folded 64-bit counter: ~10 integer ops per element, all 32-bit multiplies)
__device__ __forceinline__ uint32_t vm_hash_u32(uint64_t seed, uint64_t idx) {
    uint32_t h = (uint32_t)idx ^ (uint32_t)seed;
    h += ((uint32_t)(idx >> 32) + (uint32_t)(seed >> 32)) * 0x9E3779B1u;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    h += (uint32_t)(seed >> 32);
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    return h;
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh24) {
    return (vm_hash_u32(seed, idx) >> 8) >= thresh24;
}
static inline uint32_t dropout_thresh24(float p) { return (uint32_t)((double)p * 16777216.0); }

// ---- host-side plumbing
void vm_set_error(const char* fmt, ...);
int vm_check_launch(const char* what);

enum { VM_FAM_GEMM = 0, VM_FAM_ATTN = 1, VM_FAM_LN = 2, VM_FAM_LOSS = 3, VM_FAM_ELT = 4, VM_FAM_OPT = 5, VM_FAM_DECODE = 6, VM_FAM_N = 7 };
// RAII-ish profiler hook: records HIP events on `stream` around a launch when profiling is on.
struct VmProfScope {
    int fam; hipStream_t s; void* slot;
    VmProfScope(int family, double work, hipStream_t stream, const char* tag_fmt = nullptr, ...);
    ~VmProfScope();
};

#define VM_REQUIRE(cond, ...) do { if (!(cond)) { vm_set_error(__VA_ARGS__); return VM_EINVAL; } } while (0)
