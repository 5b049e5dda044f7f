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
// fp32 -> bf16, round-to-nearest-even (matches torch .to(bfloat16)): the gfx950 hardware convert v_cvt_pk_bf16_f32
typedef __bf16 vm_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float vm_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const vm_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vm_bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
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

// wave64 all-lanes reductions without the LDS pipeline (``__shfl_xor`` compiles to ds_bpermute_b32: six dependent LDS round trips per
// reduction): four DPP steps inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- after the quad steps
// a mirror read fetches the other quad's / half-row's total), then v_permlane16_swap / v_permlane32_swap of the register with itself
// across the rows (see col_max in attention_common.h).  Eight VALU instructions, every lane ends with the total.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
#define VM_WAVE_REDUCE(OP)                                                                                         \
    v = OP(v, dpp_f32<0xB1>(v));  /* quad_perm [1,0,3,2] */                                                        \
    v = OP(v, dpp_f32<0x4E>(v));  /* quad_perm [2,3,0,1] */                                                        \
    v = OP(v, dpp_f32<0x141>(v)); /* row_half_mirror */                                                            \
    v = OP(v, dpp_f32<0x140>(v)); /* row_mirror */                                                                 \
    { auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);             \
      v = OP(__uint_as_float(a[0]), __uint_as_float(a[1])); }                                                      \
    { auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);             \
      v = OP(__uint_as_float(b[0]), __uint_as_float(b[1])); }
__device__ __forceinline__ float vm_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum(float v) { VM_WAVE_REDUCE(vm_addf) return v; }
__device__ __forceinline__ float wave_max(float v) { VM_WAVE_REDUCE(fmaxf) return v; }

// erf-GELU and its derivative (hf:activations.py "gelu" -> nn.functional.gelu, the exact erf form).
// erf is evaluated with Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of every consumer):
// one v_rcp + one v_exp + 5 FMAs instead of libm's erff; the same exponential exp(-z^2/2) also yields the normal pdf
// that the derivative needs.
struct GeluParts { float cdf, pdf; };
__device__ __forceinline__ GeluParts gelu_parts(float z) {
    const float x = fabsf(z) * 0.70710678118654752f;
    // v_rcp_f32 (1 ulp), not __frcp_rn: the correctly rounded reciprocal compiles to the full division sequence (2 v_div_scale, v_rcp, 4 FMAs,
    // v_div_fmas, v_div_fixup + hazard s_nops: 10 of the 26 instructions per element of the GELU epilogues, which run with the MFMA pipe idle --
    // 30 us of a 100 us MLP-up launch); 1 ulp in t moves erf by ~1e-7, far below the bf16 rounding of every consumer
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
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

// counter-based RNG for dropout.  Stateless, so the backward pass regenerates the mask instead of storing it.
// One 32-bit hash decides TWO consecutive elements (16 bits each: keep iff u16 >= p * 2^16), which halves the integer
// multiplies -- v_mul_lo_u32 is a quarter-rate instruction and the mask is generated inside MFMA epilogues and the
// attention inner loop.  The 64-bit seed is expanded once per kernel (wave-uniform, scalar ALU) into two 32-bit keys;
// the second key enters between the two multiply rounds so that masks of different seeds are not shifted copies.
struct DropKey { uint32_t s0, s1; };
__device__ __forceinline__ DropKey drop_key(uint64_t seed) {          // splitmix64 finaliser
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    DropKey k; k.s0 = (uint32_t)z; k.s1 = (uint32_t)(z >> 32);
    return k;
}
__device__ __forceinline__ uint32_t drop_hash(const DropKey k, uint32_t pair) {
    uint32_t x = pair ^ k.s0;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x += k.s1; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// launch-time seed + optional device counter (HIP-graph replays draw fresh masks; see vm_gemm_epilogue.dropout_seed_dev)
__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const uint64_t* dev) { return dev ? seed + *dev : seed; }
// element idx of the logical index space (pairs are (2k, 2k+1))
__device__ __forceinline__ bool dropout_keep(const DropKey k, uint64_t idx, uint32_t thresh16) {
    const uint32_t h = drop_hash(k, (uint32_t)(idx >> 1));
    return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= thresh16;
}
// N consecutive elements starting at idx0: N/2 hashes when idx0 is even (the normal case: even row lengths)
template <int N>
__device__ __forceinline__ void dropout_keep_n(const DropKey k, uint64_t idx0, uint32_t thresh16, bool (&keep)[N]) {
    static_assert(N % 2 == 0, "pairs");
    if ((idx0 & 1) == 0) {
        const uint32_t p0 = (uint32_t)(idx0 >> 1);
#pragma unroll
        for (int j = 0; j < N / 2; ++j) {
            const uint32_t h = drop_hash(k, p0 + j);
            keep[2 * j] = (h & 0xffffu) >= thresh16;
            keep[2 * j + 1] = (h >> 16) >= thresh16;
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) keep[j] = dropout_keep(k, idx0 + j, thresh16);
    }
}
static inline uint32_t dropout_thresh16(float p) { return (uint32_t)((double)p * 65536.0 + 0.5); }

// ---- host-side plumbing
// diagnostic environment switches, read ONCE (getenv per launch costs host time; vm_reload_env() re-reads them)
struct VmEnv {
    int gemm_variant;      // VM_GEMM_VARIANT: force a tile variant (-1: cost model)
    int gemm_debug;        // VM_GEMM_DEBUG: 1 skip the epilogue, 2 one K-tile only (timing breakdowns)
    int gemm_groupw;       // VM_GEMM_GROUPW: column-group width of the tile order (0: heuristic)
    int gemm_p8_mf;        // VM_GEMM_P8_MF: A fragments per wave of the wide-tile kernel (4..8; 0: cost model)
    int wgrad_p8_min;      // VM_WGRAD_P8_MIN: fewest 256 x 256 tiles in a launch for the wide-tile kernel (default 64)
    int wgrad_p8;          // VM_WGRAD_P8: grouped weight gradients on 256 x 256 tiles: 0 off, 2 / 4 barrier pairs per K-tile (default 2)
    bool gemm_generic;     // VM_GEMM_GENERIC: register-staged fallback kernel only
    bool gemm_no_skinny;   // VM_GEMM_NO_SKINNY: never take the M <= 256 decode-step kernel
    bool attn_tile;        // VM_ATTN_TILE: tile-streaming attention kernels instead of the head-resident ones
    bool attn_stream;      // VM_ATTN_STREAM: streaming (non-resident) tile kernels
};
const VmEnv& vm_env();
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
