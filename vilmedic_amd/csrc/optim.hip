// optim.hip -- fused Adam / AdamW over a flat parameter arena: one HBM pass reads p,g,m,v (fp32) and writes
// p,m,v plus the bf16 shadow the GEMMs consume next step.  torch.optim.Adam(W) arithmetic
// (ref: vilmedic/executors/utils.py:65-94 instantiates torch.optim.<name> from YAML).
// GB = the gradient is read as bf16 -- the averaged wire buffer of the data-parallel all-reduce (parallel.ArenaDDP): the optimizer
// consumes it directly instead of a cast pass writing it back to the fp32 gradient arena first (same arithmetic: bf16 -> fp32 is exact).
#include "common.h"

#ifndef ADAM_UNROLL
#define ADAM_UNROLL 4        // 4-element groups per thread and iteration: 1 -> 1.34 - 1.38 ms per C2 step, 2 -> 1.30, 4 -> 1.14 - 1.16, 8 -> 1.28 (profiles/r05_j_ln_adam_ab.txt)
#endif
template <bool GB>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const void* __restrict__ gv, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ shadow, int64_t n,
                                                   float lr, float b1, float b2, float eps, float wd, int decoupled,
                                                   float bc1, float bc2, float gscale, const float* __restrict__ lr_dev,
                                                   const int64_t* __restrict__ step_dev, const float* __restrict__ gate_dev) {
    if (gate_dev) {                               // NaN / Inf loss: the whole update is skipped (wave-uniform: one scalar)
        const float gate = *gate_dev;
        if (!(fabsf(gate) <= 3.4028234663852886e38f)) return;
    }
    if (lr_dev) lr = *lr_dev;
    if (step_dev) { const float t = (float)(*step_dev); bc1 = 1.0f - powf(b1, t); bc2 = 1.0f - powf(b2, t); }
    const float step = lr / bc1;
    const float inv_sqrt_bc2 = rsqrtf(bc2);
    // (round 4 measured a software-pipelined form -- the next iteration's loads issued ahead of this iteration's stores, counted waits in the ISA --
    //  and dropped it: 1.45 ms against 1.21-1.36 ms for the plain loop)
    // ADAM_UNROLL 4-element groups per thread and iteration (1024 elements apart: each coalesced), all their loads requested before the arithmetic:
    // 4 x ADAM_UNROLL 16-byte loads in flight per lane instead of four (round 5; the round-4 software pipeline across ITERATIONS kept stores and the
    // next loads in one in-order counter and lost; this form has no load behind a store inside an iteration).
    constexpr int UN = ADAM_UNROLL;
    const int64_t span = (int64_t)gridDim.x * (1024 * UN);
    for (int64_t i0 = (int64_t)blockIdx.x * (1024 * UN) + threadIdx.x * 4; i0 < n; i0 += span) {
        int64_t idx[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) idx[u] = i0 + 1024 * u;
        if (idx[UN - 1] + 4 <= n) {
            float4 pp[UN], gg[UN], mm[UN], vv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t i = idx[u];
                pp[u] = *reinterpret_cast<float4*>(p + i);
                if (GB) {
                    const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(gv) + i);
                    gg[u] = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
                } else {
                    gg[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(gv) + i);
                }
                mm[u] = *reinterpret_cast<float4*>(m + i);
                vv[u] = *reinterpret_cast<float4*>(v + i);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t i = idx[u];
                float P[4] = {pp[u].x, pp[u].y, pp[u].z, pp[u].w}, G[4] = {gg[u].x, gg[u].y, gg[u].z, gg[u].w};
                float M[4] = {mm[u].x, mm[u].y, mm[u].z, mm[u].w}, V[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float gr = G[j] * gscale;
                    if (decoupled) P[j] *= 1.f - lr * wd; else gr += wd * P[j];
                    M[j] = b1 * M[j] + (1.f - b1) * gr;
                    V[j] = b2 * V[j] + (1.f - b2) * gr * gr;
                    P[j] -= step * M[j] / (sqrtf(V[j]) * inv_sqrt_bc2 + eps);
                }
                *reinterpret_cast<float4*>(p + i) = make_float4(P[0], P[1], P[2], P[3]);
                *reinterpret_cast<float4*>(m + i) = make_float4(M[0], M[1], M[2], M[3]);
                *reinterpret_cast<float4*>(v + i) = make_float4(V[0], V[1], V[2], V[3]);
                if (shadow) {
                    uint2 w; w.x = pack_bf16x2(P[0], P[1]); w.y = pack_bf16x2(P[2], P[3]);
                    *reinterpret_cast<uint2*>(shadow + i) = w;
                }
            }
        } else {                                  // the arena's tail (less than one block's span): element by element
            for (int u = 0; u < UN; ++u)
                for (int64_t k = idx[u]; k < n && k < idx[u] + 4; ++k) {
                    float gr = (GB ? bf16_to_f32(reinterpret_cast<const bf16_t*>(gv)[k]) : reinterpret_cast<const float*>(gv)[k]) * gscale, pv = p[k];
                    if (decoupled) pv *= 1.f - lr * wd; else gr += wd * pv;
                    const float mk = b1 * m[k] + (1.f - b1) * gr, vk = b2 * v[k] + (1.f - b2) * gr * gr;
                    pv -= step * mk / (sqrtf(vk) * inv_sqrt_bc2 + eps);
                    p[k] = pv; m[k] = mk; v[k] = vk;
                    if (shadow) shadow[k] = f32_to_bf16(pv);
                }
        }
    }
}

extern "C" int vm_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                            float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd,
                            float bias_corr1, float bias_corr2, float grad_scale, void* stream) {
    return vm_adam_step_dev(p, g, m, v, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, decoupled_wd, bias_corr1, bias_corr2, grad_scale,
                            nullptr, nullptr, nullptr, stream);
}

static int adam_launch(bool wire, float* p, const void* g, float* m, float* v, void* shadow_bf16, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd,
                       float bias_corr1, float bias_corr2, float grad_scale,
                       const float* lr_dev, const int64_t* step_dev, const float* gate_dev, void* stream) {
    VM_REQUIRE(p && g && m && v && n > 0, "vm_adam_step: bad arguments");
    VM_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % (wire ? 8 : 16)) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0, "vm_adam_step: buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_OPT, (wire ? 28.0 : 30.0) * n, s);
    int64_t blocks = (n + 1024 * ADAM_UNROLL - 1) / (1024 * ADAM_UNROLL); if (blocks > 2048) blocks = 2048;
    if (wire)
        hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, (bf16_t*)shadow_bf16, n, lr, beta1, beta2, eps,
                           weight_decay, decoupled_wd, bias_corr1, bias_corr2, grad_scale, lr_dev, step_dev, gate_dev);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, (bf16_t*)shadow_bf16, n, lr, beta1, beta2, eps,
                           weight_decay, decoupled_wd, bias_corr1, bias_corr2, grad_scale, lr_dev, step_dev, gate_dev);
    return vm_check_launch("vm_adam_step");
}

extern "C" int vm_adam_step_dev(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd,
                                float bias_corr1, float bias_corr2, float grad_scale,
                                const float* lr_dev, const int64_t* step_dev, const float* gate_dev, void* stream) {
    return adam_launch(false, p, g, m, v, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, decoupled_wd, bias_corr1, bias_corr2, grad_scale,
                       lr_dev, step_dev, gate_dev, stream);
}

extern "C" int vm_adam_step_wire(float* p, const void* g_bf16, float* m, float* v, void* shadow_bf16, int64_t n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd,
                                 float bias_corr1, float bias_corr2, float grad_scale,
                                 const float* lr_dev, const int64_t* step_dev, const float* gate_dev, void* stream) {
    return adam_launch(true, p, g_bf16, m, v, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, decoupled_wd, bias_corr1, bias_corr2, grad_scale,
                       lr_dev, step_dev, gate_dev, stream);
}
