// optim.hip -- fused Adam / AdamW over a flat parameter arena: one HBM pass reads p,g,m,v (fp32) and writes
// p,m,v plus the bf16 shadow the GEMMs consume next step.  torch.optim.Adam(W) arithmetic
// (ref: vilmedic/executors/utils.py:65-94 instantiates torch.optim.<name> from YAML).
// GB = the gradient is read as bf16 -- the averaged wire buffer of the data-parallel all-reduce (parallel.ArenaDDP): the optimizer
// consumes it directly instead of a cast pass writing it back to the fp32 gradient arena first (same arithmetic: bf16 -> fp32 is exact).
#include "common.h"

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
    //  and dropped it: 1.45 ms against 1.21-1.36 ms for this loop; the kernel is bandwidth-, not latency-bound, at 8 blocks per CU)
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        if (i + 4 <= n) {
            float4 pp = *reinterpret_cast<float4*>(p + i), gg;
            if (GB) {
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(gv) + i);
                gg = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
            } else {
                gg = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(gv) + i);
            }
            float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            float P[4] = {pp.x, pp.y, pp.z, pp.w}, G[4] = {gg.x, gg.y, gg.z, gg.w}, M[4] = {mm.x, mm.y, mm.z, mm.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
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
                uint2 u; u.x = pack_bf16x2(P[0], P[1]); u.y = pack_bf16x2(P[2], P[3]);
                *reinterpret_cast<uint2*>(shadow + i) = u;
            }
        } else {
            for (int64_t k = i; k < n; ++k) {
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
    int64_t blocks = (n + 1023) / 1024; if (blocks > 2048) blocks = 2048;
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
