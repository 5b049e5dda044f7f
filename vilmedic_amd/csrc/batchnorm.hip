// batchnorm.hip -- BatchNorm2d on channels-last (NHWC) activations with GROUPED batch statistics, forward and backward, with the
// residual add and the ReLU that follow it in ResNet / DenseNet blocks fused in.  HBM-bound: 3 passes over the activation forward
// (statistics; normalise), 5 backward (reductions; gradient) at 16 B (bf16) / 32 B (fp32) per lane.
//
// Why it exists (ref:vilmedic/models/selfsup/conVIRT.py:83-95, config/SELFSUP/convirt-mimic.yml:22): the reference runs its CNN tower in
// ``forward_batch_size`` micro-batches, so every BatchNorm normalises each micro-batch with ITS OWN statistics.  The tower runs once over
// the whole batch here and only the statistics are grouped (blocks/vision/micro_bn.py); as a composition of torch reductions and
// elementwise kernels on an fp32 copy of the activation that grouping cost ~130 ms of a 205 ms ConVIRT step at B = 256
// (profiles/r05_d_steady_kernel_stats_convirt.csv).  G = 1 is ordinary BatchNorm (DenseNet-169 of MVQA: 169 layers per step).
//
// Layout: x is [G * R, C] row-major (R = images per group * H * W rows of C channels), C % 8 == 0, C <= 2048.  A block of 256 threads
// owns a contiguous row range of ONE group; thread t handles the 8 channels c8 = t % (C / 8) of rows rsub = t / (C / 8) (+ k * RPI).
#include <cstdlib>
#include "common.h"

namespace {

template <typename T> struct Ld8;
template <> struct Ld8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float* f) { unpack8(*reinterpret_cast<const uint4*>(p), f); }
    static __device__ __forceinline__ void store(bf16_t* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack8(f); }
};
template <> struct Ld8<float> {
    static __device__ __forceinline__ void load(const float* p, float* f) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* f) {
        *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
};

// 8 consecutive channels of one row AS LOADED (16 B of bf16 / 32 B of fp32): the streaming loops request U rows before they unpack the first, so a
// bf16 thread keeps as many bytes in flight as an fp32 one without holding U x 8 floats (round 6: bf16 passes 3.7 -> see profiles/r06_g_bn_bench.txt)
template <typename T> struct Row8;
template <> struct Row8<bf16_t> {
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void unpack(float* f) const { unpack8(v, f); }
    __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<uint4*>(p) = v; }
};
template <> struct Row8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    __device__ __forceinline__ void unpack(float* f) const { f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w; }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = a; *reinterpret_cast<float4*>(p + 4) = b; }
};
template <typename T> constexpr int bn_rows_in_flight() { return sizeof(T) == 2 ? 4 : 2; }

struct BnGeom {
    int G, R, C, CH8, RPI, S, rows_per_split;
    int64_t ldx;      // row stride of x in elements (>= C: x may be the first C channels of a wider channels-last buffer)
    int64_t ldd;      // row stride of dx (backward)
    int ldm;          // group stride of mean / rstd / var (>= C: the statistics may live in a wider per-buffer array)
    int acc;          // backward: dx += instead of dx =
};

__device__ __forceinline__ void split_range(const BnGeom& g, int si, int& r0, int& r1) {
    r0 = si * g.rows_per_split;
    r1 = min(g.R, r0 + g.rows_per_split);
}

// ---- forward statistics: per (group, split) partial (count, mean, M2) per channel.  Sums are taken relative to the first value a thread
// sees (shifted sums), partials are merged with Chan's formula: no E[x^2] - mean^2 cancellation.
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ pmean, float* __restrict__ pm2,
                                                       float* __restrict__ pcnt, const BnGeom g, T* __restrict__ copy_dst, int64_t ldc,
                                                       int64_t* __restrict__ nbt) {
    const int gi = blockIdx.x / g.S, si = blockIdx.x % g.S;
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += g.G;      // num_batches_tracked: one update per group (bn_apply_kernel reads it afterwards)
    int r0, r1;
    split_range(g, si, r0, r1);
    const int t = threadIdx.x, c8 = t % g.CH8, rsub = t / g.CH8;
    __shared__ float red[256 * 17];
    float n = 0.f, k[8], s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { k[j] = 0.f; s[j] = 0.f; q[j] = 0.f; }
    if (rsub < g.RPI) {
        const T* base = x + ((int64_t)gi * g.R) * g.ldx + c8 * 8;
        T* cbase = copy_dst ? copy_dst + ((int64_t)gi * g.R) * ldc + c8 * 8 : nullptr;      // the rows are also copied (a dense block's feature buffer)
        int r = r0 + rsub;
        if (r < r1) {
            Ld8<T>::load(base + (int64_t)r * g.ldx, k);         // the shift: this thread's first row
            if (cbase) Ld8<T>::store(cbase + (int64_t)r * ldc, k);
            n = 1.f;
            r += g.RPI;
        }
        constexpr int U = 2 * bn_rows_in_flight<T>();           // eight (bf16) / four (fp32) rows in flight
        for (; r + (U - 1) * g.RPI < r1; r += U * g.RPI) {
            Row8<T> a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) a[u].load(base + (int64_t)(r + u * g.RPI) * g.ldx);
            if (cbase) {
#pragma unroll
                for (int u = 0; u < U; ++u) a[u].store(cbase + (int64_t)(r + u * g.RPI) * ldc);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float f[8];
                a[u].unpack(f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = f[j] - k[j]; s[j] += d; q[j] += d * d; }
            }
            n += (float)U;
        }
        for (; r < r1; r += g.RPI) {
            float a[8];
            Ld8<T>::load(base + (int64_t)r * g.ldx, a);
            if (cbase) Ld8<T>::store(cbase + (int64_t)r * ldc, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = a[j] - k[j]; s[j] += d; q[j] += d * d; }
            n += 1.f;
        }
    }
    // per-thread (n, mean, M2)
    float* mine = red + t * 17;
    mine[0] = n;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float inv = n > 0.f ? 1.f / n : 0.f;
        mine[1 + j] = k[j] + s[j] * inv;
        mine[9 + j] = q[j] - s[j] * s[j] * inv;
    }
    __syncthreads();
    if (rsub == 0) {
        float N = mine[0], m[8], M2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] = mine[1 + j]; M2[j] = mine[9 + j]; }
        for (int o = 1; o < g.RPI; ++o) {
            const float* p = red + (o * g.CH8 + c8) * 17;
            const float nb = p[0];
            if (nb <= 0.f) continue;
            const float tot = N + nb, w = nb / tot;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = p[1 + j] - m[j];
                m[j] += d * w;
                M2[j] += p[9 + j] + d * d * N * w;
            }
            N = tot;
        }
        const int64_t o = ((int64_t)gi * g.S + si) * g.C + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pmean[o + j] = m[j]; pm2[o + j] = M2[j]; }
        if (c8 == 0) pcnt[gi * g.S + si] = N;
    }
}

// merge the split partials -> mean, biased variance, rstd.  A block = (256 / LANES) channels x LANES split lanes of one group: lane l
// merges the splits l, l + LANES, ... (Chan), the lanes are merged pairwise through LDS (a single thread per (group, channel) walking up
// to 1024 dependent merges measured 30 us per layer on MVQA's DenseNet, G = 1: more than the streaming passes of its late layers)
__device__ __forceinline__ void chan_merge(float& N, float& m, float& M2, float nb, float mb, float M2b) {
    if (nb <= 0.f) return;
    const float tot = N + nb, w = nb / tot, d = mb - m;
    m += d * w;
    M2 += M2b + d * d * N * w;
    N = tot;
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ pmean, const float* __restrict__ pm2, const float* __restrict__ pcnt,
                                                          float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ var,
                                                          const BnGeom g, float eps, int lanes) {
    const int cpb = 256 / lanes, chunks = (g.C + cpb - 1) / cpb;
    const int gi = blockIdx.x / chunks, cl = threadIdx.x / lanes, l = threadIdx.x % lanes, c = (blockIdx.x % chunks) * cpb + cl;
    __shared__ float red[256 * 3];
    float N = 0.f, m = 0.f, M2 = 0.f;
    if (c < g.C)
        for (int si = l; si < g.S; si += lanes) {
            const int64_t o = ((int64_t)gi * g.S + si) * g.C + c;
            chan_merge(N, m, M2, pcnt[gi * g.S + si], pmean[o], pm2[o]);
        }
    for (int st = lanes >> 1; st > 0; st >>= 1) {
        red[threadIdx.x * 3] = N; red[threadIdx.x * 3 + 1] = m; red[threadIdx.x * 3 + 2] = M2;
        __syncthreads();
        if (l < st) { const float* p = red + (threadIdx.x + st) * 3; chan_merge(N, m, M2, p[0], p[1], p[2]); }
        __syncthreads();
    }
    if (l == 0 && c < g.C) {
        const float v = M2 / N;
        mean[gi * g.ldm + c] = m;
        var[gi * g.ldm + c] = v;
        rstd[gi * g.ldm + c] = rsqrtf(v + eps);
    }
}

// y = relu?((x - mean) * rstd * gamma + beta + residual?)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, const BnGeom g, int relu,
                                                       const float* __restrict__ var, float* __restrict__ rmean, float* __restrict__ rvar,
                                                       const int64_t* __restrict__ nbt, float momentum, float eps) {
    const int gi = blockIdx.x / g.S, si = blockIdx.x % g.S;
    int r0, r1;
    split_range(g, si, r0, r1);
    const int t = threadIdx.x, c8 = t % g.CH8, rsub = t / g.CH8;
    if (rmean) {        // running statistics: one update per group, in group order (nn.BatchNorm2d run micro-batch by micro-batch); one owner thread per channel
        const float unb = g.R > 1 ? (float)g.R / (float)(g.R - 1) : 1.f;      // running_var takes the unbiased estimate
        for (int c = blockIdx.x * 256 + t; c < g.C; c += (int)gridDim.x * 256) {
            float rm = rmean[c], rv = rvar[c];
            float n = nbt ? (float)(nbt[0] - g.G) : 0.f;                      // momentum < 0: cumulative average, factor 1 / num_batches_tracked
            for (int k = 0; k < g.G; ++k) {
                n += 1.f;
                const float f = momentum >= 0.f ? momentum : 1.f / n;
                rm = f * mean[k * g.ldm + c] + (1.f - f) * rm;
                rv = f * (var[k * g.ldm + c] * unb) + (1.f - f) * rv;
            }
            rmean[c] = rm; rvar[c] = rv;
        }
    }
    if (rsub >= g.RPI) return;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        const float rs = rstd ? rstd[gi * g.ldm + c] : rsqrtf(var[gi * g.ldm + c] + eps);      // (inference: mean / var are the running statistics)
        sc[j] = rs * ga;
        sh[j] = be - mean[gi * g.ldm + c] * sc[j];
    }
    const int64_t off = ((int64_t)gi * g.R) * g.C + c8 * 8;
    const T* xb = x + ((int64_t)gi * g.R) * g.ldx + c8 * 8;
    constexpr int U = bn_rows_in_flight<T>();
    for (int r = r0 + rsub; r < r1; r += U * g.RPI) {
        Row8<T> a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ru = r + u * g.RPI;
            if (ru < r1) {
                a[u].load(xb + (int64_t)ru * g.ldx);
                if (res) b[u].load(res + off + (int64_t)ru * g.C);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ru = r + u * g.RPI;
            if (ru >= r1) break;
            float f[8], rb[8];
            a[u].unpack(f);
            if (res) b[u].unpack(rb);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = f[j] * sc[j] + sh[j];
                if (res) v += rb[j];
                f[j] = relu ? fmaxf(v, 0.f) : v;
            }
            Ld8<T>::store(y + off + (int64_t)ru * g.C, f);
        }
    }
}

// ---- backward reductions: per (group, split) partial sums of dy' and dy' * xhat, dy' = dy * [output > 0] when the ReLU is fused
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ res,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ psum, float* __restrict__ psumx, const BnGeom g, int relu) {
    const int gi = blockIdx.x / g.S, si = blockIdx.x % g.S;
    int r0, r1;
    split_range(g, si, r0, r1);
    const int t = threadIdx.x, c8 = t % g.CH8, rsub = t / g.CH8;
    __shared__ float red[256 * 16];
    float s[8], sx[8], mu[8], rs[8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        s[j] = 0.f; sx[j] = 0.f;
        mu[j] = mean[gi * g.ldm + c]; rs[j] = rstd[gi * g.ldm + c];
        sc[j] = rs[j] * (gamma ? gamma[c] : 1.f);
        sh[j] = (beta ? beta[c] : 0.f) - mu[j] * sc[j];
    }
    if (rsub < g.RPI) {
        const int64_t off = ((int64_t)gi * g.R) * g.C + c8 * 8;
        const T* xb = x + ((int64_t)gi * g.R) * g.ldx + c8 * 8;
        constexpr int U = bn_rows_in_flight<T>();
        for (int r = r0 + rsub; r < r1; r += U * g.RPI) {
            Row8<T> d[U], a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ru = r + u * g.RPI;
                if (ru < r1) {
                    d[u].load(dy + off + (int64_t)ru * g.C);
                    a[u].load(xb + (int64_t)ru * g.ldx);
                    if (res && relu) b[u].load(res + off + (int64_t)ru * g.C);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + u * g.RPI >= r1) break;
                float df[8], af[8], bf[8];
                d[u].unpack(df); a[u].unpack(af);
                if (res && relu) b[u].unpack(bf);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (af[j] - mu[j]) * rs[j];
                    float gdy = df[j];
                    if (relu) {                                  // the forward's own expression (same rounding): y = x * sc + sh (+ res)
                        float v = af[j] * sc[j] + sh[j];
                        if (res) v += bf[j];
                        if (!(v > 0.f)) gdy = 0.f;
                    }
                    s[j] += gdy; sx[j] += gdy * xh;
                }
            }
        }
    }
    float* mine = red + t * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mine[j] = s[j]; mine[8 + j] = sx[j]; }
    __syncthreads();
    if (rsub == 0) {
        for (int o = 1; o < g.RPI; ++o) {
            const float* p = red + (o * g.CH8 + c8) * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] += p[j]; sx[j] += p[8 + j]; }
        }
        const int64_t o = ((int64_t)gi * g.S + si) * g.C + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { psum[o + j] = s[j]; psumx[o + j] = sx[j]; }
    }
}

// sums over the splits ((256 / LANES) channels x LANES split lanes per block, as bn_finalize_kernel) -> per-group sdy / sdyx.  dgamma / dbeta
// (the sums over the groups) are added up in group order by the first blocks of bn_bwd_apply_kernel: no atomics, bitwise reproducible
// (ConVIRT / GLoRIA run G = batch / 4 = 64 groups; G-way float atomics made their BatchNorm weight gradients order-dependent)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ psum, const float* __restrict__ psumx,
                                                              float* __restrict__ sdy, float* __restrict__ sdyx, const BnGeom g, int lanes) {
    const int cpb = 256 / lanes, chunks = (g.C + cpb - 1) / cpb;
    const int gi = blockIdx.x / chunks, cl = threadIdx.x / lanes, l = threadIdx.x % lanes, c = (blockIdx.x % chunks) * cpb + cl;
    __shared__ float red[256 * 2];
    float a = 0.f, b = 0.f;
    if (c < g.C)
        for (int si = l; si < g.S; si += lanes) {
            const int64_t o = ((int64_t)gi * g.S + si) * g.C + c;
            a += psum[o]; b += psumx[o];
        }
    for (int st = lanes >> 1; st > 0; st >>= 1) {
        red[threadIdx.x * 2] = a; red[threadIdx.x * 2 + 1] = b;
        __syncthreads();
        if (l < st) { a += red[(threadIdx.x + st) * 2]; b += red[(threadIdx.x + st) * 2 + 1]; }
        __syncthreads();
    }
    if (l == 0 && c < g.C) { sdy[gi * g.C + c] = a; sdyx[gi * g.C + c] = b; }
}

// dx = gamma * rstd * (dy' - mean(dy') - xhat * mean(dy' * xhat));  dres = dy'
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ res,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ sdy, const float* __restrict__ sdyx,
                                                           T* __restrict__ dx, T* __restrict__ dres, const BnGeom g, int relu, int training,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int G) {
    const int gi = blockIdx.x / g.S, si = blockIdx.x % g.S;
    int r0, r1;
    split_range(g, si, r0, r1);
    const int t = threadIdx.x, c8 = t % g.CH8, rsub = t / g.CH8;
    if (dgamma || dbeta) {   // dgamma / dbeta += sum over the groups, in group order; one owner thread per channel (grid-stride: tiny launches have < C / 256 blocks)
        for (int c = blockIdx.x * 256 + t; c < g.C; c += (int)gridDim.x * 256) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < G; ++k) { a += sdy[k * g.C + c]; b += sdyx[k * g.C + c]; }
            if (dbeta) dbeta[c] += a;
            if (dgamma) dgamma[c] += b;
        }
    }
    if (rsub >= g.RPI) return;
    float mu[8], sc[8], sh[8], c1[8], c2[8];                     // dx = sc * dy' - c1 - c2 * (x - mean): c1 = sc * mean(dy'), c2 = sc * rstd * mean(dy' * xhat) * rstd
    const float invn = training ? 1.f / (float)g.R : 0.f;       // eval mode (running statistics are constants): dx = gamma * rstd * dy'
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        const float rs = rstd[gi * g.ldm + c];
        mu[j] = mean[gi * g.ldm + c];
        sc[j] = rs * (gamma ? gamma[c] : 1.f);
        sh[j] = (beta ? beta[c] : 0.f) - mu[j] * sc[j];
        c1[j] = sc[j] * (sdy[gi * g.C + c] * invn);
        c2[j] = sc[j] * (sdyx[gi * g.C + c] * invn) * rs;
    }
    const int64_t off = ((int64_t)gi * g.R) * g.C + c8 * 8;
    const T* xb = x + ((int64_t)gi * g.R) * g.ldx + c8 * 8;
    T* dxb = dx + ((int64_t)gi * g.R) * g.ldd + c8 * 8;
    constexpr int U = bn_rows_in_flight<T>();
    for (int r = r0 + rsub; r < r1; r += U * g.RPI) {
        Row8<T> d[U], a[U], b[U], e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ru = r + u * g.RPI;
            if (ru < r1) {
                d[u].load(dy + off + (int64_t)ru * g.C);
                a[u].load(xb + (int64_t)ru * g.ldx);
                if (g.acc) e[u].load(dxb + (int64_t)ru * g.ldd);      // dx already holds the gradient of the buffer's other consumers
                if (res && relu) b[u].load(res + off + (int64_t)ru * g.C);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ru = r + u * g.RPI;
            if (ru >= r1) break;
            float df[8], af[8], bf[8], ef[8], o[8];
            d[u].unpack(df); a[u].unpack(af);
            if (g.acc) e[u].unpack(ef);
            if (res && relu) b[u].unpack(bf);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gdy = df[j];
                if (relu) {
                    float v = af[j] * sc[j] + sh[j];
                    if (res) v += bf[j];
                    if (!(v > 0.f)) gdy = 0.f;
                }
                df[j] = gdy;
                o[j] = sc[j] * gdy - c1[j] - c2[j] * (af[j] - mu[j]);
                if (g.acc) o[j] += ef[j];
            }
            Ld8<T>::store(dxb + (int64_t)ru * g.ldd, o);
            if (dres) Ld8<T>::store(dres + off + (int64_t)ru * g.C, df);
        }
    }
}

// split lanes of the finalize kernels: enough that a lane walks at most ~16 partials
static int finalize_lanes(int S) { return S > 512 ? 64 : (S > 128 ? 32 : 16); }

BnGeom geometry(int G, int R, int C) {
    BnGeom g;
    g.G = G; g.R = R; g.C = C;
    g.ldx = C; g.ldd = C; g.ldm = C; g.acc = 0;
    g.CH8 = C / 8;
    g.RPI = 256 / g.CH8;
    // splits per group.  Round 6 (tools/bn_bench.py, profiles/r06_g_bn_bench.txt): ~512 blocks in all (two per CU, each streaming many rows) instead of
    // ~4096 -- the large layers are indifferent (+-3 %), the 14 x 14 and 7 x 7 layers that make up 128 of DenseNet-169's 169 BatchNorms gain 15-40 %
    // (a block's start-up -- its per-channel constants, the reduction tail, the partials the finalize kernel has to merge -- was most of their time);
    // every split keeps >= 4 row sweeps of the block.  VM_BN_BLOCKS overrides (diagnostic).
    static const int blocks = [] { const char* e = getenv("VM_BN_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    int S = (blocks + G - 1) / G;
    const int max_s = (R + 4 * g.RPI - 1) / (4 * g.RPI);
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    if (S > 1024) S = 1024;
    g.rows_per_split = (R + S - 1) / S;
    g.S = (R + g.rows_per_split - 1) / g.rows_per_split;
    return g;
}

}  // namespace

extern "C" size_t vm_batchnorm_nhwc_ws(int G, int rows_per_group, int C) {
    if (G <= 0 || rows_per_group <= 0 || C <= 0 || (C % 8) || C > 2048) return 0;
    const BnGeom g = geometry(G, rows_per_group, C);
    return ((size_t)2 * G * g.S * C + (size_t)G * g.S + (size_t)2 * G * C) * sizeof(float);
}

// ---- forward in two calls, so that a DenseNet block can keep ONE feature buffer: the statistics of a channel do not change from layer to layer (every
// norm1 of a block normalises the same values with its own gamma / beta), so each layer computes them for its 32 NEW channels only (copying those
// rows into the buffer on the way) and normalises the first C channels of the wide buffer with the statistics array of the whole block.
extern "C" int vm_batchnorm_nhwc_stats(const void* x, int64_t ldx, void* copy_dst, int64_t ld_copy, float* mean, float* rstd, float* var, int ldm,
                                       int64_t* num_batches_tracked, int G, int rows_per_group, int C, float eps, int dtype, void* ws, size_t ws_bytes,
                                       void* stream) {
    VM_REQUIRE(x && mean && rstd && var && G > 0 && rows_per_group > 0 && C > 0 && (C % 8) == 0 && C <= 2048 && (dtype == VM_BF16 || dtype == VM_F32) &&
               ldx >= C && (ldx % 8) == 0 && ldm >= C && (!copy_dst || (ld_copy >= C && (ld_copy % 8) == 0)),
               "vm_batchnorm_nhwc_stats: bad arguments (C %% 8 == 0, C <= 2048, strides >= C and multiples of 8)");
    VM_REQUIRE(ws && ws_bytes >= vm_batchnorm_nhwc_ws(G, rows_per_group, C), "vm_batchnorm_nhwc_stats: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    BnGeom g = geometry(G, rows_per_group, C);
    g.ldx = ldx; g.ldm = ldm;
    const double bytes = (double)G * rows_per_group * C * (dtype == VM_BF16 ? 2 : 4);
    VmProfScope prof(VM_FAM_LN, (copy_dst ? 2.0 : 1.0) * bytes, s);
    float* pmean = (float*)ws;
    float* pm2 = pmean + (size_t)G * g.S * C;
    float* pcnt = pm2 + (size_t)G * g.S * C;
    if (dtype == VM_BF16) hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(G * g.S), dim3(256), 0, s, (const bf16_t*)x, pmean, pm2, pcnt, g, (bf16_t*)copy_dst, ld_copy, num_batches_tracked);
    else hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(G * g.S), dim3(256), 0, s, (const float*)x, pmean, pm2, pcnt, g, (float*)copy_dst, ld_copy, num_batches_tracked);
    const int lanes = finalize_lanes(g.S), cpb = 256 / lanes;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(G * ((C + cpb - 1) / cpb)), dim3(256), 0, s, pmean, pm2, pcnt, mean, rstd, var, g, eps, lanes);
    return vm_check_launch("vm_batchnorm_nhwc_stats");
}

// y = relu?((x - mean) * rstd * gamma + beta + residual?) over the first C channels of rows ldx apart; with running_mean / running_var the exponential
// (momentum >= 0) or cumulative (momentum < 0, needs num_batches_tracked as left by vm_batchnorm_nhwc_stats) moving averages take one update per group;
// rstd == NULL: rsqrt(var + eps) is taken in the kernel (inference straight from running_mean / running_var)
extern "C" int vm_batchnorm_nhwc_apply(const void* x, int64_t ldx, const void* residual, void* y, const float* gamma, const float* beta,
                                       const float* mean, const float* rstd, const float* var, int ldm, float* running_mean, float* running_var,
                                       const int64_t* num_batches_tracked, float momentum, float eps, int G, int rows_per_group, int C, int dtype,
                                       int relu, void* stream) {
    VM_REQUIRE(x && y && mean && (rstd || var) && G > 0 && rows_per_group > 0 && C > 0 && (C % 8) == 0 && C <= 2048 && (dtype == VM_BF16 || dtype == VM_F32) &&
               ldx >= C && (ldx % 8) == 0 && ldm >= C, "vm_batchnorm_nhwc_apply: bad arguments (C %% 8 == 0, C <= 2048, strides >= C)");
    VM_REQUIRE(!running_mean || (running_var && var && (momentum >= 0.f || num_batches_tracked)),
               "vm_batchnorm_nhwc_apply: the running update needs running_var, var and (cumulative average) num_batches_tracked");
    hipStream_t s = (hipStream_t)stream;
    BnGeom g = geometry(G, rows_per_group, C);
    g.ldx = ldx; g.ldm = ldm;
    const double bytes = (double)G * rows_per_group * C * (dtype == VM_BF16 ? 2 : 4);
    VmProfScope prof(VM_FAM_LN, (residual ? 3.0 : 2.0) * bytes, s);
    if (dtype == VM_BF16)
        hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(G * g.S), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, gamma, beta, mean, rstd, g, relu,
                           var, running_mean, running_var, num_batches_tracked, momentum, eps);
    else
        hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(G * g.S), dim3(256), 0, s, (const float*)x, (const float*)residual, (float*)y, gamma, beta, mean, rstd, g, relu,
                           var, running_mean, running_var, num_batches_tracked, momentum, eps);
    return vm_check_launch("vm_batchnorm_nhwc_apply");
}

extern "C" int vm_batchnorm_nhwc_fwd(const void* x, const void* residual, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                                     float* var, int G, int rows_per_group, int C, float eps, int dtype, int relu, int training, void* ws,
                                     size_t ws_bytes, void* stream) {
    VM_REQUIRE(x && y && mean && rstd && G > 0 && rows_per_group > 0 && C > 0 && (C % 8) == 0 && C <= 2048 && (dtype == VM_BF16 || dtype == VM_F32),
               "vm_batchnorm_nhwc_fwd: bad arguments (C %% 8 == 0, C <= 2048)");
    if (training) {
        VM_REQUIRE(var && ws && ws_bytes >= vm_batchnorm_nhwc_ws(G, rows_per_group, C), "vm_batchnorm_nhwc_fwd: workspace too small");
        const int rc = vm_batchnorm_nhwc_stats(x, C, nullptr, 0, mean, rstd, var, C, nullptr, G, rows_per_group, C, eps, dtype, ws, ws_bytes, stream);
        if (rc != VM_OK) return rc;
    }
    return vm_batchnorm_nhwc_apply(x, C, residual, y, gamma, beta, mean, rstd, var, C, nullptr, nullptr, nullptr, 0.f, eps, G, rows_per_group, C, dtype, relu, stream);
}

// x: first C channels of rows ldx apart; dx: rows lddx apart, `accumulate` adds into it (the gradient buffer of a DenseNet block: every layer's
// BatchNorm gradient lands on the channels it read, no separate add); mean / rstd: group stride ldm
extern "C" int vm_batchnorm_nhwc_bwd_ex(const void* dy, const void* x, int64_t ldx, const void* residual, const float* gamma, const float* beta,
                                        const float* mean, const float* rstd, int ldm, void* dx, int64_t lddx, int accumulate, void* dres, float* dgamma,
                                        float* dbeta, int G, int rows_per_group, int C, int dtype, int relu, int training, void* ws, size_t ws_bytes,
                                        void* stream) {
    VM_REQUIRE(dy && x && dx && mean && rstd && G > 0 && rows_per_group > 0 && C > 0 && (C % 8) == 0 && C <= 2048 && (dtype == VM_BF16 || dtype == VM_F32) &&
               ldx >= C && (ldx % 8) == 0 && lddx >= C && (lddx % 8) == 0 && ldm >= C,
               "vm_batchnorm_nhwc_bwd: bad arguments (C %% 8 == 0, C <= 2048, strides >= C and multiples of 8)");
    VM_REQUIRE(ws && ws_bytes >= vm_batchnorm_nhwc_ws(G, rows_per_group, C), "vm_batchnorm_nhwc_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    BnGeom g = geometry(G, rows_per_group, C);
    g.ldx = ldx; g.ldd = lddx; g.ldm = ldm; g.acc = accumulate ? 1 : 0;
    const double bytes = (double)G * rows_per_group * C * (dtype == VM_BF16 ? 2 : 4);
    VmProfScope prof(VM_FAM_LN, (5.0 + (accumulate ? 1.0 : 0.0) + (dres ? 1.0 : 0.0)) * bytes, s);
    float* psum = (float*)ws;
    float* psumx = psum + (size_t)G * g.S * C;
    float* sdy = psumx + (size_t)G * g.S * C + (size_t)G * g.S;
    float* sdyx = sdy + (size_t)G * C;
    if (dtype == VM_BF16)
        hipLaunchKernelGGL(bn_bwd_stats_kernel<bf16_t>, dim3(G * g.S), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)residual, gamma, beta, mean, rstd, psum, psumx, g, relu);
    else
        hipLaunchKernelGGL(bn_bwd_stats_kernel<float>, dim3(G * g.S), dim3(256), 0, s, (const float*)dy, (const float*)x, (const float*)residual, gamma, beta, mean, rstd, psum, psumx, g, relu);
    {
        const int lanes = finalize_lanes(g.S), cpb = 256 / lanes;
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(G * ((C + cpb - 1) / cpb)), dim3(256), 0, s, psum, psumx, sdy, sdyx, g, lanes);
    }
    if (dtype == VM_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(G * g.S), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)residual, gamma, beta, mean, rstd, sdy, sdyx, (bf16_t*)dx, (bf16_t*)dres, g, relu, training, dgamma, dbeta, G);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(G * g.S), dim3(256), 0, s, (const float*)dy, (const float*)x, (const float*)residual, gamma, beta, mean, rstd, sdy, sdyx, (float*)dx, (float*)dres, g, relu, training, dgamma, dbeta, G);
    return vm_check_launch("vm_batchnorm_nhwc_bwd");
}

extern "C" int vm_batchnorm_nhwc_bwd(const void* dy, const void* x, const void* residual, const float* gamma, const float* beta, const float* mean,
                                     const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta, int G, int rows_per_group, int C,
                                     int dtype, int relu, int training, void* ws, size_t ws_bytes, void* stream) {
    return vm_batchnorm_nhwc_bwd_ex(dy, x, C, residual, gamma, beta, mean, rstd, C, dx, C, 0, dres, dgamma, dbeta, G, rows_per_group, C, dtype, relu,
                                    training, ws, ws_bytes, stream);
}
