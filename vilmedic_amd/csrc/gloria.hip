// gloria.hip -- the GLoRIA local (word x image-region) contrastive loss, forward and backward, for EVERY (caption i, image j) pair of
// the batch at once (ref:vilmedic/blocks/losses/selfsup/GLoRIALoss.py:14-51 gloria_attention_fn, :78-129 local_loss; the reference loops
// over captions in Python, repeats each caption B times and runs two torch.bmm + two softmax + a cosine per caption).
//
// Per pair, with W = the caption's word embeddings [T_i, D] and C = the image's region features [D, P]:
//     S[t,p]  = <W_t, C_p>                                   (one GEMM for all pairs; "bf16 x 3" operands, see vm_split3_bf16)
//     a1[t,p] = softmax over the caption's words t of S[.,p]            \
//     a2[t,p] = softmax over the regions p of temp1 * a1[t,.]            > gloria_attn_fwd_kernel (this file)
//     dot[t]  = sum_p a2[t,p] S[t,p] = <W_t, x_t>                       /
//     x_t     = sum_p a2[t,p] C_p    (attention-weighted context)        (one GEMM per image)
//     cos[t]  = dot[t] / max(|W_t| |x_t|, eps);  sims[j,i] = temp3 * log sum_t exp(temp2 cos[t])      gloria_cos_fwd_kernel
// followed by a cross-entropy over sims and sims^T (vm_ce_smooth_fwd_bwd, smoothing 0).  The backward pass mirrors it:
// gloria_cos_bwd_kernel (d cos -> d dot, d x, the first half of d W), two GEMMs per image, gloria_attn_bwd_kernel (both softmax
// backward passes fused, d S), two GEMMs over all pairs.  All arithmetic is fp32 (the reference computes this loss in fp32 and it is
// ill-conditioned: temp2 * temp3 = 50 multiplies every error of cos in the logits); the kernels here are HBM / L2-bound row and
// column passes: one wave per word row (regions across the lanes, DPP wave reductions), one thread per region column.
//
// Ragged captions: rows t >= cap_lens[i] of a caption are masked everywhere (their a2 / d x / d S rows are written as zeros because
// the GEMMs read them).  P and D are zero-padded to multiples of 16 by the caller (Pp, the leading dimensions) for the GEMMs.
#include "common.h"

#define GL_MAXK 8            // regions per lane: P <= 512
#define GL_MAXD4 4           // float4 chunks per lane of a feature row: D <= 1024

// ------------------------------------------------------------------ batched transpose with zero padding
// dst[b][c][r] = src[b][r][c] for c < drows, r < dcols (zero where r >= rows or c >= cols); 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, int64_t sbs, int64_t lds_, float* __restrict__ dst,
                                                            int64_t dbs, int64_t ldd, int rows, int cols, int drows, int dcols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const float* s = src + (int64_t)blockIdx.z * sbs;
    float* d = dst + (int64_t)blockIdx.z * dbs;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? s[(int64_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < drows && r < dcols) d[(int64_t)c * ldd + r] = tile[tx][ty + 8 * k];
    }
}
extern "C" int vm_transpose_f32(const float* src, int64_t src_batch_stride, int64_t ld_src, float* dst, int64_t dst_batch_stride, int64_t ld_dst,
                                int batch, int rows, int cols, int dst_rows, int dst_cols, void* stream) {
    VM_REQUIRE(src && dst && batch > 0 && rows > 0 && cols > 0 && dst_rows >= cols && dst_cols >= rows && ld_src >= cols && ld_dst >= dst_cols,
               "vm_transpose_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 8.0 * batch * (double)dst_rows * dst_cols, s);
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((dst_cols + 31) / 32, (dst_rows + 31) / 32, batch), dim3(256), 0, s, src, src_batch_stride, ld_src,
                       dst, dst_batch_stride, ld_dst, rows, cols, dst_rows, dst_cols);
    return vm_check_launch("vm_transpose_f32");
}

// ------------------------------------------------------------------ fp32 -> three bf16 operand parts ("bf16 x 3" products)
// The contractions of this loss need fp32-grade accuracy (see the header) but are 0.5 TFLOP per step at B = 48: on the f32 MFMA
// (1/16 of the bf16 rate) they would dominate the GLoRIA step.  Instead every fp32 operand is split a = hi + lo with hi = bf16(a),
// lo = bf16(a - hi) (a - hi is exact in fp32), and a . b ~= hi_a hi_b + hi_a lo_b + lo_a hi_b (the dropped lo_a lo_b term is 2^-18 of
// the product, the parts themselves carry 16 mantissa bits): ONE bf16 MFMA GEMM (vm_gemm_bf16, fp32 accumulate and output) with the
// contraction three times as long, A-side parts (hi, hi, lo), B-side parts (hi, lo, hi).  This kernel lays the parts out next to each
// other along the contraction axis: along the columns (cat: [rows, 3 cols], operands stored contraction-contiguous) or along the
// rows (stack: [3 rows, cols], operands stored contraction-major), in blocks of ``block`` (the contraction length of one GEMM call,
// so that per-image sub-matrices stay contiguous: block b occupies [3 b block, 3 (b+1) block)).
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, int64_t ld_src, int rows, int cols, bf16_t* __restrict__ dst,
                                                     int64_t ld_dst, int role, int along_rows, int block) {
    const int c4n = cols >> 2;
    const int64_t total = (int64_t)rows * c4n;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int r = (int)(idx / c4n), c = (int)(idx % c4n) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)r * ld_src + c);
        const float f[4] = {v.x, v.y, v.z, v.w};
        float hi[4], lo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { hi[k] = bf16_to_f32(f32_to_bf16(f[k])); lo[k] = f[k] - hi[k]; }
        uint2 H, Lw;
        H.x = pack_bf16x2(hi[0], hi[1]); H.y = pack_bf16x2(hi[2], hi[3]);
        Lw.x = pack_bf16x2(lo[0], lo[1]); Lw.y = pack_bf16x2(lo[2], lo[3]);
        const uint2 part1 = role == 0 ? H : Lw, part2 = role == 0 ? Lw : H;       // A: (hi, hi, lo)   B: (hi, lo, hi)
        int64_t o0, step;
        if (along_rows) { o0 = ((int64_t)(r / block) * 3 * block + (r % block)) * ld_dst + c; step = (int64_t)block * ld_dst; }
        else { o0 = (int64_t)r * ld_dst + (int64_t)(c / block) * 3 * block + (c % block); step = block; }
        *reinterpret_cast<uint2*>(dst + o0) = H;
        *reinterpret_cast<uint2*>(dst + o0 + step) = part1;
        *reinterpret_cast<uint2*>(dst + o0 + 2 * step) = part2;
    }
}
extern "C" int vm_split3_bf16(const float* src, int64_t ld_src, int rows, int cols, void* dst, int64_t ld_dst, int role, int along_rows, int block,
                              void* stream) {
    VM_REQUIRE(src && dst && rows > 0 && cols > 0 && (cols % 4) == 0 && (ld_src % 4) == 0 && (ld_dst % 4) == 0 && block > 0,
               "vm_split3_bf16: bad arguments");
    VM_REQUIRE((role == 0 || role == 1) && (along_rows ? rows % block == 0 : (cols % block == 0 && block % 4 == 0)),
               "vm_split3_bf16: the split axis must be a multiple of block=%d (rows=%d cols=%d)", block, rows, cols);
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 10.0 * rows * (double)cols, s);
    const int64_t total = (int64_t)rows * (cols / 4);
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0, s, src, ld_src, rows, cols,
                       (bf16_t*)dst, ld_dst, role, along_rows, block);
    return vm_check_launch("vm_split3_bf16");
}

// ------------------------------------------------------------------ |row| of an fp32 matrix (one wave per row)
__global__ __launch_bounds__(256) void row_norm_f32_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = x[(int64_t)row * ldx + c]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) out[row] = sqrtf(s);
}
extern "C" int vm_row_norm_f32(const float* x, int64_t ldx, float* out, int rows, int cols, void* stream) {
    VM_REQUIRE(x && out && rows > 0 && cols > 0 && ldx >= cols, "vm_row_norm_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 4.0 * rows * (double)cols, s);
    hipLaunchKernelGGL(row_norm_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, out, rows, cols);
    return vm_check_launch("vm_row_norm_f32");
}

// ------------------------------------------------------------------ forward: both softmaxes + <W_t, x_t>
// grid (B images j, B captions i); S_ij[t][p] = S[(i Tp + t) ldS + j Pp + p].
// Pass 1, one thread per region p: max / sum over the caption's words (saved in colstat for the backward pass).
// Pass 2, one wave per word t: a1 from the column statistics, e = exp(temp1 a1) (a1 <= 1: no max subtraction needed), row sum by a
// wave reduction, a2 = e / sum written image-major ([j][i Tp + t][Pp], the A operand of the per-image context GEMM), dot[t].
struct GloriaArgs {
    const float* S; int64_t ldS; const int32_t* cap_lens; int Tp, P, Pp; float temp1;
    float* a2; float* dot; float* colstat;          // colstat [B cap][B img][2][Pp]
    float* da2; float* dS;                          // backward
};

__device__ __forceinline__ void gloria_col_stats(const float* Sij, int64_t ldS, int Ti, int P, float* cmax, float* cinv) {
    for (int p = threadIdx.x; p < P; p += 256) {
        float m = -INFINITY;
        for (int t = 0; t < Ti; ++t) m = fmaxf(m, Sij[(int64_t)t * ldS + p]);
        float s = 0.f;
        for (int t = 0; t < Ti; ++t) s += __expf(Sij[(int64_t)t * ldS + p] - m);
        cmax[p] = m; cinv[p] = 1.0f / s;
    }
}

__global__ __launch_bounds__(256) void gloria_attn_fwd_kernel(const GloriaArgs a) {
    extern __shared__ float sh[];
    float* cmax = sh; float* cinv = sh + a.Pp;
    const int j = blockIdx.x, i = blockIdx.y, B = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Ti = min(a.cap_lens[i], a.Tp);
    const float* Sij = a.S + (int64_t)i * a.Tp * a.ldS + (int64_t)j * a.Pp;
    gloria_col_stats(Sij, a.ldS, Ti, a.P, cmax, cinv);
    __syncthreads();
    float* cs = a.colstat + ((int64_t)i * B + j) * 2 * a.Pp;
    for (int p = threadIdx.x; p < a.Pp; p += 256) { cs[p] = p < a.P ? cmax[p] : 0.f; cs[a.Pp + p] = p < a.P ? cinv[p] : 0.f; }
    for (int t = wave; t < a.Tp; t += 4) {
        float* arow = a.a2 + ((int64_t)j * B * a.Tp + (int64_t)i * a.Tp + t) * a.Pp;
        const int64_t di = ((int64_t)i * a.Tp + t) * B + j;
        if (t >= Ti) {
            for (int p = lane; p < a.Pp; p += 64) arow[p] = 0.f;
            if (lane == 0) a.dot[di] = 0.f;
            continue;
        }
        float e[GL_MAXK], sv[GL_MAXK];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < GL_MAXK; ++k) {
            const int p = lane + 64 * k;
            e[k] = 0.f; sv[k] = 0.f;
            if (p < a.P) {
                sv[k] = Sij[(int64_t)t * a.ldS + p];
                const float a1 = __expf(sv[k] - cmax[p]) * cinv[p];
                e[k] = __expf(a.temp1 * a1);
                sum += e[k];
            }
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < GL_MAXK; ++k) {
            const int p = lane + 64 * k;
            if (p < a.Pp) {
                const float v = e[k] * inv;          // e = 0 past P: the padding columns are written as zeros
                arow[p] = v;
                d += v * sv[k];
            }
        }
        d = wave_sum(d);
        if (lane == 0) a.dot[di] = d;
    }
}

extern "C" int vm_gloria_attn_fwd(const float* S, int64_t ldS, const int32_t* cap_lens, int B, int Tp, int P, int Pp, float temp1,
                                  float* a2, float* dot, float* colstat, void* stream) {
    VM_REQUIRE(S && cap_lens && a2 && dot && colstat && B > 0 && Tp > 0 && P > 0 && Pp >= P, "vm_gloria_attn_fwd: bad arguments");
    VM_REQUIRE(P <= 64 * GL_MAXK, "vm_gloria_attn_fwd: at most %d regions per image (got %d)", 64 * GL_MAXK, P);
    GloriaArgs a = {S, ldS, cap_lens, Tp, P, Pp, temp1, a2, dot, colstat, nullptr, nullptr};
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 12.0 * B * (double)B * Tp * Pp, s);
    hipLaunchKernelGGL(gloria_attn_fwd_kernel, dim3(B, B), dim3(256), (size_t)2 * Pp * sizeof(float), s, a);
    return vm_check_launch("vm_gloria_attn_fwd");
}

// ------------------------------------------------------------------ forward: cosine, log-sum-exp over the words -> sims
// grid (B images, B captions); one wave per word row of x (|x_t| by a wave reduction over D)
__global__ __launch_bounds__(256) void gloria_cos_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ nw,
                                                             const float* __restrict__ dot, const int32_t* __restrict__ cap_lens, int Tp, int D,
                                                             float temp2, float temp3, float eps, float* __restrict__ sims,
                                                             float* __restrict__ simsT, float* __restrict__ cosv, float* __restrict__ nxv) {
    __shared__ float part[4];
    const int j = blockIdx.x, i = blockIdx.y, B = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Ti = min(cap_lens[i], Tp);
    float se = 0.f;
    for (int t = wave; t < Ti; t += 4) {
        const float* row = x + ((int64_t)j * B * Tp + (int64_t)i * Tp + t) * ldx;
        float ss = 0.f;
        for (int d4 = lane; d4 < (D >> 2); d4 += 64) {
            const float4 v = *reinterpret_cast<const float4*>(row + 4 * d4);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        const float nx = sqrtf(wave_sum(ss));
        const int64_t idx = ((int64_t)i * Tp + t) * B + j;
        const float c = dot[idx] / fmaxf(nw[i * Tp + t] * nx, eps);
        if (lane == 0) { cosv[idx] = c; nxv[idx] = nx; }
        se += __expf(temp2 * c);
    }
    if (lane == 0) part[wave] = se;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float r = temp3 * __logf(part[0] + part[1] + part[2] + part[3]);
        sims[(int64_t)j * B + i] = r;
        simsT[(int64_t)i * B + j] = r;
    }
}
extern "C" int vm_gloria_cos_fwd(const float* x, int64_t ldx, const float* word_norm, const float* dot, const int32_t* cap_lens, int B, int Tp, int D,
                                 float temp2, float temp3, float eps, float* sims, float* simsT, float* cosv, float* nxv, void* stream) {
    VM_REQUIRE(x && word_norm && dot && cap_lens && sims && simsT && cosv && nxv && B > 0 && Tp > 0 && D > 0 && (D % 4) == 0 && (ldx % 4) == 0,
               "vm_gloria_cos_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 4.0 * B * (double)B * Tp * D, s);
    hipLaunchKernelGGL(gloria_cos_fwd_kernel, dim3(B, B), dim3(256), 0, s, x, ldx, word_norm, dot, cap_lens, Tp, D, temp2, temp3, eps, sims, simsT, cosv, nxv);
    return vm_check_launch("vm_gloria_cos_fwd");
}

// ------------------------------------------------------------------ backward through the cosine
// One wave per word row (i, t), looping over the B images: with g = d L / d r (r = sims / temp3, from both cross-entropies),
//   d cos = g temp2 exp(temp2 cos - r);   d dot = d cos / (|W||x|);   d|x| = -d cos cos / |x|;   d|W| = -d cos cos / |W|
//   d x_t      = d dot W_t + (d|x| / |x|) x_t                          (written per image: the operand of two GEMMs; dot = <W_t, x_t>)
//   d W_t (1)  = sum_j [d dot x_t]  +  (sum_j d|W| / |W|) W_t          (accumulated in registers across the images: no atomics)
// (clamped pairs, |W||x| <= eps, pass d dot = d cos / eps only -- torch's clamp(min=eps) has zero gradient below the bound.)
__global__ __launch_bounds__(256) void gloria_cos_bwd_kernel(const float* __restrict__ dsims, const float* __restrict__ dsimsT,
                                                             const float* __restrict__ sims, const float* __restrict__ cosv,
                                                             const float* __restrict__ nxv, const float* __restrict__ nw, const float* __restrict__ x,
                                                             const float* __restrict__ Wt, int64_t ldx, const int32_t* __restrict__ cap_lens, int B,
                                                             int Tp, int D, float temp2, float temp3, float eps, float* __restrict__ dx,
                                                             float* __restrict__ dW1) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * Tp) return;
    const int i = row / Tp, t = row % Tp;
    const bool valid = t < cap_lens[i];
    const int n4 = D >> 2;
    float4 w[GL_MAXD4], acc[GL_MAXD4];
#pragma unroll
    for (int k = 0; k < GL_MAXD4; ++k) {
        const int d4 = lane + 64 * k;
        w[k] = d4 < n4 ? *reinterpret_cast<const float4*>(Wt + (int64_t)row * ldx + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float nwv = nw[row];
    float gsum = 0.f;
    for (int j = 0; j < B; ++j) {
        const int64_t off = ((int64_t)j * B * Tp + row) * ldx;
        const int64_t idx = (int64_t)row * B + j;
        if (!valid) {
#pragma unroll
            for (int k = 0; k < GL_MAXD4; ++k) {
                const int d4 = lane + 64 * k;
                if (d4 < n4) *reinterpret_cast<float4*>(dx + off + 4 * d4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            continue;
        }
        const float r = sims[(int64_t)j * B + i] / temp3;
        const float g = temp3 * (dsims[(int64_t)j * B + i] + dsimsT[(int64_t)i * B + j]);
        const float c = cosv[idx], nx = nxv[idx];
        const float dcos = g * temp2 * __expf(temp2 * c - r);
        const float den = nwv * nx;
        float dd, beta, gamma;
        if (den > eps) { dd = dcos / den; beta = -dcos * c / (nx * nx); gamma = -dcos * c / (nwv * nwv); }
        else { dd = dcos / eps; beta = 0.f; gamma = 0.f; }
        gsum += gamma;
#pragma unroll
        for (int k = 0; k < GL_MAXD4; ++k) {
            const int d4 = lane + 64 * k;
            if (d4 < n4) {
                const float4 xv = *reinterpret_cast<const float4*>(x + off + 4 * d4);
                *reinterpret_cast<float4*>(dx + off + 4 * d4) =
                    make_float4(dd * w[k].x + beta * xv.x, dd * w[k].y + beta * xv.y, dd * w[k].z + beta * xv.z, dd * w[k].w + beta * xv.w);
                acc[k].x += dd * xv.x; acc[k].y += dd * xv.y; acc[k].z += dd * xv.z; acc[k].w += dd * xv.w;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < GL_MAXD4; ++k) {
        const int d4 = lane + 64 * k;
        if (d4 < n4)
            *reinterpret_cast<float4*>(dW1 + (int64_t)row * ldx + 4 * d4) =
                make_float4(acc[k].x + gsum * w[k].x, acc[k].y + gsum * w[k].y, acc[k].z + gsum * w[k].z, acc[k].w + gsum * w[k].w);
    }
}
extern "C" int vm_gloria_cos_bwd(const float* dsims, const float* dsimsT, const float* sims, const float* cosv, const float* nxv, const float* word_norm,
                                 const float* x, const float* Wt, int64_t ldx, const int32_t* cap_lens, int B, int Tp, int D, float temp2, float temp3,
                                 float eps, float* dx, float* dW1, void* stream) {
    VM_REQUIRE(dsims && dsimsT && sims && cosv && nxv && word_norm && x && Wt && cap_lens && dx && dW1, "vm_gloria_cos_bwd: null pointer");
    VM_REQUIRE(B > 0 && Tp > 0 && D > 0 && (D % 4) == 0 && (ldx % 4) == 0 && D <= 256 * GL_MAXD4, "vm_gloria_cos_bwd: D=%d must be a multiple of 4 and <= %d", D,
               256 * GL_MAXD4);
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 8.0 * B * (double)B * Tp * D, s);
    hipLaunchKernelGGL(gloria_cos_bwd_kernel, dim3((B * Tp + 3) / 4), dim3(256), 0, s, dsims, dsimsT, sims, cosv, nxv, word_norm, x, Wt, ldx, cap_lens, B, Tp,
                       D, temp2, temp3, eps, dx, dW1);
    return vm_check_launch("vm_gloria_cos_bwd");
}

// ------------------------------------------------------------------ backward through both softmaxes
// grid (B images, B captions).  da2 [j][i Tp + t][Pp] arrives holding d x_t . C_p (the per-image GEMM) and is used as scratch.
// Pass A, one wave per word t (a1, e, a2 recomputed from S and the saved column statistics):
//     g[p]   = da2[t,p]      (the <W_t, x_t> term of the cosine reached a2 through d x_t = d dot W_t + ..: it is inside the GEMM result)
//     da1[p] = temp1 a2[p] (g[p] - sum_p' a2[p'] g[p'])         softmax over the regions          -> kept in da2
//     q[p]  += a1[t,p] da1[p]                                   per-lane partial of the column term, combined across the waves in LDS
// Pass B, same rows:  d S[t,p] = a1[t,p] (da1[p] - q[p])                     (softmax over the words)
__global__ __launch_bounds__(256) void gloria_attn_bwd_kernel(const GloriaArgs a) {
    extern __shared__ float sh[];
    float* cmax = sh; float* cinv = sh + a.Pp; float* q = sh + 2 * a.Pp;      // q [4][Pp]
    const int j = blockIdx.x, i = blockIdx.y, B = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Ti = min(a.cap_lens[i], a.Tp);
    const float* Sij = a.S + (int64_t)i * a.Tp * a.ldS + (int64_t)j * a.Pp;
    const float* cs = a.colstat + ((int64_t)i * B + j) * 2 * a.Pp;
    for (int p = threadIdx.x; p < a.Pp; p += 256) { cmax[p] = cs[p]; cinv[p] = cs[a.Pp + p]; }
    __syncthreads();
    float qacc[GL_MAXK];
#pragma unroll
    for (int k = 0; k < GL_MAXK; ++k) qacc[k] = 0.f;
    for (int t = wave; t < Ti; t += 4) {
        float* grow = a.da2 + ((int64_t)j * B * a.Tp + (int64_t)i * a.Tp + t) * a.Pp;
        float e[GL_MAXK], a1[GL_MAXK], g[GL_MAXK];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < GL_MAXK; ++k) {
            const int p = lane + 64 * k;
            e[k] = 0.f; a1[k] = 0.f; g[k] = 0.f;
            if (p < a.P) {
                const float sv = Sij[(int64_t)t * a.ldS + p];
                a1[k] = __expf(sv - cmax[p]) * cinv[p];
                e[k] = __expf(a.temp1 * a1[k]);
                sum += e[k];
                g[k] = grow[p];
            }
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float rd = 0.f;
#pragma unroll
        for (int k = 0; k < GL_MAXK; ++k) rd += e[k] * inv * g[k];
        rd = wave_sum(rd);
#pragma unroll
        for (int k = 0; k < GL_MAXK; ++k) {
            const int p = lane + 64 * k;
            if (p < a.P) {
                const float da1 = a.temp1 * e[k] * inv * (g[k] - rd);
                grow[p] = da1;
                qacc[k] += a1[k] * da1;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < GL_MAXK; ++k) {
        const int p = lane + 64 * k;
        if (p < a.Pp) q[wave * a.Pp + p] = qacc[k];
    }
    __syncthreads();
    for (int t = wave; t < a.Tp; t += 4) {
        float* drow = a.dS + ((int64_t)i * a.Tp + t) * a.ldS + (int64_t)j * a.Pp;
        if (t >= Ti) {
            for (int p = lane; p < a.Pp; p += 64) drow[p] = 0.f;
            continue;
        }
        const float* grow = a.da2 + ((int64_t)j * B * a.Tp + (int64_t)i * a.Tp + t) * a.Pp;
#pragma unroll
        for (int k = 0; k < GL_MAXK; ++k) {
            const int p = lane + 64 * k;
            if (p < a.P) {
                const float sv = Sij[(int64_t)t * a.ldS + p];
                const float a1 = __expf(sv - cmax[p]) * cinv[p];
                const float qq = q[p] + q[a.Pp + p] + q[2 * a.Pp + p] + q[3 * a.Pp + p];
                drow[p] = a1 * (grow[p] - qq);
            } else if (p < a.Pp) {
                drow[p] = 0.f;
            }
        }
    }
}
extern "C" int vm_gloria_attn_bwd(const float* S, int64_t ldS, const float* colstat, float* da2, const int32_t* cap_lens, int B, int Tp,
                                  int P, int Pp, float temp1, float* dS, void* stream) {
    VM_REQUIRE(S && colstat && da2 && cap_lens && dS && B > 0 && Tp > 0 && P > 0 && Pp >= P, "vm_gloria_attn_bwd: bad arguments");
    VM_REQUIRE(P <= 64 * GL_MAXK, "vm_gloria_attn_bwd: at most %d regions per image (got %d)", 64 * GL_MAXK, P);
    GloriaArgs a = {S, ldS, cap_lens, Tp, P, Pp, temp1, nullptr, nullptr, const_cast<float*>(colstat), da2, dS};
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 20.0 * B * (double)B * Tp * Pp, s);
    hipLaunchKernelGGL(gloria_attn_bwd_kernel, dim3(B, B), dim3(256), (size_t)6 * Pp * sizeof(float), s, a);
    return vm_check_launch("vm_gloria_attn_bwd");
}
