// elementwise.hip -- HBM-bound glue kernels: casts, embeddings, ViT patch gather / assembly, column sums.
// All use 16-B (bf16x8) or 32-B (fp32x8) accesses per lane and grid-stride loops (<= 2048 blocks).
#include "common.h"

static inline int grid_for(int64_t work_items, int block = 256) {
    int64_t b = (work_items + block - 1) / block;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------ casts
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
    const int64_t nvec = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const float4 a = *reinterpret_cast<const float4*>(in + i * 8), b = *reinterpret_cast<const float4*>(in + i * 8 + 4);
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        *reinterpret_cast<uint4*>(out + i * 8) = pack8(f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) out[(nvec << 3) + threadIdx.x] = f32_to_bf16(in[(nvec << 3) + threadIdx.x]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t nvec = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(in + i * 8), f);
        *reinterpret_cast<float4*>(out + i * 8) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(out + i * 8 + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) out[(nvec << 3) + threadIdx.x] = bf16_to_f32(in[(nvec << 3) + threadIdx.x]);
}
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols, int64_t ld) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        dst[(int64_t)r * ld + c] = f32_to_bf16(src[i]);
    }
}

extern "C" int vm_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
    VM_REQUIRE(in && out && n >= 0, "vm_cast_f32_to_bf16: bad arguments");
    if (n == 0) return VM_OK;
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 6.0 * n, s);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for((n + 7) / 8)), dim3(256), 0, s, in, (bf16_t*)out, n);
    return vm_check_launch("vm_cast_f32_to_bf16");
}
extern "C" int vm_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream) {
    VM_REQUIRE(in && out && n >= 0, "vm_cast_bf16_to_f32: bad arguments");
    if (n == 0) return VM_OK;
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 6.0 * n, s);
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for((n + 7) / 8)), dim3(256), 0, s, (const bf16_t*)in, out, n);
    return vm_check_launch("vm_cast_bf16_to_f32");
}
extern "C" int vm_cast_pad_f32_to_bf16(const float* src, void* dst, int rows, int cols, int64_t ld_dst, void* stream) {
    VM_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_dst >= cols, "vm_cast_pad_f32_to_bf16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 6.0 * rows * (double)cols, s);
    hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, s, src, (bf16_t*)dst, rows, cols, ld_dst);
    return vm_check_launch("vm_cast_pad_f32_to_bf16");
}

// ------------------------------------------------------------------ add
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o, int64_t n) {
    const int64_t nvec = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float x[8], y[8];
        unpack8(*reinterpret_cast<const uint4*>(a + i * 8), x);
        unpack8(*reinterpret_cast<const uint4*>(b + i * 8), y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        *reinterpret_cast<uint4*>(o + i * 8) = pack8(x);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = (nvec << 3) + threadIdx.x;
        o[i] = f32_to_bf16(bf16_to_f32(a[i]) + bf16_to_f32(b[i]));
    }
}
extern "C" int vm_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    VM_REQUIRE(a && b && out && n >= 0, "vm_add_bf16: bad arguments");
    if (n == 0) return VM_OK;
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 6.0 * n, s);
    hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for((n + 7) / 8)), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
    return vm_check_launch("vm_add_bf16");
}

// ------------------------------------------------------------------ column sum (bias gradients):  out[c] += sum_r x[r,c]
// block = 32 column-chunks (256 columns) x 8 row lanes; grid.y splits the rows; one atomicAdd per column per block
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ out, int rows, int cols, const float* __restrict__ scale_dev) {
    __shared__ float part[8][257];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + tx * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < cols) {
        const bool full = c0 + 8 <= cols;
        const int step = gridDim.y * 8;
        int r = blockIdx.y * 8 + ty;
        if (full) {
            for (; r + 3 * step < rows; r += 4 * step) {        // four independent 16-B loads in flight per thread
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(x + (int64_t)(r + u * step) * ldx + c0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[8];
                    unpack8(v[u], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += f[j];
                }
            }
        }
        for (; r < rows; r += step) {
            const bf16_t* p = x + (int64_t)r * ldx + c0;
            if (full) {
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(p), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += f[j];
            } else {
                for (int j = 0; j < cols - c0; ++j) acc[j] += bf16_to_f32(p[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[ty][tx * 8 + j] = acc[j];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
        atomicAdd(out + c, scale_dev ? t * (*scale_dev) : t);
    }
}
extern "C" int vm_colsum_bf16(const void* x, int64_t ldx, float* out, int rows, int cols, const float* scale_dev, void* stream) {
    VM_REQUIRE(x && out && rows > 0 && cols > 0 && (ldx % 8) == 0, "vm_colsum_bf16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 2.0 * rows * (double)cols, s);
    int gy = (rows + 63) / 64; if (gy > 128) gy = 128; if (gy < 1) gy = 1;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256, gy), dim3(256), 0, s, (const bf16_t*)x, ldx, out, rows, cols, scale_dev);
    return vm_check_launch("vm_colsum_bf16");
}

// ------------------------------------------------------------------ feature mask: mask[r] = sum_c |x[r,c]| != 0
// ref:vilmedic/blocks/vision/visual_encoder.py:138
__global__ __launch_bounds__(256) void feature_mask_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ mask, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int nch = cols >> 3;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        float s = 0.f;
        for (int ch = lane; ch < nch; ch += 64) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(x + (int64_t)row * cols + ch * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += fabsf(f[j]);
        }
        s = wave_sum(s);
        if (lane == 0) mask[row] = s != 0.f ? 1 : 0;
    }
}
extern "C" int vm_feature_mask(const void* feats, uint8_t* mask, int rows, int cols, void* stream) {
    VM_REQUIRE(feats && mask && rows > 0 && cols > 0 && (cols % 8) == 0, "vm_feature_mask: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 2.0 * rows * (double)cols, s);
    hipLaunchKernelGGL(feature_mask_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, s, (const bf16_t*)feats, mask, rows, cols);
    return vm_check_launch("vm_feature_mask");
}

// ------------------------------------------------------------------ embeddings
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                            const float* __restrict__ pos, bf16_t* __restrict__ out,
                                                            int rows, int L, int D, int past_len) {
    const int lane = threadIdx.x & 63;
    const int nch = D >> 3;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const int64_t id = ids[row];
        const int t = row % L + past_len;
        const float* w = word + id * D;
        const float* pp = pos + (int64_t)t * D;
        for (int ch = lane; ch < nch; ch += 64) {
            const float4 a0 = *reinterpret_cast<const float4*>(w + ch * 8), a1 = *reinterpret_cast<const float4*>(w + ch * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(pp + ch * 8), b1 = *reinterpret_cast<const float4*>(pp + ch * 8 + 4);
            const float f[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w, a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
            *reinterpret_cast<uint4*>(out + (int64_t)row * D + ch * 8) = pack8(f);
        }
    }
}
// backward of word + position embedding.  One block per (position t, column half): it walks the B rows of that position,
//   d_pos[t]      += sum_b d_out[b, t]       in registers -- every (t, column) has exactly one owner thread: no atomics, deterministic
//                                            (the previous kernel issued B-way contended atomics per element: 64 adds on each address);
//   d_word[id[b,t]] += d_out[b, t]           hardware fp32 atomics, one wave instruction = 64 lanes x 2 consecutive columns (512 B
//                                            contiguous), contended only where a token id repeats.
// The token id is block-uniform (scalar load).  Rows are read once, 4 in flight per thread.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ d_out,
                                                            float* __restrict__ d_word, float* __restrict__ d_pos,
                                                            int B, int L, int D, int padding_idx) {
    const int t = blockIdx.x;
    const int half = D >> 1;                                   // column PAIRS per row
    const int per = (half + gridDim.y - 1) / gridDim.y;
    const int p_lo = blockIdx.y * per, p_hi = min(half, p_lo + per);
    for (int cp = p_lo + threadIdx.x; cp < p_hi; cp += 256) {
        float a0 = 0.f, a1 = 0.f;
        int b = 0;
        for (; b + 4 <= B; b += 4) {
            uint32_t v[4]; int64_t id[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t row = (int64_t)(b + u) * L + t;
                id[u] = ids[row];
                v[u] = *reinterpret_cast<const uint32_t*>(d_out + row * D + 2 * cp);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float lo = __uint_as_float(v[u] << 16), hi = __uint_as_float(v[u] & 0xffff0000u);
                a0 += lo; a1 += hi;
                if (id[u] != padding_idx) { atomicAdd(d_word + id[u] * D + 2 * cp, lo); atomicAdd(d_word + id[u] * D + 2 * cp + 1, hi); }
            }
        }
        for (; b < B; ++b) {
            const int64_t row = (int64_t)b * L + t;
            const int64_t id = ids[row];
            const uint32_t v = *reinterpret_cast<const uint32_t*>(d_out + row * D + 2 * cp);
            const float lo = __uint_as_float(v << 16), hi = __uint_as_float(v & 0xffff0000u);
            a0 += lo; a1 += hi;
            if (id != padding_idx) { atomicAdd(d_word + id * D + 2 * cp, lo); atomicAdd(d_word + id * D + 2 * cp + 1, hi); }
        }
        float2* dp = reinterpret_cast<float2*>(d_pos + (int64_t)t * D + 2 * cp);
        float2 cur = *dp;
        cur.x += a0; cur.y += a1;
        *dp = cur;
    }
}
extern "C" int vm_embedding_fwd(const int64_t* ids, const float* word, const float* pos, void* out, int B, int L, int D, int past_len, void* stream) {
    VM_REQUIRE(ids && word && pos && out && B > 0 && L > 0 && D > 0 && (D % 8) == 0, "vm_embedding_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 10.0 * B * L * (double)D, s);
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(B * L, 4)), dim3(256), 0, s, ids, word, pos, (bf16_t*)out, B * L, L, D, past_len);
    return vm_check_launch("vm_embedding_fwd");
}
extern "C" int vm_embedding_bwd(const int64_t* ids, const void* d_out, float* d_word, float* d_pos, int B, int L, int D, int padding_idx, void* stream) {
    VM_REQUIRE(ids && d_out && d_word && d_pos && B > 0 && L > 0 && D > 0 && (D % 8) == 0, "vm_embedding_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 10.0 * B * L * (double)D, s);
    const int ysplit = (D / 2 + 255) / 256 > 1 ? 2 : 1;       // D = 768: 2 blocks x 192 column pairs per position
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(L, ysplit), dim3(256), 0, s, ids, (const bf16_t*)d_out, d_word, d_pos, B, L, D, padding_idx);
    return vm_check_launch("vm_embedding_bwd");
}

// ------------------------------------------------------------------ embeddings with explicit position ids and a token-type row
// hf:models/bert/modeling_bert.py BertEmbeddings.forward (token_type_ids default to zeros: one constant row) and
// hf:models/roberta/modeling_roberta.py:55-155 (position ids = cumsum(ids != pad) * (ids != pad) + pad, computed by the caller):
//   out[row] = word[ids[row]] + pos[pos_ids ? pos_ids[row] : row % L + past_len] (+ type_row)
template <typename OUT>
__global__ __launch_bounds__(256) void embedding_fwd_ex_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids,
                                                               const float* __restrict__ word, const float* __restrict__ pos,
                                                               const float* __restrict__ type_row, OUT* __restrict__ out,
                                                               int rows, int L, int D, int past_len) {
    const int lane = threadIdx.x & 63;
    const int nch = D >> 3;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const int64_t id = ids[row];
        const int64_t t = pos_ids ? pos_ids[row] : (int64_t)(row % L + past_len);
        const float* w = word + id * D;
        const float* pp = pos + t * D;
        for (int ch = lane; ch < nch; ch += 64) {
            const float4 a0 = *reinterpret_cast<const float4*>(w + ch * 8), a1 = *reinterpret_cast<const float4*>(w + ch * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(pp + ch * 8), b1 = *reinterpret_cast<const float4*>(pp + ch * 8 + 4);
            // (word + type) + pos: the order of hf's two additions
            float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            if (type_row) {
                const float4 c0 = *reinterpret_cast<const float4*>(type_row + ch * 8), c1 = *reinterpret_cast<const float4*>(type_row + ch * 8 + 4);
                f[0] += c0.x; f[1] += c0.y; f[2] += c0.z; f[3] += c0.w; f[4] += c1.x; f[5] += c1.y; f[6] += c1.z; f[7] += c1.w;
            }
            f[0] += b0.x; f[1] += b0.y; f[2] += b0.z; f[3] += b0.w; f[4] += b1.x; f[5] += b1.y; f[6] += b1.z; f[7] += b1.w;
            if constexpr (sizeof(OUT) == 2) {
                *reinterpret_cast<uint4*>(out + (int64_t)row * D + ch * 8) = pack8(f);
            } else {
                *reinterpret_cast<float4*>(out + (int64_t)row * D + ch * 8) = make_float4(f[0], f[1], f[2], f[3]);
                *reinterpret_cast<float4*>(out + (int64_t)row * D + ch * 8 + 4) = make_float4(f[4], f[5], f[6], f[7]);
            }
        }
    }
}
// backward: one block per (column t of ids, column half) as embedding_bwd_kernel.  Rows whose position id is the regular one of the
// column (t + pos_offset: every non-pad token of a right-padded batch) are summed in registers; the others go to d_pos one atomic at a time.
// Every write to d_pos / d_type is an atomic add (another column's irregular row may target the same element).  Rows with
// id == padding_idx get no word gradient and position pos_padding_idx gets no position gradient (nn.Embedding(padding_idx=...)).
__global__ __launch_bounds__(256) void embedding_bwd_ex_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids,
                                                               const bf16_t* __restrict__ d_out, float* __restrict__ d_word,
                                                               float* __restrict__ d_pos, float* __restrict__ d_type,
                                                               int B, int L, int D, int padding_idx, int pos_offset, int pos_padding_idx) {
    const int t = blockIdx.x;
    const int half = D >> 1;
    const int per = (half + gridDim.y - 1) / gridDim.y;
    const int p_lo = blockIdx.y * per, p_hi = min(half, p_lo + per);
    const int64_t regular = (int64_t)t + pos_offset;
    for (int cp = p_lo + threadIdx.x; cp < p_hi; cp += 256) {
        float a0 = 0.f, a1 = 0.f, s0 = 0.f, s1 = 0.f;
        for (int b = 0; b < B; ++b) {
            const int64_t row = (int64_t)b * L + t;
            const int64_t id = ids[row];
            const int64_t pid = pos_ids ? pos_ids[row] : regular;
            const uint32_t v = *reinterpret_cast<const uint32_t*>(d_out + row * D + 2 * cp);
            const float lo = __uint_as_float(v << 16), hi = __uint_as_float(v & 0xffff0000u);
            s0 += lo; s1 += hi;
            if (pid == regular) { a0 += lo; a1 += hi; }
            else if (pid != pos_padding_idx) { atomicAdd(d_pos + pid * D + 2 * cp, lo); atomicAdd(d_pos + pid * D + 2 * cp + 1, hi); }
            if (id != padding_idx) { atomicAdd(d_word + id * D + 2 * cp, lo); atomicAdd(d_word + id * D + 2 * cp + 1, hi); }
        }
        if (regular != pos_padding_idx) { atomicAdd(d_pos + regular * D + 2 * cp, a0); atomicAdd(d_pos + regular * D + 2 * cp + 1, a1); }
        if (d_type) { atomicAdd(d_type + 2 * cp, s0); atomicAdd(d_type + 2 * cp + 1, s1); }
    }
}
extern "C" int vm_embedding_fwd_ex(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos, const float* type_row,
                                   void* out, int out_dtype, int B, int L, int D, int past_len, void* stream) {
    VM_REQUIRE(ids && word && pos && out && B > 0 && L > 0 && D > 0 && (D % 8) == 0 && (out_dtype == VM_BF16 || out_dtype == VM_F32),
               "vm_embedding_fwd_ex: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 10.0 * B * L * (double)D, s);
    if (out_dtype == VM_BF16)
        hipLaunchKernelGGL(embedding_fwd_ex_kernel<bf16_t>, dim3(grid_for(B * L, 4)), dim3(256), 0, s, ids, pos_ids, word, pos, type_row, (bf16_t*)out, B * L, L, D, past_len);
    else
        hipLaunchKernelGGL(embedding_fwd_ex_kernel<float>, dim3(grid_for(B * L, 4)), dim3(256), 0, s, ids, pos_ids, word, pos, type_row, (float*)out, B * L, L, D, past_len);
    return vm_check_launch("vm_embedding_fwd_ex");
}
extern "C" int vm_embedding_bwd_ex(const int64_t* ids, const int64_t* pos_ids, const void* d_out, float* d_word, float* d_pos, float* d_type,
                                   int B, int L, int D, int padding_idx, int pos_offset, int pos_padding_idx, void* stream) {
    VM_REQUIRE(ids && d_out && d_word && d_pos && B > 0 && L > 0 && D > 0 && (D % 8) == 0, "vm_embedding_bwd_ex: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 10.0 * B * L * (double)D, s);
    const int ysplit = (D / 2 + 255) / 256 > 1 ? 2 : 1;
    hipLaunchKernelGGL(embedding_bwd_ex_kernel, dim3(L, ysplit), dim3(256), 0, s, ids, pos_ids, (const bf16_t*)d_out, d_word, d_pos, d_type, B, L, D,
                       padding_idx, pos_offset, pos_padding_idx);
    return vm_check_launch("vm_embedding_bwd_ex");
}

// ------------------------------------------------------------------ erf-GELU backward as an elementwise pass: dz = dy * gelu'(z)
// (the LM-head transform of BERT / RoBERTa: dense -> GELU -> LayerNorm, hf:models/roberta/modeling_roberta.py RobertaLMHead; inside the
// MLP block the same product rides in the FC2 dgrad epilogue instead, vm_gemm_epilogue.mul_gelu_z)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ z, bf16_t* __restrict__ dz, int64_t n) {
    const int64_t nvec = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float a[8], b[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + i * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(z + i * 8), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] *= gelu_grad_f(b[j]);
        *reinterpret_cast<uint4*>(dz + i * 8) = pack8(a);
    }
}
extern "C" int vm_gelu_bwd_bf16(const void* dy, const void* z, void* dz, int64_t n, void* stream) {
    VM_REQUIRE(dy && z && dz && n > 0 && (n % 8) == 0, "vm_gelu_bwd_bf16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 6.0 * n, s);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)z, (bf16_t*)dz, n);
    return vm_check_launch("vm_gelu_bwd_bf16");
}

// ------------------------------------------------------------------ ViT patches
// out[(b*gh+py)*gw+px][c*p*p + ph*p + pw] = images[b][c][py*p+ph][px*p+pw]   (p % 8 == 0)
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int C, int H, int W, int p) {
    const int gh = H / p, gw = W / p, kdim = C * p * p, cpr = p >> 3;   // chunks per patch row
    const int64_t total = (int64_t)B * gh * gw * (kdim >> 3);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int chunk = (int)(i % (kdim >> 3));
        const int64_t patch = i / (kdim >> 3);
        const int px = (int)(patch % gw), py = (int)((patch / gw) % gh), b = (int)(patch / ((int64_t)gw * gh));
        const int pwc = chunk % cpr, ph = (chunk / cpr) % p, c = chunk / (cpr * p);
        const float* src = img + (((int64_t)b * C + c) * H + py * p + ph) * W + px * p + pwc * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), bq = *reinterpret_cast<const float4*>(src + 4);
        const float f[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
        *reinterpret_cast<uint4*>(out + patch * kdim + (int64_t)chunk * 8) = pack8(f);
    }
}
extern "C" int vm_im2col_patches(const float* images, void* out, int B, int C, int H, int W, int p, void* stream) {
    VM_REQUIRE(images && out && B > 0 && C > 0 && p > 0 && (p % 8) == 0 && (H % p) == 0 && (W % p) == 0, "vm_im2col_patches: patch size must be a multiple of 8 dividing H and W");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 6.0 * B * C * (double)H * W, s);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((int64_t)B * C * H * W / 8)), dim3(256), 0, s, images, (bf16_t*)out, B, C, H, W, p);
    return vm_check_launch("vm_im2col_patches");
}

// ns special tokens in front of the n patches (ViT: [CLS]; DeiT: [CLS], distillation -- hf:models/deit/modeling_deit.py DeiTEmbeddings.forward)
__global__ __launch_bounds__(256) void vit_assemble_kernel(const bf16_t* __restrict__ patches, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, bf16_t* __restrict__ out, int B, int n, int ns, int D) {
    const int nch = D >> 3;
    const int64_t total = (int64_t)B * (n + ns) * nch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const int64_t row = i / nch;
        const int t = (int)(row % (n + ns)), b = (int)(row / (n + ns));
        float f[8];
        if (t < ns) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = cls[(int64_t)t * D + ch * 8 + j];
        } else {
            unpack8(*reinterpret_cast<const uint4*>(patches + ((int64_t)b * n + t - ns) * D + ch * 8), f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += pos[(int64_t)t * D + ch * 8 + j];
        *reinterpret_cast<uint4*>(out + row * D + ch * 8) = pack8(f);
    }
}
// d_patches = d_out[:,ns:];  d_pos[t] += sum_b d_out[b,t];  d_cls[t] += sum_b d_out[b,t] for t < ns
__global__ __launch_bounds__(256) void vit_assemble_bwd_kernel(const bf16_t* __restrict__ d_out, bf16_t* __restrict__ d_patches,
                                                               float* __restrict__ d_cls, float* __restrict__ d_pos, int B, int n, int ns, int D) {
    const int nch = D >> 3;
    const int64_t total = (int64_t)(n + ns) * nch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % nch), t = (int)(i / nch);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < B; ++b) {
            const uint4 raw = *reinterpret_cast<const uint4*>(d_out + ((int64_t)b * (n + ns) + t) * D + ch * 8);
            if (t >= ns) *reinterpret_cast<uint4*>(d_patches + ((int64_t)b * n + t - ns) * D + ch * 8) = raw;
            float f[8];
            unpack8(raw, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            d_pos[(int64_t)t * D + ch * 8 + j] += acc[j];
            if (t < ns) d_cls[(int64_t)t * D + ch * 8 + j] += acc[j];
        }
    }
}
extern "C" int vm_vit_assemble_ex(const void* patches, const float* special, const float* pos, void* out, int B, int n, int ns, int D, void* stream) {
    VM_REQUIRE(patches && special && pos && out && B > 0 && n > 0 && ns >= 1 && ns <= 4 && D > 0 && (D % 8) == 0, "vm_vit_assemble_ex: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 4.0 * B * (n + ns) * (double)D, s);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for((int64_t)B * (n + ns) * D / 8)), dim3(256), 0, s, (const bf16_t*)patches, special, pos, (bf16_t*)out, B, n, ns, D);
    return vm_check_launch("vm_vit_assemble");
}
extern "C" int vm_vit_assemble_bwd_ex(const void* d_out, void* d_patches, float* d_special, float* d_pos, int B, int n, int ns, int D, void* stream) {
    VM_REQUIRE(d_out && d_patches && d_special && d_pos && B > 0 && n > 0 && ns >= 1 && ns <= 4 && D > 0 && (D % 8) == 0, "vm_vit_assemble_bwd_ex: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 4.0 * B * (n + ns) * (double)D, s);
    hipLaunchKernelGGL(vit_assemble_bwd_kernel, dim3(grid_for((int64_t)(n + ns) * D / 8)), dim3(256), 0, s, (const bf16_t*)d_out, (bf16_t*)d_patches, d_special, d_pos, B, n, ns, D);
    return vm_check_launch("vm_vit_assemble_bwd");
}
extern "C" int vm_vit_assemble(const void* patches, const float* cls, const float* pos, void* out, int B, int n, int D, void* stream) {
    return vm_vit_assemble_ex(patches, cls, pos, out, B, n, 1, D, stream);
}
extern "C" int vm_vit_assemble_bwd(const void* d_out, void* d_patches, float* d_cls, float* d_pos, int B, int n, int D, void* stream) {
    return vm_vit_assemble_bwd_ex(d_out, d_patches, d_cls, d_pos, B, n, 1, D, stream);
}

// ------------------------------------------------------------------ dropout mask re-application (backward of the
// GEMM-epilogue dropout): out[r,c] = keep(seed, r*cols+c) ? x[r,c]/(1-p) : 0   -- same counter-based mask as gemm.hip
__global__ __launch_bounds__(256) void dropout_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int64_t n,
                                                            uint64_t seed0, const uint64_t* __restrict__ seed_dev, uint32_t thresh, float scale) {
    const int64_t nvec = n >> 3;
    const uint64_t seed = eff_seed(seed0, seed_dev);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + i * 8), f);
        bool keep[8];
        dropout_keep_n<8>(drop_key(seed), (uint64_t)i * 8, thresh, keep);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = keep[j] ? f[j] * scale : 0.f;
        *reinterpret_cast<uint4*>(out + i * 8) = pack8(f);
    }
}
extern "C" int vm_dropout_apply_bf16(const void* x, void* out, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, void* stream) {
    VM_REQUIRE(x && out && n > 0 && (n % 8) == 0 && p >= 0.f && p < 1.f, "vm_dropout_apply_bf16: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_ELT, 4.0 * n, s);
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, n, seed, seed_dev, dropout_thresh16(p), 1.0f / (1.0f - p));
    return vm_check_launch("vm_dropout_apply_bf16");
}
