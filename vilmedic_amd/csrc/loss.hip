// loss.hip -- fused softmax cross-entropy (forward + gradient in one pass over the logits), label-smoothing CE,
// and the fp32 log-softmax / argmax helpers of the decode step.  HBM-bound: one workgroup per row.
#include "common.h"

__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sh[i]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------ shifted causal-LM CE on bf16 logits
// row = (b,t); label = ids[b,t+1]; rows t == L-1 carry no loss.  One workgroup per row; dlogits may alias logits.
// RESIDENT (ldl <= 32768): the row is kept in registers (<= CE_NCH 16-B chunks per thread), so the logits are read from HBM once;
// otherwise the three passes (max, sum of exponentials, gradient) stream the row from memory again (a 60-130 KB row stays in L2).
// The kernel is VALU-bound before it is HBM-bound (119 elements per lane, three passes), so the per-element work of the common case
// -- no banned columns, no top-k threshold, chunk entirely below V -- is 1 unpack + 1 max / 1 unpack + 1 fma + v_exp + 1 add /
// 1 unpack + 1 fma + v_exp + 1 mul + pack; the label column is handled once per row by the thread that owns its chunk, not by a compare
// per element.  Chunks that touch V, banned columns or a threshold take the general path.
#define CE_NCH 16   // 256 threads * 16 chunks * 8 = 32768 columns resident
#define CE_LOG2E 1.4426950408889634f
struct CeBanned { int n; int col[4]; };
struct CeRow {
    int V; CeBanned ban; float thr; bool plain_row;        // plain_row: no banned column, no threshold (block-uniform)
    __device__ __forceinline__ bool live(int c, float f) const {
        return c < V && f >= thr && !(ban.n > 0 && (c == ban.col[0] || (ban.n > 1 && c == ban.col[1]) || (ban.n > 2 && c == ban.col[2]) || (ban.n > 3 && c == ban.col[3])));
    }
    __device__ __forceinline__ bool plain(int ch) const { return plain_row && ch * 8 + 8 <= V; }
    __device__ __forceinline__ float chunk_max(const uint4 raw, int ch, float mx) const {
        float f[8];
        unpack8(raw, f);
        if (plain(ch)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, f[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (live(ch * 8 + j, f[j])) mx = fmaxf(mx, f[j]);
        }
        return mx;
    }
    // nm = -max * log2(e): exp(f - max) = exp2(f * log2(e) + nm)
    __device__ __forceinline__ float chunk_sumexp(const uint4 raw, int ch, float nm, float se) const {
        float f[8];
        unpack8(raw, f);
        if (plain(ch)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) se += __builtin_amdgcn_exp2f(fmaf(f[j], CE_LOG2E, nm));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (live(ch * 8 + j, f[j])) se += __builtin_amdgcn_exp2f(fmaf(f[j], CE_LOG2E, nm));
        }
        return se;
    }
    __device__ __forceinline__ uint4 chunk_grad(const uint4 raw, int ch, float nm, float inv, int label, float gs) const {
        float f[8];
        unpack8(raw, f);
        if (plain(ch)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = __builtin_amdgcn_exp2f(fmaf(f[j], CE_LOG2E, nm)) * inv;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = live(ch * 8 + j, f[j]) ? __builtin_amdgcn_exp2f(fmaf(f[j], CE_LOG2E, nm)) * inv : 0.f;
        }
        if (ch == (label >> 3)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j == (label & 7)) f[j] -= gs;
        }
        return pack8(f);
    }
};

template <bool RESIDENT>
// logits and dlogits may be the SAME buffer (the training path lets the gradient overwrite the logits), so neither is __restrict__ and
// every value the row statistics need -- including the label's logit -- is read before the first barrier, i.e. before any wave can store.
__global__ __launch_bounds__(256) void ce_shift_kernel(const bf16_t* logits, int64_t ldl, const int64_t* __restrict__ ids,
                                                       int L, int V, float* __restrict__ loss_sum, float* __restrict__ row_logp,
                                                       bf16_t* dlogits, float grad_scale,
                                                       const float* __restrict__ row_weight, CeBanned ban,
                                                       const float* __restrict__ row_min_logit) {
    __shared__ float sh[4];
    const int row = blockIdx.x, t = row % L, b = row / L;
    const int nch = (int)(ldl >> 3);
    bf16_t* drow = dlogits ? dlogits + (int64_t)row * ldl : nullptr;
    const float w = row_weight ? row_weight[row] : 1.0f;
    if (t == L - 1 || w == 0.0f) {      // no loss from this row (last position / masked-out token)
        if (drow) for (int ch = threadIdx.x; ch < nch; ch += 256) *reinterpret_cast<uint4*>(drow + ch * 8) = make_uint4(0, 0, 0, 0);
        if (row_logp && threadIdx.x == 0) row_logp[row] = 0.f;
        return;
    }
    const bf16_t* lrow = logits + (int64_t)row * ldl;
    const int label = (int)ids[(int64_t)b * L + t + 1];
    CeRow R;
    R.V = V; R.ban = ban; R.thr = row_min_logit ? row_min_logit[row] : -INFINITY;
    R.plain_row = ban.n == 0 && !row_min_logit;
    // the label's logit, read by every thread from the row BEFORE the block reductions: their barriers order this load ahead of every
    // gradient store of the (possibly aliased) row.  A banned or thresholded-out label contributes lab = 0 like the per-element form did
    // (reference: its log-probability is -inf there and SCST never scores such a token)
    const float labv = bf16_to_f32(lrow[label]);
    const float lab = R.live(label, labv) ? labv : 0.f;
    uint4 raw[RESIDENT ? CE_NCH : 1];
    float mx = -INFINITY;
    if (RESIDENT) {
#pragma unroll
        for (int i = 0; i < CE_NCH; ++i) {
            const int ch = threadIdx.x + 256 * i;
            raw[i] = make_uint4(0, 0, 0, 0);
            if (ch < nch) raw[i] = *reinterpret_cast<const uint4*>(lrow + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < CE_NCH; ++i) {
            const int ch = threadIdx.x + 256 * i;
            if (ch < nch) mx = R.chunk_max(raw[i], ch, mx);
        }
    } else {
        for (int ch = threadIdx.x; ch < nch; ch += 256) mx = R.chunk_max(*reinterpret_cast<const uint4*>(lrow + ch * 8), ch, mx);
    }
    mx = block_reduce_max(mx, sh);
    const float nm = -mx * CE_LOG2E;
    float se = 0.f;
    if (RESIDENT) {
#pragma unroll
        for (int i = 0; i < CE_NCH; ++i) {
            const int ch = threadIdx.x + 256 * i;
            if (ch < nch) se = R.chunk_sumexp(raw[i], ch, nm, se);
        }
    } else {
        for (int ch = threadIdx.x; ch < nch; ch += 256) se = R.chunk_sumexp(*reinterpret_cast<const uint4*>(lrow + ch * 8), ch, nm, se);
    }
    se = block_reduce_sum(se, sh);
    const float lse = mx + __logf(se);
    if (threadIdx.x == 0) {
        atomicAdd(loss_sum, w * (lse - lab));
        if (row_logp) row_logp[row] = lab - lse;
    }
    if (drow) {
        const float gs = grad_scale * w;
        const float inv = gs / se;
        if (RESIDENT) {
#pragma unroll
            for (int i = 0; i < CE_NCH; ++i) {
                const int ch = threadIdx.x + 256 * i;
                if (ch < nch) *reinterpret_cast<uint4*>(drow + ch * 8) = R.chunk_grad(raw[i], ch, nm, inv, label, gs);
            }
        } else {
            for (int ch = threadIdx.x; ch < nch; ch += 256)
                *reinterpret_cast<uint4*>(drow + ch * 8) = R.chunk_grad(*reinterpret_cast<const uint4*>(lrow + ch * 8), ch, nm, inv, label, gs);
        }
    }
}

extern "C" int vm_ce_shift_fwd_bwd(const void* logits, int64_t ldl, const int64_t* ids, int B, int L, int V,
                                   float* loss_sum, float* row_logp, void* dlogits, float grad_scale,
                                   const float* row_weight, const int32_t* banned, int n_banned, const float* row_min_logit,
                                   void* stream) {
    VM_REQUIRE(logits && ids && loss_sum, "vm_ce_shift_fwd_bwd: null pointer");
    VM_REQUIRE(B > 0 && L > 1 && V > 0 && ldl >= V && (ldl % 8) == 0, "vm_ce_shift_fwd_bwd: bad shape");
    VM_REQUIRE(n_banned >= 0 && n_banned <= 4 && (n_banned == 0 || banned), "vm_ce_shift_fwd_bwd: at most 4 banned columns (HOST array)");
    CeBanned ban = {n_banned, {0, 0, 0, 0}};
    for (int i = 0; i < n_banned; ++i) ban.col[i] = banned[i];
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 4.0 * B * L * (double)ldl, s);
    if (ldl <= 256 * CE_NCH * 8)
        hipLaunchKernelGGL(ce_shift_kernel<true>, dim3(B * L), dim3(256), 0, s, (const bf16_t*)logits, ldl, ids, L, V, loss_sum, row_logp, (bf16_t*)dlogits,
                           grad_scale, row_weight, ban, row_min_logit);
    else        // wider vocabularies (> 32768 columns): the passes stream the row instead of holding it in registers
        hipLaunchKernelGGL(ce_shift_kernel<false>, dim3(B * L), dim3(256), 0, s, (const bf16_t*)logits, ldl, ids, L, V, loss_sum, row_logp, (bf16_t*)dlogits,
                           grad_scale, row_weight, ban, row_min_logit);
    return vm_check_launch("vm_ce_shift_fwd_bwd");
}

// ------------------------------------------------------------------ top-k threshold of bf16 logits rows
// thr[row] = k-th largest live logit of the row (HF TopKLogitsWarper keeps scores >= that value; bad-word columns are removed first,
// hf:generation/logits_process.py NoBadWordsLogitsProcessor -> TopKLogitsWarper, ref:vilmedic/blocks/rl/SCST.py:142-157).  Exact: a bf16
// has 65536 possible values, so a two-pass radix select over its order-preserving 16-bit key (256-bin histograms in LDS: high byte,
// then low byte inside the selected bin) finds the k-th value itself -- no fp32 copy of the [rows, V] logits, no sort.
__device__ __forceinline__ uint32_t bf16_order_key(bf16_t h) { return (h & 0x8000u) ? (uint32_t)(~h & 0xffffu) : (uint32_t)(h | 0x8000u); }
__global__ __launch_bounds__(256) void topk_threshold_kernel(const bf16_t* __restrict__ logits, int64_t ldl, int V, int k, CeBanned ban,
                                                             float* __restrict__ thr) {
    __shared__ uint32_t hist[256];
    __shared__ int sel[3];       // selected bin, remaining rank inside it, 1 = the row has fewer than k live columns
    const int row = blockIdx.x;
    const bf16_t* lrow = logits + (int64_t)row * ldl;
    auto live = [&](int c) { return c < V && !(ban.n > 0 && (c == ban.col[0] || (ban.n > 1 && c == ban.col[1]) || (ban.n > 2 && c == ban.col[2]) || (ban.n > 3 && c == ban.col[3]))); };
    const int nch = (V + 7) >> 3;
    int want = k;
    for (int pass = 0; pass < 2; ++pass) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t hi_sel = pass ? (uint32_t)sel[0] : 0u;
        for (int ch = threadIdx.x; ch < nch; ch += 256) {
            const uint4 raw = *reinterpret_cast<const uint4*>(lrow + ch * 8);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16_t h = (bf16_t)((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
                if (!live(ch * 8 + j) || (h & 0x7f80u) == 0x7f80u && (h & 0x7fu)) continue;      // dead column / NaN
                const uint32_t key = bf16_order_key(h);
                if (pass == 0) atomicAdd(&hist[key >> 8], 1u);
                else if ((key >> 8) == hi_sel) atomicAdd(&hist[key & 0xffu], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0, b = 255;
            for (; b > 0; --b) { if (acc + (int)hist[b] >= want) break; acc += (int)hist[b]; }
            sel[0] = b; sel[1] = want - acc;
            if (pass == 0) sel[2] = (b == 0 && acc + (int)hist[0] < want) ? 1 : 0;     // fewer than k live columns: keep everything (HF clamps top_k to the row)
        }
        __syncthreads();
        if (pass == 0) want = sel[1];
        else if (threadIdx.x == 0) {
            const uint32_t key = (hi_sel << 8) | (uint32_t)sel[0];
            const bf16_t h = (key & 0x8000u) ? (bf16_t)(key & 0x7fffu) : (bf16_t)(~key & 0xffffu);
            thr[row] = sel[2] ? -INFINITY : bf16_to_f32(h);
        }
        __syncthreads();
    }
}

extern "C" int vm_topk_threshold_bf16(const void* logits, int64_t ldl, int rows, int V, int k, const int32_t* banned, int n_banned,
                                      float* thr, void* stream) {
    VM_REQUIRE(logits && thr && rows > 0 && V > 0 && k > 0 && ldl >= V && (ldl % 8) == 0, "vm_topk_threshold_bf16: bad arguments");
    VM_REQUIRE(n_banned >= 0 && n_banned <= 4 && (n_banned == 0 || banned), "vm_topk_threshold_bf16: at most 4 banned columns (HOST array)");
    CeBanned ban = {n_banned, {0, 0, 0, 0}};
    for (int i = 0; i < n_banned; ++i) ban.col[i] = banned[i];
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 4.0 * rows * (double)ldl, s);
    hipLaunchKernelGGL(topk_threshold_kernel, dim3(rows), dim3(256), 0, s, (const bf16_t*)logits, ldl, V, k, ban, thr);
    return vm_check_launch("vm_topk_threshold_bf16");
}

// ------------------------------------------------------------------ label-smoothing CE on small fp32 logits [R,C]
// loss_row = eps/C * sum_c(-logp_c) + (1-eps) * (-logp_target)       (reduction 'mean' is applied by the caller)
__global__ __launch_bounds__(64) void ce_smooth_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int C,
                                                       float smoothing, float* __restrict__ loss_sum, float* __restrict__ dlogits, float grad_scale) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const float* lr = logits + (int64_t)row * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lr[c]);
    mx = wave_max(mx);
    float se = 0.f, sl = 0.f;
    for (int c = lane; c < C; c += 64) { se += __expf(lr[c] - mx); sl += lr[c]; }
    se = wave_sum(se); sl = wave_sum(sl);
    const float lse = mx + __logf(se);
    const int tg = (int)target[row];
    if (lane == 0) {
        const float sum_neg_logp = (float)C * lse - sl;
        atomicAdd(loss_sum, smoothing / (float)C * sum_neg_logp + (1.f - smoothing) * (lse - lr[tg]));
    }
    if (dlogits) {
        for (int c = lane; c < C; c += 64) {
            const float pr = __expf(lr[c] - lse);
            float g = pr - smoothing / (float)C - (c == tg ? (1.f - smoothing) : 0.f);
            dlogits[(int64_t)row * C + c] = g * grad_scale;
        }
    }
}
extern "C" int vm_ce_smooth_fwd_bwd(const float* logits, const int64_t* target, int R, int C, float smoothing,
                                    float* loss_sum, float* dlogits, float grad_scale, void* stream) {
    VM_REQUIRE(logits && target && loss_sum && R > 0 && C > 0, "vm_ce_smooth_fwd_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_LOSS, 8.0 * R * (double)C, s);
    hipLaunchKernelGGL(ce_smooth_kernel, dim3(R), dim3(64), 0, s, logits, target, C, smoothing, loss_sum, dlogits, grad_scale);
    return vm_check_launch("vm_ce_smooth_fwd_bwd");
}

// ------------------------------------------------------------------ decode helpers (fp32)
// thread tid visits columns tid, tid + 256, ... IN THAT ORDER (the sums below depend on it), eight loads requested before the first is
// used: a vocabulary row is read by one workgroup, and with one 4-byte load per round trip these kernels were pure latency
template <class F>
__device__ __forceinline__ void row_strided8(const float* __restrict__ x, int V, int tid, F&& fn) {
    int c = tid;
    for (; c + 7 * 256 < V; c += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[c + 256 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) fn(c + 256 * u, v[u]);
    }
    for (; c < V; c += 256) fn(c, x[c]);
}

__global__ __launch_bounds__(256) void logsoftmax_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int V) {
    __shared__ float sh[4];
    const float* r = x + (int64_t)blockIdx.x * ldx;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    row_strided8(r, V, tid, [&](int, float v) { mx = fmaxf(mx, v); });
    mx = block_reduce_max(mx, sh);
    float se = 0.f;
    row_strided8(r, V, tid, [&](int, float v) { se += expf(v - mx); });
    se = block_reduce_sum(se, sh);
    const float lse = mx + logf(se);
    float* o = out + (int64_t)blockIdx.x * V;
    row_strided8(r, V, tid, [&](int c, float v) { o[c] = v - lse; });
}
extern "C" int vm_logsoftmax_f32(const float* logits, int64_t ldl, float* out, int rows, int V, void* stream) {
    VM_REQUIRE(logits && out && rows > 0 && V > 0 && ldl >= V, "vm_logsoftmax_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_DECODE, 12.0 * rows * (double)V, s);
    hipLaunchKernelGGL(logsoftmax_kernel, dim3(rows), dim3(256), 0, s, logits, ldl, out, V);
    return vm_check_launch("vm_logsoftmax_f32");
}

// ------------------------------------------------------------------ beam search: log-softmax + running score + top-k in one launch
// hf:generation/utils.py _beam_search (:3208-3520): log_probs = log_softmax(logits.float()) + running_beam_scores[:, :, None], flattened
// over (beam, token), then topk(2 * num_beams).  Round 2 ran it as vm_logsoftmax_f32 (a second [rows, V] fp32 matrix written and re-read)
// + torch's multi-block radix top-k + gather (~300 us per step at 64 x 4 beams); here one workgroup per ROW reads its logits four times from
// L2 and leaves the row's best 2 x num_beams in a workspace, and one small workgroup per sample merges its rows' candidates (one
// workgroup per SAMPLE walking all its rows measured 110 us at 64 x 4 rows: 64 workgroups of dependent round trips).  The arithmetic is vm_logsoftmax_f32's, bit for bit (same strides, same
// reduction helpers): lse = max + log(sum exp(x - max)), value = (x - lse) + score.  The k best of the nb x V values are found exactly
// through the k-th largest of the 256 per-thread maxima (a lower bound L of the k-th value: k values >= L exist) and a candidate list in
// LDS (csrc/decode_select.hip uses the same idea); output sorted by value, ties by flat index (beam * V + token) ascending.
#define BT_CAND 1024
#define BT_MAXB 8
// one workgroup per (sample, beam) ROW: lse of the row, then the row's `keep` best values (x - lse) + score -- the sample's `keep` best all
// rank below `keep` inside their own row -- written to ws[row][rank] as (value, flat index beam * V + token); padded with (-inf, INT_MAX)
__global__ __launch_bounds__(256) void beam_rows_kernel(const float* __restrict__ logits, int64_t ldl, int nb, int V, const float* __restrict__ scores,
                                                        int keep, float* __restrict__ ws_val, int* __restrict__ ws_idx) {
    __shared__ float sh[4];
    __shared__ float s_max[256];
    __shared__ float c_val[BT_CAND];
    __shared__ int c_idx[BT_CAND];
    __shared__ int s_n;
    __shared__ float s_thr;
    const int row = blockIdx.x, r = row % nb, tid = threadIdx.x;
    const float* x = logits + (int64_t)row * ldl;
    float mx = -INFINITY;
    row_strided8(x, V, tid, [&](int, float v) { mx = fmaxf(mx, v); });
    mx = block_reduce_max(mx, sh);
    float se = 0.f;
    row_strided8(x, V, tid, [&](int, float v) { se += expf(v - mx); });
    se = block_reduce_sum(se, sh);
    const float lse = mx + logf(se), sc = scores[row];
    const int kr = min(keep, V);
    float best = -INFINITY;
    row_strided8(x, V, tid, [&](int, float v) { best = fmaxf(best, (v - lse) + sc); });
    s_max[tid] = best;
    if (tid == 0) s_n = 0;
    for (int a = tid; a < keep; a += 256) { ws_val[(int64_t)row * keep + a] = -INFINITY; ws_idx[(int64_t)row * keep + a] = 0x7fffffff; }
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < 256; ++j) { const float o = s_max[j]; rank += (o > best || (o == best && j < tid)) ? 1 : 0; }
    if (rank == min(kr, 256) - 1) s_thr = best;
    __syncthreads();
    const float L = s_thr;
    row_strided8(x, V, tid, [&](int c, float xv) {
        const float v = (xv - lse) + sc;
        if (v >= L) {
            const int at = atomicAdd(&s_n, 1);
            if (at < BT_CAND) { c_val[at] = v; c_idx[at] = r * V + c; }
        }
    });
    __syncthreads();
    const int n = min(s_n, BT_CAND);
    for (int a = tid; a < n; a += 256) {
        const float v = c_val[a]; const int ia = c_idx[a];
        int above = 0;
        for (int j = 0; j < n; ++j) { const float o = c_val[j]; above += (o > v || (o == v && c_idx[j] < ia)) ? 1 : 0; }
        if (above < kr) { ws_val[(int64_t)row * keep + above] = v; ws_idx[(int64_t)row * keep + above] = ia; }
    }
}
// one workgroup per sample: rank of every row candidate among the sample's nb * keep (value descending, flat index ascending)
__global__ __launch_bounds__(256) void beam_merge_kernel(const float* __restrict__ ws_val, const int* __restrict__ ws_idx, int nb, int keep,
                                                         float* __restrict__ out_val, int64_t* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) char bm_smem[];
    const int b = blockIdx.x, tid = threadIdx.x, n = nb * keep;
    float* cv = reinterpret_cast<float*>(bm_smem);
    int* ci = reinterpret_cast<int*>(cv + n);
    for (int a = tid; a < n; a += 256) { cv[a] = ws_val[(int64_t)b * n + a]; ci[a] = ws_idx[(int64_t)b * n + a]; }
    __syncthreads();
    for (int a = tid; a < n; a += 256) {
        const float v = cv[a]; const int ia = ci[a];
        if (ia == 0x7fffffff) continue;
        int above = 0;
        for (int j = 0; j < n; ++j) { const float o = cv[j]; above += (o > v || (o == v && ci[j] < ia)) ? 1 : 0; }
        if (above < keep) { out_val[(int64_t)b * keep + above] = v; out_idx[(int64_t)b * keep + above] = ia; }
    }
}

extern "C" size_t vm_beam_topk_ws(int B, int num_beams, int keep) { return (size_t)B * num_beams * keep * (sizeof(float) + sizeof(int)); }

extern "C" int vm_beam_topk(const float* logits, int64_t ldl, int B, int num_beams, int V, const float* running_scores, int keep,
                            float* out_values, int64_t* out_indices, void* ws, size_t ws_bytes, void* stream) {
    VM_REQUIRE(logits && running_scores && out_values && out_indices && B > 0 && V > 0 && ldl >= V, "vm_beam_topk: bad arguments");
    VM_REQUIRE(num_beams >= 1 && num_beams <= BT_MAXB && keep >= 1 && keep <= 256 && (int64_t)keep <= (int64_t)num_beams * V,
               "vm_beam_topk: 1 <= num_beams <= %d, 1 <= keep <= 256", BT_MAXB);
    VM_REQUIRE(ws && ws_bytes >= vm_beam_topk_ws(B, num_beams, keep), "vm_beam_topk: workspace of vm_beam_topk_ws(B, num_beams, keep) bytes required");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_DECODE, 16.0 * B * num_beams * (double)V, s, "beam_topk_B%d_nb%d_V%d", B, num_beams, V);
    float* wv = reinterpret_cast<float*>(ws);
    int* wi = reinterpret_cast<int*>(wv + (size_t)B * num_beams * keep);
    hipLaunchKernelGGL(beam_rows_kernel, dim3(B * num_beams), dim3(256), 0, s, logits, ldl, num_beams, V, running_scores, keep, wv, wi);
    hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(256), (size_t)num_beams * keep * 8, s, wv, wi, num_beams, keep, out_values, out_indices);
    return vm_check_launch("vm_beam_topk");
}

// argmax with lowest-index tie-break (torch.argmax returns the first maximal index on CPU)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ x, int64_t ldx, int64_t* __restrict__ idx, float* __restrict__ val, int cols) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const float* r = x + (int64_t)blockIdx.x * ldx;
    float best = -INFINITY; int bi = 0x7fffffff;
    row_strided8(r, cols, threadIdx.x, [&](int c, float v) { if (v > best || (v == best && c < bi)) { best = v; bi = c; } });
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v = sv[threadIdx.x + o]; const int i = si[threadIdx.x + o];
            if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { idx[blockIdx.x] = si[0]; if (val) val[blockIdx.x] = sv[0]; }
}
extern "C" int vm_argmax_f32(const float* x, int64_t ldx, int64_t* idx, float* val, int rows, int cols, void* stream) {
    VM_REQUIRE(x && idx && rows > 0 && cols > 0 && ldx >= cols, "vm_argmax_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_DECODE, 4.0 * rows * (double)cols, s);
    hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, s, x, ldx, idx, val, cols);
    return vm_check_launch("vm_argmax_f32");
}
