// decode_select.hip -- the token selection of one decode step in ONE launch: arg-max (greedy rows) or a draw from the bad-word + top-k
// filtered softmax (sampled rows), the finished-row bookkeeping and the write into the sequence buffer.
//
// Replaces, per step, HF's  NoBadWordsLogitsProcessor -> TopKLogitsWarper -> softmax -> multinomial / argmax -> pad finished rows ->
// append -> update unfinished  (hf:generation/utils.py _sample :2783-2975, hf:generation/logits_process.py) as driven by
// ref:vilmedic/blocks/rl/SCST.py:112-174 (greedy baseline + sampled rollout) and ref:vilmedic/blocks/huggingface/decoder/evaluation.py:73-78
// -- round 2 ran it as a dozen torch kernels per step (clone, index_fill, topk, masked_fill, softmax, multinomial, where, ...).
//
// One workgroup per row of fp32 logits [rows, V]:
//   greedy rows   arg-max of the RAW logits, lowest index among equal maxima (torch.argmax);
//   sampled rows  banned columns removed; top-k: the k-th largest live logit is found EXACTLY without sorting the row -- the k-th largest
//                 of the 256 per-thread maxima is a lower bound L of it (k values >= L exist), every logit >= L goes to a candidate list in
//                 LDS (a few dozen entries), the k-th largest candidate is the threshold (ties kept, as HF's ``scores < kth``);
//                 the draw is Gumbel-max over the candidates: arg-max of logit + G, G = -log(-log u), u from a counter-based hash of
//                 (seed, step, row, column) -- an exact sample of softmax(filtered logits) with no normalisation pass and no state.
// The drawn numbers are not torch's (multinomial's Philox stream); the distribution is the reference's (tests/: frequency test).
#include "common.h"

#define SEL_CAND 2048
struct SelArgs {
    const float* logits; int64_t ldl; int rows, V, greedy_rows, top_k, n_banned; int banned[4];
    uint64_t seed; int64_t* next; int64_t* seq; int64_t ld_seq; int cur; uint8_t* unfinished; int eos, pad;
};

__device__ __forceinline__ uint32_t sel_hash(uint32_t x, uint32_t k0, uint32_t k1) {
    x ^= k0; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x += k1; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// (value, index) arg-max merge: larger value wins, equal values -> smaller index
__device__ __forceinline__ void sel_better(float& v, int& i, float v2, int i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}
__device__ __forceinline__ void sel_block_argmax(float& v, int& i, float* sv, int* si) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float v2 = __shfl_xor(v, off);
        const int i2 = __shfl_xor(i, off);
        sel_better(v, i, v2, i2);
    }
    if (lane == 0) { sv[wave] = v; si[wave] = i; }
    __syncthreads();
    v = sv[0]; i = si[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) sel_better(v, i, sv[w], si[w]);
    __syncthreads();
}

__global__ __launch_bounds__(256) void select_tokens_kernel(const SelArgs p) {
    __shared__ float s_max[256];
    __shared__ float c_val[SEL_CAND];
    __shared__ int c_idx[SEL_CAND];
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_n;
    __shared__ float s_thr;
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* lr = p.logits + (int64_t)row * p.ldl;
    const bool sample = row >= p.greedy_rows;
    auto banned = [&](int c) {
        return p.n_banned > 0 && (c == p.banned[0] || (p.n_banned > 1 && c == p.banned[1]) || (p.n_banned > 2 && c == p.banned[2]) || (p.n_banned > 3 && c == p.banned[3]));
    };
    const DropKey key = drop_key(p.seed + ((uint64_t)p.cur << 32) + (uint64_t)row * 0x9E3779B97F4A7C15ull);
    auto gumbel = [&](int c) {
        const uint32_t h = sel_hash((uint32_t)c, key.s0, key.s1);
        // 23 random bits: (h >> 9) + 0.5 is exact in fp32 and u stays strictly inside (0, 1).  With 24 bits the + 0.5 is not representable above
        // 2^23 and 16777215.5 rounds to 2^24, i.e. u == 1 with probability 2^-24 per column: -log(-log 1) = +inf made that column win whatever
        // its logit (about V / 2^24 = 0.18 % of the sampled rows per step at V = 30522).  The inner log is the accurate one: near u = 1 the
        // fast __logf loses the relative accuracy the outer log needs.
        const float u = ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f);
        return -__logf(-logf(u));
    };
    // one pass over the row: 16-B loads, two per thread in flight (a row is 120 KB read by ONE workgroup -- with 4-B loads the launch was
    // 39 us of dependent round trips); every element is visited exactly once, so what the passes compute does not depend on the order
    const bool vec = (((uintptr_t)lr) & 15) == 0;
    auto visit = [&](auto&& fn) {
        int done = 0;
        if (vec) {
            const int nq = p.V >> 2;
            const float4* l4 = reinterpret_cast<const float4*>(lr);
            int q = tid;
            for (; q + 256 < nq; q += 512) {
                const float4 a = l4[q], b = l4[q + 256];
                fn(4 * q, a.x); fn(4 * q + 1, a.y); fn(4 * q + 2, a.z); fn(4 * q + 3, a.w);
                fn(4 * q + 1024, b.x); fn(4 * q + 1025, b.y); fn(4 * q + 1026, b.z); fn(4 * q + 1027, b.w);
            }
            if (q < nq) { const float4 a = l4[q]; fn(4 * q, a.x); fn(4 * q + 1, a.y); fn(4 * q + 2, a.z); fn(4 * q + 3, a.w); }
            done = nq << 2;
        }
        for (int c = done + tid; c < p.V; c += 256) fn(c, lr[c]);
    };
    float bv = -INFINITY; int bi = 0x7fffffff;
    if (!sample) {
        visit([&](int c, float v) { sel_better(bv, bi, v, c); });
    } else if (p.top_k <= 0 || p.top_k >= p.V - p.n_banned) {
        visit([&](int c, float v) { if (!banned(c)) sel_better(bv, bi, v + gumbel(c), c); });
    } else {
        // (1) lower bound L of the k-th largest live logit: the k-th largest of the per-thread maxima
        float mx = -INFINITY;
        visit([&](int c, float v) { if (!banned(c)) mx = fmaxf(mx, v); });
        s_max[tid] = mx;
        if (tid == 0) s_n = 0;
        __syncthreads();
        int rank = 0;
        for (int j = 0; j < 256; ++j) { const float o = s_max[j]; rank += (o > mx || (o == mx && j < tid)) ? 1 : 0; }
        const int kk = min(p.top_k, 256);
        if (rank == kk - 1) s_thr = mx;                      // exactly one thread has each rank
        __syncthreads();
        const float L = s_thr;
        // (2) every live logit >= L (at least k of them; normally a few dozen)
        visit([&](int c, float v) {
            if (!banned(c) && v >= L) {
                const int at = atomicAdd(&s_n, 1);
                if (at < SEL_CAND) { c_val[at] = v; c_idx[at] = c; }
            }
        });
        __syncthreads();
        const int n = min(s_n, SEL_CAND);
        // (3) the exact threshold: the candidate that has exactly k - 1 candidates above it (ties by index); a row with fewer than k
        // live logits, or an overflowing candidate list (degenerate: thousands of ties), keeps everything it collected
        float thr = (s_n > SEL_CAND || n < p.top_k) ? L : INFINITY;
        __syncthreads();
        if (thr == INFINITY) {
            for (int a = tid; a < n; a += 256) {
                const float v = c_val[a]; const int ia = c_idx[a];
                int above = 0;
                for (int b = 0; b < n; ++b) { const float o = c_val[b]; above += (o > v || (o == v && c_idx[b] < ia)) ? 1 : 0; }
                if (above == p.top_k - 1) s_thr = v;
            }
            __syncthreads();
            thr = s_thr;
        }
        // (4) Gumbel-max over the kept candidates
        for (int a = tid; a < n; a += 256) if (c_val[a] >= thr) sel_better(bv, bi, c_val[a] + gumbel(c_idx[a]), c_idx[a]);
    }
    sel_block_argmax(bv, bi, sv, si);
    if (tid == 0) {
        int64_t tok = (bi == 0x7fffffff) ? (int64_t)p.pad : (int64_t)bi;
        if (p.unfinished) {
            const bool live = p.unfinished[row] != 0;
            tok = live ? tok : (int64_t)p.pad;
            p.unfinished[row] = (live && tok != p.eos) ? 1 : 0;
        }
        p.next[row] = tok;
        if (p.seq) p.seq[(int64_t)row * p.ld_seq + p.cur] = tok;
    }
}

extern "C" int vm_select_tokens(const float* logits, int64_t ldl, int rows, int V, int greedy_rows, const int32_t* banned, int n_banned, int top_k,
                                uint64_t seed, int64_t* next_tokens, int64_t* seq, int64_t ld_seq, int cur, uint8_t* unfinished, int eos, int pad,
                                void* stream) {
    VM_REQUIRE(logits && next_tokens && rows > 0 && V > 0 && ldl >= V && greedy_rows >= 0, "vm_select_tokens: bad arguments");
    VM_REQUIRE(n_banned >= 0 && n_banned <= 4 && (n_banned == 0 || banned), "vm_select_tokens: at most 4 banned columns (HOST array)");
    if (top_k > 256 && top_k < V - n_banned) { vm_set_error("vm_select_tokens: top_k = %d > 256 is not supported by the one-pass threshold", top_k); return VM_EUNSUPPORTED; }
    SelArgs a = {logits, ldl, rows, V, greedy_rows, top_k, n_banned, {0, 0, 0, 0}, seed, next_tokens, seq, ld_seq, cur, unfinished, eos, pad};
    for (int i = 0; i < n_banned; ++i) a.banned[i] = banned[i];
    hipStream_t s = (hipStream_t)stream;
    VmProfScope prof(VM_FAM_DECODE, 4.0 * rows * (double)V, s, "select_r%d_V%d_k%d", rows, V, top_k);
    hipLaunchKernelGGL(select_tokens_kernel, dim3(rows), dim3(256), 0, s, a);
    return vm_check_launch("vm_select_tokens");
}
