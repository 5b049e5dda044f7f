/*
 * vmhip.h -- C ABI of libvmhip.so, the MI355X (gfx950) hot path of vilmedic_amd.
 *
 * The reference (jbdel/vilmedic) has no FFI boundary of its own: its hot path is
 * reached through Python class lookup (`eval(proto)`, vilmedic/executors/utils.py:110)
 * and executed by third-party torch/transformers kernels (SURVEY.md §2.2, §8b "B2").
 * Every entry point below therefore cites the reference *call site* whose
 * arithmetic it replaces (ref: = /root/reference, hf: = the pinned HF
 * transformers implementation the reference dispatches to).
 *
 * Conventions
 *   - plain pointers + sizes; all pointers are DEVICE pointers unless noted;
 *   - activations are bf16 (uint16 storage), statistics/params/grads fp32;
 *   - every call enqueues on the hipStream_t passed as `stream` (void*), never
 *     synchronises, never allocates; workspaces are caller-provided;
 *   - return 0 on success or a negative vm_status; vm_last_error() gives text;
 *   - row-major, leading dimensions in ELEMENTS and multiples of 8.
 */
#ifndef VMHIP_H
#define VMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { VM_OK = 0, VM_EINVAL = -1, VM_EUNSUPPORTED = -2, VM_EHIP = -3 } vm_status;
typedef enum { VM_BF16 = 1, VM_F32 = 0 } vm_dtype;

const char* vm_last_error(void);
int vm_version(void);
const char* vm_build_digest(void);   /* sha256 of the sources the library was built from (vilmedic_amd/build.py compares it, not file times) */

/* ---- lightweight per-family profiler (HIP events on the launch stream; bench.py roofline) */
int vm_prof_enable(int on);
int vm_prof_reset(void);
/* family: 0 gemm, 1 attention, 2 layernorm, 3 loss, 4 elementwise, 5 optimizer, 6 decode */
int vm_prof_read(int family, double* ms_total, double* work_total, int64_t* launches); /* syncs events */
int vm_prof_dump(const char* path);  /* per-shape breakdown: "family tag launches total_ms total_work" per line */

/* ------------------------------------------------------------------ GEMM (MFMA bf16)
 * C[M,N] = epilogue( sum_k opA(A)[m,k] * opB(B)[n,k] )
 *   a_layout 0: A is [M,K] (K contiguous)   1: A is [K,M] (M contiguous)
 *   b_layout 0: B is [N,K] (K contiguous)   1: B is [K,N] (N contiguous)
 * Replaces nn.Linear fwd/bwd inside HF BERT/ViT blocks:
 *   hf:models/bert_generation/modeling_bert_generation.py:104-106,172-174 (QKV),
 *   :45-56 (out-proj), :264-291 (MLP), :590-598 (LM head); hf:models/vit/modeling_vit.py:60 (patch-embed conv as GEMM).
 */
typedef struct {
    const float* bias;        /* [N] fp32 or NULL: added to the accumulator                     */
    int act;                  /* 0 none, 1 erf-GELU (hf:activations.py "gelu")                   */
    void* aux_out;            /* bf16 [M,N] (ldc): pre-activation z when act==1, or NULL         */
    const void* mul_gelu_z;   /* bf16 [M,N] (ldc): multiply result by gelu'(z) (MLP backward)    */
    float dropout_p;          /* inverted dropout on (acc+bias) before the residual              */
    uint64_t dropout_seed;    /* counter-based RNG: keep-mask = f(seed, m*N+n)                   */
    const void* residual;     /* bf16 [M,N] (ldr) added last, or NULL                            */
    int64_t ldr;
    float alpha;              /* scale applied to the accumulator first (1.0f default)           */
    const float* alpha_dev;   /* optional DEVICE scalar multiplied into alpha (upstream dL/dloss)  */
    int out_dtype;            /* VM_BF16 or VM_F32                                               */
    int accumulate;           /* fp32 output only: atomically add into C (wgrad / split-K)       */
    int split_k;              /* >=1; >1 requires fp32 output and a plain epilogue (alpha only)  */
    void* workspace;          /* split_k>1: fp32 scratch of >= split_k*M*ldc*4 bytes (partial slabs, */
    size_t workspace_bytes;   /*   reduced deterministically by a second kernel; no atomics)        */
    const uint64_t* dropout_seed_dev;  /* optional DEVICE counter added to dropout_seed when the kernel runs: a launch replayed
                                          from a captured HIP graph then draws a fresh mask per replay (the caller bumps the counter
                                          once per training step, inside the graph) */
} vm_gemm_epilogue;

int vm_sizeof_gemm_epilogue(void);   /* lets a foreign binding verify its struct layout */
/* re-read the VM_* diagnostic environment switches (they are cached at first use) */
void vm_reload_env(void);
int vm_gemm_bf16(const void* A, int64_t lda, int a_layout, const void* B, int64_t ldb, int b_layout,
                 void* C, int64_t ldc, int M, int N, int K, const vm_gemm_epilogue* epi, void* stream);

/* Grouped weight gradients: for i < n   dW_i[n_out, k_in] (ld_dw) += alpha_i * dY_i[rows, n_out]^T (ld_dy) . X_i[rows, k_in] (ld_x)
 * and, when db_i is given, db_i[n_out] += alpha_i * column sums of dY_i -- all problems in one launch (per 8), no split-K workspace,
 * no separate reduce / column-sum kernels.  What autograd computes for nn.Linear weight / bias in the reference's backward
 * (hf:models/bert_generation/modeling_bert_generation.py:104-106,264-291; hf:models/vit/modeling_vit.py:192-251).
 * alpha_i: optional DEVICE scalar (upstream dL/dloss).  Returns VM_EUNSUPPORTED (nothing launched) unless every problem has
 * rows % 64 == 0, leading dims % 8 == 0 and 16-byte aligned pointers -- the caller then uses vm_gemm_bf16 + vm_colsum_bf16.
 * Launches whose problems all have k_in % 256 == 0 and that hold at least 64 tiles of 256 x 256 run on the wide-tile kernel (csrc/gemm_p8w.hip:
 * one 8-wave workgroup per CU, 16 problems per launch; VM_WGRAD_P8=0 switches it off), the others on 128 x 128 tiles, 8 problems per launch. */
typedef struct {
    const void* dY; int64_t ld_dy;      /* bf16 [rows, n_out]  */
    const void* X; int64_t ld_x;        /* bf16 [rows, k_in]   */
    float* dW; int64_t ld_dw;           /* fp32 [n_out, k_in], accumulated */
    float* db;                          /* fp32 [n_out] accumulated, or NULL */
    int rows, n_out, k_in;
    const float* alpha_dev;
    int overwrite;                      /* 1: dW (and db) hold nothing yet -- this is their first contribution since the gradients were zeroed --
                                           so the kernel may STORE instead of read-add-write (only the 256 x 256-tile kernel uses it; 0 is always correct) */
} vm_wgrad_problem;
int vm_wgrad_grouped(const vm_wgrad_problem* problems, int n, void* stream);
/* n independent products C_i[M_i,N_i] (+)= A_i B_i of ONE operand layout (vm_gemm_bf16's a_layout / b_layout; NT, NN or TN) with the plain
 * epilogue, 8 problems per launch: small same-shape products (the per-image contractions of the GLoRIA local loss: 72 tiles each) fill
 * the chip together.  Returns VM_EUNSUPPORTED unless K % 64 == 0, leading dimensions % 8 == 0 and 16-byte pointers (callers then loop
 * over vm_gemm_bf16). */
typedef struct {
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    void* C; int64_t ldc;
    int M, N, K;
} vm_gemm_problem;
int vm_gemm_grouped(const vm_gemm_problem* problems, int n, int a_layout, int b_layout, int out_dtype /* VM_BF16 / VM_F32 */, int accumulate,
                    void* stream);

/* ------------------------------------------------------------------ decode-step projections (M = batch x beams <= 256 rows)
 * One launch per projection of a decoder layer with its neighbours folded in (csrc/decode_gemm.hip):
 *   C[M,N] = act(LN(A)[M,K] . W[N,K]^T + bias) + residual
 * LN (when ln_gamma is given): LayerNorm over K of A's rows, computed by every workgroup for its own row block while it loads the
 * operand; the normalised rows are also written to ln_out (the residual input of the following sub-layer).  c2: output columns
 * >= split_n go to c2[m, n - split_n] (ldc2) -- the fused Q|K|V projection writes K|V of the new token straight into its cache row.
 * dtype VM_BF16: A, W, C, c2, residual, ln_out bf16 (fp32 accumulation, v_mfma_f32_16x16x32_bf16); VM_F32: everything fp32 on the
 * exact f32-input MFMA.  bias, ln_gamma, ln_beta fp32.  Replaces per decode step hf:models/bert_generation/modeling_bert_generation.py:
 * 60-106 (self/cross attention projections), :181-231 (output dense + LayerNorm), :264-358 (intermediate / output). */
typedef struct {
    int dtype;
    const void* A; int64_t lda;
    const void* W; int64_t ldw;
    void* C; int64_t ldc;
    void* c2; int64_t ldc2; int split_n;      /* second destination, or NULL */
    int M, N, K;
    const float* bias;                        /* [N] or NULL */
    int act;                                  /* 0 none, 1 erf-GELU */
    const void* residual; int64_t ldr;        /* [M,N] or NULL */
    const float* ln_gamma; const float* ln_beta; float ln_eps;      /* NULL: A is used as it is */
    void* ln_out; int64_t ln_out_ld;          /* [M,K] or NULL */
} vm_decode_gemm_args;
int vm_decode_gemm(const vm_decode_gemm_args* args, void* stream);

/* Beam-search candidates of one step (hf:generation/utils.py _beam_search :3208-3520, driven by
 * ref:vilmedic/blocks/huggingface/decoder/evaluation.py:73-78): for every sample the ``keep`` (= 2 x num_beams) best of
 * log_softmax(logits[b * num_beams + r, :]) + running_scores[b * num_beams + r] over (r, token), sorted by value (ties: flat index
 * r * V + token ascending).  logits fp32 [B * num_beams, ldl]; out_values fp32 [B, keep], out_indices int64 [B, keep]; num_beams <= 8.
 * Two launches: one workgroup per row leaves the row's ``keep`` best in ``ws`` (vm_beam_topk_ws bytes, caller-provided), one per sample merges. */
size_t vm_beam_topk_ws(int B, int num_beams, int keep);
int vm_beam_topk(const float* logits, int64_t ldl, int B, int num_beams, int V, const float* running_scores, int keep,
                 float* out_values, int64_t* out_indices, void* ws, size_t ws_bytes, void* stream);
/* Token selection of one decode step in one launch (csrc/decode_select.hip; hf:generation/utils.py _sample :2783-2975 with
 * NoBadWordsLogitsProcessor + TopKLogitsWarper, as driven by ref:vilmedic/blocks/rl/SCST.py:112-174 and
 * ref:vilmedic/blocks/huggingface/decoder/evaluation.py:73-78).  logits fp32 [rows, ldl].  Rows < greedy_rows: arg-max of the raw logits
 * (lowest index among ties).  Other rows: banned columns (<= 4, a HOST array) removed, logits below the top_k-th largest live one removed
 * (0 = no filter, ties kept, top_k <= 256), one draw from the softmax of what is left by Gumbel-max with a counter-based hash of
 * (seed, cur, row, column).  unfinished (uint8 [rows], may be NULL): finished rows emit ``pad``; a row that emits ``eos`` becomes finished.
 * next_tokens int64 [rows]; seq (may be NULL): seq[row, cur] = the token. */
int vm_select_tokens(const float* logits, int64_t ldl, int rows, int V, int greedy_rows, const int32_t* banned, int n_banned, int top_k,
                     uint64_t seed, int64_t* next_tokens, int64_t* seq, int64_t ld_seq, int cur, uint8_t* unfinished, int eos, int pad,
                     void* stream);

/* ------------------------------------------------------------------ LayerNorm
 * hf:...bert_generation.py:49,55 (post-LN, eps from YAML), hf:models/vit/modeling_vit.py:261-262,348 (pre-LN).
 * y = (x-mean)*rstd*gamma+beta over the last dim; x,y bf16 [rows,cols]; mean/rstd fp32 [rows]. */
int vm_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     int rows, int cols, float eps, void* stream);
/* dx bf16; dgamma/dbeta fp32 [cols] are ACCUMULATED (+=); ws: fp32 workspace of vm_layernorm_bwd_ws(rows,cols) bytes */
size_t vm_layernorm_bwd_ws(int rows, int cols);
int vm_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                     void* dx, float* dgamma, float* dbeta, int rows, int cols, void* ws, void* stream);
/* same with the residual-fork gradient sums fused in (either may be NULL): dy2 = second gradient of the LN output y
   (y feeds a sub-layer and that sub-layer's residual), summed with dy in fp32; dres = gradient added to dx (the LN
   input x also feeds a residual).  Replaces the separate elementwise adds autograd would launch at the fork. */
int vm_layernorm_bwd_fused(const void* dy, const void* dy2, const void* dres, const void* x, const float* gamma,
                           const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                           int rows, int cols, void* ws, void* stream);
/* the two halves of vm_layernorm_bwd_fused, for callers that run the parameter-gradient reduction on another stream:
   _partial writes dx and the per-workgroup dgamma/dbeta partials to ws; _reduce accumulates them into dgamma/dbeta */
int vm_layernorm_bwd_partial(const void* dy, const void* dy2, const void* dres, const void* x, const float* gamma,
                             const float* mean, const float* rstd, void* dx, int rows, int cols, void* ws, void* stream);
/* _partial that also writes dx_dropped = keep(seed, row * cols + col) ? dx / (1 - p) : 0 (bf16, same shape as dx): the gradient the
   linear that produced x needs when x = dropout(linear(..)) + residual -- the mask of that linear's vm_gemm_bf16 epilogue (dropout_seed /
   dropout_seed_dev as there) -- instead of a separate vm_dropout_apply_bf16 pass over dx.  dx_dropped may be NULL (= _partial). */
int vm_layernorm_bwd_partial_dropout(const void* dy, const void* dy2, const void* dres, const void* x, const float* gamma,
                                     const float* mean, const float* rstd, void* dx, void* dx_dropped, float dropout_p, uint64_t dropout_seed,
                                     const uint64_t* dropout_seed_dev, int rows, int cols, void* ws, void* stream);
int vm_layernorm_bwd_reduce(const void* ws, float* dgamma, float* dbeta, int rows, int cols, void* stream);
/* the reduce of up to any number of LayerNorm backward launches in ONE launch (their partials wait in their workspaces): the
   training step queues its 62 LayerNorms and reduces them together when the backward pass ends -- one problem alone is latency-bound */
typedef struct {
    const void* ws;                     /* workspace a vm_layernorm_bwd_partial(rows, cols) launch filled */
    float* dgamma;                      /* fp32 [cols], accumulated; no two problems of one call may share it */
    float* dbeta;
    int rows, cols;
} vm_ln_reduce_problem;
int vm_layernorm_bwd_reduce_batched(const vm_ln_reduce_problem* problems, int n, void* stream);

/* ------------------------------------------------------------------ attention (self / causal / cross)
 * softmax(Q K^T * scale + mask) V with optional dropout on the probabilities.
 * hf:...bert_generation.py:60-85 (eager_attention_forward), :114-154 (self), :181-231 (cross).
 * q: bf16, row (b*Lq+i) at q + row*ldq + h*dh ; k,v likewise with Lk rows per batch; o: [B*Lq, H*dh] ld=ldo.
 * key_mask: uint8 [B,Lk] (1 = attend) or NULL; causal: key j visible to query i iff j<=i.
 * stats: fp32 [B,H,Lq,2] = (row max, row sum) of the scaled+masked scores (saved for backward). dh must be 64.
 * kv_row_index (forward only, may be NULL): int32 [B, kv_index_ld]; key j of batch b is row kv_row_index[b][j] of the
 * k / v matrices (absolute row, batch offset NOT added) -- the decode-time KV cache with beam indirection, so a beam
 * reorder (hf:generation/utils.py:3478-3485 copies the cache) is a gather of a small index table instead. */
int vm_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                     void* o, int64_t ldo, float* stats, const uint8_t* key_mask,
                     int B, int H, int Lq, int Lk, int dh, float scale, int causal,
                     float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_dev /* see vm_gemm_epilogue; may be NULL */,
                     const int32_t* kv_row_index, int64_t kv_index_ld, void* stream);
int vm_attention_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                     const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* stats,
                     const uint8_t* key_mask,
                     void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                     int B, int H, int Lq, int Lk, int dh, float scale, int causal,
                     float dropout_p, uint64_t dropout_seed, const uint64_t* dropout_seed_dev, float* ws_delta /* fp32 [B,H,Lq] */, void* stream);

/* ------------------------------------------------------------------ BatchNorm2d, channels-last, grouped statistics (csrc/batchnorm.hip)
 * The CNN towers of ConVIRT / GLoRIA / MVQA (ref:vilmedic/models/selfsup/conVIRT.py:83-95: towers run in forward_batch_size micro-batches,
 * so every BatchNorm sees micro-batch statistics; ref:vilmedic/models/mvqa/MVQA.py:41-43: DenseNet-169, ordinary statistics = one group).
 * x, residual, y: [G * rows_per_group, C] (NHWC memory), bf16 or fp32 (``dtype``); C % 8 == 0, C <= 2048.
 *   y = relu?( (x - mean_g) * rstd_g * gamma + beta + residual? )        per group g of rows_per_group consecutive rows
 * training != 0: mean / rstd / var ([G, C] fp32; var = biased variance, for the caller's running-statistics update) are OUTPUTS;
 * training == 0: mean / rstd are inputs (running statistics), var is not touched.  ws: vm_batchnorm_nhwc_ws(G, rows_per_group, C) bytes. */
size_t vm_batchnorm_nhwc_ws(int G, int rows_per_group, int C);
int vm_batchnorm_nhwc_fwd(const void* x, const void* residual /* or NULL */, void* y, const float* gamma /* or NULL */, const float* beta,
                          float* mean, float* rstd, float* var, int G, int rows_per_group, int C, float eps, int dtype, int relu, int training,
                          void* ws, size_t ws_bytes, void* stream);
/* dy' = dy * [y > 0] when relu;  dx = gamma * rstd * (dy' - mean_g(dy') - xhat * mean_g(dy' * xhat))  (training; eval: gamma * rstd * dy');
 * dres (NULL or same shape) = dy';  dgamma / dbeta (NULL or fp32 [C], ZEROED by the caller) += sums over all groups. */
int vm_batchnorm_nhwc_bwd(const void* dy, const void* x, const void* residual, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta, int G, int rows_per_group, int C, int dtype,
                          int relu, int training, void* ws, size_t ws_bytes, void* stream);
/* The same passes with explicit strides, for a DenseNet block that keeps ONE channels-last feature buffer instead of re-concatenating it layer by
 * layer (torchvision's torch.cat in _DenseLayer.forward, reached from ref:vilmedic/blocks/vision/cnn.py:62-66 via ref:vilmedic/models/mvqa/MVQA.py:41-43):
 *   _stats   per-group mean / rstd / biased var of the C channels of x (rows ldx elements apart) -> mean / rstd / var[g * ldm + c]; the rows are also
 *            written to copy_dst (rows ld_copy apart) when given; num_batches_tracked (NULL or a device int64) += G;
 *   _apply   the normalisation pass of vm_batchnorm_nhwc_fwd over the first C channels of rows ldx apart (y and residual stay dense [rows, C]); with
 *            running_mean / running_var (fp32 [C]) the moving averages take one update per group in group order, running_var from the unbiased
 *            estimate: r = f * s_g + (1 - f) * r, f = momentum, or 1 / (updates so far) when momentum < 0 (nn.BatchNorm2d(momentum=None));
 *            rstd == NULL: rsqrt(var + eps) in the kernel (inference from running_mean / running_var without a host-side rsqrt);
 *   _bwd_ex  vm_batchnorm_nhwc_bwd with x rows ldx apart, dx rows lddx apart and, with accumulate != 0, dx += (the block's gradient buffer). */
int vm_batchnorm_nhwc_stats(const void* x, int64_t ldx, void* copy_dst /* or NULL */, int64_t ld_copy, float* mean, float* rstd, float* var, int ldm,
                            int64_t* num_batches_tracked /* or NULL */, int G, int rows_per_group, int C, float eps, int dtype, void* ws,
                            size_t ws_bytes, void* stream);
int vm_batchnorm_nhwc_apply(const void* x, int64_t ldx, const void* residual, void* y, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, const float* var, int ldm, float* running_mean /* or NULL */, float* running_var,
                            const int64_t* num_batches_tracked, float momentum, float eps, int G, int rows_per_group, int C, int dtype, int relu,
                            void* stream);
int vm_batchnorm_nhwc_bwd_ex(const void* dy, const void* x, int64_t ldx, const void* residual, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, int ldm, void* dx, int64_t lddx, int accumulate, void* dres, float* dgamma, float* dbeta, int G,
                             int rows_per_group, int C, int dtype, int relu, int training, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ embeddings
 * hf:...bert_generation.py:394-426: out = word[ids] + pos[past_len + t]   (LayerNorm is a separate call) */
int vm_embedding_fwd(const int64_t* ids, const float* word, const float* pos, void* out /* bf16 [B*L, D] */,
                     int B, int L, int D, int past_len, void* stream);
/* scatter-add d_out into fp32 grads; rows with id==padding_idx get no word gradient (nn.Embedding(padding_idx)) */
int vm_embedding_bwd(const int64_t* ids, const void* d_out, float* d_word, float* d_pos,
                     int B, int L, int D, int padding_idx, void* stream);
/* BERT / RoBERTa embeddings behind a pretrained `proto` (ref:vilmedic/blocks/huggingface/encoder/encoder_model.py:19-22,
 * decoder/decoder_model.py:17-21 -> hf:models/bert/modeling_bert.py BertEmbeddings, hf:models/roberta/modeling_roberta.py:55-155):
 *   out[row] = (word[ids[row]] + type_row) + pos[pos_ids ? pos_ids[row] : row % L + past_len]
 * pos_ids (NULL or int64 [B*L]): explicit position ids (RoBERTa: cumsum(ids != pad) * (ids != pad) + pad, made by the caller);
 * type_row (NULL or fp32 [D]): the token-type embedding of type 0 (the reference never passes token_type_ids); out bf16 or fp32. */
int vm_embedding_fwd_ex(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos, const float* type_row,
                        void* out, int out_dtype /* VM_BF16 / VM_F32 */, int B, int L, int D, int past_len, void* stream);
/* its backward: d_word[ids] += d_out (not for id == padding_idx); d_pos[pos id] += d_out (not for pos id == pos_padding_idx, -1: none;
 * without pos_ids the position of column t is t + pos_offset); d_type (NULL or fp32 [D]) += sum of all rows.  fp32 atomics. */
int vm_embedding_bwd_ex(const int64_t* ids, const int64_t* pos_ids, const void* d_out, float* d_word, float* d_pos, float* d_type,
                        int B, int L, int D, int padding_idx, int pos_offset, int pos_padding_idx, void* stream);
/* dz = dy * gelu'(z), bf16, n % 8 == 0: backward of the dense -> GELU -> LayerNorm transform of the BERT / RoBERTa LM heads
 * (hf:models/roberta/modeling_roberta.py RobertaLMHead, hf:models/bert/modeling_bert.py BertPredictionHeadTransform) */
int vm_gelu_bwd_bf16(const void* dy, const void* z, void* dz, int64_t n, void* stream);

/* ------------------------------------------------------------------ ViT patch assembly
 * hf:models/vit/modeling_vit.py:60,69 (conv as GEMM), :146-157 (cls + position embeddings). */
int vm_im2col_patches(const float* images /* [B,C,H,W] fp32 */, void* out /* bf16 [B*gh*gw, C*p*p] */,
                      int B, int C, int H, int W, int p, void* stream);
int vm_vit_assemble(const void* patches /* bf16 [B*n, D] */, const float* cls /* [D] */, const float* pos /* [(n+1),D] */,
                    void* out /* bf16 [B,(n+1),D] */, int B, int n, int D, void* stream);
int vm_vit_assemble_bwd(const void* d_out /* bf16 [B,(n+1),D] */, void* d_patches /* bf16 [B*n,D] */,
                        float* d_cls /* += [D] */, float* d_pos /* += [(n+1),D] */, int B, int n, int D, void* stream);
/* the same with ns (1..4) special tokens in front of the patches: DeiT = [CLS], distillation token
 * (ref:vilmedic/blocks/vision/visual_encoder.py:59-61 -> hf:models/deit/modeling_deit.py DeiTEmbeddings.forward) */
int vm_vit_assemble_ex(const void* patches /* bf16 [B*n, D] */, const float* special /* [ns,D] */, const float* pos /* [(n+ns),D] */,
                       void* out /* bf16 [B,(n+ns),D] */, int B, int n, int ns, int D, void* stream);
int vm_vit_assemble_bwd_ex(const void* d_out /* bf16 [B,(n+ns),D] */, void* d_patches /* bf16 [B*n,D] */,
                           float* d_special /* += [ns,D] */, float* d_pos /* += [(n+ns),D] */, int B, int n, int ns, int D, void* stream);

/* ------------------------------------------------------------------ losses
 * Shifted causal-LM cross-entropy (hf:loss/loss_utils.py:49-72; labels = input_ids, pads included,
 * ref:vilmedic/blocks/huggingface/decoder/decoder_model.py:46).  logits bf16 [B*L, ldl]; label of row (b,t) is
 * ids[b,t+1]; rows with t==L-1 are ignored.  loss_sum: fp32[1] += sum of row losses (caller divides by B*(L-1)).
 * dlogits (bf16, same layout, may alias logits) = (softmax - onehot) * grad_scale, zero on ignored rows / pad cols.
 * row_weight (NULL = 1): fp32 [B*L]; row (b,t) contributes row_weight*CE and its gradient is scaled likewise (SCST:
 * -(logp*mask/sum(mask))*(r_sample-r_greedy), ref:vilmedic/blocks/rl/SCST.py:14-45).  banned[0..n_banned) (<=4 columns, a HOST array)
 * are treated as -inf logits (bad_words_ids=[[pad],[bos]], SCST.py:150-151).  row_logp (NULL or fp32 [B*L]) receives
 * log p(label) of each row.  row_min_logit (NULL or fp32 [B*L]): logits below it are -inf (TopKLogitsWarper:
 * the k-th largest logit of the row). */
int vm_ce_shift_fwd_bwd(const void* logits, int64_t ldl, const int64_t* ids, int B, int L, int V,
                        float* loss_sum, float* row_logp, void* dlogits, float grad_scale,
                        const float* row_weight, const int32_t* banned, int n_banned, const float* row_min_logit,
                        void* stream);
/* thr[row] = k-th largest logit of the row among the live columns (c < V, not banned): the TopKLogitsWarper threshold that
 * vm_ce_shift_fwd_bwd takes as row_min_logit (hf:generation/logits_process.py TopKLogitsWarper as configured by
 * ref:vilmedic/blocks/rl/SCST.py:142-157).  Exact radix select on the bf16 values; rows with fewer than k live columns keep everything. */
int vm_topk_threshold_bf16(const void* logits, int64_t ldl, int rows, int V, int k, const int32_t* banned, int n_banned,
                           float* thr, void* stream);
/* Generic CE with label smoothing on fp32 logits [R,C] (MVQA head; ref:...LabelSmoothingCrossEntropyLoss.py:38-48) */
int vm_ce_smooth_fwd_bwd(const float* logits, const int64_t* target, int R, int C, float smoothing,
                         float* loss_sum, float* dlogits, float grad_scale, void* stream);

/* Image-text contrastive similarity losses (ref:vilmedic/blocks/losses/selfsup/ConVIRTLoss.py:12-31, InfoNCELoss.py:11-19,
 * GLoRIALoss.py:54-75): with S[R,C] = n(a) n(b)^T * inv_tau (n = x / max(|x|, eps) when ``normalize``, identity otherwise) and row i
 * paired with column i + diag_offset (the local rows of a rank against all gathered columns):
 *     loss_rows[i] = log sum_j exp(S_ij) - S_{i,i+off}      loss_cols[j] = log sum_i exp(S_ij) - S_{j-off,j}
 * (the paired term is dropped for a row / column without a partner).
 * vm_contrastive_loss_fwd: three short dependent launches -- normalise + cast both matrices (a_hat [R,D], b_hat [C,D] bf16, the norms),
 *   one workgroup per 128 x 128 tile of S on the MFMA writing per-tile (max, sum exp) partials and the fp32 tile itself into ``ws``, one
 *   thread per row / column merging the partials into lse_rows / lse_cols and the losses.
 * vm_contrastive_loss_bwd: three launches -- a streaming pass over the stored S:
 *   G_ij = g_rows[i] softmax_row(S)_ij + g_cols[j] softmax_col(S)_ij - [j == i+off](g_rows[i] + g_cols[j])  (bf16, in ``ws``, with the
 *   sums of G.S that the normalisation backward needs); the two gradient products G b_hat / tau and G^T a_hat / tau as tiles of the
 *   library's GEMM main loop (R and C multiples of 64; 64 x 96 register-staged tiles otherwise) into
 *   da = d/da (sum_i g_rows[i] loss_rows[i] + sum_j g_cols[j] loss_cols[j]) and db likewise (fp32 [R,D] / [C,D]); the L2-normalisation
 *   backward as a row pass over both, in place.  ``ws``: vm_contrastive_ws(R, C) bytes, 256-B aligned, and THE SAME buffer, untouched, for
 *   the forward call and its backward call (it carries S; until round 4 the backward recomputed S).  (A single persistent backward launch
 *   with in-kernel hand-offs was built first and measured slower: every release / acquire hand-off costs more than a kernel boundary on
 *   this chip -- csrc/contrastive.hip.)
 * vm_rownorm_cast: the normalise + cast of one matrix on its own (GLoRIA's global embeddings share it). */
int vm_rownorm_cast(const float* x /* [rows,D] */, void* out_bf16, float* norms /* [rows] or NULL */, int rows, int D,
                    int normalize /* 1: x/max(|x|,eps) (ConVIRT cosine)  0: plain cast (InfoNCE) */, float eps, void* stream);
size_t vm_contrastive_ws(int R, int C);
int vm_contrastive_loss_fwd(const float* a /* [R,D] */, const float* b /* [C,D] */, int R, int C, int D, int normalize, float eps,
                            float inv_tau, int diag_offset, void* a_hat_bf16, void* b_hat_bf16, float* norm_a, float* norm_b,
                            float* lse_rows, float* lse_cols, float* loss_rows /* [R] */, float* loss_cols /* [C] */,
                            void* ws, size_t ws_bytes, void* stream);
int vm_contrastive_loss_bwd(const float* a, const float* b, const void* a_hat_bf16, const void* b_hat_bf16, const float* norm_a,
                            const float* norm_b, int R, int C, int D, int normalize, float eps, float inv_tau, int diag_offset,
                            const float* lse_rows, const float* lse_cols, const float* g_rows /* [R] */, const float* g_cols /* [C] */,
                            float* da /* [R,D] */, float* db /* [C,D] */, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ GLoRIA local (word x image-region) loss
 * ref:vilmedic/blocks/losses/selfsup/GLoRIALoss.py:14-51 (gloria_attention_fn), :78-129 (local_loss): every (caption i, image j) pair of
 * the batch at once, fp32 throughout (see csrc/gloria.hip for the math).  Layouts (B captions = B images, Tp / Pp = words / regions
 * padded to multiples of 16, cap_lens int32 [B] on the device):
 *   S     fp32 [B*Tp, ldS >= B*Pp]   S[(i Tp + t), j Pp + p] = <word t of caption i, region p of image j>  (one vm_gemm_f32)
 *   a2    fp32 [B img][B*Tp][Pp]     attention of word (i, t) over the regions of image j; rows t >= cap_lens[i] and columns >= P zero
 *   dot   fp32 [B*Tp][B]             sum_p a2 S = <word, attention-weighted context>
 *   colstat fp32 [B cap][B img][2][Pp]  per-region (max, 1 / sum exp) over the caption's words, kept for the backward pass
 *   x     fp32 [B img][B*Tp][ldx]    context vectors x = a2 C^T (one vm_gemm_f32 per image)
 *   sims  fp32 [B img][B cap] = temp3 log sum_t exp(temp2 cos(word, x)), simsT its transpose (the two cross-entropy inputs)
 * Backward: vm_gloria_cos_bwd turns d sims (+ d simsT) into d x [B img][B*Tp][ldx] and the first half of d words (dW1 [B*Tp][ldx]);
 * vm_gloria_attn_bwd turns da2 (= d x C per image, overwritten as scratch) into d S (layout of S). */
int vm_transpose_f32(const float* src, int64_t src_batch_stride, int64_t ld_src, float* dst, int64_t dst_batch_stride, int64_t ld_dst,
                     int batch, int rows, int cols, int dst_rows /* >= cols */, int dst_cols /* >= rows */, void* stream);   /* dst[b][c][r] = src[b][r][c], zero padded */
int vm_row_norm_f32(const float* x, int64_t ldx, float* out /* [rows] */, int rows, int cols, void* stream);
/* fp32 [rows, cols] -> the three bf16 parts of a "bf16 x 3" GEMM operand (a = hi + lo; A side (hi, hi, lo), B side (hi, lo, hi)), laid out
   along the contraction axis in blocks of ``block``: along_rows = 0 -> dst [rows, 3 cols] (columns [3 b block, 3 (b+1) block) = the parts of
   source columns [b block, (b+1) block)); along_rows = 1 -> dst [3 rows, cols] likewise over the rows.  vm_gemm_bf16 with the contraction
   3 x as long and fp32 output then gives the product to ~2^-16 relative. */
int vm_split3_bf16(const float* src, int64_t ld_src, int rows, int cols, void* dst_bf16, int64_t ld_dst, int role /* 0 = A, 1 = B */,
                   int along_rows, int block, void* stream);
int vm_gloria_attn_fwd(const float* S, int64_t ldS, const int32_t* cap_lens, int B, int Tp, int P, int Pp, float temp1,
                       float* a2, float* dot, float* colstat, void* stream);
int vm_gloria_cos_fwd(const float* x, int64_t ldx, const float* word_norm /* [B*Tp] */, const float* dot, const int32_t* cap_lens, int B, int Tp, int D,
                      float temp2, float temp3, float eps, float* sims, float* simsT, float* cosv /* [B*Tp][B] */, float* nxv /* [B*Tp][B] */, void* stream);
int vm_gloria_cos_bwd(const float* dsims, const float* dsimsT, const float* sims, const float* cosv, const float* nxv, const float* word_norm,
                      const float* x, const float* Wt /* [B*Tp][ldx] */, int64_t ldx, const int32_t* cap_lens, int B, int Tp, int D, float temp2, float temp3,
                      float eps, float* dx, float* dW1, void* stream);
int vm_gloria_attn_bwd(const float* S, int64_t ldS, const float* colstat, float* da2 /* in: d x C; scratch */, const int32_t* cap_lens,
                       int B, int Tp, int P, int Pp, float temp1, float* dS /* layout of S */, void* stream);

/* ------------------------------------------------------------------ element-wise / reductions */
int vm_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
int vm_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream);
/* dst[r, 0:cols] (ld_dst) = bf16(src fp32 [rows, cols]); pad columns/rows left untouched */
int vm_cast_pad_f32_to_bf16(const float* src, void* dst, int rows, int cols, int64_t ld_dst, void* stream);
int vm_colsum_bf16(const void* x, int64_t ldx, float* out /* += [cols] */, int rows, int cols,
                   const float* scale_dev /* optional device scalar */, void* stream);
int vm_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* out = keep(seed, idx) ? x/(1-p) : 0 with idx = flat element index (the mask vm_gemm_bf16 uses with idx = m*N+n) */
int vm_dropout_apply_bf16(const void* x, void* out, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, void* stream);
int vm_feature_mask(const void* feats /* bf16 [rows, cols] */, uint8_t* mask, int rows, int cols, void* stream);

/* ------------------------------------------------------------------ optimizer
 * torch.optim.Adam/AdamW semantics (ref:vilmedic/executors/utils.py:65-94 builds any torch.optim by name).
 * One launch over a flat arena: p,m,v fp32 [n]; g fp32 [n]; optional bf16 shadow refresh. */
int vm_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16 /* or NULL */, int64_t n,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd,
                 float bias_corr1, float bias_corr2, float grad_scale, void* stream);
/* The same update with its per-step scalars read from DEVICE memory, so that the launch can sit in a captured HIP graph:
 * lr_dev (NULL: use lr), step_dev (NULL: use bias_corr1/2; else the 1-based step count t, bias corrections 1 - beta^t computed in the
 * kernel), gate_dev (NULL: always update; else the update is skipped -- p, m, v, shadow untouched -- unless *gate_dev is finite: the
 * NaN/Inf-loss guard of ref:vilmedic/executors/trainor.py:109-112 without a host read of the loss). */
int vm_adam_step_dev(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                     float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd,
                     float bias_corr1, float bias_corr2, float grad_scale,
                     const float* lr_dev, const int64_t* step_dev, const float* gate_dev, void* stream);
/* vm_adam_step_dev with the gradient read as bf16 [n]: the averaged wire buffer of the data-parallel gradient all-reduce
 * (ref:vilmedic/executors/trainor_accelerate.py:111-156 -- accelerate's DDP averages .grad before optimizer.step()) consumed directly,
 * without a cast pass back into the fp32 gradient arena; bf16 -> fp32 is exact, so the update equals vm_adam_step_dev on the cast values. */
int vm_adam_step_wire(float* p, const void* g_bf16, float* m, float* v, void* shadow_bf16, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd,
                      float bias_corr1, float bias_corr2, float grad_scale,
                      const float* lr_dev, const int64_t* step_dev, const float* gate_dev, void* stream);

/* ------------------------------------------------------------------ decode step helpers
 * hf:generation/utils.py:3384-3389 (fp32 log_softmax), :3113-3119 (top-k over beams*V), :2925 (argmax). */
int vm_logsoftmax_f32(const float* logits, int64_t ldl, float* out, int rows, int V, void* stream);
int vm_argmax_f32(const float* x, int64_t ldx, int64_t* idx, float* val, int rows, int cols, void* stream);

/* ------------------------------------------------------------------ fp32 decode step (csrc/decode_f32.hip)
 * The same step as above with NOTHING rounded below fp32, so that greedy / beam token indices are those of the reference's
 * fp32 path (ref:vilmedic/blocks/huggingface/decoder/evaluation.py:73-78 -> hf:generation/utils.py _sample / _beam_search;
 * block arithmetic hf:models/bert_generation/modeling_bert_generation.py:45-231,264-358,394-426,590-610).
 * vm_gemm_f32: C[M,N] (ldc) = act(A[M,K] (lda) . W[N,K]^T (ldw) + bias) + residual (ldr); exact f32 MFMA; act 1 = erf-GELU (libm erff);
 *   K % 16 == 0; W is a fp32 master weight straight from the caller's parameter storage (no bf16 shadow). */
int vm_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int M, int N, int K,
                const float* bias, int act, const float* residual, int64_t ldr, void* stream);
int vm_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int rows, int cols, float eps, void* stream);
int vm_embedding_fwd_f32(const int64_t* ids, const float* word, const float* pos, float* out /* fp32 [B*L, D] */,
                         int B, int L, int D, int past_len, void* stream);
/* one query row per (row, head): softmax(q . K^T * scale + mask) V.  Query row r reads key/value batch r / q_per_kv; key j of
 * row r is K row kv_row_index[r*kv_index_ld + j] when the table is given (beam-search cache indirection), else
 * (r / q_per_kv)*Lk + j.  key_mask uint8 [rows / q_per_kv, Lk] (1 = attend) or NULL. */
int vm_attention_decode_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            float* o, int64_t ldo, const uint8_t* key_mask, const int32_t* kv_row_index, int64_t kv_index_ld,
                            int rows, int H, int Lk, int dh, int q_per_kv, float scale, void* stream);

/* ---- image input pipeline on the device (replaces, for decoded uint8 HWC images resident in HBM, the PIL / torchvision
   chain of vilmedic/datasets/base/ImageDataset.py:96-108: Resize -> RandomCrop -> RandomHorizontalFlip -> ToTensor ->
   Normalize for training (resize > 0), Resize((crop,crop)) -> ToTensor -> Normalize for evaluation (resize == 0)).
   Bit-exact with Pillow's 8-bit bilinear resampler.  src: device, packed images; src_offset [B] (bytes), src_hw [B][2],
   crop_top_left [B][2] (in the RESIZED image; NULL = 0,0), flip [B] (NULL = none), mean/std [3]: HOST arrays.
   out: device fp32 [B,3,crop,crop].  max_taps >= 2*ceil(max(scale,1))+1 of the largest down-scale; ws: device workspace
   of at most vm_image_pipeline_ws(B, max_out, max_taps) bytes (max_out = largest resized side in the batch; images of
   equal size share their coefficient tables). */
size_t vm_image_pipeline_ws(int B, int max_out, int max_taps);
int vm_image_pipeline_u8(const uint8_t* src, const int64_t* src_offset, const int32_t* src_hw, int B, int resize, int crop,
                         const int32_t* crop_top_left, const uint8_t* flip, const float* mean, const float* std,
                         float* out, int max_taps, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VMHIP_H */
