// gemm_args.h -- kernel argument block shared by the generic (gemm.hip) and tuned (gemm_fast.hip) GEMM kernels.
#pragma once
#include "common.h"

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int tiles_m, tiles_n, ktiles, ktiles_per_split;
    int group_w;             // fast path: tile columns per column group of the block-id -> tile map (>= tiles_n: plain row-major)
    vm_gemm_epilogue e;
    uint32_t drop_thresh; float drop_scale;
    int dbg;                 // VM_GEMM_DEBUG experiments (0 in production): 1 skip epilogue, 2 single K-tile, 4 epilogue operands requested late
    float* slabs;            // split-K partial slabs [split][M][ldc] fp32 (fast path), or null
    float* bias_grad;        // grouped weight-gradient launch: fp32 [M] += alpha * sum_k A(m, k), or null
};

#define VM_GEMM_MAX_GROUP 8
struct GemmGroupArgs {
    int n;
    int tile_start[VM_GEMM_MAX_GROUP + 1];
    GemmArgs g[VM_GEMM_MAX_GROUP];
};
int vm_gemm_grouped_launch(const GemmGroupArgs& ga, int nblocks, int a_layout, int b_layout, hipStream_t s);

// tuned path (gemm_fast.hip): requires K % 64 == 0
int vm_gemm_splitk_reduce(const GemmArgs& a, int nsplit, hipStream_t s);
void vm_gemm_variant_tile(int variant, int a_layout, int* bm, int* bn);
int vm_gemm_fast_dispatch(const GemmArgs& a, int a_layout, int b_layout, int nblocks, int variant, hipStream_t s);
// two GEMMs of different operand layouts in one launch of 128 x 128 tiles (a0: row-major A x k-major B, a1: k-major A x k-major B)
int vm_gemm_pair_launch(const GemmArgs& a0, const GemmArgs& a1, hipStream_t s);
// wide-tile path (gemm_p8.hip): (32 mf) x 256 tiles, 8 waves, one workgroup per CU; row-major A
int vm_gemm_p8_dispatch(const GemmArgs& a, int a_layout, int b_layout, int mf, int phases, int epi, int total, hipStream_t s);
// grouped weight gradients on 256 x 256 tiles (gemm_p8w.hip): dW[M, N] (+)= alpha * A^T B over `ktiles` 64-row steps, A = dY [rows, M] (lda),
// B = X [rows, N] (ldb), C = dW fp32 (ldc); tile_start[i] = first tile of problem i, tile_start[P8W_MAX_GROUP] = all tiles
#define P8W_MAX_GROUP 16
struct P8wProblem {
    const bf16_t* A; const bf16_t* B; void* C; float* bias_grad; const float* alpha_dev;
    int64_t lda, ldb, ldc;
    int M, N, ktiles, tiles_n, accumulate, pad_;
};
struct P8wArgs {
    int n;
    int tile_start[P8W_MAX_GROUP + 1];
    P8wProblem g[P8W_MAX_GROUP];
};
int vm_wgrad_p8w_launch(const P8wArgs& ga, int phases, hipStream_t s);
// skinny path (gemm_skinny.hip): M <= 256 rows (the decode step), K % 32 == 0, row-major A and B, plain / bias / gelu / residual epilogue
int vm_skinny_rows_per_wg(int M, int N, int max_mf);
int vm_gemm_skinny_dispatch(const GemmArgs& a, hipStream_t s);
