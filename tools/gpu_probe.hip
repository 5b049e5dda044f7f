// gpu_probe.hip -- standalone first-contact test for the GPU box (no Python):
//   1. prints the lane mapping of ds_read_b64_tr_b16 (gemm layout-1 fragments depend on it);
//   2. checks vm_gemm_bf16 in all four operand layouts against a CPU reference;
//   3. checks vm_layernorm_fwd/bwd against a CPU reference.
// build: hipcc --offload-arch=gfx950 -O2 tools/gpu_probe.hip vilmedic_amd/csrc/{gemm,layernorm,runtime}.o -o tools/gpu_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "../include/vmhip.h"

typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void tr_probe(short* out) {
    __shared__ short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    // lane l supplies address of elements [4l, 4l+3]
    const __attribute__((address_space(3))) v4s* p = (const __attribute__((address_space(3))) v4s*)(lds + threadIdx.x * 4);
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

static int test_gemm(int M, int N, int K, int la, int lb, int split, bool f32out) {
    // logical A(m,k), B(n,k)
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    for (auto& x : A) x = bf2f(f2bf(frand()));
    for (auto& x : B) x = bf2f(f2bf(frand()));
    int64_t lda = la == 0 ? (K + 7) / 8 * 8 : (M + 7) / 8 * 8;
    int64_t ldb = lb == 0 ? (K + 7) / 8 * 8 : (N + 7) / 8 * 8;
    std::vector<uint16_t> hA((size_t)(la == 0 ? M : K) * lda, 0), hB((size_t)(lb == 0 ? N : K) * ldb, 0);
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) hA[la == 0 ? (size_t)m * lda + k : (size_t)k * lda + m] = f2bf(A[(size_t)m * K + k]);
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) hB[lb == 0 ? (size_t)n * ldb + k : (size_t)k * ldb + n] = f2bf(B[(size_t)n * K + k]);
    int64_t ldc = (N + 7) / 8 * 8;
    void *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * ldc * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dC, 0, (size_t)M * ldc * 4);
    vm_gemm_epilogue e = {};
    e.alpha = 1.f; e.out_dtype = f32out ? VM_F32 : VM_BF16; e.split_k = split; e.accumulate = split > 1;
    void* ws = nullptr;
    if (split > 1) { e.workspace_bytes = (size_t)split * M * ldc * 4; hipMalloc(&ws, e.workspace_bytes); e.workspace = ws; }
    int rc = vm_gemm_bf16(dA, lda, la, dB, ldb, lb, dC, ldc, M, N, K, &e, nullptr);
    if (rc) { printf("gemm rc=%d %s\n", rc, vm_last_error()); return 1; }
    hipDeviceSynchronize();
    std::vector<float> C((size_t)M * ldc);
    if (f32out) hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    else { std::vector<uint16_t> t(C.size()); hipMemcpy(t.data(), dC, t.size() * 2, hipMemcpyDeviceToHost); for (size_t i = 0; i < t.size(); ++i) C[i] = bf2f(t[i]); }
    double maxerr = 0; int bad = 0;
    for (int m = 0; m < M; m += (M > 512 ? 37 : 1)) for (int n = 0; n < N; ++n) {
        double ref = 0; for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * B[(size_t)n * K + k];
        double err = fabs(ref - C[(size_t)m * ldc + n]);
        double tol = f32out ? 1e-3 + 1e-4 * fabs(ref) : 0.02 + 0.01 * fabs(ref);
        if (err > tol) { if (bad < 5) printf("  mismatch m=%d n=%d ref=%f got=%f\n", m, n, ref, C[(size_t)m * ldc + n]); ++bad; }
        if (err > maxerr) maxerr = err;
    }
    printf("gemm M=%d N=%d K=%d la=%d lb=%d split=%d f32=%d maxerr=%.4g bad=%d %s\n", M, N, K, la, lb, split, (int)f32out, maxerr, bad, bad ? "FAIL" : "ok");
    hipFree(dA); hipFree(dB); hipFree(dC);
    return bad != 0;
}

static void bench_gemm(int M, int N, int K, int la, int lb, int split = 1) {
    int64_t lda = la == 0 ? K : M, ldb = lb == 0 ? K : N;
    void *dA, *dB, *dC;
    size_t na = (size_t)M * K, nb = (size_t)N * K;
    std::vector<uint16_t> h(std::max(na, nb));
    for (auto& x : h) x = f2bf(frand());
    hipMalloc(&dA, na * 2); hipMalloc(&dB, nb * 2); hipMalloc(&dC, (size_t)M * N * 2);
    hipMemcpy(dA, h.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(dB, h.data(), nb * 2, hipMemcpyHostToDevice);
    vm_gemm_epilogue e = {}; e.alpha = 1.f; e.out_dtype = split > 1 ? VM_F32 : VM_BF16; e.split_k = split; e.accumulate = split > 1;
    void* ws = nullptr;
    if (split > 1) { hipFree(dC); hipMalloc(&dC, (size_t)M * N * 4); hipMemset(dC, 0, (size_t)M * N * 4); e.workspace_bytes = (size_t)split * M * N * 4; hipMalloc(&ws, e.workspace_bytes); e.workspace = ws; }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) vm_gemm_bf16(dA, lda, la, dB, ldb, lb, dC, N, M, N, K, &e, nullptr);
    hipEventRecord(a, nullptr);
    const int it = 20;
    for (int i = 0; i < it; ++i) vm_gemm_bf16(dA, lda, la, dB, ldb, lb, dC, N, M, N, K, &e, nullptr);
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
    printf("bench M=%d N=%d K=%d la=%d lb=%d split=%d: %.3f ms  %.1f TFLOP/s\n", M, N, K, la, lb, split, ms, 2.0 * M * N * K / ms * 1e-9);
    hipFree(dA); hipFree(dB); hipFree(dC);
}

// epilogue-option benchmark: flags bit0 bias, bit1 gelu act, bit2 z side output, bit3 mul_gelu_z, bit4 dropout, bit5 residual;
// rot = number of rotating buffer sets (rot > 1: operands/outputs come from HBM, not from a warm L2/MALL)
static void bench_gemm2(int M, int N, int K, int la, int lb, int flags, int rot) {
    int64_t lda = la == 0 ? K : M, ldb = lb == 0 ? K : N;
    size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
    std::vector<uint16_t> h(std::max(std::max(na, nb), nc));
    for (auto& x : h) x = f2bf(frand());
    std::vector<void*> dA(rot), dB(rot), dC(rot), dZ(rot), dR(rot);
    void* dbias; hipMalloc(&dbias, N * 4); hipMemset(dbias, 0, N * 4);
    for (int r = 0; r < rot; ++r) {
        hipMalloc(&dA[r], na * 2); hipMalloc(&dB[r], nb * 2); hipMalloc(&dC[r], nc * 2); hipMalloc(&dZ[r], nc * 2); hipMalloc(&dR[r], nc * 2);
        hipMemcpy(dA[r], h.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(dB[r], h.data(), nb * 2, hipMemcpyHostToDevice);
        hipMemcpy(dZ[r], h.data(), nc * 2, hipMemcpyHostToDevice); hipMemcpy(dR[r], h.data(), nc * 2, hipMemcpyHostToDevice);
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int it = 24;
    for (int i = -3; i < it; ++i) {
        if (i == 0) hipEventRecord(a, nullptr);
        const int r = (i + 3) % rot;
        vm_gemm_epilogue e = {}; e.alpha = 1.f; e.out_dtype = VM_BF16; e.split_k = 1;
        if (flags & 1) e.bias = (const float*)dbias;
        if (flags & 2) e.act = 1;
        if (flags & 4) e.aux_out = dZ[r];
        if (flags & 8) e.mul_gelu_z = dZ[r];
        if (flags & 16) { e.dropout_p = 0.1f; e.dropout_seed = 1234 + i; }
        if (flags & 32) { e.residual = dR[r]; e.ldr = N; }
        int rc = vm_gemm_bf16(dA[r], lda, la, dB[r], ldb, lb, dC[r], N, M, N, K, &e, nullptr);
        if (rc) { printf("rc=%d %s\n", rc, vm_last_error()); return; }
    }
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
    printf("epi M=%d N=%d K=%d l%d%d flags=%2d rot=%d: %7.1f us  %6.1f TFLOP/s\n", M, N, K, la, lb, flags, rot, ms * 1e3, 2.0 * M * N * K / ms * 1e-9);
    for (int r = 0; r < rot; ++r) { hipFree(dA[r]); hipFree(dB[r]); hipFree(dC[r]); hipFree(dZ[r]); hipFree(dR[r]); }
    hipFree(dbias);
}

static int test_ln(int rows, int cols) {
    std::vector<float> x((size_t)rows * cols), dy(x.size()), g(cols), bta(cols);
    for (auto& v : x) v = bf2f(f2bf(frand() * 2 + 0.3f));
    for (auto& v : dy) v = bf2f(f2bf(frand()));
    for (auto& v : g) v = 1.f + 0.1f * frand();
    for (auto& v : bta) v = 0.1f * frand();
    std::vector<uint16_t> hx(x.size()), hdy(x.size());
    for (size_t i = 0; i < x.size(); ++i) { hx[i] = f2bf(x[i]); hdy[i] = f2bf(dy[i]); }
    void *dx_, *ddy, *dyo, *ddx, *dg, *db, *dmean, *drstd, *dgam, *dbet, *ws;
    hipMalloc(&dx_, hx.size() * 2); hipMalloc(&ddy, hx.size() * 2); hipMalloc(&dyo, hx.size() * 2); hipMalloc(&ddx, hx.size() * 2);
    hipMalloc(&dg, cols * 4); hipMalloc(&db, cols * 4); hipMalloc(&dmean, rows * 4); hipMalloc(&drstd, rows * 4);
    hipMalloc(&dgam, cols * 4); hipMalloc(&dbet, cols * 4); hipMalloc(&ws, vm_layernorm_bwd_ws(rows, cols));
    hipMemcpy(dx_, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(ddy, hdy.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dgam, g.data(), cols * 4, hipMemcpyHostToDevice); hipMemcpy(dbet, bta.data(), cols * 4, hipMemcpyHostToDevice);
    hipMemset(dg, 0, cols * 4); hipMemset(db, 0, cols * 4);
    int rc = vm_layernorm_fwd(dx_, (float*)dgam, (float*)dbet, dyo, (float*)dmean, (float*)drstd, rows, cols, 1e-5f, nullptr);
    rc |= vm_layernorm_bwd(ddy, dx_, (float*)dgam, (float*)dmean, (float*)drstd, ddx, (float*)dg, (float*)db, rows, cols, ws, nullptr);
    if (rc) { printf("ln rc=%d %s\n", rc, vm_last_error()); return 1; }
    hipDeviceSynchronize();
    std::vector<uint16_t> hy(hx.size()), hdx(hx.size()); std::vector<float> hdg(cols), hdb(cols);
    hipMemcpy(hy.data(), dyo, hy.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hdx.data(), ddx, hy.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hdg.data(), dg, cols * 4, hipMemcpyDeviceToHost); hipMemcpy(hdb.data(), db, cols * 4, hipMemcpyDeviceToHost);
    int bad = 0; double e1 = 0, e2 = 0, e3 = 0;
    std::vector<double> rdg(cols, 0), rdb(cols, 0);
    for (int r = 0; r < rows; ++r) {
        double mu = 0, var = 0;
        for (int c = 0; c < cols; ++c) mu += x[(size_t)r * cols + c];
        mu /= cols;
        for (int c = 0; c < cols; ++c) { double d = x[(size_t)r * cols + c] - mu; var += d * d; }
        double rs = 1.0 / sqrt(var / cols + 1e-5);
        double s1 = 0, s2 = 0;
        for (int c = 0; c < cols; ++c) { double xh = (x[(size_t)r * cols + c] - mu) * rs, gy = dy[(size_t)r * cols + c] * g[c]; s1 += gy; s2 += gy * xh; rdg[c] += dy[(size_t)r * cols + c] * xh; rdb[c] += dy[(size_t)r * cols + c]; }
        s1 /= cols; s2 /= cols;
        for (int c = 0; c < cols; ++c) {
            double xh = (x[(size_t)r * cols + c] - mu) * rs;
            double yref = xh * g[c] + bta[c], dxref = rs * (dy[(size_t)r * cols + c] * g[c] - s1 - xh * s2);
            double ey = fabs(yref - bf2f(hy[(size_t)r * cols + c])), ed = fabs(dxref - bf2f(hdx[(size_t)r * cols + c]));
            if (ey > 0.02 + 0.01 * fabs(yref) || ed > 0.02 + 0.01 * fabs(dxref)) ++bad;
            if (ey > e1) e1 = ey; if (ed > e2) e2 = ed;
        }
    }
    for (int c = 0; c < cols; ++c) { double e = fmax(fabs(rdg[c] - hdg[c]), fabs(rdb[c] - hdb[c])); if (e > e3) e3 = e; if (e > 1e-2 + 1e-3 * fabs(rdg[c])) ++bad; }
    printf("layernorm rows=%d cols=%d  max|dy|=%.3g max|ddx|=%.3g max|dgamma,dbeta|=%.3g bad=%d %s\n", rows, cols, e1, e2, e3, bad, bad ? "FAIL" : "ok");
    return bad != 0;
}

// A/B of one integer switch of the library on one shape with one set of (rotating) buffers: A, B, A, B
static int g_pipe_a = 0, g_pipe_b = 1;
static const char* g_ab_env = "VM_GEMM_VARIANT";  // the switch bench_ab toggles (VM_GEMM_VARIANT, VM_GEMM_GROUPW, ...)
static void bench_ab(int M, int N, int K, int la, int lb, int flags, int split) {
    const int rot = 4;
    int64_t lda = la == 0 ? K : M, ldb = lb == 0 ? K : N;
    size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
    std::vector<uint16_t> h(std::max(std::max(na, nb), nc));
    uint32_t x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = f2bf(((x >> 8) & 0xffff) / 32768.f - 1.f); }
    std::vector<void*> dA(rot), dB(rot), dC(rot), dZ(rot);
    void *dbias, *ws = nullptr; hipMalloc(&dbias, N * 4); hipMemset(dbias, 0, N * 4);
    if (split > 1) hipMalloc(&ws, (size_t)split * nc * 4);
    for (int r = 0; r < rot; ++r) {
        hipMalloc(&dA[r], na * 2); hipMalloc(&dB[r], nb * 2); hipMalloc(&dC[r], nc * (split > 1 ? 4 : 2)); hipMalloc(&dZ[r], nc * 2);
        hipMemcpy(dA[r], h.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(dB[r], h.data(), nb * 2, hipMemcpyHostToDevice);
        hipMemcpy(dZ[r], h.data(), nc * 2, hipMemcpyHostToDevice);
        if (split > 1) hipMemset(dC[r], 0, nc * 4);
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float us[2][2];
    for (int rep = 0; rep < 2; ++rep) for (int pipe = 0; pipe < 2; ++pipe) {
        { char pb[4]; snprintf(pb, 4, "%d", pipe ? g_pipe_b : g_pipe_a); setenv(g_ab_env, pb, 1); vm_reload_env(); }
        const int it = 16;
        for (int i = -2; i < it; ++i) {
            if (i == 0) hipEventRecord(a, nullptr);
            const int r = (i + 2) % rot;
            vm_gemm_epilogue e = {}; e.alpha = 1.f; e.out_dtype = split > 1 ? VM_F32 : VM_BF16; e.split_k = split; e.accumulate = split > 1;
            if (split > 1) { e.workspace = ws; e.workspace_bytes = (size_t)split * nc * 4; }
            if (flags & 1) e.bias = (const float*)dbias;
            if (flags & 2) e.act = 1;
            if (flags & 4) e.aux_out = dZ[r];
            if (flags & 8) e.mul_gelu_z = dZ[r];
            if (flags & 16) { e.dropout_p = 0.1f; e.dropout_seed = 1234 + i; }
            if (flags & 32) { e.residual = dZ[r]; e.ldr = N; }
            int rc = vm_gemm_bf16(dA[r], lda, la, dB[r], ldb, lb, dC[r], N, M, N, K, &e, nullptr);
            if (rc) { printf("rc=%d %s\n", rc, vm_last_error()); return; }
        }
        hipEventRecord(b, nullptr); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); us[rep][pipe] = ms / it * 1e3f;
    }
    const double gf = 2.0 * M * N * K * 1e-6;
    printf("ab M=%d N=%d K=%d l%d%d flags=%2d split=%2d: pipe0 %7.1f %7.1f us (%6.1f TF)  pipe1 %7.1f %7.1f us (%6.1f TF)  ratio %.3f\n", M, N, K, la, lb, flags, split,
           us[0][0], us[1][0], gf / std::min(us[0][0], us[1][0]), us[0][1], us[1][1], gf / std::min(us[0][1], us[1][1]),
           std::min(us[0][0], us[1][0]) / std::min(us[0][1], us[1][1]));
    for (int r = 0; r < rot; ++r) { hipFree(dA[r]); hipFree(dB[r]); hipFree(dC[r]); hipFree(dZ[r]); }
    hipFree(dbias); if (ws) hipFree(ws);
}

// grouped weight gradients: the 128 x 128-tile kernel (VM_WGRAD_P8=0) against the 256 x 256-tile kernel (2 / 4 barrier pairs per K-tile) on the
// linears of `layers` transformer layers: results compared element by element (same MFMA, same K order: expected bit-identical), then timed
static int bench_wgrad(int rows, const std::vector<std::pair<int, int>>& lin, int layers, int overwrite, const char* tag) {
    std::vector<vm_wgrad_problem> pr;
    std::vector<void*> frees;
    std::vector<std::pair<float*, size_t>> outs;      // (dW | db, floats)
    uint32_t x = 777u;
    for (int l = 0; l < layers; ++l) for (auto& nk : lin) {
        const int n_out = nk.first, k_in = nk.second;
        const int64_t ld_dy = (n_out + 7) / 8 * 8;
        std::vector<uint16_t> hy((size_t)rows * ld_dy), hx((size_t)rows * k_in);
        for (auto& v : hy) { x = x * 1664525u + 1013904223u; v = f2bf((((x >> 8) & 0xffff) / 32768.f - 1.f) * 0.05f); }
        for (auto& v : hx) { x = x * 1664525u + 1013904223u; v = f2bf(((x >> 8) & 0xffff) / 32768.f - 1.f); }
        void *dY, *dX; float *dW, *db;
        hipMalloc(&dY, hy.size() * 2); hipMalloc(&dX, hx.size() * 2); hipMalloc((void**)&dW, (size_t)n_out * k_in * 4); hipMalloc((void**)&db, (size_t)n_out * 4);
        hipMemcpy(dY, hy.data(), hy.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dX, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
        frees.push_back(dY); frees.push_back(dX); frees.push_back(dW); frees.push_back(db);
        outs.push_back({dW, (size_t)n_out * k_in}); outs.push_back({db, (size_t)n_out});
        vm_wgrad_problem q = {}; q.dY = dY; q.ld_dy = ld_dy; q.X = dX; q.ld_x = k_in; q.dW = dW; q.ld_dw = k_in; q.db = db; q.rows = rows; q.n_out = n_out; q.k_in = k_in;
        q.alpha_dev = nullptr; q.overwrite = overwrite;
        pr.push_back(q);
    }
    auto fill = [&](float v) { for (auto& o : outs) { std::vector<float> h(o.second, v); hipMemcpy(o.first, h.data(), o.second * 4, hipMemcpyHostToDevice); } };
    auto fetch = [&]() { std::vector<std::vector<float>> r; for (auto& o : outs) { std::vector<float> h(o.second); hipMemcpy(h.data(), o.first, o.second * 4, hipMemcpyDeviceToHost); r.push_back(h); } return r; };
    std::vector<int> modes = {0, 2, 4};
    if (const char* only = getenv("VM_PROBE_MODE")) modes = {atoi(only)};       // one kernel only (rocprofv3 --pmc workload)
    std::vector<std::vector<float>> ref;
    int bad = 0;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode : modes) {
        { char pb[4]; snprintf(pb, 4, "%d", mode); setenv("VM_WGRAD_P8", pb, 1); vm_reload_env(); }
        fill(overwrite ? 123.f : 0.25f);                       // overwrite must erase the old contents; accumulate must keep them
        int rc = vm_wgrad_grouped(pr.data(), (int)pr.size(), nullptr);
        if (rc) { printf("wgrad rc=%d %s\n", rc, vm_last_error()); return 1; }
        hipDeviceSynchronize();
        auto got = fetch();
        if (mode == 0) {
            if (overwrite) for (auto& v : got) for (auto& e : v) e -= 123.f;      // the old kernel always accumulates
            ref = got;
        } else if (!ref.empty()) {
            size_t nd = 0; double md = 0;
            for (size_t t = 0; t < got.size(); ++t) for (size_t e = 0; e < got[t].size(); ++e) {
                const double d = fabs((double)got[t][e] - ref[t][e]);
                if (d > 0) ++nd;
                if (d > md) md = d;
                if (d > 1e-3 + 1e-4 * fabs(ref[t][e])) ++bad;
            }
            printf("wgrad %s p8w(%d) vs 128-tile kernel: %zu differing elements, max |diff| %.3g\n", tag, mode, nd, md);
        }
        double flop = 0; for (auto& q : pr) flop += 2.0 * q.rows * (double)q.n_out * q.k_in;
        hipEventRecord(a, nullptr);
        const int it = 6;
        for (int i = 0; i < it; ++i) vm_wgrad_grouped(pr.data(), (int)pr.size(), nullptr);
        hipEventRecord(b, nullptr); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
        printf("wgrad %s rows=%d problems=%zu overwrite=%d VM_WGRAD_P8=%d: %8.1f us  %7.1f TFLOP/s\n", tag, rows, pr.size(), overwrite, mode, ms * 1e3, flop / ms * 1e-9);
    }
    for (void* f : frees) hipFree(f);
    printf("wgrad %s: bad=%d %s\n", tag, bad, bad ? "FAIL" : "ok");
    return bad != 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && (!strcmp(argv[1], "wgrad") || !strcmp(argv[1], "wgrad1"))) {
        int fails = 0;
        const std::vector<std::pair<int, int>> enc = {{2304, 768}, {768, 768}, {3072, 768}, {768, 3072}};
        const std::vector<std::pair<int, int>> dec = {{2304, 768}, {768, 768}, {768, 768}, {768, 768}, {3072, 768}, {768, 3072}};
        if (!strcmp(argv[1], "wgrad1")) {           // gpu_probe.bin wgrad1: the two production groups only (VM_PROBE_MODE picks the kernel)
            fails += bench_wgrad(12608, enc, 2, 1, "enc2");
            fails += bench_wgrad(8192, dec, 2, 1, "dec2");
            return fails;
        }
        fails += bench_wgrad(12608, enc, 2, 0, "enc2");
        fails += bench_wgrad(12608, enc, 2, 1, "enc2");
        fails += bench_wgrad(8192, dec, 2, 0, "dec2");
        fails += bench_wgrad(8192, dec, 2, 1, "dec2");
        fails += bench_wgrad(12608, enc, 1, 1, "enc1");
        fails += bench_wgrad(12608, enc, 4, 1, "enc4");
        fails += bench_wgrad(8192, {{30522, 768}}, 1, 0, "lmhead");
        fails += bench_wgrad(12608, {{18432, 768}}, 1, 1, "crosskv");
        setenv("VM_WGRAD_P8_MIN", "1", 1);
        fails += bench_wgrad(320, {{1000, 512}, {520, 256}}, 2, 0, "ragged");
        fails += bench_wgrad(64, {{256, 256}, {130, 256}}, 1, 1, "tiny");
        printf("wgrad fails=%d\n", fails);
        return fails;
    }
    if (argc >= 8 && !strcmp(argv[1], "one")) {     // gpu_probe.bin one M N K la lb split     (production kernel only: rocprofv3 --pmc workload)
        bench_gemm(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]));
        return 0;
    }
    if (argc >= 8 && !strcmp(argv[1], "bench")) {   // gpu_probe.bin bench M N K la lb split   (for rocprofv3 --pmc runs)
        for (int dbg = 0; dbg < 4; ++dbg) {
            char b[4]; snprintf(b, 4, "%d", dbg); setenv("VM_GEMM_DEBUG", b, 1); vm_reload_env();
            printf("dbg=%d ", dbg);
            bench_gemm(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]));
        }
        return 0;
    }
    // gpu_probe.bin ab ENV A B        the same for any integer switch of the library, e.g.  ab VM_GEMM_VARIANT 0 7
    if (argc >= 5 && !strcmp(argv[1], "ab")) {
        int fails = 0;
        g_ab_env = argv[2]; g_pipe_a = atoi(argv[3]); g_pipe_b = atoi(argv[4]);
        printf("A = %s=%d (pipe0 columns), B = %s=%d (pipe1 columns)\n", g_ab_env, g_pipe_a, g_ab_env, g_pipe_b);
        { char pb[4]; snprintf(pb, 4, "%d", g_pipe_b); setenv(g_ab_env, pb, 1); }
        const bool variant_ab = !strcmp(g_ab_env, "VM_GEMM_VARIANT");
        for (int variant = 0; variant <= 4; variant += 4) {
            if (variant_ab) { if (variant) break; vm_reload_env(); variant = g_pipe_b; }      // correctness of setting B itself
            else { char b[4]; snprintf(b, 4, "%d", variant); setenv("VM_GEMM_VARIANT", b, 1); vm_reload_env(); }
            for (int la = 0; la < (variant == 4 ? 1 : 2); ++la) for (int lb = 0; lb < 2; ++lb) {
                fails += test_gemm(200, 136, 192, la, lb, 1, false);
                fails += test_gemm(333, 97, 128, la, lb, 1, true);
                fails += test_gemm(130, 140, 64, la, lb, 1, true);
                fails += test_gemm(700, 260, 448, la, lb, 1, true);
            }
            if (variant != 4) fails += test_gemm(256, 256, 1024, 1, 1, 4, true);
            fails += test_gemm(1000, 768, 768, 0, 0, 1, false);
            if (variant_ab) break;
        }
        if (!variant_ab) unsetenv("VM_GEMM_VARIANT");
        printf("correctness with setting B: fails=%d\n", fails);
        struct { int M, N, K, la, lb, flags, split; } cs[] = {
            {12608, 2304, 768, 0, 0, 1, 1}, {12608, 3072, 768, 0, 0, 7, 1}, {12608, 768, 3072, 0, 0, 49, 1}, {12608, 768, 768, 0, 0, 49, 1},
            {8192, 2304, 768, 0, 0, 1, 1}, {8192, 768, 768, 0, 0, 49, 1}, {8192, 3072, 768, 0, 0, 7, 1}, {8192, 768, 3072, 0, 0, 49, 1},
            {8192, 30528, 768, 0, 0, 1, 1},
            {12608, 3072, 768, 0, 1, 8, 1}, {12608, 768, 3072, 0, 1, 0, 1}, {12608, 768, 768, 0, 1, 0, 1}, {12608, 768, 2304, 0, 1, 0, 1},
            {8192, 768, 768, 0, 1, 0, 1}, {8192, 768, 30528, 0, 1, 0, 1},
            {2304, 768, 12608, 1, 1, 0, 4}, {3072, 768, 12608, 1, 1, 0, 3}, {768, 3072, 12608, 1, 1, 0, 3}, {768, 768, 8192, 1, 1, 0, 14},
            {30528, 768, 8192, 1, 1, 0, 1},
        };
        for (auto& c : cs) bench_ab(c.M, c.N, c.K, c.la, c.lb, c.flags, c.split);
        return fails;
    }
    if (argc >= 8 && !strcmp(argv[1], "onef")) {    // gpu_probe.bin onef M N K la lb flags   (one shape with epilogue options: rocprofv3 --pmc workload)
        bench_gemm2(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), 6);
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "epiab")) {      // epilogue operands requested in front of the last K-tile (0) vs behind the main loop (VM_GEMM_DEBUG=4)
        struct { int M, N, K, la, lb, flags; } cs[] = {
            {12608, 3072, 768, 0, 1, 8}, {8192, 3072, 768, 0, 1, 8}, {12608, 3072, 768, 0, 0, 7}, {8192, 3072, 768, 0, 0, 7},
            {12608, 768, 3072, 0, 0, 33}, {12608, 768, 768, 0, 0, 33}, {8192, 768, 768, 0, 0, 49}, {8192, 768, 3072, 0, 0, 49},
            {12608, 2304, 768, 0, 0, 1}, {8192, 2304, 768, 0, 0, 1}, {12608, 768, 2304, 0, 1, 0}, {8192, 768, 768, 0, 1, 0},
        };
        for (int rep = 0; rep < 2; ++rep)
            for (auto& c : cs) for (int dbg : {0, 4}) {
                { char b[4]; snprintf(b, 4, "%d", dbg); setenv("VM_GEMM_DEBUG", b, 1); vm_reload_env(); }
                printf("dbg%d ", dbg);
                bench_gemm2(c.M, c.N, c.K, c.la, c.lb, c.flags, 6);
            }
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "epi")) {
        struct { int M, N, K, la, lb, flags; } cs[] = {
            {12608, 3072, 768, 0, 0, 0}, {12608, 3072, 768, 0, 0, 1}, {12608, 3072, 768, 0, 0, 3}, {12608, 3072, 768, 0, 0, 7}, {12608, 3072, 768, 0, 0, 5},
            {12608, 3072, 768, 0, 1, 0}, {12608, 3072, 768, 0, 1, 8},
            {12608, 768, 768, 0, 0, 0}, {12608, 768, 768, 0, 0, 1}, {12608, 768, 768, 0, 0, 33}, {8192, 768, 768, 0, 0, 49},
            {12608, 768, 3072, 0, 0, 0}, {12608, 768, 3072, 0, 0, 33}, {12608, 2304, 768, 0, 0, 0}, {12608, 2304, 768, 0, 0, 1},
            {12608, 768, 768, 0, 1, 0}, {8192, 768, 768, 0, 1, 0}, {12608, 768, 3072, 0, 1, 0}, {12608, 768, 2304, 0, 1, 0}, {12608, 1536, 768, 0, 0, 1},
            {8192, 2304, 768, 0, 0, 1}, {8192, 768, 3072, 0, 0, 49}, {12608, 2304, 768, 0, 1, 0},
        };
        const int variants[] = {0, 4, 8, 9};
        for (auto& c : cs) for (int variant : variants) {
            { char b[4]; snprintf(b, 4, "%d", variant); setenv("VM_GEMM_VARIANT", b, 1); vm_reload_env(); }
            printf("v%d ", variant);
            bench_gemm2(c.M, c.N, c.K, c.la, c.lb, c.flags, 6);
        }
        return 0;
    }
    short* d; hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16 with lane l -> &lds[4l] (values are source element indices):\n");
    for (int l = 0; l < 64; ++l) { printf("  lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : ""); }
    int fails = 0;
    for (int variant = 0; variant <= 8; variant += 4) {
        { char b[4]; snprintf(b, 4, "%d", variant); setenv("VM_GEMM_VARIANT", b, 1); vm_reload_env(); }
        printf("---- VM_GEMM_VARIANT=%d correctness\n", variant);
        for (int la = 0; la < (variant == 4 ? 1 : 2); ++la) for (int lb = 0; lb < 2; ++lb) {
            fails += test_gemm(200, 136, 192, la, lb, 1, false);
            fails += test_gemm(128, 128, 64, la, lb, 1, true);
            fails += test_gemm(333, 97, 104, la, lb, 1, true);
            fails += test_gemm(700, 260, 448, la, lb, 1, true);
        }
        if (variant != 4) fails += test_gemm(256, 256, 1024, 1, 1, 4, true);
        fails += test_gemm(1000, 768, 768, 0, 0, 1, false);
    }
    unsetenv("VM_GEMM_VARIANT"); vm_reload_env();
    fails += test_ln(1000, 768);
    fails += test_ln(37, 64);
    fails += test_ln(50, 1664);
    struct { int M, N, K, la, lb, split; } shapes[] = {
        {12608, 2304, 768, 0, 0, 1}, {12608, 768, 768, 0, 0, 1}, {12608, 3072, 768, 0, 0, 1}, {12608, 768, 3072, 0, 0, 1},
        {8192, 2304, 768, 0, 0, 1}, {8192, 768, 768, 0, 0, 1}, {8192, 3072, 768, 0, 0, 1}, {8192, 768, 3072, 0, 0, 1},
        {12608, 1536, 768, 0, 0, 1}, {8192, 30528, 768, 0, 0, 1},
        {12608, 768, 2304, 0, 1, 1}, {12608, 768, 768, 0, 1, 1}, {12608, 768, 3072, 0, 1, 1}, {12608, 3072, 768, 0, 1, 1},
        {8192, 768, 30528, 0, 1, 1},
        {2304, 768, 12608, 1, 1, 2}, {768, 768, 12608, 1, 1, 8}, {768, 768, 12608, 1, 1, 14}, {3072, 768, 12608, 1, 1, 2}, {3072, 768, 12608, 1, 1, 4},
        {768, 3072, 12608, 1, 1, 4}, {30528, 768, 8192, 1, 1, 1}, {8192, 8192, 8192, 0, 0, 1}, {8192, 8192, 8192, 1, 1, 1},
    };
    for (auto& sh : shapes) for (int variant = 0; variant <= 4; variant += 4) {
        { char b[4]; snprintf(b, 4, "%d", variant); setenv("VM_GEMM_VARIANT", b, 1); vm_reload_env(); }
        printf("v%d ", variant);
        bench_gemm(sh.M, sh.N, sh.K, sh.la, sh.lb, sh.split);
    }
    printf("fails=%d\n", fails);
    return fails;
}
