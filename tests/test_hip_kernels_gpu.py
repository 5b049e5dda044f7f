"""GPU parity tests of the individual HIP kernels (through the C ABI) against plain-PyTorch fp32 math on the
same bf16-rounded inputs.  Tolerances are stated per test; integer outputs are bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


@pytest.mark.parametrize("la,lb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(200, 136, 192), (128, 128, 64), (333, 97, 104), (1, 8, 8)])
def test_gemm_layouts(la, lb, M, N, K):
    from vilmedic_amd import ops
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    ref = A.float() @ B.float().t()
    pad = lambda n: (n + 7) // 8 * 8
    Ad = torch.zeros(M, pad(K), dtype=BF) if la == 0 else torch.zeros(K, pad(M), dtype=BF)
    Bd = torch.zeros(N, pad(K), dtype=BF) if lb == 0 else torch.zeros(K, pad(N), dtype=BF)
    (Ad[:, :K] if la == 0 else Ad[:, :M]).copy_(A if la == 0 else A.t())
    (Bd[:, :K] if lb == 0 else Bd[:, :N]).copy_(B if lb == 0 else B.t())
    Ad, Bd = Ad.to(dev()), Bd.to(dev())
    C = torch.zeros(M, pad(N), dtype=torch.float32, device=dev())
    ops.gemm(Ad, la, Bd, lb, C, M, N, K)
    torch.testing.assert_close(C[:, :N].cpu(), ref, rtol=1e-4, atol=1e-3)     # fp32 accumulate of exact bf16 products
    Cb = torch.zeros(M, pad(N), dtype=BF, device=dev())
    ops.gemm(Ad, la, Bd, lb, Cb, M, N, K)
    torch.testing.assert_close(Cb[:, :N].float().cpu(), ref, rtol=1e-2, atol=2e-2)   # one bf16 rounding of the output


def test_gemm_epilogues_and_splitk():
    from vilmedic_amd import ops
    M, N, K = 300, 256, 512
    A, B = rnd(M, K, seed=3).to(dev()), rnd(N, K, scale=0.05, seed=4).to(dev())
    bias = torch.randn(N, device=dev())
    res = rnd(M, N, seed=5).to(dev())
    z = torch.empty(M, N, dtype=BF, device=dev())
    y = torch.empty(M, N, dtype=BF, device=dev())
    ops.gemm(A, 0, B, 0, y, M, N, K, bias=bias, act=1, aux_out=z, residual=res)
    zr = A.float() @ B.float().t() + bias
    torch.testing.assert_close(z.float(), zr, rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(y.float(), torch.nn.functional.gelu(zr) + res.float(), rtol=1e-2, atol=3e-2)
    # gelu' multiply (MLP backward epilogue)
    dz = torch.empty(M, N, dtype=BF, device=dev())
    ops.gemm(A, 0, B, 0, dz, M, N, K, mul_gelu_z=z)
    zf = z.float().requires_grad_(True)
    torch.nn.functional.gelu(zf).sum().backward()
    torch.testing.assert_close(dz.float(), (A.float() @ B.float().t()) * zf.grad, rtol=2e-2, atol=3e-2)
    # split-K fp32 accumulate == single pass, and accumulates on top of existing content
    C1 = torch.ones(M, N, dtype=torch.float32, device=dev())
    ops.gemm(A, 0, B, 0, C1, M, N, K, accumulate=True, split_k=4)
    torch.testing.assert_close(C1, A.float() @ B.float().t() + 1.0, rtol=1e-4, atol=1e-3)
    # device-scalar alpha
    sc = torch.tensor([0.25], device=dev())
    C2 = torch.zeros(M, N, dtype=torch.float32, device=dev())
    ops.gemm(A, 0, B, 0, C2, M, N, K, alpha_dev=sc)
    torch.testing.assert_close(C2, 0.25 * (A.float() @ B.float().t()), rtol=1e-4, atol=1e-3)


def test_gemm_dropout_mask_is_reproducible_and_unbiased():
    from vilmedic_amd import ops
    M, N, K = 512, 512, 64
    A, B = torch.ones(M, K, dtype=BF, device=dev()), torch.ones(N, K, dtype=BF, device=dev()) / K
    y = torch.empty(M, N, dtype=BF, device=dev())
    ops.gemm(A, 0, B, 0, y, M, N, K, dropout_p=0.1, dropout_seed=1234)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01
    assert torch.allclose(y[y != 0].float(), torch.tensor(1 / 0.9, device=dev()), rtol=1e-2)
    # the stand-alone mask kernel (used in backward) regenerates the same mask
    ones = torch.ones(M, N, dtype=BF, device=dev())
    m2 = ops.dropout_apply(ones, 0.1, 1234)
    assert torch.equal(m2 != 0, y != 0)


@pytest.mark.parametrize("rows,cols", [(1000, 768), (37, 64), (50, 1664), (3, 2048)])
def test_layernorm_fwd_bwd(rows, cols):
    from vilmedic_amd import ops
    x = rnd(rows, cols, scale=2.0, seed=6).to(dev()).requires_grad_(True)
    g = (1 + 0.1 * torch.randn(cols)).to(dev())
    b = (0.1 * torch.randn(cols)).to(dev())
    gg, gb = torch.zeros(cols, device=dev()), torch.zeros(cols, device=dev())
    y = ops.layer_norm(x, g, b, 1e-5, gg, gb)
    dy = rnd(rows, cols, seed=7).to(dev())
    y.backward(dy)
    xf = x.detach().float().requires_grad_(True)
    gf, bf_ = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xf, (cols,), gf, bf_, 1e-5)
    yr.backward(dy.float())
    torch.testing.assert_close(y.float(), yr, rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(x.grad.float(), xf.grad, rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(gg, gf.grad, rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(gb, bf_.grad, rtol=1e-3, atol=2e-2)


def _attn_ref(q, k, v, H, key_mask, causal):
    B, Lq, D = q.shape
    Lk = k.shape[1]
    dh = D // H
    qh = q.view(B, Lq, H, dh).transpose(1, 2)
    kh = k.view(B, Lk, H, dh).transpose(1, 2)
    vh = v.view(B, Lk, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(2, 3) * dh ** -0.5
    neg = torch.finfo(torch.float32).min
    if key_mask is not None:
        s = s.masked_fill(~key_mask.bool()[:, None, None, :], neg)
    if causal:
        s = s.masked_fill(~torch.tril(torch.ones(Lq, Lk, dtype=torch.bool, device=q.device)), neg)
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Lq, D)


@pytest.mark.parametrize("case", [
    (2, 2, 17, 17, False, False),      # ViT-tiny shaped, ragged tile
    (2, 8, 49, 49, False, False, 96),  # MVQA transformer: d=768, 8 heads -> head_dim 96 (config/MVQA/vqa.yml:39-42)
    (2, 3, 70, 90, False, True, 32),
    (1, 2, 130, 130, True, True, 128),
    (3, 12, 197, 197, False, False),   # ViT-B/16 sequence
    (2, 4, 128, 128, True, True),      # decoder self-attention: causal + padding
    (2, 4, 128, 197, False, True),     # cross-attention with a key-padding mask
    (1, 1, 1, 300, False, True),       # single query row (decode step), > 4 key tiles
    (2, 2, 70, 64, False, False),
    (1, 2, 256, 256, True, False),     # the longest head-resident sequence: every kernel of the three needs > 64 KiB of dynamic LDS
    (1, 2, 250, 256, False, False),
])
def test_attention_fwd_bwd(case):
    from vilmedic_amd import ops
    B, H, Lq, Lk, causal, masked = case[:6]
    dh = case[6] if len(case) > 6 else 64
    D = H * dh
    q = rnd(B, Lq, D, seed=10).to(dev()).requires_grad_(True)
    k = rnd(B, Lk, D, seed=11).to(dev()).requires_grad_(True)
    v = rnd(B, Lk, D, seed=12).to(dev()).requires_grad_(True)
    km = None
    if masked:
        km = torch.ones(B, Lk, dtype=torch.uint8, device=dev())
        km[0, Lk // 2:] = 0
        if B > 1:
            km[1, Lk - 3:] = 0
    o = ops.AttentionFn.apply(q, k, v, km, H, causal, 0.0)
    do = rnd(B, Lq, D, seed=13).to(dev())
    o.backward(do)
    qf, kf, vf = [t.detach().float().requires_grad_(True) for t in (q, k, v)]
    orf = _attn_ref(qf, kf, vf, H, km, causal)
    orf.backward(do.float())
    torch.testing.assert_close(o.float(), orf, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(q.grad.float(), qf.grad, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(k.grad.float(), kf.grad, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(v.grad.float(), vf.grad, rtol=3e-2, atol=3e-2)


def test_attention_fully_masked_row_is_uniform():
    """HF adds finfo.min to masked scores: a row whose keys are ALL masked attends uniformly (softmax of equal values)."""
    from vilmedic_amd import ops
    B, H, Lq, Lk = 1, 1, 5, 9
    q, k, v = [rnd(B, L, 64, seed=s).to(dev()) for L, s in ((Lq, 1), (Lk, 2), (Lk, 3))]
    km = torch.zeros(B, Lk, dtype=torch.uint8, device=dev())
    o = ops.AttentionFn.apply(q, k, v, km, H, False, 0.0)
    torch.testing.assert_close(o.float()[0], v.float()[0].mean(0, keepdim=True).expand(Lq, -1), rtol=2e-2, atol=2e-2)


def test_attention_dropout_statistics_and_backward_consistency():
    from vilmedic_amd import ops
    B, H, L = 2, 2, 64
    D = H * 64
    q = rnd(B, L, D, seed=20).to(dev()).requires_grad_(True)
    k = rnd(B, L, D, seed=21).to(dev()).requires_grad_(True)
    v = torch.ones(B, L, D, dtype=BF, device=dev()).requires_grad_(True)
    ops.manual_seed(7)
    o = ops.AttentionFn.apply(q, k, v, None, H, False, 0.5)
    # with V == 1 each output equals sum_j dropped P_ij; its mean over many rows is ~1
    assert abs(o.float().mean().item() - 1.0) < 0.05
    # finite-difference-free check: dV = P_drop^T dO, so sum(dV) == sum_i dO_i * sum_j P_drop_ij == sum(o * dO)
    do = torch.ones_like(o)
    o.backward(do)
    assert abs(v.grad.float().sum().item() / (o.float() * do.float()).sum().item() - 1) < 2e-2


@pytest.mark.parametrize("nprob,rows,N,K", [(1, 512, 256, 256), (4, 1024, 2304, 768)])      # 128 x 128 grouped kernel / 256 x 256 tiles (>= 64 of them)
def test_param_grads_first_touch_stores_then_accumulates(nprob, rows, N, K):
    """vm_wgrad_problem.overwrite through ops.param_grads: the first launch after the arena reported a zeroing STORES (shown by poisoning the
    buffers behind the tracker's back), the second one accumulates; weight and bias gradients of every problem of the launch"""
    from vilmedic_amd import ops
    g = torch.zeros(nprob * (N * K + N), device=dev())
    dYs = [rnd(rows, N, scale=0.1, seed=40 + i).to(dev()) for i in range(nprob)]
    Xs = [rnd(rows, K, seed=50 + i).to(dev()) for i in range(nprob)]
    dWs = [g[i * (N * K + N):i * (N * K + N) + N * K].view(N, K) for i in range(nprob)]
    dbs = [g[i * (N * K + N) + N * K:(i + 1) * (N * K + N)] for i in range(nprob)]
    ops.flush_param_grads()
    ops.grads_zeroed(g)
    if nprob > 1:                                  # the 256-tile kernel honours the flag (the 128-tile kernel always adds: the flag is a permission, and
        g.fill_(7.0)                               # adding to a clean buffer is the same thing) -- no caller poisons a buffer: a stale value survives only if the launch adds
    for rep in (1, 2):
        ops._side["defer"] = True                  # (outside a backward pass every param_grads() call would flush on its own: one launch per problem)
        try:
            for dY, X, dW, db in zip(dYs, Xs, dWs, dbs):
                ops.param_grads(dY, X, dW, db)
        finally:
            ops._side["defer"] = False
        ops.join_side()
        torch.cuda.synchronize()
        for dY, X, dW, db in zip(dYs, Xs, dWs, dbs):
            ref = dY.float().t() @ X.float()
            torch.testing.assert_close(dW, rep * ref, rtol=1e-4, atol=2e-3 * rep)
            torch.testing.assert_close(db, rep * dY.float().sum(0), rtol=1e-4, atol=2e-3 * rep)


def test_embedding_and_ce():
    from vilmedic_amd import ops
    B, L, V, D = 3, 10, 50, 64
    ids = torch.randint(0, V, (B, L), device=dev())
    ids[0, 3] = 1
    word = torch.randn(V, D, device=dev())
    pos = torch.randn(32, D, device=dev())
    gw, gp = torch.zeros_like(word), torch.zeros_like(pos)
    anchor = torch.zeros(1, device=dev(), requires_grad=True)
    out = ops.embedding(anchor, ids, word, pos, padding_idx=1, g_word=gw, g_pos=gp)
    torch.testing.assert_close(out.float(), (word[ids] + pos[:L]).to(BF).float())
    dout = rnd(B, L, D, seed=3).to(dev())
    out.backward(dout)
    wr = word.clone().requires_grad_(True)
    pr = pos.clone().requires_grad_(True)
    (torch.nn.functional.embedding(ids, wr, padding_idx=1) + pr[:L]).backward(dout.float())
    torch.testing.assert_close(gw, wr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gp, pr.grad, rtol=1e-5, atol=1e-5)
    # shifted CE fused fwd+bwd vs torch (pads included, last position ignored)
    Vp = 56
    logits = torch.zeros(B * L, Vp, dtype=BF, device=dev())
    logits[:, :V] = rnd(B * L, V, scale=2.0, seed=4).to(dev())
    loss_sum = torch.zeros(1, device=dev())
    dl = torch.empty_like(logits)
    from vilmedic_amd._lib import lib, ptr, stream, check
    check(lib().vm_ce_shift_fwd_bwd(ptr(logits), Vp, ptr(ids), B, L, V, ptr(loss_sum), None, ptr(dl), 1.0 / (B * (L - 1)), None, None, 0, None, stream()))
    lf = logits[:, :V].float().view(B, L, V).requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lf[:, :-1].reshape(-1, V), ids[:, 1:].reshape(-1))
    ref.backward()
    torch.testing.assert_close(loss_sum[0] / (B * (L - 1)), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dl[:, :V].float().view(B, L, V), lf.grad, rtol=1e-2, atol=1e-4)
    assert torch.count_nonzero(dl[:, V:]) == 0


@pytest.mark.parametrize("V,Vp,banned,thr", [(30522, 30528, 0, False),      # the production row: fast path + the chunk that touches V
                                              (30522, 30528, 2, True),       # banned columns + top-k threshold: general path
                                              (40003, 40008, 0, False),      # > 32768 columns: the streaming variant
                                              (40003, 40008, 1, True)])
def test_ce_shift_wide_rows_and_filtered_rows_vs_torch(V, Vp, banned, thr):
    """vm_ce_shift_fwd_bwd against torch cross_entropy on the same bf16 logits: register-resident rows (<= 32768 columns) and the
    streaming variant for wider vocabularies, each with and without the SCST filters (banned columns and a per-row top-k threshold,
    below which logits are removed from the distribution); loss sum, per-row log-probability and gradient (bf16 output: rtol 1e-2)."""
    from vilmedic_amd._lib import lib, ptr, stream, check
    import ctypes
    B, L = 2, 6
    g = torch.Generator().manual_seed(V + banned)
    ids = torch.randint(5, V, (B, L), generator=g).to(dev())
    logits = torch.zeros(B * L, Vp, dtype=BF, device=dev())
    logits[:, :V] = (torch.randn(B * L, V, generator=g) * 2.0).to(BF).to(dev())
    ban = [0, 3][:banned]
    thr_t = None
    lf = logits[:, :V].float()
    if banned:
        lf[:, ban] = float("-inf")
    if thr:
        thr_t = torch.topk(lf, 50, dim=-1).values[:, -1].contiguous()
        lf = torch.where(lf < thr_t[:, None], torch.full_like(lf, float("-inf")), lf)
        # the label must survive the filter (SCST only scores tokens it sampled from the filtered distribution)
        best = lf.argmax(-1).view(B, L)
        ids[:, 1:] = best[:, :-1]
    lf = lf.view(B, L, V).requires_grad_(True)
    loss_sum, row_logp, dl = torch.zeros(1, device=dev()), torch.empty(B * L, device=dev()), torch.empty_like(logits)
    barr = (ctypes.c_int32 * max(1, banned))(*ban) if banned else None
    check(lib().vm_ce_shift_fwd_bwd(ptr(logits), Vp, ptr(ids), B, L, V, ptr(loss_sum), ptr(row_logp), ptr(dl), 1.0 / (B * (L - 1)), None,
                                    barr, banned, ptr(thr_t) if thr_t is not None else None, stream()))
    logp = torch.log_softmax(lf[:, :-1], -1).gather(-1, ids[:, 1:, None])[..., 0]
    ref = -logp.mean()
    ref.backward()
    torch.testing.assert_close(loss_sum[0] / (B * (L - 1)), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(row_logp.view(B, L)[:, :-1], logp.detach(), rtol=1e-5, atol=1e-5)
    assert torch.count_nonzero(row_logp.view(B, L)[:, -1]) == 0
    gref = torch.nan_to_num(lf.grad, nan=0.0)
    torch.testing.assert_close(dl[:, :V].float().view(B, L, V), gref, rtol=1e-2, atol=1e-6)
    assert torch.count_nonzero(dl[:, V:]) == 0


def test_layernorm_backward_writes_the_dropout_masked_gradient():
    """vm_layernorm_bwd_partial_dropout: dx is unchanged and dx_dropped == vm_dropout_apply_bf16(dx) with the same (seed, element index)
    mask -- up to one bf16 rounding, because the fused kernel masks the fp32 value before it is rounded -- and the mask is the one the
    GEMM epilogue draws (zeros at exactly the same positions)."""
    from vilmedic_amd._lib import lib, ptr, stream, check
    rows, cols, p, seed = 300, 768, 0.25, 0x1234ABCD5678
    g = torch.Generator().manual_seed(5)
    x, dy, dy2, dres = ((torch.randn(rows, cols, generator=g)).to(BF).to(dev()) for _ in range(4))
    gamma = (torch.rand(cols, generator=g) + 0.5).to(dev())
    mean, rstd = x.float().mean(1), (x.float().var(1, unbiased=False) + 1e-12).rsqrt()
    ws = torch.empty(lib().vm_layernorm_bwd_ws(rows, cols) // 4, device=dev())
    dx0, dx1, dxd = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    check(lib().vm_layernorm_bwd_partial(ptr(dy), ptr(dy2), ptr(dres), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx0), rows, cols, ptr(ws), stream()))
    check(lib().vm_layernorm_bwd_partial_dropout(ptr(dy), ptr(dy2), ptr(dres), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx1), ptr(dxd), p, seed,
                                                 None, rows, cols, ptr(ws), stream()))
    assert torch.equal(dx0, dx1)
    ref = torch.empty_like(x)
    check(lib().vm_dropout_apply_bf16(ptr(dx0), ptr(ref), dx0.numel(), p, seed, None, stream()))
    assert torch.equal(dxd == 0, ref == 0) or ((dxd == 0) != (ref == 0)).sum() <= (dx0 == 0).sum()      # identical mask
    torch.testing.assert_close(dxd.float(), ref.float(), rtol=2 ** -7, atol=1e-6)
    kept = (ref != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.01, kept


def test_adam_matches_torch():
    from vilmedic_amd._lib import lib, ptr, stream, check
    n = 10007
    p = torch.randn(n, device=dev()); g = torch.randn(n, device=dev())
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    m = torch.zeros(n, device=dev()); v = torch.zeros(n, device=dev()); sh = torch.empty(n, dtype=BF, device=dev())
    for step in range(1, 4):
        pr.grad = g.clone()
        opt.step()
        check(lib().vm_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(sh), n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1,
                                 1 - 0.9 ** step, 1 - 0.999 ** step, 1.0, stream()))
        torch.testing.assert_close(p, pr.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sh.float(), p.to(BF).float())


def test_argmax_logsoftmax():
    from vilmedic_amd._lib import lib, ptr, stream, check
    x = torch.randn(7, 1000, device=dev())
    x[2, 5] = x[2, 900] = 50.0     # tie -> lowest index
    out = torch.empty_like(x)
    check(lib().vm_logsoftmax_f32(ptr(x), 1000, ptr(out), 7, 1000, stream()))
    torch.testing.assert_close(out, torch.log_softmax(x, -1), rtol=1e-5, atol=1e-5)
    idx = torch.empty(7, dtype=torch.long, device=dev())
    check(lib().vm_argmax_f32(ptr(x), 1000, ptr(idx), None, 7, 1000, stream()))
    assert torch.equal(idx, x.argmax(-1)) and idx[2] == 5
