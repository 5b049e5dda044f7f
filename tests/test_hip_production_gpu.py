"""GPU parity at the PRODUCTION shapes of BASELINE.json configs[1] (ViT-B/16 + 12-layer decoder, B=64, L=128, S=197, V=30522).

The small-shape suites (test_hip_kernels_gpu / test_hip_models_gpu) never reach the launch configurations the training step
actually runs: the 160x128 / 256x256 GEMM tiles, column-group tile order, the Vp = 30528 LM head, the wgrad split-K slabs,
the all-layer cross K|V projection, head-resident attention at B*H = 768.  Here every GEMM shape of the step runs once per
tile variant against fp32 math on the same bf16-rounded operands, one full-width decoder layer and the 12-layer cross
K|V path run against the CPU oracle, and the contrastive losses run at B = 2048, D = 768 (BASELINE configs[2]).

Every test prints its measured error (``-s`` shows it; tools/collect_errors.py gathers them into profiles/) and asserts a
bound of about 2x that measurement -- see DESIGN.md §4 for why each bound is what it is.
"""
import os

import pytest
import torch

import golden_recipes as R
from test_hip_models_gpu import build_decoder, cosine, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def report(name, **vals):
    print(f"[parity] {name}: " + " ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in vals.items()), flush=True)


@pytest.fixture
def gemm_variant():
    """sets VM_GEMM_VARIANT (tile) for the duration of a test and restores the default (the cost model) afterwards"""
    from vilmedic_amd._lib import lib

    def setv(v):
        if v is None:
            os.environ.pop("VM_GEMM_VARIANT", None)
        else:
            os.environ["VM_GEMM_VARIANT"] = str(v)
        lib().vm_reload_env()
    yield setv
    os.environ.pop("VM_GEMM_VARIANT", None)
    lib().vm_reload_env()


def _rand_bf16(rows, cols, seed, scale=1.0):
    g = torch.Generator(device=dev()).manual_seed(seed)       # generated on the device: 232 M elements for the largest operand
    return (torch.randn(rows, cols, generator=g, device=dev()) * scale).to(BF)


def _check_sampled_rows_f64(C, A, B, la, lb, rows):
    """independent of the GPU fp32 matmul: a few output rows recomputed on the CPU in float64"""
    Af = (A if la == 0 else A.t()).double().cpu()
    Bf = (B if lb == 0 else B.t()).double().cpu()
    ref = Af[rows] @ Bf.t()
    return (C[rows].double().cpu() - ref).abs().max().item(), ref.abs().max().item()


# (M, N, K, a_layout, b_layout, epilogue, split_k): the launches of one C2 training step (profiles/r01_d_kernel_shape_breakdown.txt)
FWD = [
    (12608, 2304, 768, 0, 0, "bias", 1),          # ViT QKV projection
    (12608, 768, 768, 0, 0, "bias_res", 1),       # ViT attention output + residual
    (12608, 3072, 768, 0, 0, "bias_gelu_z", 1),   # ViT MLP up + erf-GELU (+ pre-activation side output)
    (12608, 768, 3072, 0, 0, "bias_res", 1),      # ViT MLP down + residual
    (12608, 18432, 768, 0, 0, "bias", 1),         # cross-attention K|V of all 12 decoder layers
    (8192, 2304, 768, 0, 0, "bias", 1),           # decoder QKV
    (8192, 30522, 768, 0, 0, "bias", 1),          # tied LM head, V = 30522 (padded leading dim 30528)
]
DGRAD = [
    (12608, 768, 2304, 0, 1, "plain", 1),
    (12608, 3072, 768, 0, 1, "gelu_grad", 1),
    (12608, 768, 3072, 0, 1, "plain", 1),
    (12608, 768, 18432, 0, 1, "plain", 1),
    (8192, 768, 30528, 0, 1, "plain", 1),         # LM-head dgrad: contraction over the padded vocabulary
]
WGRAD = [
    (2304, 768, 12608, 1, 1, "acc", 4),
    (3072, 768, 12608, 1, 1, "acc", 3),
    (768, 3072, 12608, 1, 1, "acc", 3),
    (768, 768, 8192, 1, 1, "acc", 14),
    (30528, 768, 8192, 1, 1, "acc", 1),
    (18432, 768, 12608, 1, 1, "acc", 1),
]


def _run_gemm_case(M, N, K, la, lb, epi, split, seed=0):
    from vilmedic_amd import ops
    A = _rand_bf16(M if la == 0 else K, K if la == 0 else M, seed + 1)
    Bm = _rand_bf16(N if lb == 0 else K, K if lb == 0 else N, seed + 2, scale=0.05)
    Af = (A if la == 0 else A.t()).float()
    Bf = (Bm if lb == 0 else Bm.t()).float()
    ref = Af @ Bf.t()
    ldc = (N + 7) // 8 * 8
    kw = {}
    if epi == "acc":
        C = torch.full((M, ldc), 0.5, dtype=torch.float32, device=dev())
        ops.gemm(A, la, Bm, lb, C, M, N, K, accumulate=True, split_k=split)
        ref = ref + 0.5
        got = C[:, :N]
        err = (got - ref).abs()
        bound = 2e-3 + 2e-5 * ref.abs()           # fp32 accumulation of exact bf16 products; only the summation order differs
    else:
        bias = torch.randn(N, device=dev()) if "bias" in epi else None
        res = _rand_bf16(M, N, seed + 3) if "res" in epi else None
        z = torch.empty(M, ldc, dtype=BF, device=dev()) if "gelu_z" in epi else None
        zin = _rand_bf16(M, N, seed + 4) if epi == "gelu_grad" else None
        C = torch.empty(M, ldc, dtype=BF, device=dev())
        ops.gemm(A, la, Bm, lb, C, M, N, K, bias=bias, act=1 if "gelu_z" in epi else 0, aux_out=z, mul_gelu_z=zin, residual=res)
        C2 = torch.empty_like(C)          # race screen: a second launch must reproduce the first bit for bit (LDS-DMA ordering bugs do not)
        ops.gemm(A, la, Bm, lb, C2, M, N, K, bias=bias, act=1 if "gelu_z" in epi else 0, aux_out=z, mul_gelu_z=zin, residual=res)
        assert torch.equal(C[:, :N], C2[:, :N]), "two launches of the same GEMM differ"
        if bias is not None:
            ref = ref + bias
        if z is not None:
            zerr = (z[:, :N].float() - ref).abs()
            assert bool((zerr <= 8e-3 * ref.abs() + 1e-3).all()), zerr.max()     # one bf16 rounding (2^-8 relative)
            ref = torch.nn.functional.gelu(ref)
        if zin is not None:
            zf = zin.float().requires_grad_(True)
            torch.nn.functional.gelu(zf).sum().backward()
            ref = ref * zf.grad
        if res is not None:
            ref = ref + res.float()
        got = C[:, :N].float()
        err = (got - ref).abs()
        bound = 8e-3 * ref.abs() + 2e-3           # one bf16 rounding of the output (half ulp = 2^-9 relative) + fp32 order noise
    assert bool((err <= bound).all()), (err.max().item(), (err - bound).max().item())
    rows = torch.tensor([0, 1, M // 2, M - 1])
    if epi in ("plain", "acc"):
        e64, mag = _check_sampled_rows_f64(got - (0.5 if epi == "acc" else 0.0), A, Bm, la, lb, rows)
        assert e64 <= (3e-3 if epi == "acc" else 8e-3 * mag + 2e-3), e64
    return err.max().item(), (err / (ref.abs() + 1e-2)).max().item()


@pytest.mark.parametrize("variant", [None, 0, 4, 1, 5])
def test_gemm_forward_shapes_every_tile_variant(variant, gemm_variant):
    """forward (row-major x row-major) launches of the step, cost-model choice and every forced tile variant"""
    gemm_variant(variant)
    worst = 0.0
    cases = FWD[:2] + FWD[6:] if variant == 1 else FWD
    for c in cases:
        mx, rel = _run_gemm_case(*c)
        worst = max(worst, rel)
    report(f"gemm fwd variant={variant}", shapes=len(cases), max_rel_err=worst)


@pytest.mark.parametrize("variant", [None, 0, 4, 5])
def test_gemm_dgrad_shapes(variant, gemm_variant):
    gemm_variant(variant)
    worst = 0.0
    for c in DGRAD:
        mx, rel = _run_gemm_case(*c)
        worst = max(worst, rel)
    report(f"gemm dgrad variant={variant}", shapes=len(DGRAD), max_rel_err=worst)


@pytest.mark.parametrize("variant", [None, 0])
def test_gemm_wgrad_shapes_splitk(variant, gemm_variant):
    gemm_variant(variant)
    worst = 0.0
    for c in WGRAD:
        mx, rel = _run_gemm_case(*c)
        worst = max(worst, mx)
    report(f"gemm wgrad variant={variant}", shapes=len(WGRAD), max_abs_err=worst)


def test_wgrad_wrapper_picks_production_split_and_matches_fp32():
    """ops.wgrad (the call the autograd functions make): its split-K heuristic at the step's shapes, accumulating twice"""
    from vilmedic_amd import ops
    for (N, K, M) in [(2304, 768, 12608), (768, 768, 8192), (3072, 768, 8192)]:
        dY, X = _rand_bf16(M, N, 5, 0.1), _rand_bf16(M, K, 6)
        dW = torch.zeros(N, K, dtype=torch.float32, device=dev())
        ops.wgrad(dY, X, dW)
        ops.wgrad(dY, X, dW)
        ref = 2.0 * (dY.float().t() @ X.float())
        err = (dW - ref).abs().max().item()
        report(f"wgrad {N}x{K}x{M}", max_abs_err=err, ref_max=ref.abs().max().item())
        assert err <= 2e-3 + 2e-5 * ref.abs().max().item()


def test_grouped_param_grads_match_fp32_at_layer_shapes():
    """ops.param_grads / vm_wgrad_grouped: the weight AND bias gradients of one ViT layer (rows = 12608) and of the LM head
    (V = 30522 of 30528 padded columns, device-scalar alpha) as grouped launches without split-K, accumulated on top of existing
    gradients; the same parameter queued twice must not share a launch"""
    from vilmedic_amd import ops
    rows = 12608
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    probs = []
    for i, (N, K) in enumerate(shapes):
        dY, X = _rand_bf16(rows, N, 40 + i, 0.1), _rand_bf16(rows, K, 50 + i)
        dW = torch.full((N, K), 0.25, dtype=torch.float32, device=dev())
        db = torch.full((N,), -1.0, dtype=torch.float32, device=dev())
        probs.append((dY, X, dW, db))
    for dY, X, dW, db in probs:
        ops.param_grads(dY, X, dW, db)
    ops.join_side()
    dY0, X0, dW0, db0 = probs[1]
    ops.param_grads(dY0, X0, dW0, db0)          # same parameter again (a module used twice in one graph): separate launch
    ops.param_grads(dY0, X0, dW0, db0)
    ops.join_side()
    torch.cuda.synchronize()
    worst_w = worst_b = 0.0
    for i, (dY, X, dW, db) in enumerate(probs):
        k = 3.0 if i == 1 else 1.0
        rw = 0.25 + k * (dY.float().t() @ X.float())
        rb = -1.0 + k * dY.float().sum(0)
        # fp32 accumulation of exact bf16 products over 12608 rows: only the summation order differs from the reference product
        worst_w = max(worst_w, ((dW - rw).abs().max() / rw.abs().max()).item())
        worst_b = max(worst_b, ((db - rb).abs().max() / rb.abs().max()).item())
    report("grouped wgrad ViT layer", max_err_over_max_w=worst_w, max_err_over_max_b=worst_b)
    assert worst_w <= 2e-5 and worst_b <= 2e-5
    # LM head: rows = 8192, dlogits [8192, 30528] of which 30522 columns count, alpha on the device
    M, V, Vp, D = 8192, 30522, 30528, 768
    dl = _rand_bf16(M, Vp, 60, 0.05)
    dl[:, V:] = 0
    h = _rand_bf16(M, D, 61)
    gE = torch.zeros(Vp, D, dtype=torch.float32, device=dev())
    gb = torch.zeros(V, dtype=torch.float32, device=dev())
    sc = torch.tensor(0.5, device=dev())
    ops.param_grads(dl, h, gE[:V], gb, alpha_dev=sc, cols=V)
    ops.join_side()
    torch.cuda.synchronize()
    rw = 0.5 * (dl[:, :V].float().t() @ h.float())
    rb = 0.5 * dl[:, :V].float().sum(0)
    ew, eb = (gE[:V] - rw).abs().max().item(), (gb - rb).abs().max().item()
    report("grouped wgrad LM head", max_abs_err_w=ew, max_abs_err_b=eb, ref_max=rw.abs().max().item())
    assert ew <= 2e-3 + 2e-5 * rw.abs().max().item() and eb <= 2e-3 + 2e-5 * rb.abs().max().item()
    assert gE[V:].abs().max().item() == 0.0


FULL = dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, vocab_size=1024, max_position_embeddings=130,
            layer_norm_eps=1e-5, bos_token_id=0, pad_token_id=1, eos_token_id=2)


def _decoder_vs_oracle(cfg, B, L, S, seed, masked_keys, grad_names):
    from oracle import torch_ref as O
    dec, st = build_decoder(cfg, seed, std=0.03)
    ids, am = R.make_reports(B, L, cfg["vocab_size"], seed=seed)
    gen = torch.Generator().manual_seed(seed + 1)
    enc = torch.randn(B, S, cfg["hidden_size"], generator=gen).to(BF).float()       # bf16-representable: both sides see the same numbers
    enc_mask = torch.ones(B, S, dtype=torch.bool)
    if masked_keys:
        enc_mask[1::2, S - masked_keys:] = False
    enc_d = enc.to(dev()).to(BF).requires_grad_(True)
    dec.train()
    out = dec(input_ids=ids.to(dev()), attention_mask=am.to(dev()), encoder_outputs=enc_d, encoder_attention_mask=enc_mask.to(dev()))
    out["loss"].backward()
    st_r = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    enc_r = enc.clone().requires_grad_(True)
    ref_loss, ref_logits = O.decoder_forward(ids, am, enc_r, enc_mask, st_r, cfg)
    ref_loss.backward()
    res = {"loss_err": abs(out["loss"].item() - ref_loss.item()), "loss": ref_loss.item()}
    lerr = (out["logits"].float().cpu() - ref_logits.detach()).abs()
    res["logits_max_err"], res["logits_mean_err"] = lerr.max().item(), lerr.mean().item()
    res["logits_absmax"] = ref_logits.detach().abs().max().item()
    named = dict(dec.decoder.named_parameters())
    worst_cos, worst_rel = 1.0, 0.0
    for n in grad_names:
        got, ref = named[n].grad.float().cpu(), st_r[n].grad
        worst_cos, worst_rel = min(worst_cos, cosine(got, ref)), max(worst_rel, rel_l2(got, ref))
    eg = enc_d.grad.float().cpu()
    res["grad_min_cos"], res["grad_max_rel_l2"] = min(worst_cos, cosine(eg, enc_r.grad)), max(worst_rel, rel_l2(eg, enc_r.grad))
    return res


def test_full_width_decoder_layer_vs_oracle():
    """one decoder layer at the production width and batch (d=768, h=12, ff=3072, B=64, L=128, S=197): QKV 8192x2304x768, the
    head-resident attention kernels at B*H = 768 (causal+padding self-attention, key-masked cross-attention), cross K|V
    12608x1536x768, MLP 8192x3072x768, every dgrad / wgrad at M = 8192 / 12608 -- loss, logits, parameter and encoder
    gradients against the fp32 CPU oracle"""
    cfg = dict(FULL, num_hidden_layers=1)
    p = "bert.encoder.layer.0."
    names = [p + "attention.self.query.weight", p + "attention.self.value.bias", p + "attention.output.dense.weight",
             p + "attention.output.LayerNorm.weight", p + "crossattention.self.query.weight", p + "crossattention.self.key.weight",
             p + "crossattention.self.value.weight", p + "crossattention.output.dense.bias", p + "intermediate.dense.weight",
             p + "output.dense.weight", p + "output.LayerNorm.bias", "bert.embeddings.word_embeddings.weight",
             "bert.embeddings.position_embeddings.weight", "lm_head.bias"]
    r = _decoder_vs_oracle(cfg, 64, 128, 197, 21, masked_keys=40, grad_names=names)
    report("full-width decoder layer B=64", **r)
    assert r["loss_err"] <= 1e-3 * max(1.0, abs(r["loss"]))
    assert r["logits_max_err"] <= 2e-2 + 8e-3 * r["logits_absmax"] and r["logits_mean_err"] <= 4e-3
    assert r["grad_min_cos"] >= 0.9995 and r["grad_max_rel_l2"] <= 2e-2


def test_twelve_layer_cross_kv_all_vs_oracle():
    """12 decoder layers: CrossKVAllFn's single [B*S, 12*2*768] projection, its 12 gradient slots, the one dgrad with an
    18432-deep contraction and the one wgrad -- parameter gradients of the first / middle / last layer's K and V
    projections and the encoder gradient against the fp32 CPU oracle"""
    cfg = dict(FULL, num_hidden_layers=12, vocab_size=512)
    names = []
    for i in (0, 5, 11):
        p = f"bert.encoder.layer.{i}.crossattention.self."
        # (no key.bias: softmax is invariant to a constant added to every key's score, so its true gradient is exactly zero
        #  and a relative comparison of rounding noise means nothing -- it is bounded in absolute terms by the key.weight check)
        names += [p + "key.weight", p + "value.weight", p + "value.bias"]
    r = _decoder_vs_oracle(cfg, 8, 32, 197, 22, masked_keys=17, grad_names=names)
    report("12-layer decoder cross K|V all B=8", **r)
    assert r["loss_err"] <= 2e-3 * max(1.0, abs(r["loss"]))
    # 12 post-LN layers of bf16 activations: ~10 roundings of 2^-9 per layer accumulate as a random walk (measured on MI355X:
    # max 6.4e-2, mean 1.0e-2 on logits of magnitude <= 3.9); bounds = 2x the measurement
    assert r["logits_max_err"] <= 0.13 and r["logits_mean_err"] <= 2e-2
    assert r["grad_min_cos"] >= 0.999 and r["grad_max_rel_l2"] <= 3e-2


def test_vit_b16_layer_batch64_vs_oracle():
    """one ViT-B/16 layer at the production batch (12608 token rows, 768 resident attention heads): features against the CPU oracle"""
    from oracle import torch_ref as O
    from test_hip_models_gpu import build_vit
    cfg = dict(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072, image_size=224, patch_size=16,
               num_channels=3, layer_norm_eps=1e-12)
    enc, st = build_vit(cfg, 23)
    images = R.make_images(64, 224, seed=23)
    enc.eval()
    with torch.no_grad():
        feats = enc(images.to(dev())).float().cpu()
    ref = O.vit_forward(images, st, cfg)
    err = (feats - ref).abs()
    report("ViT-B/16 1 layer B=64", max_err=err.max().item(), mean_err=err.mean().item(), ref_absmax=ref.abs().max().item())
    # bf16 activations through patch embedding, pre-LN layer and final LayerNorm (measured on MI355X: max 5.6e-2, mean 5.2e-3 on
    # features of magnitude <= 5.9); bounds = 2x the measurement
    assert err.max().item() <= 0.11 and err.mean().item() <= 1.1e-2


@pytest.mark.parametrize("kind", ["convirt", "infonce"])
def test_contrastive_losses_global_batch_2048(kind):
    """BASELINE configs[2]: ConVIRT / InfoNCE on the [2048, 768] embeddings of the global batch -- loss, per-row losses and both
    input gradients against the oracle (ref: ConVIRTLoss.py:12-31, InfoNCELoss.py:11-19)"""
    from oracle import torch_ref as O
    from vilmedic_amd.blocks.losses import ConVIRTLoss, InfoNCELoss
    g = torch.Generator().manual_seed(31)
    B, D = 2048, 768
    l = torch.randn(B, D, generator=g)
    v = (0.6 * l + 0.8 * torch.randn(B, D, generator=g))
    if kind == "infonce":
        l, v = l * 0.1, v * 0.1              # raw dot products (InfoNCE applies no normalisation and no tau): off-diagonal ~0.3, diagonal ~4.6
    ld, vd = l.to(dev()).requires_grad_(True), v.to(dev()).requires_grad_(True)
    crit = ConVIRTLoss(tau=0.1, lambda_=0.75) if kind == "convirt" else InfoNCELoss(tau=0.1)
    loss, la, lb = crit(ld, vd)
    loss.backward()
    lr, vr = l.clone().requires_grad_(True), v.clone().requires_grad_(True)
    if kind == "convirt":
        rl, ra, rb = O.convirt_loss(lr, vr, 0.1, 0.75)
    else:
        rl, ra, rb = O.infonce_loss(lr, vr)
    rl.backward()
    r = dict(loss_err=abs(loss.item() - rl.item()), loss=rl.item(),
             rows_err=max((la.float().cpu() - ra.detach()).abs().max().item(), (lb.float().cpu() - rb.detach()).abs().max().item()),
             grad_cos=min(cosine(ld.grad.cpu(), lr.grad), cosine(vd.grad.cpu(), vr.grad)),
             grad_rel=max(rel_l2(ld.grad.cpu(), lr.grad), rel_l2(vd.grad.cpu(), vr.grad)))
    report(f"{kind} loss B=2048 D=768", **r)
    assert r["loss_err"] <= 2e-3 * max(1.0, abs(r["loss"]))
    assert r["rows_err"] <= 5e-2
    assert r["grad_cos"] >= 0.999 and r["grad_rel"] <= 3e-2


def _sim_ref(a, b, normalize, inv_tau, off):
    """fp32 reference of _SimilarityLossFn: row / column losses of the pairs (i, i + off) that exist"""
    ah = a / a.norm(dim=1, keepdim=True).clamp_min(1e-8) if normalize else a
    bh = b / b.norm(dim=1, keepdim=True).clamp_min(1e-8) if normalize else b
    S = ah @ bh.t() * inv_tau
    lo, hi = max(0, -off), min(a.shape[0], b.shape[0] - off)
    idx = torch.arange(lo, hi)
    d = S[idx, idx + off]
    return torch.logsumexp(S, 1)[lo:hi] - d, torch.logsumexp(S, 0)[lo + off:hi + off] - d


@pytest.mark.parametrize("R,C,D,off,normalize", [(2048, 2048, 768, 0, True), (256, 2048, 768, 512, True), (300, 1000, 96, 37, True),
                                                  (77, 50, 40, -5, False), (1000, 300, 264, 0, True),
                                                  (192, 320, 136, 64, False), (320, 192, 72, -64, True)])      # multiples of 64 that are not tiles of 128: the GEMM-tile gradient path's edges
def test_similarity_loss_kernels_rectangular_ragged_offset(R, C, D, off, normalize):
    """csrc/contrastive.hip through _SimilarityLossFn: square (BASELINE configs[2]), a rank's row block of the global similarity
    (256 local rows against 2048 gathered columns, paired column = row + 512), ragged tile edges in every dimension, a negative offset,
    no normalisation; both outputs weighted by random upstream gradients; a second run reproduces the first bit for bit."""
    from vilmedic_amd.blocks.losses.selfsup import _SimilarityLossFn
    g = torch.Generator().manual_seed(R + C + D)
    sc = 1.0 if normalize else 0.15
    a, b = torch.randn(R, D, generator=g) * sc, torch.randn(C, D, generator=g) * sc
    lo, hi = max(0, -off), min(R, C - off)
    b[lo + off:hi + off] += 0.7 * a[lo:hi]                       # the pairs are positively correlated
    wr, wc = torch.rand(hi - lo, generator=g), torch.rand(hi - lo, generator=g)
    inv_tau = 10.0 if normalize else 1.0
    ad, bd = a.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    lr, lc = _SimilarityLossFn.apply(ad, bd, normalize, inv_tau, 1e-8, off)
    ((lr * wr.to(dev())).sum() + (lc * wc.to(dev())).sum()).backward()
    torch.cuda.synchronize()
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rr, rc = _sim_ref(ar, br, normalize, inv_tau, off)
    ((rr * wr).sum() + (rc * wc).sum()).backward()
    res = dict(rows_err=max((lr.cpu() - rr.detach()).abs().max().item(), (lc.cpu() - rc.detach()).abs().max().item()), loss_absmax=rr.abs().max().item(),
               grad_cos=min(cosine(ad.grad.cpu(), ar.grad), cosine(bd.grad.cpu(), br.grad)),
               grad_rel=max(rel_l2(ad.grad.cpu(), ar.grad), rel_l2(bd.grad.cpu(), br.grad)))
    report(f"similarity loss R={R} C={C} D={D} off={off} normalize={normalize}", **res)
    assert lr.shape == rr.shape and lc.shape == rc.shape
    assert res["rows_err"] <= 5e-2 and res["grad_cos"] >= 0.999 and res["grad_rel"] <= 3e-2
    # determinism + the backward of a second forward reuses nothing stale: run again, must reproduce bit for bit
    ad2, bd2 = a.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    lr2, lc2 = _SimilarityLossFn.apply(ad2, bd2, normalize, inv_tau, 1e-8, off)
    ((lr2 * wr.to(dev())).sum() + (lc2 * wc.to(dev())).sum()).backward()
    torch.cuda.synchronize()
    assert torch.equal(lr2, lr) and torch.equal(lc2, lc)
    gerr = max(rel_l2(ad2.grad, ad.grad), rel_l2(bd2.grad, bd.grad))
    assert gerr == 0.0, gerr                         # no atomics, no order-dependent reduction anywhere in the six launches
