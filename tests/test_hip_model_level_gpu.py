"""Model-level GPU parity: the composed ``forward`` of ConVIRT, MVQA and RRG_SCST on the HIP path
against oracle compositions of the pinned pieces (oracle/torch_ref.py: convirt_forward, mvqa_forward, scst_forward).

Integer outputs (MVQA ``answer``, the top-k filtered token set) are compared bit for bit; floating-point outputs at the bf16
tolerances of tests/test_hip_models_gpu.py with the measured error printed.
"""
import pytest
import torch

import golden_recipes as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


TXT = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=211, max_position_embeddings=40,
           layer_norm_eps=1e-5, bos_token_id=0, pad_token_id=1, eos_token_id=2)
RESNET = dict(num_channels=3, embedding_size=16, hidden_sizes=[16, 32], depths=[1, 1], layer_type="basic", hidden_act="relu")


@pytest.mark.parametrize("training", [False, True])
def test_convirt_forward_vs_oracle(training):
    """ConVIRT.forward (ref: conVIRT.py:75-102): text tower + pooler, hfresnet image tower, both projection MLPs (on the bf16 MFMA
    GEMM since round 2), ConVIRTLoss -- loss, loss_l / loss_v, both embeddings; in training mode with forward_batch_size 3 < batch 6
    (per-micro-batch BatchNorm statistics) and gradients of one parameter per sub-module.  (The image tower's last stage normalises
    over 3 values per channel there -- 1 x 1 maps, micro-batch 3 -- which amplifies the bf16 rounding of the projection input in
    its gradient: the CNN gradient gets the looser of the two bounds.)"""
    from oracle import torch_ref as O
    from vilmedic_amd.models import ConVIRT
    torch.manual_seed(7)
    B, L, fbs = 6, 16, 3
    model = ConVIRT(encoder=dict(proto=None, add_pooling_layer=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **TXT),
                    cnn=dict(proto="VisualEncoder", backbone="hfresnet", permute="batch_first", dropout_out=0.0, **RESNET),
                    projection=dict(visual_embedding_dim=32, textual_embedding_dim=128, projection_dim=64),
                    loss=dict(proto="ConVIRTLoss", tau=0.1, lambda_=0.75), forward_batch_size=fbs).to(dev())
    with torch.no_grad():                       # non-trivial BatchNorm affine / running statistics and projection weights
        for n, p in model.named_parameters():
            if "normalization" in n or "proj" in n:
                p.add_(0.2 * torch.randn_like(p))
        for n, b in model.named_buffers():
            if "running_mean" in n:
                b.normal_(0, 0.1)
            elif "running_var" in n:
                b.uniform_(0.8, 1.2)
    state = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    images = R.make_images(B, 8, seed=3)
    ids, am = R.make_reports(B, L, TXT["vocab_size"], seed=3)
    model.train(training)
    out = model(input_ids=ids.to(dev()), attention_mask=am.to(dev()), images=images.to(dev()))
    # oracle: the image tower restated (hf_resnet_forward is pinned against transformers.ResNetModel), BatchNorm mode as the model's
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and "num_batches" not in k) for k, v in state.items()}

    def visual(im):
        fmap = O.hf_resnet_forward(im, st, RESNET, prefix="visual.model.", training=training)
        f = fmap.view(*fmap.shape[:2], -1).permute(0, 2, 1)
        return f.squeeze(1) if f.shape[1] == 1 else f
    ref = O.convirt_forward(images, ids, am, st, TXT, visual, 0.1, 0.75, fbs if training else B)
    loss, loss_l, loss_v, lin, vis = ref
    r = dict(loss_err=abs(out["loss"].item() - loss.item()), loss=loss.item(),
             rows_err=max((out["loss_l"].float().cpu() - loss_l.detach()).abs().max().item(), (out["loss_v"].float().cpu() - loss_v.detach()).abs().max().item()),
             lin_err=(out["linguistic"].float().cpu() - lin.detach()).abs().max().item(), lin_absmax=lin.abs().max().item(),
             vis_err=(out["visual"].float().cpu() - vis.detach()).abs().max().item(), vis_absmax=vis.abs().max().item())
    if training:
        out["loss"].backward()
        loss.backward()
        named = dict(model.named_parameters())
        for n in ("lin_proj.0.weight", "vis_proj.2.weight", "vis_proj.2.bias", "linguistic.pooler.dense.weight",
                  "linguistic.encoder.encoder.layer.1.output.dense.weight", "visual.model.embedder.embedder.convolution.weight"):
            got, want = named[n].grad.float().cpu(), st[n].grad
            key = "cnn" if n.startswith("visual.") else "hip"
            print(f"    grad {n}: cos {_cos(got, want):.5f} rel {_rel(got, want):.3e}", flush=True)
            r[key + "_grad_min_cos"] = min(r.get(key + "_grad_min_cos", 1.0), _cos(got, want))
            r[key + "_grad_max_rel"] = max(r.get(key + "_grad_max_rel", 0.0), _rel(got, want))
    print(f"[parity] ConVIRT.forward training={training}: " + " ".join(f"{k}={v:.3e}" for k, v in r.items()), flush=True)
    assert r["loss_err"] <= 5e-3 * max(1.0, abs(r["loss"]))
    assert r["rows_err"] <= 6e-2
    assert r["lin_err"] <= 2e-2 + 2e-2 * r["lin_absmax"] and r["vis_err"] <= 2e-2 + 2e-2 * r["vis_absmax"]
    if training:
        assert r["hip_grad_min_cos"] >= 0.995 and r["hip_grad_max_rel"] <= 0.1
        assert r["cnn_grad_min_cos"] >= 0.9 and r["cnn_grad_max_rel"] <= 0.5


def test_mvqa_forward_vs_oracle_answer_bit_exact():
    """MVQA.forward (ref: MVQA.py:40-54) at the reference's transformer shape (d = 768, 8 heads -> head_dim 96, ff = 2048, 330
    classes; 2 layers): the CNN output is shared by both sides (the DenseNet is a MIOpen-backed torch module), everything
    behind it -- adapter, BertEncoder, pooler, classifier, label-smoothing CE -- against the oracle: ``output`` within tolerance,
    ``loss`` within tolerance, ``answer`` BIT-EXACT on every row.  (The classifier is given unit-scale weights so that the class
    logits' top-2 gap exceeds what bf16 activations can move them by; the gap and the logit error are printed.)"""
    from oracle import torch_ref as O
    from vilmedic_amd.models import MVQA
    torch.manual_seed(11)
    tcfg = dict(hidden_size=768, intermediate_size=2048, num_hidden_layers=2, num_attention_heads=8, attention_probs_dropout_prob=0.0,
                hidden_dropout_prob=0.0, hidden_act="gelu", initializer_range=0.02, layer_norm_eps=1e-12)
    model = MVQA(cnn=dict(proto="VisualEncoder", backbone="densenet169", output_layer="features", dropout_out=0.0, permute="batch_first",
                          freeze=False),
                 adapter=dict(input_size=1664, output_size=768), transformer=dict(tcfg),
                 classifier=dict(proto="Classifier", input_size=768, num_classes=330, dropout=0.0),
                 loss=dict(proto="LabelSmoothingCrossEntropy")).to(dev())
    with torch.no_grad():
        model.classifier.classifier[0].weight.normal_(0, 1.0)
        model.classifier.classifier[0].bias.normal_(0, 1.0)
    model.eval()                                   # running BatchNorm statistics: the CNN output is a pure function of the images
    B = 16
    images = R.make_images(B, 64, seed=5).to(dev())
    labels = torch.randint(0, 330, (B,), generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        feats = model.cnn(images).float()
        out = model(images=images, labels=labels.to(dev()), from_training=True)
    state = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if not k.startswith("cnn.")}
    ocfg = dict(tcfg)
    ref_loss, ref_out, ref_answer = O.mvqa_forward(feats.cpu(), labels, state, ocfg)
    top2 = ref_out.topk(2, dim=-1)[0]
    gap = (top2[:, 0] - top2[:, 1]).min().item()
    err = (out["output"].float().cpu() - ref_out).abs().max().item()
    print(f"[parity] MVQA.forward: max |logit err| {err:.3e} (|logits| <= {ref_out.abs().max().item():.2f}), smallest top-2 gap {gap:.3e}, "
          f"loss {out['loss'].item():.5f} vs {ref_loss.item():.5f}", flush=True)
    assert out["answer"].dtype == torch.int64 and torch.equal(out["answer"].cpu(), ref_answer)
    assert err <= 5e-2 + 1e-2 * ref_out.abs().max().item()
    assert abs(out["loss"].item() - ref_loss.item()) <= 5e-3 * max(1.0, abs(ref_loss.item()))
    assert gap > 2 * err, "fixture too weak: a top-2 gap is within the logit error"
    # the reference's use_amp switch (fp16 autocast there) = bf16 channels-last convolutions in the CNN tower here (Trainor sets it from
    # ``use_amp``; default off = fp32 convolutions as the reference's default): the CNN features move by bf16 rounding, the answers must not
    from vilmedic_amd.blocks.vision import visual_encoder as VE
    try:
        VE.CNN_AMP = True
        with torch.no_grad():
            feats_amp = model.cnn(images).float()
            out_amp = model(images=images, labels=labels.to(dev()), from_training=True)
    finally:
        VE.CNN_AMP = False
    ferr = ((feats_amp - feats).norm() / feats.norm()).item()
    print(f"[parity] MVQA with the CNN tower under bf16 autocast: CNN feature rel-l2 {ferr:.2e}, loss {out_amp['loss'].item():.5f}", flush=True)
    # (a randomly initialised 169-layer DenseNet amplifies the bf16 rounding of its convolutions: ~8 % here -- which is why the switch is
    # opt-in; the class logits move by less than their top-2 gap)
    assert 0 < ferr < 0.15 and torch.equal(out_amp["answer"].cpu(), ref_answer)


def test_rrg_scst_forward_with_fixed_rollouts_and_top_k_vs_oracle():
    """RRG_SCST.forward (ref: RRG_SCST.py:59-85) composed end to end with ``top_k`` set: the two rollouts are replaced by fixed
    sequences (sampling is stochastic; everything else -- the two encoder passes, rewards, the bad-word + top-k filtered log-probs of
    the sampled tokens, the policy-gradient loss and its gradients -- is what runs in training) and compared with the oracle's
    scst_forward on the oracle-encoded image features"""
    from oracle import torch_ref as O
    from vilmedic_amd.datasets import SyntheticImSeq
    from vilmedic_amd.models import RRG_SCST
    top_k = 7
    ds = SyntheticImSeq(num_samples=4, image_size=32, vocab_size=97, tokenizer_max_len=12)
    dl = torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=ds.get_collate_fn())
    dcfg = dict(R.DEC_TINY, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    import zlib

    def toy_reward(refs, hyps):
        """deterministic stand-in for a text metric (random-token hypotheses share nothing with the references, so ROUGE-L would make
        every reward -- and with it the loss -- exactly zero): (corpus score, per-sample scores) like the reference's scorers"""
        vals = [(zlib.crc32((r + "|" + h).encode()) % 1000) / 1000.0 for r, h in zip(refs, hyps)]
        return sum(vals) / len(vals), vals
    model = RRG_SCST(decoder=dict(proto=None, **dcfg), cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", **R.VIT_TINY),
                     dl=dl, scores=[toy_reward], top_k=top_k).to(dev())
    vst = R.rand_state(R.vit_shapes(R.VIT_TINY), 31)
    dst = R.rand_state(R.decoder_shapes(R.DEC_TINY), 32, std=0.08)
    dst["lm_head.bias"][2] += 6.0                # eos is always among the top-k candidates (a row may then legitimately end early)
    sd = {"enc.model." + k: v for k, v in vst.items()}
    sd.update({"dec.decoder." + k: v for k, v in dst.items()})
    sd["dec.decoder.lm_head.decoder.weight"] = dst["bert.embeddings.word_embeddings.weight"]
    sd["dec.decoder.lm_head.decoder.bias"] = dst["lm_head.bias"]
    model.model.load_state_dict(sd, strict=True)
    batch = next(iter(dl))
    B, T = batch["input_ids"].shape[0], 10
    # the oracle decides which tokens survive bad-word + top-k filtering at every step; the fixed "sampled" sequence alternates
    # between the best and a low-ranked surviving candidate (two ranks inside the threshold: the HIP path ranks bf16 logits, so
    # the k-th and (k+1)-th candidate may swap against the fp32 oracle -- a token AT the threshold would make the fixture a coin flip)
    feats = O.vit_forward(batch["images"], vst, R.VIT_TINY)
    enc_o, mask_o = O.visual_encode(feats, {})
    seq = torch.zeros(B, T, dtype=torch.long)
    for t in range(1, T):
        h = O.decoder_hidden(seq[:, :t], None, enc_o, mask_o, dst, R.DEC_TINY)
        lg = O.lm_logits(h, dst).float()[:, -1]
        lg[:, [1, 0]] = -float("inf")
        cand = lg.topk(top_k, dim=-1)[1]
        pick = cand[:, (top_k - 3) if t % 2 else 0]
        if t % 2 == 0:                           # never eos except where placed below
            pick = torch.where(pick == 2, cand[:, 1], pick)
        else:
            pick = torch.where(pick == 2, cand[:, top_k - 4], pick)
        seq[:, t] = pick
        if t == 6:
            assert bool((cand[1] == 2).any()), "fixture: eos must be a top-k candidate"
            seq[1, 6] = 2                        # one row ends early: eos, then pads (masked out of the loss)
        if t > 6:
            seq[1, t] = 1
    greedy = seq.clone()
    greedy[:, 1:] = torch.roll(seq[:, 1:], 1, dims=1)

    class _Gen:
        def __init__(self, sequences):
            self.sequences = sequences

    dec = model.model.dec.decoder
    calls = []

    def fake_generate(input_ids=None, do_sample=False, greedy_rows=None, encoder_hidden_states=None, **kw):
        calls.append((bool(do_sample), greedy_rows, tuple(encoder_hidden_states.shape[:1])))
        if greedy_rows:                              # the paired rollout: greedy rows first, sampled rows behind them
            return _Gen(torch.cat([greedy, seq]).to(dev()))
        return _Gen((seq if do_sample else greedy).to(dev()))
    dec.generate = fake_generate
    out = model(**batch)
    assert calls == [(True, B, (2 * B,))]            # ONE decode loop of 2B rows serves the baseline and the sampled rollout
    out["loss"].backward()
    # oracle side: rewards by the same scorer on the same strings, loss by scst_forward
    tok = ds.tokenizer
    dec_str = lambda rows: [tok.decode(r, skip_special_tokens=True, clean_up_tokenization_spaces=False) for r in rows]
    refs = dec_str(batch["input_ids"])
    scorer = model.scst.scorers[0]
    r_greedy = [scorer(refs, dec_str(greedy))[model.scst.scorers_index[0]]]
    r_sample = [scorer(refs, dec_str(seq[:, 1:]))[model.scst.scorers_index[0]]]
    vst_r = {k: v.clone().requires_grad_(True) for k, v in vst.items()}
    dst_r = {k: v.clone().requires_grad_(True) for k, v in dst.items()}
    feats_r = O.vit_forward(batch["images"], vst_r, R.VIT_TINY)
    enc_r, mask_r = O.visual_encode(feats_r, {})
    ref_loss, ref_logp = O.scst_forward(seq, enc_r, mask_r, dst_r, R.DEC_TINY, r_sample, r_greedy, [1.0], 1, 0, top_k=top_k)
    ref_loss.backward()
    assert torch.isfinite(ref_logp[seq[:, 1:] > 1]).all(), "fixture: every sampled (non-pad) token must survive the oracle's top-k filter"
    named = dict(model.model.named_parameters())
    g1 = named["dec.decoder.bert.encoder.layer.1.output.dense.weight"].grad.float().cpu()
    g2 = named["enc.model.encoder.layer.0.intermediate.dense.weight"].grad.float().cpu()
    r1 = dst_r["bert.encoder.layer.1.output.dense.weight"].grad
    r2 = vst_r["encoder.layer.0.intermediate.dense.weight"].grad
    print(f"[parity] RRG_SCST.forward top_k={top_k}: loss {out['loss'].item():.6f} vs {ref_loss.item():.6f}; decoder grad cos {_cos(g1, r1):.5f} rel {_rel(g1, r1):.3e}; "
          f"encoder grad cos {_cos(g2, r2):.5f} rel {_rel(g2, r2):.3e}", flush=True)
    assert abs(ref_loss.item()) > 1e-3, "fixture: the reward difference must not vanish"
    assert abs(out["loss"].item() - ref_loss.item()) <= 5e-3 * max(1e-1, abs(ref_loss.item()))
    assert _cos(g1, r1) >= 0.99 and _cos(g2, r2) >= 0.99 and _rel(g1, r1) <= 0.15 and _rel(g2, r2) <= 0.15


def test_topk_threshold_kernel_is_exact():
    """vm_topk_threshold_bf16 against torch.topk on the same bf16 logits with banned columns, at the LM head's width (V = 30522 of
    30528 padded columns) and on a tie-heavy row set: the threshold must be the k-th largest live VALUE exactly"""
    import ctypes as C
    from vilmedic_amd._lib import check, lib, ptr, stream
    rows, V, Vp = 64, 30522, 30528
    g = torch.Generator().manual_seed(9)
    logits = (torch.randn(rows, Vp, generator=g) * 3).to(BF)
    logits[7] = (torch.randint(0, 4, (Vp,), generator=g).float() - 2).to(BF)          # ~7600 copies of each value
    logits[:, V:] = 50.0                                                              # pad columns must never count
    ld = logits.to(dev())
    for k, banned in ((1, []), (20, [1, 0]), (50, [1, 0]), (1000, [3])):
        thr = torch.empty(rows, dtype=torch.float32, device=dev())
        ban = (C.c_int32 * 4)(*(banned + [0] * (4 - len(banned)))) if banned else None
        check(lib().vm_topk_threshold_bf16(ptr(ld), Vp, rows, V, k, ban, len(banned), ptr(thr), stream()), "vm_topk_threshold_bf16")
        ref = logits[:, :V].float()
        if banned:
            ref[:, banned] = -float("inf")
        want = ref.topk(k, dim=-1)[0][:, -1]
        assert torch.equal(thr.cpu(), want), (k, banned, (thr.cpu() - want).abs().max())


def test_rrg_scst_graphed_step_equals_the_eager_step():
    """BASELINE configs[4] "HIP-graph-captured step" (ref: vilmedic/models/rrg/RRG_SCST.py:59-85): RRG_SCST.graphed_step -- rollouts, then the
    train-mode encoder pass + teacher-forced decoder pass + policy-gradient loss + backward + fused Adam replayed from ONE captured graph
    on the rollout padded to max_length -- against forward() + backward() + optimizer.step() on the same fixed rollouts: the loss of every
    step (warm-up, capture, replays) and the parameters after 5 steps."""
    import zlib
    from vilmedic_amd.arena import arena_of
    from vilmedic_amd.datasets import SyntheticImSeq
    from vilmedic_amd.models import RRG_SCST
    from vilmedic_amd.optim import FusedAdam
    ds = SyntheticImSeq(num_samples=4, image_size=32, vocab_size=97, tokenizer_max_len=12)
    dl = torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=ds.get_collate_fn())
    dcfg = dict(R.DEC_TINY, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)

    def toy_reward(refs, hyps):
        vals = [(zlib.crc32((r + "|" + h).encode()) % 1000) / 1000.0 for r, h in zip(refs, hyps)]
        return sum(vals) / len(vals), vals
    batch = next(iter(dl))
    B = batch["input_ids"].shape[0]
    g = torch.Generator().manual_seed(3)
    rollouts = []
    for _ in range(5):                                   # a different fixed (greedy, sampled) pair per step, different lengths
        T = int(torch.randint(5, 11, (1,), generator=g))
        sq = torch.randint(3, 97, (2 * B, T), generator=g)
        sq[:, 0] = 0
        sq[1, 3] = 2; sq[1, 4:] = 1
        sq[B + 2, T - 2] = 2; sq[B + 2, T - 1:] = 1
        rollouts.append(sq)

    class _Gen:
        def __init__(self, sequences):
            self.sequences = sequences

    def build():
        torch.manual_seed(5)
        m = RRG_SCST(decoder=dict(proto=None, **dcfg), cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", **R.VIT_TINY),
                     dl=dl, scores=[toy_reward], top_k=None).to(dev())
        calls = {"n": 0}

        def fake_generate(**kw):
            sq = rollouts[calls["n"]]
            calls["n"] += 1
            return _Gen(sq.to(dev()))
        m.model.dec.decoder.generate = fake_generate
        return m, FusedAdam(m, lr=2e-3)
    m1, o1 = build()
    m2, o2 = build()
    m2.load_state_dict(m1.state_dict())
    l1, l2, modes = [], [], []
    for _ in range(5):
        out = m1(**batch)
        o1.zero_grad()
        out["loss"].backward()
        o1.step()
        l1.append(out["loss"].detach().float().item())
        out2 = m2.graphed_step(o2, **batch)
        l2.append(float(out2["loss"]))
        modes.append(out2["launch_mode"])
    torch.cuda.synchronize()
    err = max(abs(a - b) for a, b in zip(l1, l2))
    perr = _rel(arena_of(m2).flat, arena_of(m1).flat)
    print(f"[parity] RRG_SCST.graphed_step vs eager: losses {l2} vs {l1}; max |diff| {err:.3e}; parameters rel L2 {perr:.3e}; modes {modes}", flush=True)
    assert modes[:2] == ["eager (graph warm-up)"] * 2 and modes[2:] == ["hip-graph replay"] * 3
    assert all(abs(x) > 1e-4 for x in l1), "fixture: the reward difference must not vanish"
    assert err <= 2e-3 and perr <= 2e-3


def test_hfpoolformer_visual_encoder_on_the_gpu_vs_the_cpu_module():
    """``VisualEncoder(backbone='hfpoolformer')`` (ref:vilmedic/blocks/vision/visual_encoder.py:67-69,192-208): the tower on the GPU (MIOpen convolutions,
    channels-last input) against the SAME modules on the CPU -- which tests/test_host_cpu.py pins against transformers' PoolFormerModel --: the
    ``batch_first`` features, the input gradient, and ``encode`` (bf16 features + the all-zero-feature mask of visual_encoder.py:138-140)"""
    import copy
    from vilmedic_amd.blocks.vision import VisualEncoder
    torch.manual_seed(11)
    kw = dict(depths=[1, 2, 1, 1], hidden_sizes=[16, 32, 48, 64], mlp_ratio=2.0, layer_scale_init_value=0.5, initializer_range=0.2)
    enc = VisualEncoder(backbone="hfpoolformer", permute="batch_first", dropout_out=0.0, **kw)
    cpu = copy.deepcopy(enc.model).eval()
    enc = enc.to(dev()).eval()
    x = torch.randn(3, 3, 96, 96)
    a, b = x.clone().requires_grad_(True), x.to(dev()).requires_grad_(True)
    fm = cpu(a)
    want = fm.view(*fm.shape[:2], -1).permute(0, 2, 1)
    got = enc(b)
    assert got.shape == want.shape == (3, 9, 64)
    e = _rel(got.float().cpu(), want)
    want.square().mean().backward(); got.float().square().mean().backward()
    eg = _rel(b.grad.cpu(), a.grad)
    print(f"[parity] hfpoolformer tower GPU vs CPU: features rel {e:.2e}, input gradient rel {eg:.2e}")
    assert e <= 2e-3 and eg <= 5e-3, (e, eg)
    with torch.no_grad():
        feats, mask = enc.encode(x)
    assert feats.dtype == BF and feats.shape == (3, 9, 64) and mask.shape == (3, 9) and bool(mask.all())
    assert _rel(feats.float().cpu(), want.detach()) <= 1e-2


def test_3d_densenet_visual_encoder_volume_and_per_slice_encoding():
    """``VisualEncoder(backbone='_3d_densenet121', ...)`` (ref:vilmedic/blocks/vision/visual_encoder.py:71,144-157): a 5-D volume through the restated
    N-d DenseNet (Conv3d / BatchNorm3d on MIOpen) against the same modules on the CPU; ``encode`` of the full volume (features cut, batch_first) and of
    slices (``slice_encode``, the ``class_layers`` vector per slice stacked along dim 1) with the all-zero-feature mask"""
    import copy
    from vilmedic_amd.blocks.vision import VisualEncoder
    torch.manual_seed(4)
    kw = dict(spatial_dims=3, in_channels=1, out_channels=16, block_config=(2, 2), init_features=8, growth_rate=4)
    enc = VisualEncoder(backbone="_3d_densenet121", permute="batch_first", dropout_out=0.0, output_layer="features", **kw)
    cpu = copy.deepcopy(enc.model).eval()
    enc = enc.to(dev()).eval()
    vol = torch.randn(2, 1, 32, 32, 32)
    fm = cpu(vol)
    want = fm.view(*fm.shape[:2], -1).permute(0, 2, 1)
    with torch.no_grad():
        feats, mask = enc.encode(vol)
    assert feats.shape == want.shape == (2, 64, 16) and mask.shape == (2, 64)
    e = _rel(feats.float().cpu(), want.detach())
    print(f"[parity] 3-D DenseNet tower GPU vs CPU (bf16 features): rel {e:.2e}")
    assert e <= 1e-2, e
    sl = VisualEncoder(backbone="_3d_densenet121", permute="batch_first", dropout_out=0.0, output_layer="class_layers", slice_encode=True, slice_dim=2,
                       **dict(kw, spatial_dims=2)).to(dev()).eval()
    with torch.no_grad():
        f2, m2 = sl.encode(vol[:, :, :5])
    assert f2.shape == (2, 5, 16) and m2.shape == (2, 5)
    with pytest.raises(Exception, match="slice_dim"):
        VisualEncoder(backbone="_3d_densenet121", permute="batch_first", output_layer="class_layers", slice_encode=True, **kw)
