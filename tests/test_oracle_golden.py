"""Pin the CPU oracle (oracle/torch_ref.py) to the reference: every fixture under
tests/golden/ was produced by tools/make_golden.py from the reference's own block
files; the oracle must reproduce them (fp32, tight tolerance; token ids bit-exact)."""
import pytest
import torch

import golden_recipes as R
from oracle import torch_ref as O


def close(a, b, rtol=2e-4, atol=2e-5):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def test_g1_vit_tiny(golden):
    g = golden("g1_vit_tiny")
    st = R.rand_state(R.vit_shapes(g["cfg"]), g["seed"])
    assert abs(R.state_checksum(st) - g["checksum"]) < 1e-6 * g["checksum"]
    images = R.make_images(g["B"], g["cfg"]["image_size"], seed=g["seed"])
    images[g["blank_image"]] = 0.0
    feats, mask = O.visual_encode(O.vit_forward(images, st, g["cfg"]), {})
    close(feats, g["features"])
    assert torch.equal(mask, g["mask"])


def test_g2_vit_b16_layer(golden):
    g = golden("g2_vit_b16_1layer")
    st = R.rand_state(R.vit_shapes(g["cfg"]), g["seed"])
    images = R.make_images(g["B"], 224, seed=g["seed"])
    feats = O.vit_forward(images, st, g["cfg"])
    close(feats[:, ::8], g["features"], rtol=1e-3, atol=1e-4)
    assert abs(float(feats.double().sum()) - g["features_full_sum"]) < 1e-2


def _decoder_inputs(g):
    cfg = g["cfg"]
    ids, am = R.make_reports(g["B"], g["L"], cfg["vocab_size"], seed=g["seed"])
    gen = torch.Generator().manual_seed(g["seed"] + 1)
    enc = torch.randn(g["B"], g["S"], cfg["hidden_size"], generator=gen)
    enc[~g["enc_mask"]] = 0.0
    return ids, am, enc


def test_g3_decoder_loss_logits_grads(golden):
    g = golden("g3_decoder_tiny")
    cfg = g["cfg"]
    st = R.rand_state(R.decoder_shapes(cfg), g["seed"])
    assert abs(R.state_checksum(st) - g["checksum"]) < 1e-6 * g["checksum"]
    st = {k: v.requires_grad_(True) for k, v in st.items()}
    ids, am, enc = _decoder_inputs(g)
    enc.requires_grad_(True)
    loss, logits = O.decoder_forward(ids, am, enc, g["enc_mask"], st, cfg)
    close(loss, g["loss"])
    close(logits, g["logits"])
    loss.backward()
    for n, ref in g["grads"].items():
        close(st[n].grad, ref, rtol=1e-3, atol=1e-6)
    close(enc.grad, g["enc_grad"], rtol=1e-3, atol=1e-7)
    assert {"loss", "logits", "past_key_values", "hidden_states", "attentions", "cross_attentions"} <= set(g["out_keys"])


def test_g5_rrg_adam_trajectory(golden):
    g = golden("g5_rrg_tiny")
    vst = {"enc.model." + k: v for k, v in R.rand_state(R.vit_shapes(g["vit_cfg"]), g["seed"]).items()}
    dst = {"dec.decoder." + k: v for k, v in R.rand_state(R.decoder_shapes(g["dec_cfg"]), g["seed"] + 1).items()}
    st = {k: v.requires_grad_(True) for k, v in {**vst, **dst}.items()}
    images = R.make_images(g["B"], g["vit_cfg"]["image_size"], seed=g["seed"])
    ids, am = R.make_reports(g["B"], g["L"], g["dec_cfg"]["vocab_size"], seed=g["seed"])
    opt = torch.optim.Adam(list(st.values()), lr=g["lr"])
    for step in range(3):
        loss, logits = O.rrg_vit_forward(images, ids, am, st, g["vit_cfg"], g["dec_cfg"])
        if step == 0:
            close(logits, g["logits0"])
        close(loss.detach(), g["losses"][step], rtol=1e-4, atol=1e-5)
        opt.zero_grad()
        loss.backward()
        opt.step()


@pytest.mark.parametrize("B", [8, 64])
def test_g6_contrastive_and_ce_losses(golden, B):
    g = golden("g6_losses")
    gen = torch.Generator().manual_seed(1234 + B)
    l = torch.randn(B, 96, generator=gen).requires_grad_(True)
    v = torch.randn(B, 96, generator=gen).requires_grad_(True)
    ref = g[f"convirt_{B}"]
    loss, ll, lv = O.convirt_loss(l, v, 0.1, 0.75)
    loss.backward()
    close(loss, ref["loss"]), close(ll, ref["loss_l"]), close(lv, ref["loss_v"])
    close(l.grad, ref["gl"], atol=1e-6), close(v.grad, ref["gv"], atol=1e-6)
    l2 = (0.2 * l.detach()).requires_grad_(True)
    v2 = (0.2 * v.detach()).requires_grad_(True)
    ref = g[f"infonce_{B}"]
    loss, lt, li = O.infonce_loss(l2, v2)
    loss.backward()
    close(loss, ref["loss"]), close(lt, ref["loss_t"]), close(li, ref["loss_i"])
    close(l2.grad, ref["gl"], atol=1e-6), close(v2.grad, ref["gv"], atol=1e-6)
    logits = torch.randn(B, 33, generator=gen).requires_grad_(True)
    ref = g[f"lsce_{B}"]
    loss = O.label_smoothing_ce(logits, ref["target"], 0.1)
    loss.backward()
    close(loss, ref["loss"]), close(logits.grad, ref["g"], atol=1e-7)
    # the reference's own statement: equals F.cross_entropy(label_smoothing=eps)
    close(loss.detach(), torch.nn.functional.cross_entropy(logits.detach(), ref["target"], label_smoothing=0.1))


def test_g6_convirt_known_answer(golden):
    g = golden("g6_losses")
    torch.manual_seed(1234)
    a, b = torch.randn(8, 768), torch.randn(8, 768)
    loss = O.convirt_loss(a, b, 0.1, 0.75)[0]
    close(loss, g["convirt_known_answer"])
    assert abs(float(loss) - 1.9949309826) < 1e-5


def test_g6_gloria(golden):
    g = golden("g6_losses")["gloria"]
    B, D, T, hw = g["B"], g["D"], g["T"], g["hw"]
    gen = torch.Generator().manual_seed(99)
    glob = torch.randn(B, D, generator=gen).requires_grad_(True)
    loc = torch.randn(B, D, hw, hw, generator=gen).requires_grad_(True)
    words = torch.randn(B, D, T, generator=gen).requires_grad_(True)
    sent = torch.randn(B, D, generator=gen).requires_grad_(True)
    l0, l1 = O.gloria_local_loss(loc, words, g["cap_lens"], 4.0, 5.0, 10.0)
    g0, g1 = O.gloria_global_loss(glob, sent, 10.0)
    loss = (l0 + l1) * 1.0 + (g0 + g1) * 1.0
    loss.backward()
    close(loss, g["loss"])
    close(glob.grad, g["g_glob"], atol=1e-6), close(loc.grad, g["g_loc"], atol=1e-6)
    close(words.grad, g["g_words"], atol=1e-6), close(sent.grad, g["g_sent"], atol=1e-6)


def test_g7_greedy_and_beam_ids_bit_exact(golden):
    g = golden("g7_decode")
    cfg = g["cfg"]
    rc = g["recipe"]
    st = R.rand_state(R.decoder_shapes(cfg), g["seed"], std=rc["std"], emb_std=rc["emb_std"], qk_std=rc.get("qk_std"), pos_std=rc.get("pos_std"))
    st["lm_head.bias"][cfg["eos_token_id"]] += rc["eos_bias"]
    assert abs(R.state_checksum(st) - g["checksum"]) < 1e-6 * g["checksum"]
    gen = torch.Generator().manual_seed(g["seed"] + 1)
    enc = torch.randn(g["B"], g["S"], cfg["hidden_size"], generator=gen)
    enc[~g["enc_mask"]] = 0.0
    ids = O.greedy_decode(enc, g["enc_mask"], st, cfg, 0, 2, 1, g["max_len"])
    assert torch.equal(ids, g["beams1_lp1.0"]["sequences"])
    for lp in (1.0, 2.0):
        ref = g[f"beams4_lp{lp}"]
        seqs, scores = O.beam_decode(enc, g["enc_mask"], st, cfg, 0, 2, 1, g["max_len"], 4, lp)
        assert torch.equal(seqs, ref["sequences"]), (lp, seqs, ref["sequences"])
        close(scores, ref["scores"], rtol=1e-4, atol=1e-4)


def test_g8_scst_loss(golden):
    g = golden("g8_scst")
    inp = g["logp"].clone().requires_grad_(True)
    loss = O.scst_loss(inp * 1.0, g["seq"], g["rs"], g["rg"], g["w"], g["pad"])
    loss.backward()
    close(loss, g["loss"])
    close(inp.grad, g["grad"])


def test_g9_mvqa_core_and_text_encoder(golden):
    g = golden("g9_mvqa_text")
    m = g["mvqa"]
    st = R.rand_state(R.bert_stack_shapes(m["cfg"]), m["seed"])
    h = O.bert_encoder_forward(m["x"], st, m["cfg"], "")
    close(h, m["hidden"])
    pooled = O.bert_pooler(h, {"p.dense.weight": m["pw"], "p.dense.bias": m["pb"]}, "p")
    close(pooled, m["pooled"])
    close(torch.nn.functional.linear(pooled, m["cw"], m["cb"]), m["logits"])
    t = g["text"]
    st = R.rand_state(R.text_encoder_shapes(t["cfg"]), t["seed"])
    ids, am = R.make_reports(t["B"], t["L"], t["cfg"]["vocab_size"], seed=t["seed"])
    h = O.text_encoder_forward(ids, am, st, t["cfg"])
    close(h, t["last_hidden_state"])
    close(O.bert_pooler(h, {"p.dense.weight": t["pw"], "p.dense.bias": t["pb"]}, "p"), t["pooler_output"])


def test_gloria_aggregate_tokens_oracle_and_device_segment_sum_vs_reference(golden):
    """G11: the reference's GLoRIA.aggregate_tokens (word-piece merge) -- the oracle restatement and the product's
    one-shot segment-sum (vilmedic_amd.models.selfsup.GLoRIA.aggregate_tokens, torch index_add, runs on any device)."""
    import types
    from vilmedic_amd.models.selfsup.GLoRIA import GLoRIA, word_segments
    g = golden("g11_gloria_aggregate")
    idxtoword = dict(enumerate(g["vocab"]))
    out, sents = O.gloria_aggregate_tokens(g["embeddings"], g["input_ids"], idxtoword)
    assert torch.equal(out, g["out"]) and sents == g["sentences"]
    out2, sents2 = GLoRIA.aggregate_tokens(types.SimpleNamespace(idxtoword=idxtoword), g["embeddings"], g["input_ids"])
    assert torch.allclose(out2, g["out"], atol=1e-6) and sents2 == g["sentences"]
    # a caption without [SEP] loses its last open word (the reference never flushes it)
    seg, words = word_segments(["[CLS]", "no", "eff", "##usion"])
    assert seg == [0, 1, -1, -1] and words == ["[CLS]", "no"]


def _rrs_state(g):
    est = R.rand_state(R.text_encoder_shapes(g["enc_cfg"]), g["seed"])
    dst = R.rand_state(R.decoder_shapes(g["dec_cfg"]), g["seed"] + 1)
    assert abs(R.state_checksum(est) + R.state_checksum(dst) - g["checksum"]) < 1e-6 * g["checksum"]
    st = {"enc.encoder." + k: v for k, v in est.items()}
    st.update({"dec.decoder." + k: v for k, v in dst.items()})
    src = R.make_reports(g["B"], g["Ls"], g["enc_cfg"]["vocab_size"], seed=g["seed"])
    tgt = R.make_reports(g["B"], g["Lt"], g["dec_cfg"]["vocab_size"], seed=g["seed"] + 1)
    return st, src, tgt


def test_g13_rrs_loss_logits_grads(golden):
    """RRS: the reference's EncoderModel -> DecoderModel chain (models/rrs/RRS.py:30-52): loss, logits, encoder memory and
    gradients on both sides of the cross-attention."""
    g = golden("g13_rrs_tiny")
    st, (sid, sam), (tid, tam) = _rrs_state(g)
    st = {k: v.requires_grad_(True) for k, v in st.items()}
    loss, logits, hidden = O.rrs_forward(sid, sam, tid, tam, st, g["enc_cfg"], g["dec_cfg"])
    close(hidden, g["encoder_hidden"])
    close(loss, g["loss"])
    close(logits, g["logits"])
    loss.backward()
    for n, ref in g["enc_grads"].items():
        close(st["enc.encoder." + n].grad, ref, rtol=1e-3, atol=1e-6)
    for n, ref in g["dec_grads"].items():
        close(st["dec.decoder." + n].grad, ref, rtol=1e-3, atol=1e-6)


def _vicreg_inputs(N, D):
    g = torch.Generator().manual_seed(4321 + N)
    z1 = 0.7 * torch.randn(N, D, generator=g) + 0.1
    z2 = z1 + 0.3 * torch.randn(N, D, generator=g)
    return z1, z2


def test_g14_vicreg_loss(golden):
    for case in golden("g14_vicreg").values():
        z1, z2 = (z.requires_grad_(True) for z in _vicreg_inputs(case["N"], case["D"]))
        loss = O.vicreg_loss(z1, z2)
        loss.backward()
        close(loss, case["loss"])
        close(z1.grad, case["g1"], rtol=1e-3, atol=1e-6)
        close(z2.grad, case["g2"], rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("layer_type,hidden_sizes,depths,extra", [
    ("basic", [16, 32, 48, 64], [2, 1, 2, 1], {}),
    ("bottleneck", [32, 64, 96, 128], [1, 2, 1, 1], {"downsample_in_bottleneck": True, "downsample_in_first_stage": True}),
])
def test_hf_resnet_oracle_matches_transformers(layer_type, hidden_sizes, depths, extra):
    """the functional hfresnet restatement (BASELINE configs[0]'s encoder) against the installed transformers ResNetModel: feature
    map and input gradient, train-mode and eval-mode BatchNorm"""
    tr = pytest.importorskip("transformers")
    cfg = dict(num_channels=3, embedding_size=8, hidden_sizes=hidden_sizes, depths=depths, layer_type=layer_type, hidden_act="relu", **extra)
    ref = tr.ResNetModel(tr.ResNetConfig(**cfg))
    x = torch.randn(3, 3, 64, 64)
    for training in (True, False):
        ref.train(training)
        st = {k: v.detach().clone() for k, v in ref.state_dict().items()}      # (the train-mode pass moves the running statistics)
        a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        got, want = O.hf_resnet_forward(a, st, cfg, training=training), ref(b).last_hidden_state
        close(got, want, rtol=1e-5, atol=1e-5)
        got.square().mean().backward(); want.square().mean().backward()
        close(a.grad, b.grad, rtol=1e-4, atol=1e-6)


def test_g16_multi_image_encode(golden):
    """VisualEncoder.encode on 5-D images with images_mask: masked images give zero features -> mask False, projected rows = bias"""
    g = golden("g16_vit_multi_image")
    cfg, vp = g["cfg"], g["visual_projection"]
    st = {"model." + k: v for k, v in R.rand_state(R.vit_shapes(cfg), g["seed"]).items()}
    gen = torch.Generator().manual_seed(g["seed"] + 77)                       # tools/make_golden.py build_ref_vit
    st["visual_projection.weight"] = 0.05 * torch.randn(vp["out_features"], vp["in_features"], generator=gen)
    st["visual_projection.bias"] = 0.02 * torch.randn(vp["out_features"], generator=gen)
    assert abs(R.state_checksum(st) - g["checksum"]) < 1e-6 * g["checksum"]
    B, N, size = g["B"], g["N"], cfg["image_size"]
    images = R.make_images(B * N, size, seed=g["seed"]).view(B, N, 3, size, size)
    feats, mask = O.visual_encode_multi(images, g["images_mask"], st, cfg)
    close(feats, g["features"])
    assert torch.equal(mask, g["mask"])
    S = feats.shape[1] // N
    assert not mask[0, 2 * S:].any() and mask[0, :2 * S].all() and not mask[1, S:2 * S].any()
    close(feats[0, 2 * S], st["visual_projection.bias"])


def test_g18_model_level_compositions_vs_the_reference_classes(golden):
    """G18: the reference's own ``MVQA`` and ``ConVIRT`` class bodies (lifted by AST in tools/make_golden.py, stand-in CNNs) pin the
    oracle's model-level compositions: adapter + LayerNorm -> encoder -> pooler -> classifier -> label-smoothing CE with the arg-max
    answers (MVQA.py:40-54), and both towers in ``forward_batch_size`` micro-batches (training-mode BatchNorm statistics per micro-batch)
    -> projection heads -> ConVIRTLoss (conVIRT.py:75-102)."""
    g = golden("g18_model_compositions")
    m = g["mvqa"]
    loss, out, answer = O.mvqa_forward(m["features"], m["labels"], m["state"], dict(m["cfg"], hidden_act="gelu"))
    torch.testing.assert_close(out, m["output"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss, m["loss"], rtol=1e-5, atol=1e-5)
    assert torch.equal(answer, m["answer"])
    c = g["convirt"]
    ids, am = R.make_reports(c["B"], c["L"], c["cfg"]["vocab_size"], seed=185)
    st = R.rand_state(R.text_encoder_shapes(c["cfg"]), c["encoder_seed"])
    assert R.state_checksum(st) == c["encoder_checksum"]
    state = dict(c["state"], **{"linguistic.encoder." + k: v for k, v in st.items()})

    def visual(x):                                     # the fixture's stand-in CNN: flatten -> Linear -> training-mode BatchNorm1d
        y = x.flatten(1) @ c["visual_fc_w"].t() + c["visual_fc_b"]
        return (y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
    loss, loss_l, loss_v, lin, vis = O.convirt_forward(c["images"], ids, am, state, c["cfg"], visual, c["tau"], c["lambda_"], c["fbs"])
    torch.testing.assert_close(vis, c["visual"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lin, c["linguistic"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(torch.stack([loss_l, loss_v]), torch.stack([c["loss_l"], c["loss_v"]]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(loss, c["loss"], rtol=1e-4, atol=1e-5)


def test_g19_scst_forward_vs_the_reference_forward_sampling(golden):
    """G19: the reference's own ``SCST.forward_sampling`` body (lifted by AST) on its DecoderModel with HF ``generate`` (sampling, top-k,
    bad words, output_scores) produced a sampled batch, the gathered log-probabilities, the policy-gradient loss and the gradient w.r.t.
    the encoder states; oracle.scst_forward, given that sampled batch, must reproduce all of it (teacher forcing == the per-step
    distributions of the un-wrapped generate loop)."""
    g = golden("g19_scst_sampling")
    rc = dict(g["recipe"])
    eos_bias = rc.pop("eos_bias")
    st = R.rand_state(R.decoder_shapes(g["cfg"]), g["seed"], **rc)
    st["lm_head.bias"][g["cfg"]["eos_token_id"]] += eos_bias
    assert R.state_checksum(st) == g["checksum"]
    enc = g["enc"].clone().requires_grad_(True)
    loss, logp = O.scst_forward(g["sequences"], enc, g["enc_mask"], st, g["cfg"], g["rs"], g["rg"], [1.0], 1, 0, top_k=g["top_k"])
    loss.backward()
    live = g["sequences"][:, 1:] > 1                        # sampled, non-pad tokens (what the loss sees)
    torch.testing.assert_close(logp[live], g["logp"][live], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(loss.detach(), g["loss"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(enc.grad, g["g_enc"], rtol=1e-3, atol=1e-6)


def test_g20_rrg_hf_forward_vs_the_reference_method(golden):
    """G20: the reference's own ``RRG_HF.forward`` body on a VisionEncoderDecoderModel (ViT of width 64 -> enc_to_dec_proj -> decoder of
    width 128): 5-D images with an images_mask (crops concatenated along the sequence, masked per crop in the cross-attention) and 4-D
    images; oracle.rrg_hf_forward must reproduce loss and logits."""
    g = golden("g20_rrg_hf")
    vst = R.rand_state(R.vit_shapes(g["vit_cfg"]), g["seed"])
    dst = R.rand_state(R.decoder_shapes(g["dec_cfg"]), g["seed"] + 1)
    assert R.state_checksum(vst) == g["vit_checksum"] and R.state_checksum(dst) == g["dec_checksum"]
    state = {"model.encoder." + k: v for k, v in vst.items()}
    state.update({"model.decoder." + k: v for k, v in dst.items()})
    state["model.enc_to_dec_proj.weight"], state["model.enc_to_dec_proj.bias"] = g["proj_w"], g["proj_b"]
    B, N, size = g["B"], g["N"], g["vit_cfg"]["image_size"]
    images = R.make_images(B * N, size, seed=g["seed"]).view(B, N, 3, size, size)
    ids, am = R.make_reports(B, g["L"], g["dec_cfg"]["vocab_size"], seed=g["seed"])
    with torch.no_grad():
        loss5, logits5 = O.rrg_hf_forward(images, ids, am, state, g["vit_cfg"], g["dec_cfg"], images_mask=g["images_mask"])
        loss4, logits4 = O.rrg_hf_forward(images[:, 0], ids, am, state, g["vit_cfg"], g["dec_cfg"])
    torch.testing.assert_close(logits5, g["logits5"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(loss5, g["loss5"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(logits4, g["logits4"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(loss4, g["loss4"], rtol=1e-5, atol=1e-5)


def _g25_states(g):
    """(VisionEncoderDecoder state under ``model.``, checkpoint-directory writer) for both G25 cases, rebuilt from the recipe"""
    a, b = g["ved"], g["strings"]
    vst = R.rand_state(R.vit_pooled_shapes(a["vit_cfg"]), a["seed"])
    dst = R.rand_state(R.decoder_shapes(a["dec_cfg"]), a["seed"] + 1)
    gen = torch.Generator().manual_seed(a["seed"] + 2)
    ved = {"encoder." + k: v for k, v in vst.items()}
    ved.update({"decoder." + k: v for k, v in dst.items()})
    ved["enc_to_dec_proj.weight"] = 0.1 * torch.randn(a["dec_cfg"]["hidden_size"], a["vit_cfg"]["hidden_size"], generator=gen)
    ved["enc_to_dec_proj.bias"] = 0.02 * torch.randn(a["dec_cfg"]["hidden_size"], generator=gen)
    assert R.state_checksum(ved) == a["checksum"]
    vst2 = R.rand_state(R.vit_pooled_shapes(b["vit_cfg"]), b["seed"])
    dst2 = R.rand_state(R.decoder_shapes(b["dec_cfg"]), b["seed"] + 1)
    assert R.state_checksum(vst2) + R.state_checksum(dst2) == b["checksum"]
    return ved, vst2, dst2


def test_g25_rrg_hf_from_local_checkpoints_vs_the_reference_class(golden):
    """G25: the reference's ``RRG_HF`` class built by HF's own ``from_pretrained`` calls -- ``encoderdecoder=<dir>`` (RRG_HF.py:24-25) and
    ``vision`` / ``decoder`` given as strings (:48-49, :86-87) -- on checkpoint directories written from the recipe; oracle.rrg_hf_forward on
    the same tensors must reproduce loss and logits."""
    g = golden("g25_rrg_hf_pretrained")
    ved, vst2, dst2 = _g25_states(g)
    a, b = g["ved"], g["strings"]
    ids, am = R.make_reports(a["B"], a["L"], a["dec_cfg"]["vocab_size"], seed=a["seed"])
    with torch.no_grad():
        loss, logits = O.rrg_hf_forward(R.make_images(a["B"], a["vit_cfg"]["image_size"], seed=a["seed"]), ids, am,
                                        {"model." + k: v for k, v in ved.items()}, a["vit_cfg"], a["dec_cfg"])
        st2 = {"model.encoder." + k: v for k, v in vst2.items()}
        st2.update({"model.decoder." + k: v for k, v in dst2.items()})
        loss2, logits2 = O.rrg_hf_forward(R.make_images(b["B"], b["vit_cfg"]["image_size"], seed=b["seed"]), ids, am, st2, b["vit_cfg"], b["dec_cfg"])
    torch.testing.assert_close(logits, a["logits"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(loss, a["loss"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(logits2, b["logits"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(loss2, b["loss"], rtol=1e-5, atol=1e-5)


def test_g21_gloria_forward_vs_the_reference_class(golden):
    """G21: the reference's own ``GLoRIA`` class (lifted by AST; its EncoderModel and GLoRIALoss, a stand-in CNN whose [6] is the hooked
    local feature map, a stand-in tokenizer vocabulary) -- towers in forward_batch_size chunks with training-mode BatchNorm, up-sampling
    to 299 x 299, hidden-state stacking, word-piece aggregation, embeddings, loss -- against oracle.gloria_forward."""
    g = golden("g21_gloria_model")
    nn_ = torch.nn
    cnn = nn_.Sequential(nn_.Conv2d(3, 6, 7, stride=8, padding=3), nn_.ReLU(), nn_.Identity(), nn_.Identity(), nn_.Identity(),
                         nn_.Conv2d(6, g["interm"], 3, stride=4, padding=1), nn_.BatchNorm2d(g["interm"]), nn_.ReLU(),
                         nn_.Conv2d(g["interm"], g["feat"], 1), nn_.AdaptiveAvgPool2d(1))
    cnn.load_state_dict(g["cnn_state"], strict=False)
    cnn.train()
    st = R.rand_state(R.text_encoder_shapes(g["cfg"]), g["encoder_seed"])
    assert R.state_checksum(st) == g["encoder_checksum"]
    state = dict(g["state"], **{"linguistic.encoder." + k: v for k, v in st.items()})
    with torch.no_grad():
        loss, gf, lf, word, sent = O.gloria_forward(g["images"], g["input_ids"], g["attention_mask"], state, g["cfg"], cnn,
                                                    dict(enumerate(g["vocab"])), g["last_n_layers"], g["fbs"])
    torch.testing.assert_close(gf, g["global_features"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(lf, g["local_features"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(word, g["word_embeddings"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(sent, g["sent_embeddings"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(loss, g["loss"], rtol=1e-4, atol=1e-4)


def test_g22_ensemble_decode_summed_logits_bit_exact(golden):
    """the ENSEMBLE branch of ``decoder_step_logits`` / ``greedy_decode`` / ``beam_decode`` against HF ``generate`` run over the
    summed logits of two reference DecoderModels (tools/make_golden.py:gen_ensemble_decode; the change the reference's
    beam_search.py:243-262 makes to HF's beam search)"""
    g = golden("g22_ensemble_decode")
    cfg, rc = g["cfg"], g["recipe"]
    states, encs = [], []
    for i, sd in enumerate(g["seeds"]):
        st = R.rand_state(R.decoder_shapes(cfg), sd, std=rc["std"], emb_std=rc["emb_std"], qk_std=rc["qk_std"], pos_std=rc["pos_std"])
        st["lm_head.bias"][cfg["eos_token_id"]] += rc["eos_bias"]
        assert abs(R.state_checksum(st) - g["checksums"][i]) < 1e-6 * g["checksums"][i]
        gen = torch.Generator().manual_seed(sd + 1)
        e = torch.randn(g["B"], g["S"] - i, cfg["hidden_size"], generator=gen)
        e[~g["enc_masks"][i]] = 0.0
        states.append(st)
        encs.append(e)
    masks = list(g["enc_masks"])
    ids = O.greedy_decode(encs, masks, states, cfg, 0, 2, 1, g["max_len"])
    assert torch.equal(ids, g["beams1_lp1.0"]["sequences"])
    for lp in (1.0, 2.0):
        ref = g[f"beams4_lp{lp}"]
        seqs, scores = O.beam_decode(encs, masks, states, cfg, 0, 2, 1, g["max_len"], 4, lp)
        assert torch.equal(seqs, ref["sequences"]), (lp, seqs, ref["sequences"])
        close(scores, ref["scores"], rtol=1e-4, atol=1e-4)
    # and the sum matters: model 0 alone decodes something else
    alone = O.greedy_decode(encs[0], masks[0], states[0], cfg, 0, 2, 1, g["max_len"])
    assert torch.equal(alone, g["model0_alone_greedy"]) and not torch.equal(alone, ids)


# ----------------------------------------------------------------------------- G23 / G24: pretrained `proto` towers, DeiT
def _thin(g):
    return g[::4] if (g.dim() == 2 and g.shape[0] >= 128) else g


def proto_enc_case(g, mt):
    e = g[mt + "_enc"]
    cfg = dict(e["cfg"], model_type=mt)
    st = R.rand_state(R.text_model_shapes(e["cfg"]), e["seed"])
    assert abs(R.state_checksum(st) - e["checksum"]) < 1e-6 * e["checksum"]
    ids, am = R.make_reports(e["B"], e["L"], cfg["vocab_size"], seed=e["seed"], **e["specials"])
    return e, cfg, st, ids, am


def proto_dec_case(g, mt):
    d = g[mt + "_dec"]
    cfg = dict(d["cfg"], model_type=mt)
    st = R.rand_state(R.causal_lm_shapes(d["cfg"], mt), d["seed"], **d["recipe"])
    st["lm_head.bias" if mt == "roberta" else "cls.predictions.bias"][d["specials"]["sep"]] += d["eos_bias"]
    assert abs(R.state_checksum(st) - d["checksum"]) < 1e-6 * d["checksum"]
    ids, am = R.make_reports(d["B"], d["L"], cfg["vocab_size"], seed=d["seed"] - 1, **d["specials"])
    gen = torch.Generator().manual_seed(d["seed"] + 1)
    enc = torch.randn(d["B"], d["S"], cfg["hidden_size"], generator=gen)
    enc[~d["enc_mask"]] = 0.0
    return d, cfg, st, ids, am, enc


@pytest.mark.parametrize("mt", ["roberta", "bert"])
def test_g23_proto_encoder_tower(golden, mt):
    """EncoderModel(proto=<dir>) -> AutoModel.from_pretrained (ref:encoder_model.py:19-22): hidden states of every layer, the built-in
    pooler, and the gradients of the three embedding tables (RoBERTa: position ids from the pad mask, no gradient into the pad rows)"""
    e, cfg, st, ids, am = proto_enc_case(golden("g23_proto_towers"), mt)
    assert e["cls_name"] == ("RobertaModel" if mt == "roberta" else "BertModel")
    st = {k: v.requires_grad_(True) for k, v in st.items()}
    x = O.text_embeddings(ids, st, "embeddings.", cfg)
    hs = [x]
    m = O.key_padding_mask(am)
    for i in range(cfg["num_hidden_layers"]):
        x = O.bert_layer(x, st, f"encoder.layer.{i}.", cfg, m)
        hs.append(x)
    close(torch.stack(hs), e["hidden_states"])
    close(O.text_encoder_forward(ids, am, st, cfg), e["last_hidden_state"])
    pooled = O.bert_pooler(x, st, "pooler")
    close(pooled, e["pooler_output"])
    gen = torch.Generator().manual_seed(e["seed"] + 5)
    wl, wp = torch.randn(x.shape, generator=gen), torch.randn(pooled.shape, generator=gen)
    ((x * wl * am[..., None]).sum() + (pooled * wp).sum()).backward()
    for n, ref in e["grads"].items():
        close(_thin(st[n].grad), ref, rtol=1e-3, atol=1e-5)
    if mt == "roberta":          # nn.Embedding(padding_idx): the pad rows of BOTH tables get no gradient
        assert float(st["embeddings.position_embeddings.weight"].grad[cfg["pad_token_id"]].abs().sum()) == 0.0
    assert float(st["embeddings.word_embeddings.weight"].grad[cfg["pad_token_id"]].abs().sum()) == 0.0


@pytest.mark.parametrize("mt", ["roberta", "bert"])
def test_g23_proto_decoder_loss_grads_and_decode_ids(golden, mt):
    """DecoderModel(proto=<dir>) -> AutoModelForCausalLM.from_pretrained(is_decoder, add_cross_attention) (ref:decoder_model.py:17-21):
    loss / logits / gradients, and greedy + beam-4 token ids of HF generate() on it (bit-exact)"""
    d, cfg, st, ids, am, enc = proto_dec_case(golden("g23_proto_towers"), mt)
    assert d["cls_name"] == ("RobertaForCausalLM" if mt == "roberta" else "BertLMHeadModel")
    stg = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    encg = enc.clone().requires_grad_(True)
    loss, logits = O.decoder_forward(ids, am, encg, d["enc_mask"], stg, cfg)
    close(loss, d["loss"])
    close(logits, d["logits"], rtol=1e-3, atol=1e-4)
    loss.backward()
    for n, ref in d["grads"].items():
        close(_thin(stg[n].grad), ref, rtol=2e-3, atol=1e-6)
    close(encg.grad, d["enc_grad"], rtol=2e-3, atol=1e-7)
    sp = d["specials"]
    with torch.no_grad():
        seq = O.greedy_decode(enc, d["enc_mask"], st, cfg, sp["cls"], sp["sep"], sp["pad"], d["max_len"])
        assert torch.equal(seq, d["beams1"]["sequences"]), (seq.tolist(), d["beams1"]["sequences"].tolist())
        seqs, scores = O.beam_decode(enc, d["enc_mask"], st, cfg, sp["cls"], sp["sep"], sp["pad"], d["max_len"], 4)
        assert torch.equal(seqs, d["beams4"]["sequences"]), (seqs.tolist(), d["beams4"]["sequences"].tolist())
        close(scores, d["beams4"]["scores"], rtol=1e-4, atol=1e-5)


def test_g24_deit_visual_encoder_and_rrg_hf(golden):
    g = golden("g24_deit")
    assert g["cls_name"] == "DeiTModel"
    cfg = g["cfg"]
    st = R.rand_state(R.deit_shapes(cfg), g["seed"])
    assert abs(R.state_checksum(st) - g["checksum"]) < 1e-6 * g["checksum"]
    images = R.make_images(g["B"], cfg["image_size"], seed=g["seed"])
    feats, mask = O.visual_encode(O.vit_forward(images, st, cfg), {})
    assert feats.shape[1] == (cfg["image_size"] // cfg["patch_size"]) ** 2 + 2
    close(feats, g["features"])
    assert torch.equal(mask, g["mask"])
    dst = R.rand_state(R.decoder_shapes(g["dec_cfg"]), g["seed"] + 1)
    assert abs(R.state_checksum(dst) - g["dec_checksum"]) < 1e-6 * g["dec_checksum"]
    full = {"model.encoder." + k: v for k, v in st.items()}
    full.update({"model.decoder." + k: v for k, v in dst.items()})
    ids, am = R.make_reports(g["B"], g["L"], g["dec_cfg"]["vocab_size"], seed=g["seed"])
    loss, logits = O.rrg_hf_forward(images, ids, am, full, cfg, g["dec_cfg"])
    close(loss, g["loss4"])
    close(logits, g["logits4"], rtol=1e-3, atol=1e-4)
