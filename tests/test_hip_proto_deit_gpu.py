"""GPU parity of the pretrained-`proto` text towers (BERT / RoBERTa from a local checkpoint directory: ref:encoder_model.py:19-22,
decoder_model.py:17-21), of DeiT (ref:visual_encoder.py:59-61, config/RRG/baseline-HF.yml:22) and of head widths outside the kernels'
native set, against fixtures G23 / G24 (generated from the reference on HF) and the pinned oracle.  Tolerances as in
tests/test_hip_models_gpu.py: loss 2e-3, bf16 tensors 3e-2 + 3e-2|ref| elementwise with mean <= 1e-2, gradients cosine >= 0.999 and
rel-L2 <= 3e-2, token ids bit-exact (fp32 decode step)."""
import pytest
import torch

import golden_recipes as R
from test_hip_models_gpu import BF, close_bf16, cosine, dev, rel_l2
from test_oracle_golden import _thin, proto_dec_case, proto_enc_case

pytestmark = pytest.mark.gpu


def grads_close(got, ref, cos=0.999, rel=3e-2):
    return cosine(got, ref) >= cos and rel_l2(got, ref) <= rel


@pytest.mark.parametrize("mt", ["roberta", "bert"])
def test_proto_encoder_tower_vs_golden(golden, tmp_path, mt):
    from vilmedic_amd.blocks.huggingface.encoder.encoder_model import EncoderModel
    e, cfg, st, ids, am = proto_enc_case(golden("g23_proto_towers"), mt)
    enc = EncoderModel(dict(proto=R.write_proto_dir(str(tmp_path / "ckpt"), mt, e["cfg"], st))).to(dev())
    enc.train()                                       # dropout probabilities are 0 in the checkpoint's config
    o = enc(input_ids=ids.to(dev()), attention_mask=am.to(dev()), output_hidden_states=True)
    hs = torch.stack([h.float().cpu() for h in o.hidden_states])
    live = am.bool()
    assert close_bf16(hs[:, live], e["hidden_states"][:, live]), (hs - e["hidden_states"])[:, live].abs().max()
    assert close_bf16(o["pooler_output"].float().cpu(), e["pooler_output"])
    gen = torch.Generator().manual_seed(e["seed"] + 5)
    wl, wp = torch.randn(e["last_hidden_state"].shape, generator=gen), torch.randn(e["pooler_output"].shape, generator=gen)
    loss = (o.last_hidden_state.float() * (wl * am[..., None]).to(dev())).sum() + (o.pooler_output.float() * wp.to(dev())).sum()
    loss.backward()
    named = dict(enc.encoder.named_parameters())
    for n, ref in e["grads"].items():
        got = _thin(named[n].grad.float().cpu())
        assert grads_close(got, ref, rel=4e-2), (n, cosine(got, ref), rel_l2(got, ref))
    pad = cfg["pad_token_id"]
    assert float(named["embeddings.word_embeddings.weight"].grad[pad].abs().sum()) == 0.0
    if mt == "roberta":
        assert float(named["embeddings.position_embeddings.weight"].grad[pad].abs().sum()) == 0.0


@pytest.mark.parametrize("mt", ["roberta", "bert"])
def test_proto_decoder_loss_logits_grads_vs_oracle(golden, tmp_path, mt):
    """RobertaForCausalLM / BertLMHeadModel with cross-attention: token-type row, RoBERTa position ids, dense -> GELU -> LayerNorm head,
    on trained-scale weights (std 0.05) against the G23-pinned oracle"""
    from oracle import torch_ref as O
    from vilmedic_amd.blocks.huggingface.decoder.decoder_model import DecoderModel
    d = golden("g23_proto_towers")[mt + "_dec"]
    cfg = dict(d["cfg"], model_type=mt)
    st = R.rand_state(R.causal_lm_shapes(d["cfg"], mt), 77)
    dec = DecoderModel(dict(proto=R.write_proto_dir(str(tmp_path / "ckpt"), mt, d["cfg"], st))).to(dev())
    B, L, S = 4, 24, 10
    ids, am = R.make_reports(B, L, cfg["vocab_size"], seed=78, **d["specials"])
    gen = torch.Generator().manual_seed(79)
    enc = torch.randn(B, S, cfg["hidden_size"], generator=gen)
    enc_mask = torch.ones(B, S, dtype=torch.bool)
    enc_mask[2, 6:] = False
    enc[~enc_mask] = 0.0
    stg = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    encg = enc.clone().requires_grad_(True)
    ref_loss, ref_logits = O.decoder_forward(ids, am, encg, enc_mask, stg, cfg)
    ref_loss.backward()
    enc_d = enc.to(dev()).to(BF).requires_grad_(True)
    dec.train()
    out = dec(input_ids=ids.to(dev()), attention_mask=am.to(dev()), encoder_outputs=enc_d, encoder_attention_mask=enc_mask.to(dev()))
    assert abs(out["loss"].item() - ref_loss.item()) <= 2e-3 * max(1.0, abs(ref_loss.item())), (out["loss"].item(), ref_loss.item())
    assert close_bf16(out["logits"].float().cpu(), ref_logits.detach())
    out["loss"].backward()
    named = dict(dec.decoder.named_parameters())
    for n in stg:
        if stg[n].grad is None:
            continue
        got, ref = named[n].grad.float().cpu(), stg[n].grad
        if ref.norm() <= 1e-8 * max(1.0, ref.numel() ** 0.5):       # pad rows, and key biases (the softmax is invariant to them: rounding noise only)
            assert got.norm() <= 1e-4, (n, got.norm())
            continue
        assert grads_close(got, ref, cos=0.998, rel=6e-2), (n, cosine(got, ref), rel_l2(got, ref))
    assert grads_close(enc_d.grad.float().cpu(), encg.grad)


@pytest.mark.parametrize("mt", ["roberta", "bert"])
def test_proto_decoder_greedy_and_beam_ids_vs_golden(golden, tmp_path, mt):
    """HF generate() on the reference's DecoderModel(proto=dir): greedy and beam-4 token ids, bit-exact (fp32 decode step)"""
    from vilmedic_amd.blocks.huggingface.decoder.decoder_model import DecoderModel
    d, cfg, st, ids, am, enc = proto_dec_case(golden("g23_proto_towers"), mt)
    dec = DecoderModel(dict(proto=R.write_proto_dir(str(tmp_path / "ckpt"), mt, d["cfg"], st))).to(dev()).eval()
    sp = d["specials"]
    args = dict(bos_token_id=sp["cls"], eos_token_id=sp["sep"], pad_token_id=sp["pad"], max_length=d["max_len"], return_dict_in_generate=True)
    start = torch.full((d["B"], 1), sp["cls"], dtype=torch.long, device=dev())
    with torch.no_grad():
        o1 = dec.generate(input_ids=start, encoder_hidden_states=enc.to(dev()), encoder_attention_mask=d["enc_mask"].to(dev()), num_beams=1, **args)
        o4 = dec.generate(input_ids=start, encoder_hidden_states=enc.to(dev()), encoder_attention_mask=d["enc_mask"].to(dev()), num_beams=4, **args)
    assert torch.equal(o1.sequences.cpu(), d["beams1"]["sequences"]), (o1.sequences.tolist(), d["beams1"]["sequences"].tolist())
    assert torch.equal(o4.sequences.cpu(), d["beams4"]["sequences"]), (o4.sequences.tolist(), d["beams4"]["sequences"].tolist())
    torch.testing.assert_close(o4.sequences_scores.cpu(), d["beams4"]["scores"], rtol=1e-4, atol=1e-4)


def test_deit_visual_encoder_and_rrg_hf_vs_golden(golden):
    from oracle import torch_ref as O
    from vilmedic_amd.blocks.vision import VisualEncoder
    from vilmedic_amd.models import RRG_HF
    g = golden("g24_deit")
    cfg = g["cfg"]
    st = R.rand_state(R.deit_shapes(cfg), g["seed"])
    enc = VisualEncoder(backbone="deit", permute="no_permute", dropout_out=0.0, **cfg).to(dev())
    enc.model.load_state_dict(st, strict=True)
    images = R.make_images(g["B"], cfg["image_size"], seed=g["seed"])
    enc.eval()
    with torch.no_grad():
        feats, mask = enc.encode(images.to(dev()))
    assert feats.shape == g["features"].shape and close_bf16(feats.float().cpu(), g["features"])
    assert torch.equal(mask.cpu(), g["mask"])
    # gradients of the special tokens / position table through the two-token assemble kernel, vs the oracle
    stg = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    gen = torch.Generator().manual_seed(5)
    w = torch.randn(g["features"].shape, generator=gen)
    (O.vit_forward(images, stg, cfg) * w).sum().backward()
    enc.train()
    (enc(images.to(dev())).float() * w.to(dev())).sum().backward()
    named = dict(enc.model.named_parameters())
    for n in ("embeddings.cls_token", "embeddings.distillation_token", "embeddings.position_embeddings",
              "embeddings.patch_embeddings.projection.weight", "encoder.layer.0.attention.attention.query.weight"):
        got, ref = named[n].grad.float().cpu(), stg[n].grad
        assert grads_close(got, ref), (n, cosine(got, ref), rel_l2(got, ref))
    # RRG_HF with proto_model: deit (config/RRG/baseline-HF.yml)
    dcfg = g["dec_cfg"]
    m = RRG_HF(vision=dict(proto_model="deit", proto_config="deit", proto_config_args=dict(cfg)),
               decoder=dict(proto_model="bert-generation", proto_config="bert-generation",
                            proto_config_args=dict(dcfg, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))).to(dev())
    dst = R.rand_state(R.decoder_shapes(dcfg), g["seed"] + 1)
    sd = m.state_dict()
    for k, v in st.items():
        sd["model.encoder." + k] = v
    for k, v in dst.items():
        sd["model.decoder." + k] = v
    sd["model.decoder.lm_head.decoder.weight"] = dst["bert.embeddings.word_embeddings.weight"]
    sd["model.decoder.lm_head.decoder.bias"] = dst["lm_head.bias"]
    m.load_state_dict(sd, strict=True)
    ids, am = R.make_reports(g["B"], g["L"], dcfg["vocab_size"], seed=g["seed"])
    m.eval()
    with torch.no_grad():
        out = m(input_ids=ids.to(dev()), attention_mask=am.to(dev()), images=images.to(dev()))
    assert abs(out["loss"].item() - g["loss4"].item()) <= 2e-3 * max(1.0, abs(g["loss4"].item()))
    assert close_bf16(out["logits"].float().cpu(), g["logits4"])


def test_rrg_hf_from_local_checkpoint_directories_vs_golden(golden, tmp_path):
    """RRG_HF(encoderdecoder=<dir>) and RRG_HF(vision=<dir>, decoder=<dir>) (ref:models/rrg/RRG_HF.py:24-25, 48-49, 86-87) on the checkpoint
    directories fixture G25 was written from: loss and logits of the reference's own class (HF from_pretrained + RRG_HF.forward), and the
    gradient of the loaded enc_to_dec_proj against the oracle"""
    from oracle import torch_ref as O
    from vilmedic_amd.models import RRG_HF
    g = golden("g25_rrg_hf_pretrained")
    a, b = g["ved"], g["strings"]
    vst = R.rand_state(R.vit_pooled_shapes(a["vit_cfg"]), a["seed"])
    dst = R.rand_state(R.decoder_shapes(a["dec_cfg"]), a["seed"] + 1)
    gen = torch.Generator().manual_seed(a["seed"] + 2)
    ved = {"encoder." + k: v for k, v in vst.items()}
    ved.update({"decoder." + k: v for k, v in dst.items()})
    ved["enc_to_dec_proj.weight"] = 0.1 * torch.randn(a["dec_cfg"]["hidden_size"], a["vit_cfg"]["hidden_size"], generator=gen)
    ved["enc_to_dec_proj.bias"] = 0.02 * torch.randn(a["dec_cfg"]["hidden_size"], generator=gen)
    assert R.state_checksum(ved) == a["checksum"]
    m = RRG_HF(encoderdecoder=R.write_ved_dir(str(tmp_path / "ved"), "vit", a["vit_cfg"], a["dec_cfg"], ved)).to(dev())
    images = R.make_images(a["B"], a["vit_cfg"]["image_size"], seed=a["seed"])
    ids, am = R.make_reports(a["B"], a["L"], a["dec_cfg"]["vocab_size"], seed=a["seed"])
    m.train()                                        # (the checkpoint's dropout probabilities are 0)
    out = m(input_ids=ids.to(dev()), attention_mask=am.to(dev()), images=images.to(dev()))
    assert abs(out["loss"].item() - a["loss"].item()) <= 2e-3 * max(1.0, abs(a["loss"].item())), (out["loss"].item(), a["loss"].item())
    assert close_bf16(out["logits"].float().cpu(), a["logits"])
    out["loss"].backward()
    stg = {"model." + k: v.clone().requires_grad_(True) for k, v in ved.items()}
    O.rrg_hf_forward(images, ids, am, stg, a["vit_cfg"], a["dec_cfg"])[0].backward()
    got, ref = m.model.enc_to_dec_proj.weight.grad.float().cpu(), stg["model.enc_to_dec_proj.weight"].grad
    assert grads_close(got, ref), (cosine(got, ref), rel_l2(got, ref))
    # strings: AutoModel / AutoModelForCausalLM directories
    vst2 = R.rand_state(R.vit_pooled_shapes(b["vit_cfg"]), b["seed"])
    dst2 = R.rand_state(R.decoder_shapes(b["dec_cfg"]), b["seed"] + 1)
    dv = R.write_proto_dir(str(tmp_path / "vit"), "vit", dict(b["vit_cfg"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), vst2)
    dd = R.write_proto_dir(str(tmp_path / "dec"), "bert-generation", dict(b["dec_cfg"], is_decoder=True, add_cross_attention=True,
                                                                         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), dst2)
    m2 = RRG_HF(vision=dv, decoder=dd).to(dev())
    m2.eval()
    with torch.no_grad():
        out2 = m2(input_ids=ids.to(dev()), attention_mask=am.to(dev()), images=R.make_images(b["B"], b["vit_cfg"]["image_size"], seed=b["seed"]).to(dev()))
    assert abs(out2["loss"].item() - b["loss"].item()) <= 2e-3 * max(1.0, abs(b["loss"].item()))
    assert close_bf16(out2["logits"].float().cpu(), b["logits"])


def test_heads_of_48_columns_train_through_the_padded_path_vs_oracle():
    """BertGenerationConfig's default 16 heads on hidden_size 768 (ref:config/RRG/baseline-HF.yml:26-30) are 48 columns wide: here 2 heads on
    hidden_size 96, self- and cross-attention, loss and gradients against the oracle"""
    from oracle import torch_ref as O
    from vilmedic_amd.blocks.huggingface.decoder.decoder_model import DecoderModel
    cfg = dict(R.DEC_TINY, hidden_size=96, num_attention_heads=2, intermediate_size=192)
    st = R.rand_state(R.decoder_shapes(cfg), 91)
    dec = DecoderModel(dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfg)).to(dev())
    full = dict(st)
    full["lm_head.decoder.weight"], full["lm_head.decoder.bias"] = st["bert.embeddings.word_embeddings.weight"], st["lm_head.bias"]
    dec.decoder.load_state_dict(full, strict=True)
    B, L, S = 3, 18, 9
    ids, am = R.make_reports(B, L, cfg["vocab_size"], seed=92)
    gen = torch.Generator().manual_seed(93)
    enc = torch.randn(B, S, cfg["hidden_size"], generator=gen)
    enc_mask = torch.ones(B, S, dtype=torch.bool)
    enc_mask[1, 5:] = False
    enc[~enc_mask] = 0.0
    stg = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    ref_loss, ref_logits = O.decoder_forward(ids, am, enc, enc_mask, stg, cfg)
    ref_loss.backward()
    dec.train()
    out = dec(input_ids=ids.to(dev()), attention_mask=am.to(dev()), encoder_outputs=enc.to(dev()).to(BF), encoder_attention_mask=enc_mask.to(dev()))
    assert abs(out["loss"].item() - ref_loss.item()) <= 2e-3 * max(1.0, abs(ref_loss.item()))
    assert close_bf16(out["logits"].float().cpu(), ref_logits.detach())
    out["loss"].backward()
    named = dict(dec.decoder.named_parameters())
    for n in ("bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.0.attention.self.value.weight",
              "bert.encoder.layer.1.crossattention.self.key.weight", "bert.encoder.layer.1.crossattention.self.value.bias",
              "bert.encoder.layer.0.attention.output.dense.weight"):
        got, ref = named[n].grad.float().cpu(), stg[n].grad
        assert grads_close(got, ref), (n, cosine(got, ref), rel_l2(got, ref))


def test_heads_of_48_columns_decode_on_padded_projections_vs_oracle():
    """the cached decode step of the same head width: zero-padded projection weights (generation.DecodeState._build_padded_weights), greedy
    and beam-4 token ids against the oracle's full-prefix recompute, bit-exact (fp32 step)"""
    from oracle import torch_ref as O
    from vilmedic_amd.blocks.huggingface.decoder.decoder_model import DecoderModel
    cfg = dict(R.DEC_TINY, hidden_size=96, num_attention_heads=2, intermediate_size=192)
    st = R.rand_state(R.decoder_shapes(cfg), 95, std=0.6, emb_std=0.2, qk_std=0.15, pos_std=0.6)
    st["lm_head.bias"][cfg["eos_token_id"]] += 3.0
    dec = DecoderModel(dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfg)).to(dev()).eval()
    full = dict(st)
    full["lm_head.decoder.weight"], full["lm_head.decoder.bias"] = st["bert.embeddings.word_embeddings.weight"], st["lm_head.bias"]
    dec.decoder.load_state_dict(full, strict=True)
    B, S, T = 4, 9, 20
    gen = torch.Generator().manual_seed(96)
    enc = torch.randn(B, S, cfg["hidden_size"], generator=gen)
    enc_mask = torch.ones(B, S, dtype=torch.bool)
    enc_mask[3, 4:] = False
    enc[~enc_mask] = 0.0
    with torch.no_grad():
        ref1 = O.greedy_decode(enc, enc_mask, st, cfg, 0, 2, 1, T)
        ref4, sc4 = O.beam_decode(enc, enc_mask, st, cfg, 0, 2, 1, T, 4)
        start = torch.zeros((B, 1), dtype=torch.long, device=dev())
        kw = dict(encoder_hidden_states=enc.to(dev()), encoder_attention_mask=enc_mask.to(dev()), bos_token_id=0, eos_token_id=2, pad_token_id=1,
                  max_length=T, return_dict_in_generate=True)
        o1 = dec.generate(input_ids=start, **kw)
        o4 = dec.generate(input_ids=start, num_beams=4, **kw)
    assert torch.equal(o1.sequences.cpu(), ref1), (o1.sequences.tolist(), ref1.tolist())
    assert torch.equal(o4.sequences.cpu(), ref4), (o4.sequences.tolist(), ref4.tolist())
    torch.testing.assert_close(o4.sequences_scores.cpu(), sc4, rtol=1e-4, atol=1e-4)
