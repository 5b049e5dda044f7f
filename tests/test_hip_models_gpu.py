"""GPU parity of the composed hot path (ViT encoder, cross-attention decoder, RRG train steps) against
(a) the CPU oracle on the same weights/inputs and (b) the golden fixtures generated from the reference.

Tolerance (north_star: "outputs equal to reference within 1e-3 bf16"): activations travel in bf16 (8-bit
mantissa, eps = 3.9e-3) with fp32 accumulation/statistics, so we require
    loss:   |hip - ref| <= 2e-3 * max(1, |ref|)
    logits / features: |err| <= 3e-2 + 3e-2*|ref| elementwise (a few bf16 ulps: the OUTPUT itself is rounded to bf16,
                       ulp(2.0) = 1.6e-2) and mean-abs error <= 1e-2
    gradients: cosine similarity >= 0.999 and relative L2 error <= 3e-2 vs the fp32 oracle.
"""
import pytest
import torch

import golden_recipes as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def rel_l2(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def close_bf16(got, ref):
    err = (got - ref).abs()
    return bool((err <= 3e-2 + 3e-2 * ref.abs()).all()) and err.mean().item() <= 1e-2


def cosine(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


def build_vit(cfg, seed):
    from vilmedic_amd.blocks.vision import VisualEncoder
    enc = VisualEncoder(backbone="vit", permute="no_permute", dropout_out=0.0, **cfg).to(dev())
    st = R.rand_state(R.vit_shapes(cfg), seed)
    enc.model.load_state_dict(st, strict=True)
    return enc, st


def build_decoder(cfg, seed, **recipe):
    from vilmedic_amd.blocks.huggingface.decoder.decoder_model import DecoderModel
    d = dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfg)
    dec = DecoderModel(d).to(dev())
    st = R.rand_state(R.decoder_shapes(cfg), seed, std=recipe.get("std", 0.05), emb_std=recipe.get("emb_std"),
                      qk_std=recipe.get("qk_std"), pos_std=recipe.get("pos_std"))
    st["lm_head.bias"][cfg["eos_token_id"]] += recipe.get("eos_bias", 0.0)
    full = dict(st)
    full["lm_head.decoder.weight"] = st["bert.embeddings.word_embeddings.weight"]
    full["lm_head.decoder.bias"] = st["lm_head.bias"]
    dec.decoder.load_state_dict(full, strict=True)
    return dec, st


def test_state_dict_keys_match_reference_names():
    enc, st = build_vit(R.VIT_TINY, 1)
    assert set(enc.model.state_dict().keys()) == set(st.keys())
    dec, st = build_decoder(R.DEC_TINY, 2)
    assert set(dec.decoder.state_dict().keys()) == set(st.keys()) | {"lm_head.decoder.weight", "lm_head.decoder.bias"}


@pytest.mark.parametrize("name", ["g1_vit_tiny", "g2_vit_b16_1layer"])
def test_vit_features_vs_golden_and_oracle(golden, name):
    from oracle import torch_ref as O
    g = golden(name)
    cfg = g["cfg"]
    enc, st = build_vit(cfg, g["seed"])
    images = R.make_images(g["B"], cfg["image_size"], seed=g["seed"])
    if g.get("blank_image") is not None:
        images[g["blank_image"]] = 0.0
    enc.eval()
    with torch.no_grad():
        feats, mask = enc.encode(images.to(dev()))
    feats = feats.float().cpu()
    ref = O.vit_forward(images, st, cfg)
    assert close_bf16(feats, ref), ((feats - ref).abs().max(), (feats - ref).abs().mean())
    if name == "g1_vit_tiny":
        assert close_bf16(feats, g["features"])
        assert torch.equal(mask.cpu(), g["mask"])
    else:
        assert close_bf16(feats[:, ::8], g["features"])


def test_decoder_loss_logits_grads_vs_golden(golden):
    g = golden("g3_decoder_tiny")
    cfg = g["cfg"]
    dec, st = build_decoder(cfg, g["seed"])
    ids, am = R.make_reports(g["B"], g["L"], cfg["vocab_size"], seed=g["seed"])
    gen = torch.Generator().manual_seed(g["seed"] + 1)
    enc = torch.randn(g["B"], g["S"], cfg["hidden_size"], generator=gen)
    enc[~g["enc_mask"]] = 0.0
    enc_d = enc.to(dev()).to(BF).requires_grad_(True)
    dec.train()
    out = dec(input_ids=ids.to(dev()), attention_mask=am.to(dev()), encoder_outputs=enc_d,
              encoder_attention_mask=g["enc_mask"].to(dev()))
    assert {"loss", "logits", "past_key_values", "hidden_states", "attentions", "cross_attentions"} <= set(out.keys())
    loss = out["loss"]
    assert abs(loss.item() - g["loss"].item()) <= 2e-3 * max(1.0, abs(g["loss"].item()))
    logits = out["logits"].float().cpu()
    assert logits.shape == g["logits"].shape
    assert close_bf16(logits, g["logits"])
    loss.backward()
    named = dict(dec.decoder.named_parameters())
    for n, ref in g["grads"].items():
        got = named[n].grad.float().cpu()
        assert cosine(got, ref) >= 0.999 and rel_l2(got, ref) <= 3e-2, (n, cosine(got, ref), rel_l2(got, ref))
    eg = enc_d.grad.float().cpu()
    assert cosine(eg, g["enc_grad"]) >= 0.999 and rel_l2(eg, g["enc_grad"]) <= 3e-2


def test_rrg_adam_trajectory_vs_golden(golden):
    """three Adam steps of RRG(ViT + decoder): the loss trajectory of the reference (fixture G5/G10)."""
    from vilmedic_amd.models.rrg.RRG import RRG
    g = golden("g5_rrg_tiny")
    model = RRG(decoder=dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **g["dec_cfg"]),
                cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", dropout_out=0.0, **g["vit_cfg"])).to(dev())
    vst = R.rand_state(R.vit_shapes(g["vit_cfg"]), g["seed"])
    dst = R.rand_state(R.decoder_shapes(g["dec_cfg"]), g["seed"] + 1)
    sd = {"enc.model." + k: v for k, v in vst.items()}
    sd.update({"dec.decoder." + k: v for k, v in dst.items()})
    sd["dec.decoder.lm_head.decoder.weight"] = dst["bert.embeddings.word_embeddings.weight"]
    sd["dec.decoder.lm_head.decoder.bias"] = dst["lm_head.bias"]
    model.load_state_dict(sd, strict=True)
    images = R.make_images(g["B"], g["vit_cfg"]["image_size"], seed=g["seed"]).to(dev())
    ids, am = R.make_reports(g["B"], g["L"], g["dec_cfg"]["vocab_size"], seed=g["seed"])
    ids, am = ids.to(dev()), am.to(dev())
    from vilmedic_amd.optim import FusedAdam
    opt = FusedAdam(model, lr=g["lr"])
    model.train()
    for step in range(3):
        out = model(input_ids=ids, attention_mask=am, images=images)
        if step == 0:
            lg = out["logits"].float().cpu()
            assert close_bf16(lg, g["logits0"])
        ref = g["losses"][step].item()
        assert abs(out["loss"].item() - ref) <= 2e-3 * max(1.0, abs(ref)), (step, out["loss"].item(), ref)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()


def test_rrg_with_torch_optim_and_grad_accumulation_matches_fused_path(golden):
    """any torch.optim named in a YAML still works on the arena parameters (ref: executors/utils.py:65-94),
    and loss/grad_accu scaling reaches the weight gradients through the device-scalar alpha."""
    from vilmedic_amd.models.rrg.RRG import RRG
    g = golden("g5_rrg_tiny")

    def make():
        torch.manual_seed(0)
        return RRG(decoder=dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **g["dec_cfg"]),
                   cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", dropout_out=0.0, **g["vit_cfg"])).to(dev())
    m1, m2 = make(), make()
    m2.load_state_dict(m1.state_dict())
    images = R.make_images(4, g["vit_cfg"]["image_size"], seed=3).to(dev())
    ids, am = R.make_reports(4, 20, g["dec_cfg"]["vocab_size"], seed=3)
    ids, am = ids.to(dev()), am.to(dev())
    (m1(input_ids=ids, attention_mask=am, images=images)["loss"]).backward()
    (m2(input_ids=ids, attention_mask=am, images=images)["loss"] / 4).backward()
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert rel_l2(b.grad * 4, a.grad) <= 2e-2, n
    opt = torch.optim.Adam(m1.parameters(), lr=1e-3)
    l0 = m1(input_ids=ids, attention_mask=am, images=images)["loss"].item()
    for _ in range(5):
        opt.zero_grad(set_to_none=True)      # the reference's default path drops .grad; the arena re-attaches it
        out = m1(input_ids=ids, attention_mask=am, images=images)
        out["loss"].backward()
        opt.step()
    assert m1(input_ids=ids, attention_mask=am, images=images)["loss"].item() < l0 - 0.05


def _oracle_logp_of(seq, enc, enc_mask, st, cfg):
    """teacher-force ``seq`` through the fp32 oracle: log-softmax [B, T-1, V] for predicting tokens 1..T-1"""
    from oracle import torch_ref as O
    h = O.decoder_hidden(seq[:, :-1], None, enc, enc_mask, st, cfg)
    return torch.log_softmax(O.lm_logits(h, st).float(), -1)


def _g7_setup(golden):
    g = golden("g7_decode")
    cfg, rc = g["cfg"], g["recipe"]
    dec, st = build_decoder(cfg, g["seed"], **rc)
    dec.eval()
    gen = torch.Generator().manual_seed(g["seed"] + 1)
    enc = torch.randn(g["B"], g["S"], cfg["hidden_size"], generator=gen)
    enc[~g["enc_mask"]] = 0.0
    start = torch.zeros(g["B"], 1, dtype=torch.long, device=dev())
    common = dict(bos_token_id=0, eos_token_id=2, pad_token_id=1, max_length=g["max_len"])
    return g, cfg, dec, st, enc, start, common


def test_greedy_and_beam_decode_bit_exact_vs_golden(golden):
    """north_star: "bit-exact token indices for greedy decode".  The default decode step is the fp32 one (csrc/decode_f32.hip):
    EVERY row of the reference's generate() output (fixture G7, written by the reference's own DecoderModel + HF generate) must
    be reproduced token for token -- greedy, and beam-4 under both length penalties, with the hypothesis scores within 1e-4.
    The fixture decoder is deliberately chaotic (large random weights): top-2 logit gaps below 0.1 nat are common,
    which is what makes this a test."""
    g, cfg, dec, st, enc, start, common = _g7_setup(golden)
    enc_d, mask_d = enc.to(dev()), g["enc_mask"].to(dev())
    ids = dec.generate(input_ids=start, encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, **common).cpu()
    ref = g["beams1_lp1.0"]["sequences"]
    assert ids.shape == ref.shape and torch.equal(ids, ref), (ids, ref)
    # eager launches (no HIP graph) and a second call on the cached state give the same ids
    again = dec.generate(input_ids=start, encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, decode_dtype="fp32", **common).cpu()
    assert torch.equal(again, ref)
    for lpen in (1.0, 2.0):
        refb = g[f"beams4_lp{lpen}"]
        out = dec.generate(input_ids=start, encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, num_beams=4,
                           length_penalty=lpen, return_dict_in_generate=True, **common)
        seq = out.sequences.cpu()
        assert seq.shape == refb["sequences"].shape and torch.equal(seq, refb["sequences"]), (lpen, seq, refb["sequences"])
        err = (out.sequences_scores.cpu() - refb["scores"]).abs().max().item()
        print(f"[parity] beam-4 lp={lpen}: sequences identical on all {g['B']} rows, max |score err| = {err:.2e}")
        assert err <= 1e-4


def test_greedy_with_output_scores_and_bad_words_takes_the_library_argmax_loop(golden):
    """generation.sample's second loop (callers: ``output_scores=True``, a greedy call with ``bad_words_ids``): same ids as the one-kernel
    selection, one fp32 score tensor per emitted position whose arg-max is that token; a greedy call with a banned token never emits it;
    sampling outside vm_select_tokens' domain (top_k > 256, output_scores) is served by the torch.multinomial path (advisor, round 5)"""
    g, cfg, dec, st, enc, start, common = _g7_setup(golden)
    enc_d, mask_d = enc.to(dev()), g["enc_mask"].to(dev())
    ref = g["beams1_lp1.0"]["sequences"]
    out = dec.generate(input_ids=start, encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, output_scores=True,
                       return_dict_in_generate=True, **common)
    assert torch.equal(out.sequences.cpu(), ref)
    assert len(out.scores) == ref.shape[1] - 1 and out.scores[0].shape == (g["B"], cfg["vocab_size"])
    assert torch.equal(out.scores[0].argmax(-1).cpu(), ref[:, 1])
    banned = int(ref[0, 1])
    ids = dec.generate(input_ids=start, encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, bad_words_ids=[[banned]], **common).cpu()
    assert int(ids[0, 1]) != banned and not bool((ids[:, 1:] == banned).any())
    # sampling outside vm_select_tokens' domain (ref:blocks/rl/SCST.py:142-157 passes the YAML's top_k and output_scores=True) runs the
    # step-by-step path: HF's processors on the fp32 logits + torch.multinomial on the device
    V = cfg["vocab_size"]
    gen = torch.Generator(device=dev()).manual_seed(11)
    out = dec.generate(input_ids=start, encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, do_sample=True, top_k=min(1000, V), output_scores=True,
                       bad_words_ids=[[1], [0]], return_dict_in_generate=True, generator=gen, **common)
    seq = out.sequences.cpu()
    assert seq.shape[0] == g["B"] and len(out.scores) == seq.shape[1] - 1
    ended = torch.cumsum((seq[:, 1:] == 2).int(), 1) - (seq[:, 1:] == 2).int() > 0
    assert not (((seq[:, 1:] == 0) | (seq[:, 1:] == 1)) & ~ended).any()                # banned tokens only as padding behind a row's eos
    for t, sc in enumerate(out.scores):                                                  # every live token has a finite processed score
        tok = seq[:, t + 1]
        live = ~ended[:, t]
        assert torch.isfinite(sc.cpu()[torch.arange(g["B"]), tok][live]).all()
        assert bool((sc[:, [0, 1]] == -float("inf")).all())
    # top_k = 1 sampling is the arg-max: the path's filter is exact
    one = dec.generate(input_ids=start, encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, do_sample=True, top_k=1, output_scores=True,
                       return_dict_in_generate=True, generator=gen, **common)
    assert torch.equal(one.sequences.cpu(), ref)
    # greedy_rows on that path: the first rows are the arg-max of the raw logits
    both = dec.generate(input_ids=torch.cat([start, start]), encoder_hidden_states=torch.cat([enc_d, enc_d]), encoder_attention_mask=torch.cat([mask_d, mask_d]),
                        do_sample=True, greedy_rows=g["B"], top_k=min(1000, V), bad_words_ids=[[1], [0]], generator=gen, **common).cpu()
    n = min(both.shape[1], ref.shape[1])
    assert torch.equal(both[:g["B"], :n], ref[:, :n])


def test_bf16_decode_step_stays_within_margin_of_fp32_oracle(golden):
    """the training-precision decode step (decode_dtype="bf16": SCST rollouts, throughput runs): every emitted token is an
    fp32-oracle arg-max of ITS OWN prefix up to a margin, where the margin is what bf16 activations can move a logit gap by
    (measured and printed); rows whose reference path never passes a near-tie are identical to the reference."""
    g, cfg, dec, st, enc, start, common = _g7_setup(golden)
    enc_d, mask_d = enc.to(dev()), g["enc_mask"].to(dev())
    ids = dec.generate(input_ids=start, encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, decode_dtype="bf16", **common).cpu()
    ref = g["beams1_lp1.0"]["sequences"]
    lp = _oracle_logp_of(ids, enc, g["enc_mask"], st, cfg)
    worst = 0.0
    for b in range(g["B"]):
        for t in range(1, ids.shape[1]):
            if ids[b, t] == 1:        # padding after eos
                break
            worst = max(worst, float(lp[b, t - 1].max() - lp[b, t - 1, ids[b, t]]))
    lp_ref = _oracle_logp_of(ref, enc, g["enc_mask"], st, cfg)
    n_clear = n_same = 0
    for b in range(g["B"]):
        n = int((ref[b, 1:] != 1).sum())
        top2 = lp_ref[b, :n].topk(2, dim=-1)[0]
        L = min(ids.shape[1], ref.shape[1])
        same = torch.equal(ids[b, :L], ref[b, :L])
        n_same += int(same)
        if n == 0 or float((top2[:, 0] - top2[:, 1]).min()) > 0.25:
            n_clear += 1
            assert same, (b, ids[b], ref[b])
    print(f"[parity] bf16 greedy decode: worst arg-max margin {worst:.3f} nat, {n_same}/{g['B']} rows identical to the reference "
          f"({n_clear} rows have no near-tie)")
    assert worst <= 0.25


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_paired_rollout_greedy_rows_equal_a_separate_greedy_decode(golden, dtype):
    """generation.sample(greedy_rows=B) -- SCST's two rollouts in ONE decode loop of 2B rows (blocks/rl/SCST.forward_rollouts): the first B
    rows must emit exactly the tokens of a separate greedy generate() on the same encoder states (rows are independent of their batch
    mates in every kernel of the step), the other B rows sample from the bad-word + top-k filtered distribution (never a banned token
    before their eos) on THEIR encoder states."""
    from vilmedic_amd.generation import trim_to_last_eos
    g, cfg, dec, st, enc, start, common = _g7_setup(golden)
    enc_d, mask_d = enc.to(dev()).to(BF), g["enc_mask"].to(dev())
    alone = dec.generate(input_ids=start, encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, decode_dtype=dtype, **common)
    B = enc_d.shape[0]
    enc2 = torch.cat([enc_d, enc_d.flip(0)])                          # the sampled rows run on different encoder states
    mask2 = torch.cat([mask_d, mask_d.flip(0)])
    both = dec.generate(input_ids=torch.zeros(2 * B, 1, dtype=torch.long, device=dev()), encoder_hidden_states=enc2, encoder_attention_mask=mask2,
                        do_sample=True, greedy_rows=B, top_k=5, bad_words_ids=[[1], [0]], decode_dtype=dtype,
                        generator=torch.Generator(device=dev()).manual_seed(3), **common)
    greedy = trim_to_last_eos(both[:B], 2)
    n = min(greedy.shape[1], alone.shape[1])
    assert torch.equal(greedy[:, :n], alone[:, :n]) and (greedy[:, n:] == 1).all() and (alone[:, n:] == 1).all()
    sampled = both[B:, 1:]
    ended = torch.cumsum((sampled == 2).int(), 1) - (sampled == 2).int() > 0          # positions after a row's eos hold pads
    assert not (((sampled == 0) | (sampled == 1)) & ~ended).any()


def test_two_phase_backward_equals_single_backward(golden):
    """ArenaDDP's overlap trick (decoder consumes detached features; encoder backward runs as a second phase) yields the
    same gradients as one loss.backward()."""
    from vilmedic_amd.models.rrg.RRG import RRG
    g = golden("g5_rrg_tiny")

    def make():
        torch.manual_seed(0)
        return RRG(decoder=dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **g["dec_cfg"]),
                   cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", dropout_out=0.0, **g["vit_cfg"])).to(dev())
    m1, m2 = make(), make()
    m2.load_state_dict(m1.state_dict())
    images = R.make_images(4, g["vit_cfg"]["image_size"], seed=3).to(dev())
    ids, am = R.make_reports(4, 20, g["dec_cfg"]["vocab_size"], seed=3)
    ids, am = ids.to(dev()), am.to(dev())
    m1.train(), m2.train()
    m1(input_ids=ids, attention_mask=am, images=images)["loss"].backward()
    m2.split_backward = True
    out = m2(input_ids=ids, attention_mask=am, images=images)
    out["loss"].backward()
    feats, leaf = m2._split
    assert m2.enc.model.layernorm.weight.grad.abs().sum() == 0          # encoder untouched by phase 1
    feats.backward(leaf.grad)
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        torch.testing.assert_close(b.grad, a.grad, rtol=1e-3, atol=1e-5, msg=n)
    # arena layout assumption of ArenaDDP: [decoder parameters | encoder parameters]
    assert min(p._vm_off for p in m2.enc.parameters()) >= max(p._vm_off + p.numel() for p in m2.dec.parameters())


def test_dropout_mask_fused_into_layernorm_backward_equals_the_separate_pass(golden):
    """With dropout on, the gradient of  y = dropout(linear(x)) + residual  is masked either by vm_dropout_apply_bf16 in the linear's
    backward or (default) by the LayerNorm backward kernel that produced it (vm_layernorm_bwd_partial_dropout -> ops._masked_grad).
    Same seeds, same model, both ways: same loss, every parameter gradient equal up to the single bf16 rounding the fused form
    skips, and the fused run must actually have used the fused path (no dropout_apply launch)."""
    from vilmedic_amd import ops
    from vilmedic_amd.arena import arena_of
    from vilmedic_amd.models.rrg.RRG import RRG
    g = golden("g5_rrg_tiny")
    torch.manual_seed(0)
    m = RRG(decoder=dict(proto=None, hidden_dropout_prob=0.2, attention_probs_dropout_prob=0.1, **g["dec_cfg"]),
            cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", dropout_out=0.0, **g["vit_cfg"])).to(dev()).train()
    images = R.make_images(4, g["vit_cfg"]["image_size"], seed=3).to(dev())
    ids, am = R.make_reports(4, 20, g["dec_cfg"]["vocab_size"], seed=3)
    ids, am = ids.to(dev()), am.to(dev())
    arena = arena_of(m)
    calls = {"n": 0}
    real = ops.dropout_apply

    def counting(x, p, seed):
        calls["n"] += 1
        return real(x, p, seed)
    out = {}
    try:
        ops.dropout_apply = counting
        for fuse in (True, False):
            ops.FUSE_LN_DROPOUT = fuse
            ops.manual_seed(77)
            arena.zero_grad()
            calls["n"] = 0
            loss = m(input_ids=ids, attention_mask=am, images=images)["loss"]
            loss.backward()
            torch.cuda.synchronize()
            out[fuse] = (loss.item(), arena.gflat.clone(), calls["n"])
    finally:
        ops.dropout_apply = real
        ops.FUSE_LN_DROPOUT = True
    assert abs(out[True][0] - out[False][0]) <= 1e-6 * abs(out[False][0])        # same forward; the CE kernel's atomic loss sum is order-dependent in the last bits
    assert out[False][2] > 0 and out[True][2] < out[False][2], (out[True][2], out[False][2])      # the fused run skipped the separate passes
    err = rel_l2(out[True][1], out[False][1])
    print(f"[parity] fused LN-backward dropout mask vs separate pass: grad rel-l2 {err:.2e}, dropout_apply launches {out[True][2]} vs {out[False][2]}")
    assert err < 5e-3, err


def _rrg_hf_pair(vit_cfg, dec_cfg, seed):
    from vilmedic_amd.models import RRG_HF
    m = RRG_HF(vision=dict(proto_model="vit", proto_config="vit", proto_config_args=dict(vit_cfg)),
               decoder=dict(proto_model="bert-generation", proto_config="bert-generation",
                            proto_config_args=dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **dec_cfg))).to(dev())
    vst = R.rand_state(R.vit_shapes(vit_cfg), seed)
    dst = R.rand_state(R.decoder_shapes(dec_cfg), seed + 1)
    sd = {"model.encoder." + k: v for k, v in vst.items()}
    sd.update({"model.decoder." + k: v for k, v in dst.items()})
    sd["model.decoder.lm_head.decoder.weight"] = dst["bert.embeddings.word_embeddings.weight"]
    sd["model.decoder.lm_head.decoder.bias"] = dst["lm_head.bias"]
    own = m.state_dict()
    for k in own:                      # pooler (unused by the path) and enc_to_dec_proj keep their own init
        sd.setdefault(k, own[k])
    m.load_state_dict(sd, strict=True)
    return m, {k: v.detach().float().cpu() for k, v in m.state_dict().items()}


def test_rrg_hf_single_image_equals_rrg_and_multi_image_matches_oracle(golden):
    """RRG_HF: the VisionEncoderDecoder wiring of the same kernels.  4-D batch: identical loss / logits to
    RRG on the same weights apart from the key mask (RRG masks all-zero feature rows, RRG_HF passes None: both attend
    everything here).  5-D batch with images_mask and an enc_to_dec_proj: against the fp32 oracle."""
    from oracle import torch_ref as O
    from vilmedic_amd.models.rrg.RRG import RRG
    g = golden("g5_rrg_tiny")
    vit_cfg, dec_cfg = g["vit_cfg"], g["dec_cfg"]
    hf, st = _rrg_hf_pair(vit_cfg, dec_cfg, g["seed"])
    rrg = RRG(decoder=dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **dec_cfg),
              cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", dropout_out=0.0, **vit_cfg)).to(dev())
    sd = {k.replace("model.encoder.", "enc.model.").replace("model.decoder.", "dec.decoder."): v for k, v in st.items() if "pooler" not in k}
    rrg.load_state_dict(sd, strict=True)
    images = R.make_images(g["B"], vit_cfg["image_size"], seed=g["seed"]).to(dev())
    ids, am = R.make_reports(g["B"], g["L"], dec_cfg["vocab_size"], seed=g["seed"])
    ids, am = ids.to(dev()), am.to(dev())
    hf.train(); rrg.train()
    a = hf(input_ids=ids, attention_mask=am, images=images)
    b = rrg(input_ids=ids, attention_mask=am, images=images)
    assert abs(a["loss"].item() - b["loss"].item()) <= 1e-4 and abs(a["loss"].item() - g["losses"][0].item()) <= 2e-3 * max(1.0, abs(g["losses"][0].item()))
    assert torch.equal(a["logits"], b["logits"])
    a["loss"].backward(); b["loss"].backward()
    ga = hf.model.decoder.bert.encoder.layer[0].crossattention.self.key.weight.grad
    gb = rrg.dec.decoder.bert.encoder.layer[0].crossattention.self.key.weight.grad
    assert rel_l2(ga, gb) <= 1e-3 and ga.abs().sum().item() > 0

    # ---- multi-image + projection against the oracle (decoder hidden 128, encoder hidden 64 -> enc_to_dec_proj)
    vit2 = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128, image_size=32, patch_size=16,
                num_channels=3, layer_norm_eps=1e-12)
    hf2, st2 = _rrg_hf_pair(vit2, dec_cfg, 11)
    B, N = 3, 2
    imgs = R.make_images(B * N, 32, seed=5).view(B, N, 3, 32, 32)
    imask = torch.tensor([[1, 1], [1, 0], [1, 1]], dtype=torch.bool)
    ids2, am2 = R.make_reports(B, 12, dec_cfg["vocab_size"], seed=5)
    ref_loss, ref_logits = O.rrg_hf_forward(imgs, ids2, am2, st2, vit2, dec_cfg, images_mask=imask)
    hf2.train()
    out = hf2(input_ids=ids2.to(dev()), attention_mask=am2.to(dev()), images=imgs.to(dev()), images_mask=imask.to(dev()))
    assert abs(out["loss"].item() - ref_loss.item()) <= 2e-3 * max(1.0, abs(ref_loss.item())), (out["loss"].item(), ref_loss.item())
    assert close_bf16(out["logits"].float().cpu(), ref_logits)
    out["loss"].backward()
    assert hf2.model.enc_to_dec_proj.weight.grad.abs().sum().item() > 0


def test_ensemble_greedy_and_beam_decode_vs_oracle(golden):
    """n-best checkpoint ensembling: two decoders with their own encoder states and KV caches, logits
    summed before the log-softmax (ref: blocks/huggingface/decoder/beam_search.py:243-262) -- against the fp32 oracle's
    ensemble decode with the same margin-aware criterion as the single-model test (the reference's own ensemble file does not
    import at HEAD; the oracle's ensemble branch is pinned by fixture G22, see the next test)."""
    from oracle import torch_ref as O
    g = golden("g7_decode")
    cfg, rc = g["cfg"], g["recipe"]
    dec_a, st_a = build_decoder(cfg, g["seed"], **rc)
    dec_b, st_b = build_decoder(cfg, g["seed"] + 40, **rc)
    dec_a.eval(); dec_b.eval()
    gen = torch.Generator().manual_seed(g["seed"] + 1)
    B, S = g["B"], g["S"]
    enc_a = torch.randn(B, S, cfg["hidden_size"], generator=gen)
    enc_b = torch.randn(B, S, cfg["hidden_size"], generator=gen)
    mask = g["enc_mask"]
    encs, masks, sts = [enc_a, enc_b], [mask, mask], [st_a, st_b]
    eo = [dict(encoder_hidden_states=e.to(dev()), encoder_attention_mask=mask.to(dev())) for e in encs]
    start = torch.zeros(B, 1, dtype=torch.long, device=dev())
    common = dict(bos_token_id=0, eos_token_id=2, pad_token_id=1, max_length=g["max_len"])

    def ens_logp(seq):          # teacher-forced ensemble log-probs under the fp32 oracle
        logits = sum(O.lm_logits(O.decoder_hidden(seq[:, :-1], None, e, m, st, cfg), st).float() for e, m, st in zip(encs, masks, sts))
        return torch.log_softmax(logits, -1)

    ids = dec_a.generate(input_ids=start, hf_models=[dec_a.decoder, dec_b.decoder], encoders_outputs=eo, **common).cpu()
    single = dec_a.generate(input_ids=start, encoder_hidden_states=eo[0]["encoder_hidden_states"],
                            encoder_attention_mask=eo[0]["encoder_attention_mask"], **common).cpu()
    assert not torch.equal(ids[:, :min(ids.shape[1], single.shape[1])], single[:, :min(ids.shape[1], single.shape[1])])   # the 2nd model matters
    # fp32 decode step (the default): token for token the oracle's ensemble decode, on every row
    ref = O.greedy_decode(encs, masks, sts, cfg, 0, 2, 1, g["max_len"])
    assert ids.shape == ref.shape and torch.equal(ids, ref), (ids, ref)
    refb, refs = O.beam_decode(encs, masks, sts, cfg, 0, 2, 1, g["max_len"], 4)
    out = dec_a.generate(input_ids=start, hf_models=[dec_a.decoder, dec_b.decoder], encoders_outputs=eo, num_beams=4, return_dict_in_generate=True, **common)
    seq = out.sequences.cpu()
    assert seq.shape == refb.shape and torch.equal(seq, refb), (seq, refb)
    assert (out.sequences_scores.cpu() - torch.as_tensor(refs)).abs().max().item() <= 1e-4
    # training-precision step: margin criterion under the fp32 oracle's ensemble log-probs
    idb = dec_a.generate(input_ids=start, hf_models=[dec_a.decoder, dec_b.decoder], encoders_outputs=eo, decode_dtype="bf16", **common).cpu()
    lp = ens_logp(idb)
    for b in range(B):
        for t in range(1, idb.shape[1]):
            if idb[b, t] == 1:
                break
            assert lp[b, t - 1].max() - lp[b, t - 1, idb[b, t]] <= 0.25, (b, t)


def test_ensemble_decode_vs_hf_generate_fixture(golden):
    """fixture G22: HF ``generate`` over the summed logits of two reference DecoderModels (different encoder lengths and masks) --
    the HIP ensemble decode (fp32 decode step, per-model KV caches) gives the same greedy and beam-4 token ids, row for row"""
    g = golden("g22_ensemble_decode")
    cfg, rc = g["cfg"], g["recipe"]
    decs, eo = [], []
    for i, sd in enumerate(g["seeds"]):
        d, _ = build_decoder(cfg, sd, **rc)
        decs.append(d.eval())
        gen = torch.Generator().manual_seed(sd + 1)
        e = torch.randn(g["B"], g["S"] - i, cfg["hidden_size"], generator=gen)
        e[~g["enc_masks"][i]] = 0.0
        eo.append(dict(encoder_hidden_states=e.to(dev()), encoder_attention_mask=g["enc_masks"][i].to(dev())))
    start = torch.zeros(g["B"], 1, dtype=torch.long, device=dev())
    common = dict(bos_token_id=0, eos_token_id=2, pad_token_id=1, max_length=g["max_len"],
                  hf_models=[d.decoder for d in decs], encoders_outputs=eo)
    ids = decs[0].generate(input_ids=start, **common).cpu()
    ref = g["beams1_lp1.0"]["sequences"]
    assert ids.shape == ref.shape and torch.equal(ids, ref), (ids, ref)
    for lp in (1.0, 2.0):
        ref = g[f"beams4_lp{lp}"]
        out = decs[0].generate(input_ids=start, num_beams=4, length_penalty=lp, return_dict_in_generate=True, **common)
        seq = out.sequences.cpu()
        assert seq.shape == ref["sequences"].shape and torch.equal(seq, ref["sequences"]), (lp, seq, ref["sequences"])
        assert (out.sequences_scores.cpu() - ref["scores"]).abs().max().item() <= 1e-4


def build_rrs(g, device=None):
    """RRS on the fixture's weights; returns (model, source batch, target batch)"""
    from vilmedic_amd.models import RRS
    nodrop = dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = RRS(encoder=dict(nodrop, **g["enc_cfg"]), decoder=dict(nodrop, **g["dec_cfg"]))
    est = R.rand_state(R.text_encoder_shapes(g["enc_cfg"]), g["seed"])
    dst = R.rand_state(R.decoder_shapes(g["dec_cfg"]), g["seed"] + 1)
    sd = {"enc.encoder." + k: v for k, v in est.items()}
    sd.update({"dec.decoder." + k: v for k, v in dst.items()})
    sd["dec.decoder.lm_head.decoder.weight"] = dst["bert.embeddings.word_embeddings.weight"]
    sd["dec.decoder.lm_head.decoder.bias"] = dst["lm_head.bias"]
    m.load_state_dict(sd, strict=True)
    src = R.make_reports(g["B"], g["Ls"], g["enc_cfg"]["vocab_size"], seed=g["seed"])
    tgt = R.make_reports(g["B"], g["Lt"], g["dec_cfg"]["vocab_size"], seed=g["seed"] + 1)
    return (m.to(device) if device is not None else m), src, tgt


def test_rrs_loss_logits_grads_vs_golden_and_decode_vs_oracle(golden):
    """RRS (the text -> text caller of the same encoder / decoder kernels): loss, logits, encoder memory and
    gradients on both sides of the cross-attention against the reference's fixture (G13); greedy decode against the oracle."""
    g = golden("g13_rrs_tiny")
    m, (sid, sam), (tid, tam) = build_rrs(g, dev())
    m.train()
    out = m(input_ids=sid.to(dev()), attention_mask=sam.to(dev()), decoder_input_ids=tid.to(dev()), decoder_attention_mask=tam.to(dev()))
    assert abs(out["loss"].item() - g["loss"].item()) <= 2e-3 * max(1.0, abs(g["loss"].item())), (out["loss"].item(), g["loss"].item())
    assert close_bf16(out["logits"].float().cpu(), g["logits"])
    out["loss"].backward()
    en, dn = dict(m.enc.encoder.named_parameters()), dict(m.dec.decoder.named_parameters())
    for n, ref in list(g["enc_grads"].items()) + list(g["dec_grads"].items()):
        got = (en[n] if n in g["enc_grads"] else dn[n]).grad.float().cpu()
        assert cosine(got, ref) >= 0.999 and rel_l2(got, ref) <= 3e-2, (n, cosine(got, ref), rel_l2(got, ref))
    m.eval()
    with torch.no_grad():
        hidden, mask = m.encode(sid.to(dev()), sam.to(dev()))
    assert close_bf16(hidden.float().cpu(), g["encoder_hidden"])
    # greedy decode from the encoder memory: every emitted token is an fp32-oracle arg-max of its own prefix up to a 0.25-nat
    # margin (same criterion as test_greedy_and_beam_decode_vs_golden_and_oracle)
    dcfg = g["dec_cfg"]
    dst = R.rand_state(R.decoder_shapes(dcfg), g["seed"] + 1)
    start = torch.full((g["B"], 1), dcfg["bos_token_id"], dtype=torch.long, device=dev())
    with torch.no_grad():
        hyp = m.dec.decoder.generate(input_ids=start, encoder_hidden_states=hidden, encoder_attention_mask=mask,
                                     bos_token_id=dcfg["bos_token_id"], eos_token_id=dcfg["eos_token_id"], pad_token_id=dcfg["pad_token_id"],
                                     max_length=10).cpu()
    assert hyp.shape[0] == g["B"] and 2 <= hyp.shape[1] <= 10 and (hyp[:, 0] == dcfg["bos_token_id"]).all()
    lp = _oracle_logp_of(hyp, g["encoder_hidden"], sam.bool(), dst, dcfg)
    for b in range(g["B"]):
        for t in range(1, hyp.shape[1]):
            if hyp[b, t] == dcfg["pad_token_id"]:
                break
            margin = lp[b, t - 1].max() - lp[b, t - 1, hyp[b, t]]
            assert margin <= 0.25, (b, t, margin)


def test_graphed_train_step_follows_the_golden_trajectory_and_redraws_dropout(golden):
    """vilmedic_amd.graph.GraphedTrainStep: the whole RRG step (forward, autograd backward with the side stream and the grouped
    weight-gradient launches, fused Adam) replayed from ONE captured HIP graph.  (a) dropout off: step 1 eager, step 2 = capture +
    first replay, step 3 = replay reproduce the reference's 3-step Adam loss trajectory (fixture G5) -- the device-side step counter
    drives Adam's bias correction; (b) dropout on, learning rate 0: two replays on the same batch give DIFFERENT losses (the device seed
    counter advances inside the graph) that are both finite; (c) a NaN loss skips the update on the device (no host read)."""
    from vilmedic_amd.graph import GraphedTrainStep
    from vilmedic_amd.models.rrg.RRG import RRG
    from vilmedic_amd.optim import FusedAdam
    g = golden("g5_rrg_tiny")

    def make(drop):
        m = RRG(decoder=dict(proto=None, hidden_dropout_prob=drop, attention_probs_dropout_prob=drop, **g["dec_cfg"]),
                cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", dropout_out=0.0, **g["vit_cfg"])).to(dev())
        vst = R.rand_state(R.vit_shapes(g["vit_cfg"]), g["seed"])
        dst = R.rand_state(R.decoder_shapes(g["dec_cfg"]), g["seed"] + 1)
        sd = {"enc.model." + k: v for k, v in vst.items()}
        sd.update({"dec.decoder." + k: v for k, v in dst.items()})
        sd["dec.decoder.lm_head.decoder.weight"] = dst["bert.embeddings.word_embeddings.weight"]
        sd["dec.decoder.lm_head.decoder.bias"] = dst["lm_head.bias"]
        m.load_state_dict(sd, strict=True)
        return m.train()
    images = R.make_images(g["B"], g["vit_cfg"]["image_size"], seed=g["seed"]).to(dev())
    ids, am = R.make_reports(g["B"], g["L"], g["dec_cfg"]["vocab_size"], seed=g["seed"])
    batch = dict(input_ids=ids.to(dev()), attention_mask=am.to(dev()), images=images)

    def stepper(model, opt):
        def step(input_ids, attention_mask, images):
            out = model(input_ids=input_ids, attention_mask=attention_mask, images=images, return_logits=False)
            opt.zero_grad()
            opt.gate = out["loss"].detach()
            out["loss"].backward()
            opt.step()
            return out["loss"]
        return step
    # (a) golden trajectory through eager -> capture -> replay
    model = make(0.0)
    opt = FusedAdam(model, lr=g["lr"])
    gs = GraphedTrainStep(stepper(model, opt), batch, optimizer=opt, warmup=1)
    losses = [gs(**batch).item() for _ in range(3)]
    assert gs.graph is not None and opt.steps == 3
    for got, ref in zip(losses, g["losses"]):
        assert abs(got - ref.item()) <= 2e-3 * max(1.0, abs(ref.item())), (losses, g["losses"])
    # (b) fresh dropout masks per replay
    model = make(0.3)
    opt = FusedAdam(model, lr=0.0)
    gs = GraphedTrainStep(stepper(model, opt), batch, optimizer=opt, warmup=1)
    ls = [gs(**batch).item() for _ in range(4)]
    assert all(torch.isfinite(torch.tensor(ls))) and abs(ls[2] - ls[3]) > 1e-4, ls
    # (c) NaN gate inside the graph
    model = make(0.0)
    opt = FusedAdam(model, lr=1e-2)
    gs = GraphedTrainStep(stepper(model, opt), batch, optimizer=opt, warmup=1)
    gs(**batch), gs(**batch)
    from vilmedic_amd.arena import arena_of
    before = arena_of(model).flat.clone()
    bad = dict(batch, images=torch.full_like(images, float("nan")))
    assert not torch.isfinite(gs(**bad)).item()
    assert torch.equal(arena_of(model).flat, before)                   # update skipped on the device
    assert torch.isfinite(gs(**batch)).item() and not torch.equal(arena_of(model).flat, before)
