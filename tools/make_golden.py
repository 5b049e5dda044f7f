#!/usr/bin/env python3
"""Generate tests/golden/*.pt by running the REFERENCE's own block files.

Runs only in the build container (needs /root/reference + HF transformers).
The reference's python never ships: this script loads its block files *by
path* (``vilmedic/__init__`` cannot be imported here: omegaconf/torchvision are
absent), with three shims:
  1. ``Tensor.cuda``/``Module.cuda`` -> identity (reference hard-calls ``.cuda()``);
  2. ``AttrDict`` standing in for OmegaConf ``DictConfig``;
  3. stub ``torchvision`` / ``monai`` modules so ``visual_encoder.py`` imports.
Fixtures hold inputs' recipes (seed/shape, see tests/golden_recipes.py), a
checksum of the generated weights, and the reference's outputs.

    python tools/make_golden.py            # writes tests/golden/*.pt
"""
import importlib.util
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_recipes as R  # noqa: E402

REF = "/root/reference/vilmedic/"
OUT = os.path.join(ROOT, "tests", "golden")

import transformers  # noqa: E402  (must precede the stubs)
import transformers.models.vit.modeling_vit  # noqa: E402,F401
import transformers.models.resnet.modeling_resnet  # noqa: E402,F401
import transformers.models.deit.modeling_deit  # noqa: E402,F401
import transformers.models.poolformer.modeling_poolformer  # noqa: E402,F401

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self


class AttrDict(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def _stub_modules():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.__all__ = []
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    for n in ["monai", "monai.networks", "monai.networks.nets", "monai.networks.nets.densenet"]:
        sys.modules[n] = types.ModuleType(n)
    dn = sys.modules["monai.networks.nets.densenet"]
    for n in ["densenet121", "densenet169", "densenet201", "densenet264"]:
        setattr(dn, n, None)


def load_ref(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF + rel)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_stub_modules()
ve = load_ref("ref_visual_encoder", "blocks/vision/visual_encoder.py")
dm = load_ref("ref_decoder_model", "blocks/huggingface/decoder/decoder_model.py")
em = load_ref("ref_encoder_model", "blocks/huggingface/encoder/encoder_model.py")
lc = load_ref("ref_convirt", "blocks/losses/selfsup/ConVIRTLoss.py")
li = load_ref("ref_infonce", "blocks/losses/selfsup/InfoNCELoss.py")
lg = load_ref("ref_gloria", "blocks/losses/selfsup/GLoRIALoss.py")
ll = load_ref("ref_lsce", "blocks/losses/mvqa/LabelSmoothingCrossEntropyLoss.py")
cl = load_ref("ref_classifier", "blocks/classifier/classifier.py")


# ------------------------------------------------------------------ name maps (HF 5.x <-> pinned 4.55.3)
def vit_to_hf5(name):
    """canonical (4.55.3) ViT name -> transformers 5.x name."""
    name = name.replace("encoder.layer.", "layers.")
    name = name.replace("attention.attention.query", "attention.q_proj")
    name = name.replace("attention.attention.key", "attention.k_proj")
    name = name.replace("attention.attention.value", "attention.v_proj")
    name = name.replace("attention.output.dense", "attention.o_proj")
    name = name.replace("intermediate.dense", "mlp.fc1")
    name = name.replace("output.dense", "mlp.fc2")
    return name


def load_into(module, state, rename=lambda n: n, extra_alias=None):
    sd = {rename(k): v.clone() for k, v in state.items()}
    for dst, src in (extra_alias or {}).items():
        sd[dst] = sd[src]
    missing, unexpected = module.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m]
    assert not missing and not unexpected, (missing, unexpected)


def build_ref_vit(cfg, seed, visual_projection=None):
    enc = ve.VisualEncoder(backbone="vit", permute="no_permute", dropout_out=0.0,
                           visual_projection=AttrDict(visual_projection) if visual_projection else None,
                           **{k: v for k, v in cfg.items()}, attn_implementation="eager")
    st = R.rand_state(R.vit_shapes(cfg), seed)
    load_into(enc.model, st, vit_to_hf5)
    full = {"model." + k: v for k, v in st.items()}
    if visual_projection:
        g = torch.Generator().manual_seed(seed + 77)
        w = 0.05 * torch.randn(visual_projection["out_features"], visual_projection["in_features"], generator=g)
        b = 0.02 * torch.randn(visual_projection["out_features"], generator=g)
        enc.visual_projection.weight.data.copy_(w)
        enc.visual_projection.bias.data.copy_(b)
        full["visual_projection.weight"], full["visual_projection.bias"] = w, b
    return enc.eval(), full


def build_ref_decoder(cfg, seed, std=0.05, emb_std=None, eos_bias=0.0, qk_std=None, pos_std=None):
    d = AttrDict(proto=None, add_cross_attention=True, is_decoder=True, hidden_act="gelu",
                 attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0,
                 position_embedding_type="absolute", use_cache=True, **cfg)
    dec = dm.DecoderModel(d)
    dec.decoder.config._attn_implementation = "eager"
    st = R.rand_state(R.decoder_shapes(cfg), seed, std=std, emb_std=emb_std, qk_std=qk_std, pos_std=pos_std)
    st["lm_head.bias"][cfg["eos_token_id"]] += eos_bias
    load_into(dec.decoder, st, extra_alias={"lm_head.decoder.weight": "bert.embeddings.word_embeddings.weight",
                                            "lm_head.decoder.bias": "lm_head.bias"})
    return dec.eval(), st


def save(name, obj):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".pt")
    torch.save(obj, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------ G1/G2: ViT through VisualEncoder
def gen_vit():
    for tag, cfg, B, seed in [("g1_vit_tiny", R.VIT_TINY, 3, 11), ("g2_vit_b16_1layer", R.VIT_B16_1L, 1, 12)]:
        enc, st = build_ref_vit(cfg, seed)
        images = R.make_images(B, cfg["image_size"], seed=seed)
        if tag == "g1_vit_tiny":
            images[1] = 0.0  # exercises the "mask from magnitude" rule on a blank image (still non-zero features)
        with torch.no_grad():
            feats, mask = enc.encode(images)
        save(tag, dict(cfg=cfg, seed=seed, B=B, blank_image=1 if tag == "g1_vit_tiny" else None,
                       checksum=R.state_checksum(st), features=feats if tag == "g1_vit_tiny" else feats[:, ::8].clone(),
                       features_full_sum=float(feats.double().sum()), mask=mask))


# ------------------------------------------------------------------ G16: multi-image encode (5-D images + images_mask)
def gen_vit_multi():
    """VisualEncoder.encode on [B, N, C, H, W] (visual_encoder.py:161-178).  The reference reads the image count from
    ``images.shape[1]`` AFTER flattening to 4-D, i.e. it takes the CHANNEL count for it: the path is only
    self-consistent when N == C == 3, which is what this fixture uses; with a visual_projection so that masked rows (zero
    features, mask False) still come out as the projection's bias."""
    cfg, seed, B, N = R.VIT_TINY, 41, 2, 3
    vp = dict(in_features=cfg["hidden_size"], out_features=96)
    enc, st = build_ref_vit(cfg, seed, visual_projection=vp)
    images = R.make_images(B * N, cfg["image_size"], seed=seed).view(B, N, 3, cfg["image_size"], cfg["image_size"])
    images_mask = torch.tensor([[1, 1, 0], [1, 0, 1]], dtype=torch.bool)
    with torch.no_grad():
        feats, mask = enc.encode(images, images_mask)
    save("g16_vit_multi_image", dict(cfg=cfg, seed=seed, B=B, N=N, visual_projection=vp, checksum=R.state_checksum(st),
                                     images_mask=images_mask, features=feats, mask=mask))


# ------------------------------------------------------------------ G3/G4: decoder fwd + grads
def gen_decoder():
    cfg, seed, B, L, S = R.DEC_TINY, 21, 4, 24, 10
    dec, st = build_ref_decoder(cfg, seed)
    ids, am = R.make_reports(B, L, cfg["vocab_size"], seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    enc = torch.randn(B, S, cfg["hidden_size"], generator=g)
    enc_mask = torch.ones(B, S, dtype=torch.bool)
    enc_mask[1, 7:] = False
    enc_mask[3, 3:] = False
    enc[~enc_mask] = 0.0
    enc = enc.requires_grad_(True)
    dec.train()  # dropout probs are 0; exercise the training graph
    out = dec(input_ids=ids, attention_mask=am, encoder_outputs=enc, encoder_attention_mask=enc_mask)
    assert set(out.keys()) >= {"loss", "logits"}
    out["loss"].backward()
    named = dict(dec.decoder.named_parameters())
    grad_names = ["bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
                  "bert.embeddings.LayerNorm.weight", "bert.encoder.layer.0.attention.self.query.weight",
                  "bert.encoder.layer.0.crossattention.self.key.weight", "bert.encoder.layer.0.crossattention.self.value.bias",
                  "bert.encoder.layer.1.crossattention.output.dense.weight", "bert.encoder.layer.1.intermediate.dense.weight",
                  "bert.encoder.layer.1.output.LayerNorm.bias", "lm_head.bias"]
    save("g3_decoder_tiny", dict(cfg=cfg, seed=seed, B=B, L=L, S=S, checksum=R.state_checksum(st),
                                 enc_mask=enc_mask, loss=out["loss"].detach(), logits=out["logits"].detach(),
                                 out_keys=sorted(out.keys()),
                                 grads={n: named[n].grad.clone() for n in grad_names},
                                 enc_grad=enc.grad.clone()))


# ------------------------------------------------------------------ G5/G10: RRG (ViT + decoder) loss and Adam trajectory
def gen_rrg():
    vcfg, dcfg, seed, B, L = R.VIT_TINY, R.DEC_TINY, 31, 4, 20
    enc, est = build_ref_vit(vcfg, seed)
    dec, dst = build_ref_decoder(dcfg, seed + 1)
    images = R.make_images(B, vcfg["image_size"], seed=seed)
    ids, am = R.make_reports(B, L, dcfg["vocab_size"], seed=seed)
    # RRG.forward == enc.encode -> dec(...)  (ref: vilmedic/models/rrg/RRG.py:25-41)
    params = list(enc.parameters()) + list(dec.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    enc.train(), dec.train()
    losses, logits0 = [], None
    for step in range(3):
        feats, fmask = enc.encode(images)
        out = dec(input_ids=ids, attention_mask=am, encoder_outputs=feats, encoder_attention_mask=fmask)
        if step == 0:
            logits0 = out["logits"].detach().clone()
        losses.append(out["loss"].detach().clone())
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
    save("g5_rrg_tiny", dict(vit_cfg=vcfg, dec_cfg=dcfg, seed=seed, B=B, L=L,
                             checksum=R.state_checksum(est) + R.state_checksum(dst),
                             losses=torch.stack(losses), logits0=logits0, lr=1e-3))


# ------------------------------------------------------------------ G6: losses
def gen_losses():
    out = {}
    for B in (8, 64):
        g = torch.Generator().manual_seed(1234 + B)
        l = torch.randn(B, 96, generator=g).requires_grad_(True)
        v = torch.randn(B, 96, generator=g).requires_grad_(True)
        loss, ll_, lv = lc.ConVIRTLoss(tau=0.1, lambda_=0.75)(l, v)
        loss.backward()
        out[f"convirt_{B}"] = dict(loss=loss.detach(), loss_l=ll_.detach(), loss_v=lv.detach(), gl=l.grad.clone(), gv=v.grad.clone())
        l2 = (0.2 * l.detach()).requires_grad_(True)
        v2 = (0.2 * v.detach()).requires_grad_(True)
        loss, lt, li_ = li.InfoNCELoss(tau=0.1)(l2, v2)
        loss.backward()
        out[f"infonce_{B}"] = dict(loss=loss.detach(), loss_t=lt.detach(), loss_i=li_.detach(), gl=l2.grad.clone(), gv=v2.grad.clone())
        logits = torch.randn(B, 33, generator=g).requires_grad_(True)
        tgt = torch.randint(0, 33, (B,), generator=g)
        loss = ll.LabelSmoothingCrossEntropy(smoothing=0.1)(logits, tgt)
        loss.backward()
        out[f"lsce_{B}"] = dict(loss=loss.detach(), target=tgt, g=logits.grad.clone())
    # known answer quoted in SURVEY §8(a) a13
    torch.manual_seed(1234)
    a, b = torch.randn(8, 768), torch.randn(8, 768)
    out["convirt_known_answer"] = lc.ConVIRTLoss(tau=0.1, lambda_=0.75)(a, b)[0]
    # GLoRIA
    B, D, T, hw = 6, 32, 9, 5
    g = torch.Generator().manual_seed(99)
    glob = torch.randn(B, D, generator=g).requires_grad_(True)
    loc = torch.randn(B, D, hw, hw, generator=g).requires_grad_(True)
    words = torch.randn(B, D, T, generator=g).requires_grad_(True)
    sent = torch.randn(B, D, generator=g).requires_grad_(True)
    cap_lens = [4, 7, 3, 6, 5, 7]
    sents = [["[CLS]"] + ["w"] * (n - 1) + ["[SEP]"] + ["[PAD]"] * (T - n - 1) for n in cap_lens]
    gl = lg.GLoRIALoss(local_loss_weight=1.0, global_loss_weight=1.0, temp1=4.0, temp2=5.0, temp3=10.0)
    loss, attn_maps = gl(glob, loc, words, sent, sents)
    loss.backward()
    out["gloria"] = dict(B=B, D=D, T=T, hw=hw, cap_lens=cap_lens, loss=loss.detach(),
                         g_glob=glob.grad.clone(), g_loc=loc.grad.clone(), g_words=words.grad.clone(), g_sent=sent.grad.clone(),
                         attn0=attn_maps[0].detach())
    save("g6_losses", out)


# ------------------------------------------------------------------ G7: greedy + beam decode
def gen_decode():
    from transformers import GenerationConfig
    from transformers.cache_utils import DynamicCache, EncoderDecoderCache
    cfg, seed, B, S, max_len = R.DEC_TINY, 41, 5, 10, 24
    # random tied-embedding decoders repeat their last token under argmax (SURVEY §7 hard parts):
    # large layer weights + small embeddings + an eos bias make greedy/beam paths non-degenerate; small query/key
    # projections keep the attention softmax smooth (large ones make it arg-max-like: a chaotic network in which one
    # bf16 rounding flips a whole context vector -- useless as a parity fixture).
    recipe = dict(std=float(os.environ.get("G7_STD", 0.6)), emb_std=0.2, eos_bias=float(os.environ.get("G7_EOS", 6.0)), qk_std=float(os.environ.get("G7_QK", 0.15)),
                  pos_std=float(os.environ.get("G7_POS", 0.6)))
    dec, st = build_ref_decoder(cfg, seed, **recipe)
    hf = dec.decoder
    g = torch.Generator().manual_seed(seed + 1)
    enc = torch.randn(B, S, cfg["hidden_size"], generator=g)
    enc_mask = torch.ones(B, S, dtype=torch.bool)
    enc_mask[2, 6:] = False
    enc[~enc_mask] = 0.0
    res = dict(cfg=cfg, seed=seed, B=B, S=S, max_len=max_len, recipe=recipe, enc_mask=enc_mask, checksum=R.state_checksum(st))
    for nb in (1, 4):
        for lp in ((1.0,) if nb == 1 else (1.0, 2.0)):
            args = dict(bos_token_id=0, eos_token_id=2, pad_token_id=1, num_return_sequences=1, max_length=max_len,
                        use_cache=True, num_beams=nb, length_penalty=lp, return_dict_in_generate=True, output_scores=True)
            # evaluation.py:73-78 call; shim 3 of SURVEY §8c: give HF 5.x the cache type cross-attention expects
            with torch.no_grad():
                o = hf.generate(input_ids=torch.ones((B, 1), dtype=torch.long) * 0,
                                generation_config=GenerationConfig(**args),
                                encoder_hidden_states=enc, encoder_attention_mask=enc_mask,
                                past_key_values=EncoderDecoderCache(DynamicCache(config=hf.config), DynamicCache(config=hf.config)))
            key = f"beams{nb}_lp{lp}"
            res[key] = dict(sequences=o.sequences, scores=getattr(o, "sequences_scores", None))
            print(key, o.sequences.tolist())
    save("g7_decode", res)


# ------------------------------------------------------------------ G9: MVQA core + text encoder
def gen_mvqa_text():
    from transformers.models.bert.modeling_bert import BertEncoder, BertPooler
    from transformers.models.bert_generation import BertGenerationConfig
    cfg, seed, B, S, C = R.MVQA_TINY, 51, 4, 9, 13
    hfcfg = BertGenerationConfig(hidden_act="gelu", attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0, **cfg)
    hfcfg._attn_implementation = "eager"
    enc = BertEncoder(hfcfg).eval()
    pool = BertPooler(hfcfg).eval()
    clf = cl.Classifier(input_size=cfg["hidden_size"], num_classes=C, dropout=0.0).eval()
    st = R.rand_state(R.bert_stack_shapes(cfg), seed)
    load_into(enc, st)
    g = torch.Generator().manual_seed(seed + 1)
    H = cfg["hidden_size"]
    pw, pb = 0.1 * torch.randn(H, H, generator=g), 0.02 * torch.randn(H, generator=g)
    cw, cb = 0.1 * torch.randn(C, H, generator=g), 0.02 * torch.randn(C, generator=g)
    pool.dense.weight.data.copy_(pw), pool.dense.bias.data.copy_(pb)
    lin = [m for m in clf.modules() if isinstance(m, torch.nn.Linear)][0]
    lin.weight.data.copy_(cw), lin.bias.data.copy_(cb)
    x = torch.randn(B, S, H, generator=g)
    with torch.no_grad():
        h = enc(x).last_hidden_state          # MVQA.py:43
        pooled = pool(h)                      # MVQA.py:47
        logits = clf(pooled)                  # MVQA.py:49
    out = dict(mvqa=dict(cfg=cfg, seed=seed, B=B, S=S, C=C, checksum=R.state_checksum(st), x=x, pw=pw, pb=pb, cw=cw, cb=cb,
                         hidden=h, pooled=pooled, logits=logits))
    # EncoderModel(proto=None): random BertGenerationEncoder + BertPooler (encoder_model.py:18-29,44-62)
    tcfg, seed = R.TXT_TINY, 61
    e = em.EncoderModel(AttrDict(proto=None, add_pooling_layer=True, hidden_act="gelu", attention_probs_dropout_prob=0.0,
                                 hidden_dropout_prob=0.0, **tcfg))
    e.encoder.config._attn_implementation = "eager"
    st = R.rand_state(R.text_encoder_shapes(tcfg), seed)
    load_into(e.encoder, st)
    e.pooler.dense.weight.data.copy_(pw), e.pooler.dense.bias.data.copy_(pb)
    ids, am = R.make_reports(4, 16, tcfg["vocab_size"], seed=seed)
    with torch.no_grad():
        o = e.eval()(input_ids=ids, attention_mask=am)
    out["text"] = dict(cfg=tcfg, seed=seed, B=4, L=16, checksum=R.state_checksum(st), pw=pw, pb=pb,
                       last_hidden_state=o.last_hidden_state, pooler_output=o.pooler_output)
    save("g9_mvqa_text", out)


# ------------------------------------------------------------------ G8: scst_loss (pure function, load source by exec of the def only)
def gen_scst():
    import ast
    src = open(REF + "blocks/rl/SCST.py").read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "scst_loss"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "scst_loss", "exec"), ns)
    g = torch.Generator().manual_seed(71)
    B, T = 5, 11
    logp = -torch.rand(B, T, 1, generator=g) * 5
    logp[1, 4, 0] = -float("inf")
    logp[3, 9, 0] = -float("inf")
    seq = torch.randint(3, 50, (B, T), generator=g)
    seq[0, 7:] = 1
    seq[2, 3:] = 1
    seq[4, 10:] = 0
    rs = [torch.rand(B, generator=g).tolist(), torch.rand(B, generator=g).tolist()]
    rg = [torch.rand(B, generator=g).tolist(), torch.rand(B, generator=g).tolist()]
    w = [0.7, 0.3]
    inp = logp.clone().requires_grad_(True)
    loss, dr, drm = ns["scst_loss"](inp * 1.0, seq, rs, rg, w, 1)
    loss.backward()
    save("g8_scst", dict(logp=logp, seq=seq, rs=rs, rg=rg, w=w, pad=1, loss=loss.detach(), grad=inp.grad.clone(),
                         delta_reward=dr, delta_reward_per_metric=drm))


def gen_gloria_aggregate():
    """G11: GLoRIA.aggregate_tokens (models/selfsup/GLoRIA.py:123-177) -- the method is lifted out of the class by AST
    (the class itself cannot be constructed here: pretrained tokenizer / encoder downloads) and run with a stand-in
    ``self`` that only carries ``idxtoword``."""
    import ast
    src = open(REF + "models/selfsup/GLoRIA.py").read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "GLoRIA"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "aggregate_tokens"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "aggregate_tokens", "exec"), ns)
    vocab = ["[PAD]", "[CLS]", "[SEP]", "the", "heart", "##s", "is", "en", "##larg", "##ed", "no", "pleural", "eff", "##usion", ".", "lung"]
    idxtoword = dict(enumerate(vocab))
    W = {w: i for i, w in idxtoword.items()}
    caps = [["[CLS]", "the", "heart", "##s", "is", "en", "##larg", "##ed", ".", "[SEP]", "[PAD]", "[PAD]"],
            ["[CLS]", "no", "pleural", "eff", "##usion", "[SEP]", "[PAD]", "[PAD]", "[PAD]", "[PAD]", "[PAD]", "[PAD]"],
            ["[CLS]", "lung", "##s", "##s", "is", "the", "lung", ".", "no", "eff", "##usion", "[SEP]"],
            ["[CLS]", "##s", "the", "[SEP]", "heart", "[SEP]", "[PAD]", "[PAD]", "[PAD]", "[PAD]", "[PAD]", "[PAD]"]]
    input_ids = torch.tensor([[W[w] for w in c] for c in caps])
    g = torch.Generator().manual_seed(111)
    emb = torch.randn(3, len(caps), input_ids.shape[1], 8, generator=g)
    out, sents = ns["aggregate_tokens"](types.SimpleNamespace(idxtoword=idxtoword), emb, input_ids)
    save("g11_gloria_aggregate", dict(vocab=vocab, input_ids=input_ids, embeddings=emb, out=out.clone(), sentences=sents))


def gen_report_cleaning():
    """G12: the report normalisations the RRG / RRS configs name (``processing: r2gen_clean_report`` / ``rouge``,
    datasets/base/papers/report_preprocessing.py:8-23,69-108), lifted out by AST and run on synthetic report strings.
    (``ifcc_clean_report`` / ``gloria_clean_report_chexpert`` need nltk, which this image lacks: not pinned.)"""
    import ast
    import re
    src = open(REF + "datasets/base/papers/report_preprocessing.py").read()
    import six
    fns = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in ("r2gen_clean_report", "rouge")]
    ns = {"re": re, "six": six}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "report_preprocessing", "exec"), ns)
    reports = ["1. No acute cardiopulmonary process. 2. Stable cardiomegaly.  3. Small left pleural effusion...",
               "FINDINGS:  The heart is ENLARGED (moderate); lungs are clear!\nNo pneumothorax / effusion.",
               "____ was removed. __ ___ . There is a 5 mm nodule, unchanged.. ..",
               "", ".", "  Multiple    spaces   and 'quotes' \"double\" back\\slash [brackets] {braces} 100% a+b a_b",
               "Compared to prior: 1. improved aeration 4. new line placement. 5. ET tube 3 cm above carina",
               "x" + "_" * 300 + "y" + " " * 70 + "z" + "." * 300 + " end"]
    reports_rouge = reports[:7] + ["Ünïcode résumé, CT-scan #2: 3.5cm\tmass;\nT1/T2 weighted", "already clean tokens 123"]
    save("g12_report_cleaning", dict(reports=reports, cleaned=[ns["r2gen_clean_report"](r) for r in reports],
                                     reports_rouge=reports_rouge, rouge=[ns["rouge"](r) for r in reports_rouge]))


# ------------------------------------------------------------------ G17: the reference's own LR schedulers
def gen_schedulers():
    lw = load_ref("ref_lwca", "blocks/schedulers/LinearWarmupCosineAnnealingLR.py")
    out = {}

    def run(make, steps):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=0.02)
        sch = make(opt)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(steps):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        return lrs
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["lwca_10_40"] = run(lambda o: lw.LinearWarmupCosineAnnealingLR(o, warmup_epochs=10, max_epochs=40), 55)
        out["lwca_5_20_start_eta"] = run(lambda o: lw.LinearWarmupCosineAnnealingLR(o, warmup_epochs=5, max_epochs=20, warmup_start_lr=0.001, eta_min=0.002), 24)
    fn = lw.linear_warmup_decay(3, 10, cosine=True)
    out["lambda_cosine"] = [fn(i) for i in range(12)]
    save("g17_schedulers", out)


# ------------------------------------------------------------------ G15: BLEU (the vendored COCO-caption scorer)
def gen_bleu():
    """the reference's Bleu wrapper (blocks/scorers/NLG/bleu/bleu.py:24-46) adds (hypothesis, [reference]) pairs to its vendored
    BleuScorer and asks for option='closest'; the scorer module is loaded by path (pure Python, needs ``six``)."""
    bs = load_ref("ref_bleu_scorer", "blocks/scorers/NLG/bleu/bleu_scorer.py")
    refs = ["the heart is normal in size . no pleural effusion .", "no acute cardiopulmonary process .", "lungs are clear",
            "there is a small left pleural effusion with adjacent atelectasis .", "a", "stable cardiomegaly . no edema .",
            "no focal consolidation pleural effusion or pneumothorax"]
    hyps = ["the heart is normal . no effusion .", "no acute cardiopulmonary process .", "the lungs are clear bilaterally without focal consolidation",
            "small left effusion .", "", "cardiomegaly is stable . no pulmonary edema . no effusion .", "no focal consolidation pleural effusion or pneumothorax is seen"]
    out = {}
    for n in (4, 2):
        scorer = bs.BleuScorer(n=n)
        for r, h in zip(refs, hyps):
            scorer += (h, [r])
        score, scores = scorer.compute_score(option="closest", verbose=0)
        out[f"n{n}"] = dict(corpus=score[n - 1], per_sentence=list(scores[n - 1]))
    save("g15_bleu", dict(refs=refs, hyps=hyps, **out))


# ------------------------------------------------------------------ G14: VICReg loss
def gen_vicreg():
    lv = load_ref("ref_vicreg", "blocks/losses/selfsup/VICREGLoss.py")
    out = {}
    for N, D in ((16, 64), (50, 128)):
        g = torch.Generator().manual_seed(4321 + N)
        z1 = (0.7 * torch.randn(N, D, generator=g) + 0.1).requires_grad_(True)      # std < 1: the variance hinge is active
        z2 = (z1.detach() + 0.3 * torch.randn(N, D, generator=g)).requires_grad_(True)
        crit = lv.VICREGLoss(sim_loss_weight=25.0, var_loss_weight=25.0, cov_loss_weight=1.0)
        loss = crit(z1, z2)
        loss.backward()
        out[f"n{N}_d{D}"] = dict(N=N, D=D, loss=loss.detach(), sim=lv.VICREGLoss.invariance_loss(z1, z2).detach(),
                                 var=lv.VICREGLoss.variance_loss(z1, z2).detach(), cov=lv.VICREGLoss.covariance_loss(z1, z2).detach(),
                                 g1=z1.grad.clone(), g2=z2.grad.clone())
    save("g14_vicreg", out)


# ------------------------------------------------------------------ G13: RRS (text encoder -> cross-attending decoder)
def gen_rrs():
    """RRS.forward == enc(input_ids, attention_mask).last_hidden_state -> dec(decoder ids, encoder mask = source attention mask)
    (ref: vilmedic/models/rrs/RRS.py:30-52), with the reference's EncoderModel and DecoderModel blocks."""
    tcfg, dcfg, seed, B, Ls, Lt = R.TXT_TINY, R.DEC_TINY, 71, 4, 18, 12
    assert tcfg["hidden_size"] == dcfg["hidden_size"]
    e = em.EncoderModel(AttrDict(proto=None, add_pooling_layer=False, hidden_act="gelu", attention_probs_dropout_prob=0.0,
                                 hidden_dropout_prob=0.0, **tcfg))
    e.encoder.config._attn_implementation = "eager"
    est = R.rand_state(R.text_encoder_shapes(tcfg), seed)
    load_into(e.encoder, est)
    dec, dst = build_ref_decoder(dcfg, seed + 1)
    src_ids, src_am = R.make_reports(B, Ls, tcfg["vocab_size"], seed=seed)
    tgt_ids, tgt_am = R.make_reports(B, Lt, dcfg["vocab_size"], seed=seed + 1)
    e.train(), dec.train()
    hidden = e(src_ids, src_am, return_dict=True).last_hidden_state
    out = dec(input_ids=tgt_ids, attention_mask=tgt_am, encoder_outputs=hidden, encoder_attention_mask=src_am)
    out["loss"].backward()
    en, dn = dict(e.encoder.named_parameters()), dict(dec.decoder.named_parameters())
    enc_grads = ["embeddings.word_embeddings.weight", "encoder.layer.0.attention.self.query.weight", "encoder.layer.1.output.dense.weight",
                 "encoder.layer.1.output.LayerNorm.bias"]
    dec_grads = ["bert.encoder.layer.0.crossattention.self.key.weight", "bert.encoder.layer.1.intermediate.dense.weight", "lm_head.bias"]
    save("g13_rrs_tiny", dict(enc_cfg=tcfg, dec_cfg=dcfg, seed=seed, B=B, Ls=Ls, Lt=Lt,
                              checksum=R.state_checksum(est) + R.state_checksum(dst),
                              encoder_hidden=hidden.detach(), loss=out["loss"].detach(), logits=out["logits"].detach(),
                              enc_grads={n: en[n].grad.clone() for n in enc_grads},
                              dec_grads={n: dn[n].grad.clone() for n in dec_grads}))


# ------------------------------------------------------------------ G18: the reference's MVQA and ConVIRT classes themselves
def _lift(rel, names, ns):
    """compile the named top-level classes / functions of a reference file into ``ns`` (the file itself star-imports the
    ``vilmedic`` package, which cannot be imported here: omegaconf / torchvision are absent)"""
    import ast
    tree = ast.parse(open(REF + rel).read())
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert len(body) == len(names), [n.name for n in body]
    exec(compile(ast.Module(body=body, type_ignores=[]), rel, "exec"), ns)
    return ns


class _Identity(torch.nn.Module):
    """stand-in for the CNN of MVQA: the fixture feeds the CNN OUTPUT [B, S, C] as ``images``"""

    def forward(self, x):
        return x


class _StubVisual(torch.nn.Module):
    """stand-in for ConVIRT's VisualEncoder: flatten -> Linear -> BatchNorm1d, so the forward_batch_size micro-batching is visible in
    the result (training-mode batch statistics per micro-batch)"""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.fc = torch.nn.Linear(in_dim, out_dim)
        self.bn = torch.nn.BatchNorm1d(out_dim)

    def forward(self, x):
        return self.bn(self.fc(x.flatten(1)))


def gen_model_compositions():
    """G18: ``MVQA.forward`` (models/mvqa/MVQA.py:13-54) and ``ConVIRT.forward`` (models/selfsup/conVIRT.py:47-102) run from the
    reference's own class bodies (lifted by AST) on the reference's own blocks (EncoderModel, ConVIRTLoss, Classifier,
    LabelSmoothingCrossEntropy) with stand-in CNNs -- pins oracle.mvqa_forward / oracle.convirt_forward, i.e. the model-level
    composition (adapter + LayerNorm, micro-batched towers, projection heads, loss wiring), not only the pieces."""
    from transformers.models.bert.modeling_bert import BertEncoder, BertPooler
    from transformers.models.bert_generation import BertGenerationConfig
    out = {}
    # ---- MVQA
    ns = _lift("models/mvqa/MVQA.py", ["MVQA"], dict(torch=torch, nn=torch.nn, BertEncoder=BertEncoder, BertPooler=BertPooler,
                                                      BertGenerationConfig=BertGenerationConfig, Classifier=cl.Classifier,
                                                      LabelSmoothingCrossEntropy=ll.LabelSmoothingCrossEntropy, evaluation=None,
                                                      get_n_params=lambda m: 0, _Identity=_Identity))
    cfg, B, S, Cin, NC = R.MVQA_TINY, 5, 7, 24, 11
    torch.manual_seed(181)
    tcfg = AttrDict(hidden_act="gelu", attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0, **cfg)
    m = ns["MVQA"](cnn=dict(proto="_Identity"), classifier=dict(proto="Classifier", input_size=cfg["hidden_size"], num_classes=NC, dropout=0.0),
                   adapter=dict(input_size=Cin, output_size=cfg["hidden_size"]), transformer=tcfg,
                   loss=dict(proto="LabelSmoothingCrossEntropy")).eval()
    m.transformer.config._attn_implementation = "eager"
    g = torch.Generator().manual_seed(182)
    feats, labels = torch.randn(B, S, Cin, generator=g), torch.randint(0, NC, (B,), generator=g)
    with torch.no_grad():
        o = m(images=feats, labels=labels, from_training=True)
    out["mvqa"] = dict(cfg=dict(cfg), features=feats, labels=labels, state={k: v.clone() for k, v in m.state_dict().items()},
                       loss=o["loss"].clone(), output=o["output"].clone(), answer=o["answer"].clone())
    # ---- ConVIRT
    class _EncoderModelItems(em.EncoderModel):
        """the reference reads ``linguistic['pooler_output']`` (conVIRT.py:91), which only exists as an ITEM for hub encoders that return
        it natively; its own EncoderModel(proto=None) attaches the pooled output as an attribute (encoder_model.py:58-60) -> expose both"""

        def forward(self, *a, **k):
            o = super().forward(*a, **k)
            return {"last_hidden_state": o.last_hidden_state, "pooler_output": o.pooler_output}

    ns = _lift("models/selfsup/conVIRT.py", ["chunks", "ConVIRT"], dict(torch=torch, nn=torch.nn, EncoderModel=_EncoderModelItems,
                                                                        ConVIRTLoss=lc.ConVIRTLoss, InfoNCELoss=li.InfoNCELoss,
                                                                        evaluation=None, get_n_params=lambda m: 0, _StubVisual=_StubVisual))
    tcfg, B, L, C, Pd, fbs = R.TXT_TINY, 6, 12, 20, 16, 4
    torch.manual_seed(183)
    m = ns["ConVIRT"](encoder=AttrDict(proto=None, add_pooling_layer=True, hidden_act="gelu", attention_probs_dropout_prob=0.0,
                                       hidden_dropout_prob=0.0, **tcfg),
                      cnn=dict(proto="_StubVisual", in_dim=3 * 4 * 4, out_dim=C),
                      projection=AttrDict(visual_embedding_dim=C, textual_embedding_dim=tcfg["hidden_size"], projection_dim=Pd),
                      loss=dict(proto="ConVIRTLoss", tau=0.1, lambda_=0.75), forward_batch_size=fbs)
    m.linguistic.encoder.config._attn_implementation = "eager"
    st = R.rand_state(R.text_encoder_shapes(tcfg), 184)
    load_into(m.linguistic.encoder, st)
    m.train()                                              # BatchNorm batch statistics per micro-batch of ``fbs`` (6 = 4 + 2 rows)
    ids, am = R.make_reports(B, L, tcfg["vocab_size"], seed=185)
    images = torch.randn(B, 3, 4, 4, generator=torch.Generator().manual_seed(186))
    o = m(input_ids=ids, attention_mask=am, images=images)
    # the text encoder's weights are a recipe (R.rand_state(R.text_encoder_shapes(cfg), encoder_seed)); only the small heads are stored
    state = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith(("linguistic.encoder.", "visual."))}
    out["convirt"] = dict(cfg=dict(tcfg), B=B, L=L, fbs=fbs, tau=0.1, lambda_=0.75, images=images, state=state, encoder_seed=184,
                          encoder_checksum=R.state_checksum(st),
                          visual_fc_w=m.visual.fc.weight.detach().clone(), visual_fc_b=m.visual.fc.bias.detach().clone(),
                          loss=o["loss"].detach().clone(), loss_l=o["loss_l"].detach().clone(), loss_v=o["loss_v"].detach().clone(),
                          linguistic=o["linguistic"].detach().clone(), visual=o["visual"].detach().clone())
    save("g18_model_compositions", out)


# ------------------------------------------------------------------ G19: SCST.forward_sampling itself
def gen_scst_sampling():
    """G19: ``SCST.forward_sampling`` (blocks/rl/SCST.py:142-185) lifted out of its class and run on the reference's DecoderModel with
    HF ``generate`` (do_sample, top_k, bad_words_ids, output_scores -- the un-wrapped, differentiable call): the sampled sequence, the
    gathered log-probabilities of the processed scores, the loss and a decoder gradient.  Pins oracle.scst_forward, which takes that
    sampled sequence as input.  Stand-ins: fixed per-sample rewards instead of text scorers (``get_reward``), and the HF 5.x cache
    object the cross-attention expects (as in G7)."""
    import ast
    import inspect
    from transformers.cache_utils import DynamicCache, EncoderDecoderCache
    tree = ast.parse(open(REF + "blocks/rl/SCST.py").read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "scst_loss"]
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SCST"][0]
    fns += [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward_sampling"]
    ns = {"torch": torch, "F": torch.nn.functional, "inspect": inspect}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "SCST.py", "exec"), ns)
    cfg, B, S, T, top_k = R.DEC_TINY, 4, 6, 24, 6
    recipe = dict(std=0.3, emb_std=0.2, eos_bias=7.0, qk_std=0.15, pos_std=0.3)
    # the reference passes forced_eos_token_id=True, which HF turns into a BOOLEAN index tensor that fails the moment a row reaches
    # max_length - 1 (SCST.py:163; transformers ForcedEOSTokenLogitsProcessor): the call only works while every sampled row ends early.
    # A random decoder often falls into a repeating token, so the first seed whose sampled rows all end before max_length is taken.
    for seed in range(91, 131):
        dec, st = build_ref_decoder(cfg, seed, **recipe)
        hf = dec.decoder
        hf.train()                                          # dropout probabilities are 0 in the recipe: train() only matters for autograd
        g = torch.Generator().manual_seed(seed + 1)
        enc = torch.randn(B, S, cfg["hidden_size"], generator=g).requires_grad_(True)
        enc_mask = torch.ones(B, S, dtype=torch.bool)
        enc_mask[1, 4:] = False
        rs, rg = [torch.rand(B, generator=g).tolist()], [torch.rand(B, generator=g).tolist()]
        raw = inspect.unwrap(hf.generate)

        def _generate(self, hf=hf, raw=raw, **kw):
            return raw(hf, past_key_values=EncoderDecoderCache(DynamicCache(config=hf.config), DynamicCache(config=hf.config)), **kw)
        torch.manual_seed(seed + 2)
        with torch.no_grad():
            probe = _generate(None, input_ids=torch.zeros(B, 1, dtype=torch.long), max_length=T, num_beams=1, encoder_hidden_states=enc.detach(),
                              encoder_attention_mask=enc_mask, bad_words_ids=[[1], [0]], top_k=top_k, do_sample=True, use_cache=True)
        lens = (probe != 1).sum(1)
        if probe.shape[1] < T - 1 and int(lens.min()) >= 3 and len(set(lens.tolist())) > 1:
            break
    else:
        raise RuntimeError("no seed gives sampled rows that all end before max_length")
    print("G19 seed", seed, "sampled lengths", lens.tolist())
    standin = types.SimpleNamespace(
        decoder=types.SimpleNamespace(generate=torch.no_grad()(_generate)), use_nll=False, max_length=T, bos_token_id=0, pad_token_id=1,
        top_k=top_k, scores=["toy"], scores_weights=[1.0], get_reward=lambda ids, ref: (rs, None, None))
    torch.manual_seed(seed + 2)                             # the sampling draws
    ids = torch.zeros(B, T, dtype=torch.long)
    loss, dr, drm, reward_sampling, _ = ns["forward_sampling"](standin, ids, None, enc, enc_mask, rg)
    loss.backward()
    # recover what the method computed internally: re-run generate with the same draws for the sequence and the processed scores
    torch.manual_seed(seed + 2)
    with torch.no_grad():
        o = standin.decoder.generate(self=None, input_ids=torch.zeros(B, 1, dtype=torch.long), max_length=T, num_beams=1, num_return_sequences=1,
                                     encoder_hidden_states=enc.detach(), encoder_attention_mask=enc_mask, bad_words_ids=[[1], [0]],
                                     top_k=top_k, forced_eos_token_id=True, output_scores=True, do_sample=True, use_cache=True,
                                     return_dict_in_generate=True)
    logp = torch.log_softmax(torch.stack(o.scores, dim=1), -1).gather(2, o.sequences[:, 1:].unsqueeze(-1)).squeeze(-1)
    print("sampled", o.sequences.tolist(), "loss", loss.item())
    save("g19_scst_sampling", dict(cfg=cfg, seed=seed, B=B, S=S, T=T, top_k=top_k, recipe=recipe,
                                   checksum=R.state_checksum(st), enc=enc.detach().clone(), enc_mask=enc_mask, rs=rs, rg=rg,
                                   sequences=o.sequences.clone(), logp=logp.clone(), loss=loss.detach().clone(), delta_reward=dr.detach().clone(),
                                   g_enc=enc.grad.clone()))


# ------------------------------------------------------------------ G20: RRG_HF.forward itself (multi-image + enc_to_dec_proj)
def gen_rrg_hf():
    """G20: ``RRG_HF.forward`` (models/rrg/RRG_HF.py:105-177), lifted out of its class and run on a ``VisionEncoderDecoderModel`` built
    from a ViT whose width differs from the decoder's (so ``enc_to_dec_proj`` is in the path) -- 5-D images with an ``images_mask``
    (crops encoded flat, concatenated along the sequence, masked per crop in the cross-attention) and 4-D images.  Pins
    oracle.rrg_hf_forward."""
    import ast
    from transformers import ViTConfig, ViTModel, VisionEncoderDecoderModel
    tree = ast.parse(open(REF + "models/rrg/RRG_HF.py").read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "RRG_HF"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "RRG_HF.py", "exec"), ns)
    vcfg = dict(R.VIT_TINY, hidden_size=64, intermediate_size=128)
    dcfg, seed, B, N, L = R.DEC_TINY, 201, 3, 2, 14
    vit = ViTModel(ViTConfig(**vcfg, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager"), add_pooling_layer=False)
    vst = R.rand_state(R.vit_shapes(vcfg), seed)
    load_into(vit, vst, vit_to_hf5)
    dec, dst = build_ref_decoder(dcfg, seed + 1)
    model = VisionEncoderDecoderModel(encoder=vit, decoder=dec.decoder).eval()
    g = torch.Generator().manual_seed(seed + 2)
    pw = 0.1 * torch.randn(dcfg["hidden_size"], vcfg["hidden_size"], generator=g)
    pb = 0.02 * torch.randn(dcfg["hidden_size"], generator=g)
    model.enc_to_dec_proj.weight.data.copy_(pw), model.enc_to_dec_proj.bias.data.copy_(pb)
    if not hasattr(model.decoder.config, "cross_attention_hidden_size"):
        model.decoder.config.cross_attention_hidden_size = None      # PretrainedConfig default in the pinned 4.55.3; gone in 5.x
    self_ = types.SimpleNamespace(model=model)
    size = vcfg["image_size"]
    images = R.make_images(B * N, size, seed=seed).view(B, N, 3, size, size)
    images_mask = torch.tensor([[1, 1], [1, 0], [1, 1]], dtype=torch.bool)
    ids, am = R.make_reports(B, L, dcfg["vocab_size"], seed=seed)
    with torch.no_grad():
        o5 = ns["forward"](self_, ids, am, images, images_mask=images_mask)
        o4 = ns["forward"](self_, ids, am, images[:, 0])
    save("g20_rrg_hf", dict(vit_cfg=vcfg, dec_cfg=dcfg, seed=seed, B=B, N=N, L=L, images_mask=images_mask, proj_w=pw, proj_b=pb,
                            vit_checksum=R.state_checksum(vst), dec_checksum=R.state_checksum(dst),
                            loss5=o5["loss"].clone(), logits5=o5["logits"].clone(), loss4=o4["loss"].clone(), logits4=o4["logits"].clone()))


# ------------------------------------------------------------------ G21: the GLoRIA class itself
class _StubCnnEncoder(torch.nn.Module):
    """stand-in for GLoRIA's VisualEncoder(resnet50, avgpool, batch_first): ``cnn`` is an nn.Sequential whose [6] yields the local feature
    map (the hook point of GLoRIA.py:75) and whose end is a global average pool; forward applies visual_encoder.py's batch_first permute"""

    def __init__(self, interm, feat):
        super().__init__()
        nn_ = torch.nn
        self.cnn = nn_.Sequential(nn_.Conv2d(3, 6, 7, stride=8, padding=3), nn_.ReLU(), nn_.Identity(), nn_.Identity(), nn_.Identity(),
                                  nn_.Conv2d(6, interm, 3, stride=4, padding=1), nn_.BatchNorm2d(interm), nn_.ReLU(),
                                  nn_.Conv2d(interm, feat, 1), nn_.AdaptiveAvgPool2d(1))

    def forward(self, x):
        out = self.cnn(x)
        out = out.view(*out.size()[:2], -1).permute(0, 2, 1)
        return out.squeeze(1) if out.shape[1] == 1 else out


def gen_gloria_model():
    """G21: the reference's own ``GLoRIA`` class (models/selfsup/GLoRIA.py:46-121 + aggregate_tokens), lifted by AST, on its own
    EncoderModel (proto None) and GLoRIALoss with a stand-in CNN and a stand-in tokenizer vocabulary: both towers in forward_batch_size
    chunks (training-mode BatchNorm per chunk), the layer-3 hook, up-sampling to 299 x 299, hidden-state stacking, word-piece
    aggregation, sentence / word embeddings and the loss.  Pins oracle.gloria_forward."""
    vocab = ["[PAD]", "[CLS]", "[SEP]", "the", "heart", "##s", "is", "en", "##larg", "##ed", "no", "pleural", "eff", "##usion", ".", "lung",
             "clear", "small", "##er", "normal"]
    W = {w: i for i, w in enumerate(vocab)}
    caps = [["[CLS]", "the", "heart", "##s", "is", "en", "##larg", "##ed", ".", "[SEP]", "[PAD]", "[PAD]"],
            ["[CLS]", "no", "pleural", "eff", "##usion", "[SEP]", "[PAD]", "[PAD]", "[PAD]", "[PAD]", "[PAD]", "[PAD]"],
            ["[CLS]", "lung", "##s", "is", "clear", ".", "no", "eff", "##usion", "[SEP]", "[PAD]", "[PAD]"],
            ["[CLS]", "small", "##er", "heart", "[SEP]", "[PAD]", "[PAD]", "[PAD]", "[PAD]", "[PAD]", "[PAD]", "[PAD]"],
            ["[CLS]", "the", "lung", "is", "normal", "the", "heart", "is", "normal", ".", "[SEP]", "[PAD]"]]
    ids = torch.tensor([[W[w] for w in c] for c in caps])
    am = (ids != W["[PAD]"]).long()
    tcfg = dict(R.TXT_TINY, vocab_size=len(vocab), pad_token_id=0)
    interm, feat, B, fbs, last_n = 10, 14, len(caps), 3, 2
    ns = _lift("models/selfsup/GLoRIA.py", ["chunks", "GLoRIA"], dict(torch=torch, nn=torch.nn, EncoderModel=em.EncoderModel,
                                                                      GLoRIALoss=lg.GLoRIALoss, evaluation=None, get_n_params=lambda m: 0,
                                                                      _StubCnnEncoder=_StubCnnEncoder))
    tok = types.SimpleNamespace(get_vocab=lambda: dict(W))
    torch.manual_seed(211)
    m = ns["GLoRIA"](encoder=AttrDict(proto=None, last_n_layers=last_n, hidden_act="gelu", attention_probs_dropout_prob=0.0,
                                      hidden_dropout_prob=0.0, **tcfg),
                     cnn=dict(proto="_StubCnnEncoder", interm=interm, feat=feat),
                     visual_embedder=AttrDict(feature_dim=feat, interm_feature_dim=interm),
                     loss=dict(local_loss_weight=1.0, global_loss_weight=1.0, temp1=4.0, temp2=5.0, temp3=10.0),
                     dl=types.SimpleNamespace(dataset=types.SimpleNamespace(tokenizer=tok)), forward_batch_size=fbs)
    m.linguistic.encoder.config._attn_implementation = "eager"
    st = R.rand_state(R.text_encoder_shapes(tcfg), 212)
    load_into(m.linguistic.encoder, st)
    m.train()                                              # BatchNorm batch statistics per chunk of ``fbs`` images (5 = 3 + 2)
    images = torch.randn(B, 3, 40, 40, generator=torch.Generator().manual_seed(213))
    o = m(input_ids=ids, attention_mask=am, images=images)
    state = {k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith(("global_embedder.", "local_embedder."))}
    save("g21_gloria_model", dict(vocab=vocab, cfg=tcfg, interm=interm, feat=feat, fbs=fbs, last_n_layers=last_n, input_ids=ids, attention_mask=am,
                                  images=images, state=state, encoder_seed=212, encoder_checksum=R.state_checksum(st),
                                  cnn_state={k: v.detach().clone() for k, v in m.visual.cnn.state_dict().items() if "running" not in k and "num_batches" not in k},
                                  loss=o["loss"].detach().clone(), global_features=o["global_features"].detach().clone(),
                                  local_features=o["local_features"].detach().clone(), word_embeddings=o["word_embeddings"].detach().clone(),
                                  sent_embeddings=o["sent_embeddings"].detach().clone()))


# ------------------------------------------------------------------ G22: ensemble decode = HF generate over SUMMED logits
def gen_ensemble_decode():
    """The reference's ensemble path (blocks/huggingface/decoder/beam_search.py:165-400) is a copy of the beam_search of the HF release it
    was written against with ONE arithmetic change, the marked lines 243-262: every model of the ensemble runs on the same
    ``input_ids`` (each with its own encoder states) and the next-token logits are SUMMED before the log-softmax.  That file
    imports ``transformers.generation_utils`` / ``generation_beam_search`` (removed long before the pinned 4.55.3), so it cannot be
    imported at this snapshot.  The fixture therefore runs the SAME change through the installed HF beam search: the reference's
    own DecoderModel twice (two seeds), the first one's ``forward`` wrapped to add the second one's logits, decoded by HF
    ``generate`` exactly as evaluation.py:73-78 calls it (no KV cache: the cache is not part of the arithmetic)."""
    from transformers import GenerationConfig
    cfg, seeds, B, S, max_len = R.DEC_TINY, (41, 43), 5, 10, 24
    recipe = dict(std=0.6, emb_std=0.2, eos_bias=3.0, qk_std=0.15, pos_std=0.6)      # the eos biases add up: 3 + 3 = G7's 6
    decs = [build_ref_decoder(cfg, sd, **recipe) for sd in seeds]
    hf1, hf2 = decs[0][0].decoder, decs[1][0].decoder
    encs, masks = [], []
    for i, sd in enumerate(seeds):
        g = torch.Generator().manual_seed(sd + 1)
        e = torch.randn(B, S - i, cfg["hidden_size"], generator=g)           # the models see different numbers of encoder positions
        m = torch.ones(B, S - i, dtype=torch.bool)
        m[2 + i, 6:] = False
        e[~m] = 0.0
        encs.append(e)
        masks.append(m)
    inner = hf1.forward

    def summed(input_ids=None, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None, enc2=None, mask2=None, **kw):
        o1 = inner(input_ids=input_ids, attention_mask=attention_mask, encoder_hidden_states=encoder_hidden_states,
                   encoder_attention_mask=encoder_attention_mask, use_cache=False, return_dict=True)
        o2 = hf2(input_ids=input_ids, attention_mask=attention_mask, encoder_hidden_states=enc2, encoder_attention_mask=mask2,
                 use_cache=False, return_dict=True)
        o1.logits = o1.logits + o2.logits                                       # beam_search.py:262
        return o1

    hf1.forward = summed
    res = dict(cfg=cfg, seeds=seeds, B=B, S=S, max_len=max_len, recipe=recipe, enc_masks=masks,
               checksums=[R.state_checksum(st) for _, st in decs])
    for nb in (1, 4):
        for lp in ((1.0,) if nb == 1 else (1.0, 2.0)):
            args = dict(bos_token_id=0, eos_token_id=2, pad_token_id=1, num_return_sequences=1, max_length=max_len,
                        use_cache=False, num_beams=nb, length_penalty=lp, return_dict_in_generate=True, output_scores=True)
            with torch.no_grad():
                o = hf1.generate(input_ids=torch.zeros((B, 1), dtype=torch.long), generation_config=GenerationConfig(**args),
                                 encoder_hidden_states=encs[0], encoder_attention_mask=masks[0], enc2=encs[1], mask2=masks[1])
            key = f"beams{nb}_lp{lp}"
            res[key] = dict(sequences=o.sequences, scores=getattr(o, "sequences_scores", None))
            print(key, o.sequences.tolist())
    # the single models decode differently from the ensemble (otherwise the fixture would not see the sum)
    hf1.forward = inner
    with torch.no_grad():
        alone = hf1.generate(input_ids=torch.zeros((B, 1), dtype=torch.long), encoder_hidden_states=encs[0], encoder_attention_mask=masks[0],
                             generation_config=GenerationConfig(bos_token_id=0, eos_token_id=2, pad_token_id=1, max_length=max_len, use_cache=False, num_beams=1))
    res["model0_alone_greedy"] = alone
    print("model 0 alone", alone.tolist())
    save("g22_ensemble_decode", res)

# ------------------------------------------------------------------ G23: pretrained `proto` towers (BERT / RoBERTa) from a local directory
def _specials(mt):
    return dict(cls=0, pad=1, sep=2) if mt == "roberta" else dict(cls=3, pad=0, sep=4)


def _thin(g):
    """weight-matrix gradients are stored every 4th row (fixture size); embedding tables and vectors whole"""
    return g[::4].clone() if (g.dim() == 2 and g.shape[0] >= 128) else g.clone()


def gen_proto_towers():
    """The reference's two `proto` paths on local checkpoint directories written from the recipe (R.write_proto_dir):
      EncoderModel(proto=dir)  -> AutoModel.from_pretrained (RobertaModel / BertModel, built-in pooler)     encoder_model.py:19-22
      DecoderModel(proto=dir)  -> AutoModelForCausalLM.from_pretrained(config: is_decoder, add_cross_attention)  decoder_model.py:17-21
    Pins oracle.text_embeddings (token-type row, RoBERTa position ids and their pad-row gradient rule), oracle.lm_logits' dense -> GELU ->
    LayerNorm transform and, through HF generate() on the loaded decoder (evaluation.py:73-78), the position numbering of decode."""
    import tempfile
    from transformers import GenerationConfig
    from transformers.cache_utils import DynamicCache, EncoderDecoderCache
    out = {}
    tmp = tempfile.mkdtemp()
    for mt, cfg, seed in (("roberta", R.ROBERTA_TINY, 231), ("bert", R.BERT_TINY, 241)):
        sp = _specials(mt)
        # ---- encoder
        st = R.rand_state(R.text_model_shapes(cfg), seed)
        d = R.write_proto_dir(os.path.join(tmp, mt + "_enc"), mt, cfg, st)
        e = em.EncoderModel(AttrDict(proto=d))
        e.encoder.config._attn_implementation = "eager"
        B, L = 4, 20
        ids, am = R.make_reports(B, L, cfg["vocab_size"], seed=seed, **sp)
        e.train()
        o = e(input_ids=ids, attention_mask=am, output_hidden_states=True)
        g = torch.Generator().manual_seed(seed + 5)
        wl, wp = torch.randn(o.last_hidden_state.shape, generator=g), torch.randn(o.pooler_output.shape, generator=g)
        ((o.last_hidden_state * wl * am[..., None]).sum() + (o.pooler_output * wp).sum()).backward()
        named = dict(e.encoder.named_parameters())
        gn = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight", "embeddings.token_type_embeddings.weight",
              "embeddings.LayerNorm.weight", "encoder.layer.0.attention.self.query.weight", "encoder.layer.1.output.dense.weight", "pooler.dense.weight"]
        out[mt + "_enc"] = dict(cfg=cfg, seed=seed, B=B, L=L, specials=sp, checksum=R.state_checksum(st), cls_name=type(e.encoder).__name__,
                                last_hidden_state=o.last_hidden_state.detach(), pooler_output=o.pooler_output.detach(),
                                hidden_states=torch.stack([h.detach() for h in o.hidden_states]),
                                grads={n: _thin(named[n].grad) for n in gn})
        # ---- decoder (non-degenerate decode recipe of G7)
        recipe = dict(std=0.6, emb_std=0.2, qk_std=0.15, pos_std=0.6)
        eos_bias = float(os.environ.get("G23_EOS", 2.5))     # (the head's LayerNorm bounds the logits: G7's 6.0 ends every row at once)
        dst = R.rand_state(R.causal_lm_shapes(cfg, mt), seed + 1, **recipe)
        bias_key = "lm_head.bias" if mt == "roberta" else "cls.predictions.bias"
        dst[bias_key][sp["sep"]] += eos_bias
        dd = R.write_proto_dir(os.path.join(tmp, mt + "_dec"), mt, cfg, dst)
        dec = dm.DecoderModel(AttrDict(proto=dd))
        hf = dec.decoder
        hf.config._attn_implementation = "eager"
        assert hf.config.is_decoder and hf.config.add_cross_attention
        S = 10
        g = torch.Generator().manual_seed(seed + 2)
        enc = torch.randn(B, S, cfg["hidden_size"], generator=g)
        enc_mask = torch.ones(B, S, dtype=torch.bool)
        enc_mask[1, 7:] = False
        enc[~enc_mask] = 0.0
        enc = enc.requires_grad_(True)
        dec.train()
        o = dec(input_ids=ids, attention_mask=am, encoder_outputs=enc, encoder_attention_mask=enc_mask)
        o["loss"].backward()
        named = dict(hf.named_parameters())
        base = mt + "."
        head = ["lm_head.dense.weight", "lm_head.layer_norm.weight", "lm_head.bias"] if mt == "roberta" else \
            ["cls.predictions.transform.dense.weight", "cls.predictions.transform.LayerNorm.weight", "cls.predictions.bias"]
        gn = [base + "embeddings.word_embeddings.weight", base + "embeddings.position_embeddings.weight", base + "embeddings.token_type_embeddings.weight",
              base + "encoder.layer.0.crossattention.self.key.weight", base + "encoder.layer.1.intermediate.dense.weight"] + head
        res = dict(cfg=cfg, seed=seed + 1, B=B, L=L, S=S, specials=sp, recipe=recipe, eos_bias=eos_bias, checksum=R.state_checksum(dst),
                   cls_name=type(hf).__name__, enc_mask=enc_mask, loss=o["loss"].detach(), logits=o["logits"].detach(),
                   grads={n: _thin(named[n].grad) for n in gn}, enc_grad=enc.grad.clone())
        dec.eval()
        max_len = 24
        for nb in (1, 4):
            args = dict(bos_token_id=sp["cls"], eos_token_id=sp["sep"], pad_token_id=sp["pad"], num_return_sequences=1, max_length=max_len,
                        use_cache=True, num_beams=nb, length_penalty=1.0, return_dict_in_generate=True, output_scores=True)
            with torch.no_grad():
                og = hf.generate(input_ids=torch.ones((B, 1), dtype=torch.long) * sp["cls"], generation_config=GenerationConfig(**args),
                                 encoder_hidden_states=enc.detach(), encoder_attention_mask=enc_mask,
                                 past_key_values=EncoderDecoderCache(DynamicCache(config=hf.config), DynamicCache(config=hf.config)))
            res[f"beams{nb}"] = dict(sequences=og.sequences, scores=getattr(og, "sequences_scores", None))
            print(mt, "beams", nb, og.sequences.tolist())
        res["max_len"] = max_len
        out[mt + "_dec"] = res
    save("g23_proto_towers", out)


# ------------------------------------------------------------------ G24: DeiT through VisualEncoder and RRG_HF
def gen_deit():
    """``backbone: deit`` (visual_encoder.py:59-61: DeiTModel(DeiTConfig(**kwargs), add_pooling_layer=False)) through VisualEncoder.encode,
    and the RRG_HF forward body on a VisionEncoderDecoderModel whose encoder is a DeiTModel (config/RRG/baseline-HF.yml:22)."""
    import ast
    from transformers import DeiTConfig, DeiTModel, VisionEncoderDecoderModel
    cfg, seed, B = R.DEIT_TINY, 251, 3
    enc = ve.VisualEncoder(backbone="deit", permute="no_permute", dropout_out=0.0, **cfg, attn_implementation="eager")
    st = R.rand_state(R.deit_shapes(cfg), seed)
    load_into(enc.model, st, vit_to_hf5)
    enc.eval()
    images = R.make_images(B, cfg["image_size"], seed=seed)
    with torch.no_grad():
        feats, mask = enc.encode(images)
    res = dict(cfg=cfg, seed=seed, B=B, checksum=R.state_checksum(st), cls_name=type(enc.model).__name__, features=feats, mask=mask)
    # RRG_HF.forward on a DeiT encoder (4-D images; same width as the decoder: no enc_to_dec_proj)
    tree = ast.parse(open(REF + "models/rrg/RRG_HF.py").read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "RRG_HF"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "RRG_HF.py", "exec"), ns)
    dcfg, L = R.DEC_TINY, 14
    deit = DeiTModel(DeiTConfig(**cfg, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager"), add_pooling_layer=False)
    load_into(deit, st, vit_to_hf5)
    dec, dst = build_ref_decoder(dcfg, seed + 1)
    model = VisionEncoderDecoderModel(encoder=deit, decoder=dec.decoder).eval()
    if not hasattr(model.decoder.config, "cross_attention_hidden_size"):
        model.decoder.config.cross_attention_hidden_size = None
    assert not hasattr(model, "enc_to_dec_proj")
    ids, am = R.make_reports(B, L, dcfg["vocab_size"], seed=seed)
    with torch.no_grad():
        o4 = ns["forward"](types.SimpleNamespace(model=model), ids, am, images)
    res.update(dec_cfg=dcfg, L=L, dec_checksum=R.state_checksum(dst), loss4=o4["loss"].clone(), logits4=o4["logits"].clone())
    save("g24_deit", res)


# ------------------------------------------------------------------ G25: RRG_HF built from local checkpoints
def gen_rrg_hf_pretrained():
    """G25: the reference's ``RRG_HF`` class itself (``__init__`` + ``forward``, lifted by AST; HF's own ``from_pretrained`` does the loading)
    on checkpoint directories written from the recipe:
      (a) ``encoderdecoder=<dir>``  -> VisionEncoderDecoderModel.from_pretrained             models/rrg/RRG_HF.py:24-25
          (ViT of width 64 with its pooler -> enc_to_dec_proj -> BertGenerationDecoder of width 128)
      (b) ``vision=<dir>``, ``decoder=<dir>`` (strings) -> AutoModel.from_pretrained / AutoModelForCausalLM.from_pretrained(add_cross_attention=True)
          :48-49, :86-87 (equal widths: the container then has no freshly initialised projection)."""
    import ast
    import tempfile
    from importlib import import_module
    from transformers import VisionEncoderDecoderModel, AutoModel, AutoModelForCausalLM
    tree = ast.parse(open(REF + "models/rrg/RRG_HF.py").read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "RRG_HF"][0]

    class DictConfig(dict):
        pass
    ns = {"torch": torch, "nn": torch.nn, "VisionEncoderDecoderModel": VisionEncoderDecoderModel, "AutoModel": AutoModel,
          "AutoModelForCausalLM": AutoModelForCausalLM, "DictConfig": DictConfig, "import_module": import_module,
          "evaluation": None, "evaluation_multi": None, "get_n_params": lambda m: 0}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "RRG_HF.py", "exec"), ns)
    Ref = ns["RRG_HF"]
    tmp = tempfile.mkdtemp()
    seed, B, L = 261, 3, 14
    dcfg = R.DEC_TINY
    out = {}
    # ---- (a)
    vcfg = dict(R.VIT_TINY, hidden_size=64, intermediate_size=128, num_attention_heads=1)
    vst = R.rand_state(R.vit_pooled_shapes(vcfg), seed)
    dst = R.rand_state(R.decoder_shapes(dcfg), seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    state = {"encoder." + k: v for k, v in vst.items()}
    state.update({"decoder." + k: v for k, v in dst.items()})
    state["enc_to_dec_proj.weight"] = 0.1 * torch.randn(dcfg["hidden_size"], vcfg["hidden_size"], generator=g)
    state["enc_to_dec_proj.bias"] = 0.02 * torch.randn(dcfg["hidden_size"], generator=g)
    d = R.write_ved_dir(os.path.join(tmp, "ved"), "vit", vcfg, dcfg, state)
    m = Ref(encoderdecoder=d).eval()
    assert type(m.model).__name__ == "VisionEncoderDecoderModel" and hasattr(m.model, "enc_to_dec_proj")
    m.model.encoder.config._attn_implementation = "eager"
    m.model.decoder.config._attn_implementation = "eager"
    if not hasattr(m.model.decoder.config, "cross_attention_hidden_size"):
        m.model.decoder.config.cross_attention_hidden_size = None      # PretrainedConfig default in the pinned 4.55.3; gone in 5.x
    torch.testing.assert_close(m.model.enc_to_dec_proj.weight.detach(), state["enc_to_dec_proj.weight"])
    torch.testing.assert_close(m.model.decoder.bert.encoder.layer[1].crossattention.self.key.weight.detach(), dst["bert.encoder.layer.1.crossattention.self.key.weight"])
    images = R.make_images(B, vcfg["image_size"], seed=seed)
    ids, am = R.make_reports(B, L, dcfg["vocab_size"], seed=seed)
    with torch.no_grad():
        o = m(ids, am, images)
    out["ved"] = dict(vit_cfg=vcfg, dec_cfg=dcfg, seed=seed, B=B, L=L, checksum=R.state_checksum(state), loss=o["loss"].clone(), logits=o["logits"].clone())
    # ---- (b)
    vcfg2 = dict(R.VIT_TINY)
    vst2 = R.rand_state(R.vit_pooled_shapes(vcfg2), seed + 10)
    dst2 = R.rand_state(R.decoder_shapes(dcfg), seed + 11)
    dv = R.write_proto_dir(os.path.join(tmp, "vit"), "vit", dict(vcfg2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, qkv_bias=True), vst2)
    dd = R.write_proto_dir(os.path.join(tmp, "dec"), "bert-generation", dict(dcfg, is_decoder=True, add_cross_attention=True, hidden_dropout_prob=0.0,
                                                                            attention_probs_dropout_prob=0.0), dst2)
    m2 = Ref(vision=dv, decoder=dd).eval()
    assert not hasattr(m2.model, "enc_to_dec_proj") and type(m2.model.decoder).__name__ == "BertGenerationDecoder"
    m2.model.encoder.config._attn_implementation = "eager"
    m2.model.decoder.config._attn_implementation = "eager"
    if not hasattr(m2.model.decoder.config, "cross_attention_hidden_size"):
        m2.model.decoder.config.cross_attention_hidden_size = None
    torch.testing.assert_close(m2.model.decoder.bert.encoder.layer[1].crossattention.self.key.weight.detach(), dst2["bert.encoder.layer.1.crossattention.self.key.weight"])
    torch.testing.assert_close(m2.model.encoder.embeddings.cls_token.detach(), vst2["embeddings.cls_token"])
    images2 = R.make_images(B, vcfg2["image_size"], seed=seed + 10)
    with torch.no_grad():
        o2 = m2(ids, am, images2)
    out["strings"] = dict(vit_cfg=vcfg2, dec_cfg=dcfg, seed=seed + 10, B=B, L=L, checksum=R.state_checksum(vst2) + R.state_checksum(dst2),
                          loss=o2["loss"].clone(), logits=o2["logits"].clone())
    print("G25 losses", float(o["loss"]), float(o2["loss"]))
    save("g25_rrg_hf_pretrained", out)


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["vit", "decoder", "rrg", "losses", "decode", "mvqa_text", "scst", "gloria_aggregate", "report_cleaning", "rrs", "vicreg", "bleu", "vit_multi", "schedulers", "model_compositions", "scst_sampling", "rrg_hf", "gloria_model", "ensemble_decode", "proto_towers", "deit", "rrg_hf_pretrained"]
    for w in which:
        globals()["gen_" + w]()
