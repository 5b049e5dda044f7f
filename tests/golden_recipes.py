"""Deterministic weight / input recipes shared by ``tools/make_golden.py`` (which
feeds them to the *reference* blocks to produce ``tests/golden/*.pt``) and by
the tests (which feed the same tensors to the oracle and to the HIP path).

Storing the recipe (seed + shapes) instead of the tensors keeps fixtures small;
every fixture also stores a checksum of the generated state so RNG drift
between torch builds is detected instead of silently mis-compared.

Parameter names are the ones a reference checkpoint holds (HF transformers
4.55.3, the reference's pin in ``setup.py:29``).
"""
import torch


# ----------------------------------------------------------------------------- shapes
def vit_shapes(cfg, prefix=""):
    d, ff, p, c = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"], cfg.get("num_channels", 3)
    n = (cfg["image_size"] // p) ** 2
    s = {
        "embeddings.cls_token": (1, 1, d),
        "embeddings.position_embeddings": (1, n + 1, d),
        "embeddings.patch_embeddings.projection.weight": (d, c, p, p),
        "embeddings.patch_embeddings.projection.bias": (d,),
        "layernorm.weight": (d,), "layernorm.bias": (d,),
    }
    for i in range(cfg["num_hidden_layers"]):
        L = f"encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            s[L + f"attention.attention.{nm}.weight"] = (d, d)
            s[L + f"attention.attention.{nm}.bias"] = (d,)
        s[L + "attention.output.dense.weight"] = (d, d)
        s[L + "attention.output.dense.bias"] = (d,)
        s[L + "intermediate.dense.weight"] = (ff, d)
        s[L + "intermediate.dense.bias"] = (ff,)
        s[L + "output.dense.weight"] = (d, ff)
        s[L + "output.dense.bias"] = (d,)
        for nm in ("layernorm_before", "layernorm_after"):
            s[L + nm + ".weight"] = (d,)
            s[L + nm + ".bias"] = (d,)
    return {prefix + k: v for k, v in s.items()}


def bert_layer_shapes(cfg, prefix, cross, enc_dim=None):
    d, ff = cfg["hidden_size"], cfg["intermediate_size"]
    enc_dim = enc_dim or d
    s = {}
    blocks = [("attention", d)] + ([("crossattention", enc_dim)] if cross else [])
    for blk, kv_in in blocks:
        s[f"{blk}.self.query.weight"] = (d, d)
        s[f"{blk}.self.query.bias"] = (d,)
        for nm in ("key", "value"):
            s[f"{blk}.self.{nm}.weight"] = (d, kv_in)
            s[f"{blk}.self.{nm}.bias"] = (d,)
        s[f"{blk}.output.dense.weight"] = (d, d)
        s[f"{blk}.output.dense.bias"] = (d,)
        s[f"{blk}.output.LayerNorm.weight"] = (d,)
        s[f"{blk}.output.LayerNorm.bias"] = (d,)
    s["intermediate.dense.weight"] = (ff, d)
    s["intermediate.dense.bias"] = (ff,)
    s["output.dense.weight"] = (d, ff)
    s["output.dense.bias"] = (d,)
    s["output.LayerNorm.weight"] = (d,)
    s["output.LayerNorm.bias"] = (d,)
    return {prefix + k: v for k, v in s.items()}


def bert_embedding_shapes(cfg, prefix):
    d = cfg["hidden_size"]
    return {
        prefix + "word_embeddings.weight": (cfg["vocab_size"], d),
        prefix + "position_embeddings.weight": (cfg["max_position_embeddings"], d),
        prefix + "LayerNorm.weight": (d,), prefix + "LayerNorm.bias": (d,),
    }


def deit_shapes(cfg, prefix=""):
    """HF DeiTModel(add_pooling_layer=False): the ViT tree + ``embeddings.distillation_token`` and n + 2 positions"""
    s = vit_shapes(cfg)
    n = (cfg["image_size"] // cfg["patch_size"]) ** 2
    s["embeddings.distillation_token"] = (1, 1, cfg["hidden_size"])
    s["embeddings.position_embeddings"] = (1, n + 2, cfg["hidden_size"])
    return {prefix + k: v for k, v in s.items()}


def text_model_shapes(cfg, prefix="", pooler=True, cross=False):
    """HF BertModel / RobertaModel: embeddings (word, position, token_type, LayerNorm) + encoder.layer.* (+ pooler.dense)"""
    d = cfg["hidden_size"]
    s = bert_embedding_shapes(cfg, "embeddings.")
    s["embeddings.token_type_embeddings.weight"] = (cfg["type_vocab_size"], d)
    for i in range(cfg["num_hidden_layers"]):
        s.update(bert_layer_shapes(cfg, f"encoder.layer.{i}.", cross=cross))
    if pooler:
        s["pooler.dense.weight"], s["pooler.dense.bias"] = (d, d), (d,)
    return {prefix + k: v for k, v in s.items()}


def causal_lm_shapes(cfg, model_type):
    """RobertaForCausalLM / BertLMHeadModel with cross-attention (the tied ``*.decoder.weight`` / ``.bias`` aliases are not generated)"""
    d, V = cfg["hidden_size"], cfg["vocab_size"]
    s = text_model_shapes(cfg, model_type + ".", pooler=False, cross=True)
    if model_type == "roberta":
        s.update({"lm_head.bias": (V,), "lm_head.dense.weight": (d, d), "lm_head.dense.bias": (d,),
                  "lm_head.layer_norm.weight": (d,), "lm_head.layer_norm.bias": (d,)})
    else:
        s.update({"cls.predictions.bias": (V,), "cls.predictions.transform.dense.weight": (d, d), "cls.predictions.transform.dense.bias": (d,),
                  "cls.predictions.transform.LayerNorm.weight": (d,), "cls.predictions.transform.LayerNorm.bias": (d,)})
    return s


def write_proto_dir(path, model_type, cfg, state, architectures=None):
    """a local pretrained-checkpoint directory (config.json + model.safetensors) holding ``state`` -- what ``proto: <dir>`` of
    EncoderModel / DecoderModel points at.  Written from the recipe so that fixtures stay small; make_golden.py loads the SAME
    directory through the reference's ``AutoModel.from_pretrained`` / ``AutoModelForCausalLM.from_pretrained``."""
    import json
    import os
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    c = dict(cfg, model_type=model_type, hidden_act="gelu", position_embedding_type="absolute")
    if architectures:
        c["architectures"] = architectures
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(c, f, indent=1, sort_keys=True)
    save_file({k: v.contiguous() for k, v in state.items()}, os.path.join(path, "model.safetensors"))
    return path


def write_ved_dir(path, enc_type, vit_cfg, dec_cfg, state):
    """a ``VisionEncoderDecoderModel.save_pretrained`` directory (config.json with the two sub-configs + model.safetensors named
    ``encoder.*`` / ``decoder.*`` / ``enc_to_dec_proj.*``) -- what ``RRG_HF(encoderdecoder=<dir>)`` loads (ref:models/rrg/RRG_HF.py:24-25)"""
    import json
    import os
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    c = dict(model_type="vision-encoder-decoder", is_encoder_decoder=True, tie_word_embeddings=False,
             architectures=["VisionEncoderDecoderModel"],
             encoder=dict(vit_cfg, model_type=enc_type, hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, qkv_bias=True),
             decoder=dict(dec_cfg, model_type="bert-generation", hidden_act="gelu", is_decoder=True, add_cross_attention=True,
                          hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, position_embedding_type="absolute"))
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(c, f, indent=1, sort_keys=True)
    save_file({k: v.contiguous() for k, v in state.items()}, os.path.join(path, "model.safetensors"))
    return path


def vit_pooled_shapes(cfg, prefix=""):
    """HF ViTModel(config) with its default pooler (what VisionEncoderDecoderModel / AutoModel hold)"""
    s = vit_shapes(cfg)
    d = cfg["hidden_size"]
    s["pooler.dense.weight"], s["pooler.dense.bias"] = (d, d), (d,)
    return {prefix + k: v for k, v in s.items()}


def decoder_shapes(cfg, prefix=""):
    """BertGenerationDecoder with cross-attention; LM head tied to word embeddings
    (``lm_head.decoder.weight`` is an alias and is not generated)."""
    s = bert_embedding_shapes(cfg, "bert.embeddings.")
    for i in range(cfg["num_hidden_layers"]):
        s.update(bert_layer_shapes(cfg, f"bert.encoder.layer.{i}.", cross=True))
    s["lm_head.bias"] = (cfg["vocab_size"],)
    return {prefix + k: v for k, v in s.items()}


def text_encoder_shapes(cfg, prefix=""):
    s = bert_embedding_shapes(cfg, "embeddings.")
    for i in range(cfg["num_hidden_layers"]):
        s.update(bert_layer_shapes(cfg, f"encoder.layer.{i}.", cross=False))
    return {prefix + k: v for k, v in s.items()}


def bert_stack_shapes(cfg, prefix=""):
    s = {}
    for i in range(cfg["num_hidden_layers"]):
        s.update(bert_layer_shapes(cfg, f"layer.{i}.", cross=False))
    return {prefix + k: v for k, v in s.items()}


# ----------------------------------------------------------------------------- tensors
def rand_state(shapes, seed, std=0.05, emb_std=None, qk_std=None, pos_std=None):
    """LayerNorm weights ~ 1+0.1N, biases ~ 0.02N, everything else ~ std*N
    (embedding tables ~ emb_std*N, query/key projections ~ qk_std*N when given)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shape = shapes[name]
        x = torch.randn(*shape, generator=g)
        low = name.lower()
        if ("layernorm" in low) and name.endswith(".weight"):
            x = 1.0 + 0.1 * x
        elif name.endswith(".bias"):
            x = 0.02 * x
        elif pos_std is not None and "position_embeddings" in low:
            x = pos_std * x
        elif emb_std is not None and "embeddings" in low:
            x = emb_std * x
        elif qk_std is not None and (".query.weight" in low or ".key.weight" in low):
            x = qk_std * x
        else:
            x = std * x
        out[name] = x
    return out


def state_checksum(state):
    return float(sum(v.double().abs().sum() for v in state.values()))


def make_images(B, size, seed=0, channels=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, channels, size, size, generator=g)


def make_reports(B, L, V, seed=0, cls=0, pad=1, sep=2):
    """[CLS]=0, body U{3..V-1} of length U{L/2..L-2}, [SEP]=2, [PAD]=1."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.full((B, L), pad, dtype=torch.long)
    mask = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(L // 2, L - 1, (1,), generator=g))
        ids[b, 0] = cls
        ids[b, 1:1 + n - 1] = torch.randint(3, V, (n - 1,), generator=g)
        ids[b, n] = sep
        mask[b, :n + 1] = 1
    return ids, mask


# ----------------------------------------------------------------------------- configs
# (head_dim = 64 everywhere: the MFMA attention kernels contract over 32-wide k-steps of the head dimension)
VIT_TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                image_size=32, patch_size=8, num_channels=3, layer_norm_eps=1e-12)
VIT_B16_1L = dict(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072,
                  image_size=224, patch_size=16, num_channels=3, layer_norm_eps=1e-12)
DEC_TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                vocab_size=97, max_position_embeddings=64, layer_norm_eps=1e-5,
                bos_token_id=0, pad_token_id=1, eos_token_id=2)
DEC_768_2L = dict(hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                  vocab_size=1000, max_position_embeddings=514, layer_norm_eps=1e-5,
                  bos_token_id=0, pad_token_id=1, eos_token_id=2)
# pretrained-`proto` towers (ref:encoder_model.py:19-22, decoder_model.py:17-21): RoBERTa-shaped (pad 1, one token type, 1e-5) and
# BERT-shaped (pad 0, two token types, 1e-12) tiny checkpoints
ROBERTA_TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=97,
                    max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5, bos_token_id=0, pad_token_id=1, eos_token_id=2,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
BERT_TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=97,
                 max_position_embeddings=64, type_vocab_size=2, layer_norm_eps=1e-12, pad_token_id=0,
                 hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
DEIT_TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                 image_size=32, patch_size=8, num_channels=3, layer_norm_eps=1e-12)
MVQA_TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=192,
                 layer_norm_eps=1e-12)
TXT_TINY = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                vocab_size=97, max_position_embeddings=64, layer_norm_eps=1e-12, pad_token_id=1)
