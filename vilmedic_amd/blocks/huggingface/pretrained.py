"""``from_pretrained`` for the text towers, from files on disk: what ``AutoConfig.from_pretrained(path)`` +
``AutoModel.from_pretrained(path)`` / ``AutoModelForCausalLM.from_pretrained(path, config=...)`` do for the reference
(ref:vilmedic/blocks/huggingface/encoder/encoder_model.py:19-22, decoder/decoder_model.py:17-21), restricted to the architectures of
the HIP path (``model_type`` bert / roberta / bert-generation) and to local data: a checkpoint directory, or a hub name that is
already in the local HuggingFace cache (there is no network on the training nodes this package targets).

Loading follows HF's rules: keys are matched after adding / stripping the base-model prefix (``roberta.`` / ``bert.``), tied weights
fill each other, checkpoint keys the model does not have are ignored (a masked-LM checkpoint's head when only the encoder is built),
and parameters the checkpoint lacks keep their fresh initialisation (the cross-attention blocks and the pooler of a decoder built
from an encoder checkpoint) -- both lists are logged and kept on the module (``_vm_missing_keys`` / ``_vm_unexpected_keys``).
Anything else that is missing raises: a silently half-loaded tower is never returned.
"""
import json
import logging
import os
import re

import torch

log = logging.getLogger(__name__)

# parameters HF itself initialises afresh when a checkpoint of another head / task is loaded
_FRESH_OK = re.compile(r"(crossattention\.|pooler\.|lm_head\.|cls\.predictions\.|position_ids$|token_type_ids$)")
_WEIGHT_FILES = ("model.safetensors", "pytorch_model.bin")


def resolve(proto):
    """-> directory holding config.json + weights.  A directory is taken as is; a hub name is looked up in the local HF cache only."""
    if os.path.isdir(proto):
        return proto
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(proto, local_files_only=True, allow_patterns=["config.json", *_WEIGHT_FILES])
    except Exception as e:          # not cached / hub library unusable
        raise NotImplementedError(f"proto={proto!r} is neither a local checkpoint directory nor in the local HuggingFace cache, and "
                                  f"this package never downloads (no network): point `proto` at a directory with config.json and "
                                  f"model.safetensors / pytorch_model.bin  [{type(e).__name__}]") from None


_VISION_DROP = ("encoder_stride", "pooler_act", "pooler_output_size", "attn_implementation")    # keys of ViT / DeiT configs the HIP encoder has no use for


def clean_config(cfg):
    """drop the bookkeeping keys of a (sub-)config dict read from config.json"""
    cfg = dict(cfg)
    for k in ("architectures", "transformers_version", "dtype", "torch_dtype", "_name_or_path", "auto_map", "return_dict", "output_hidden_states",
              "output_attentions", "tie_word_embeddings", "classifier_dropout", "gradient_checkpointing", "_attn_implementation",
              "id2label", "label2id", "problem_type", "finetuning_task", "tokenizer_class", "task_specific_params", "chunk_size_feed_forward",
              "tie_encoder_decoder", "is_encoder_decoder", "pruned_heads", "torchscript", "cross_attention_hidden_size"):
        cfg.pop(k, None)
    return cfg


def read_config(path):
    with open(os.path.join(path, "config.json")) as f:
        return clean_config(json.load(f))


def read_state(path):
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"{path}: no model.safetensors / pytorch_model.bin")


def load_into(module, state, base_prefix):
    """copy ``state`` (a checkpoint's tensors) into ``module`` under HF's matching rules -> (missing, unexpected)"""
    own = module.state_dict()
    state = {k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias"): v for k, v in state.items()}
    has_prefix_model = any(k.startswith(base_prefix + ".") for k in own)
    has_prefix_ckpt = any(k.startswith(base_prefix + ".") for k in state)
    if has_prefix_model and not has_prefix_ckpt:            # head model <- base-model checkpoint
        state = {(base_prefix + "." + k): v for k, v in state.items()}
    dropped = []
    if has_prefix_ckpt and not has_prefix_model:            # base model <- head-model checkpoint: its head is dropped
        dropped = [k for k in state if not k.startswith(base_prefix + ".")]
        state = {k[len(base_prefix) + 1:]: v for k, v in state.items() if k.startswith(base_prefix + ".")}
    merged = dict(own)
    unexpected = sorted([k for k in state if k not in own] + dropped)
    present = set()
    for k, v in state.items():
        if k in own:
            if tuple(v.shape) != tuple(own[k].shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}")
            merged[k] = v
            present.add(k)
    # tied parameters (one Parameter under two names: lm_head.decoder.weight <-> word embeddings; safetensors keeps one name only)
    by_ptr = {}
    for k, v in own.items():
        by_ptr.setdefault(v.data_ptr(), []).append(k)
    for names in by_ptr.values():
        src = [k for k in names if k in present]
        if src:
            for k in names:
                if k not in present:
                    merged[k] = merged[src[0]]
                    present.add(k)
    missing = sorted(k for k in own if k not in present)
    bad = [k for k in missing if not _FRESH_OK.search(k)]
    if bad:
        raise RuntimeError(f"checkpoint lacks parameters the model needs: {bad[:8]}{' ...' if len(bad) > 8 else ''}")
    module.load_state_dict(merged, strict=True)
    if missing:
        log.warning("newly initialised (absent from the checkpoint): %s", missing)
    if unexpected:
        log.warning("ignored checkpoint keys: %s", unexpected)
    module._vm_missing_keys, module._vm_unexpected_keys = missing, unexpected
    return missing, unexpected


def auto_model(proto, add_pooling_layer=True):
    """AutoModel.from_pretrained(proto) for the text towers -> nn.Module (BertModel / RobertaModel / BertGenerationEncoder)"""
    from .bert_models import BertModel, RobertaModel, text_config
    from .decoder.bert_generation import BertGenerationEncoder
    from ...nn import BERT_GEN_DEFAULTS, make_config
    path = resolve(proto)
    cfg = read_config(path)
    mt = cfg.pop("model_type", None)
    if mt == "bert-generation":
        c = make_config(BERT_GEN_DEFAULTS, cfg)
        model, prefix = BertGenerationEncoder(c), "bert"
    else:
        c = text_config(mt, cfg)
        model = (RobertaModel if mt == "roberta" else BertModel)(c, add_pooling_layer=add_pooling_layer)
        prefix = mt
    c.model_type = mt
    load_into(model, read_state(path), prefix)
    return model


def auto_causal_lm(proto):
    """AutoModelForCausalLM.from_pretrained(proto, config=<is_decoder, add_cross_attention>) -> decoder module"""
    path = resolve(proto)
    model, prefix = _causal_lm_from_config(read_config(path))
    load_into(model, read_state(path), prefix)
    return model


def _causal_lm_from_config(cfg):
    """decoder module for a cleaned config dict (``model_type`` inside), cross-attention switched on -> (module, base-model prefix)"""
    from .bert_models import BertLMHeadModel, RobertaForCausalLM, text_config
    from .decoder.bert_generation import BertGenerationDecoder, decoder_config
    cfg = dict(cfg)
    mt = cfg.pop("model_type", None)
    if mt == "bert-generation":
        c = decoder_config(cfg)
        model, prefix = BertGenerationDecoder(c), "bert"
    else:
        c = text_config(mt, cfg)
        c.is_decoder = True
        c.add_cross_attention = True
        model = (RobertaForCausalLM if mt == "roberta" else BertLMHeadModel)(c)
        prefix = mt
    c.model_type = mt
    return model, prefix


def _vision_from_config(cfg, vit_cls):
    """ViTModel / DeiTModel (with HF's default pooler) for a cleaned config dict"""
    from ...nn import VIT_DEFAULTS, make_config
    cfg = dict(cfg)
    mt = cfg.pop("model_type", None)
    if mt not in ("vit", "deit"):
        raise NotImplementedError(f"vision model_type {mt!r}: the HIP path builds 'vit' and 'deit' encoders")
    for k in _VISION_DROP:
        cfg.pop(k, None)
    c = make_config(VIT_DEFAULTS, cfg)
    c.model_type = mt
    return vit_cls(c, distillation=(mt == "deit"))


def auto_vision_model(proto, vit_cls):
    """AutoModel.from_pretrained(proto) for the image tower of RRG_HF (ref:vilmedic/models/rrg/RRG_HF.py:48-49): a ViT / DeiT checkpoint
    directory -> ``vit_cls`` (the encoder class with HF's pooler) loaded under HF's matching rules"""
    path = resolve(proto)
    cfg = read_config(path)
    model = _vision_from_config(cfg, vit_cls)
    load_into(model, read_state(path), cfg.get("model_type"))
    return model


def vision_encoder_decoder(proto, vit_cls, wrap):
    """VisionEncoderDecoderModel.from_pretrained(proto) (ref:vilmedic/models/rrg/RRG_HF.py:24-25): config.json holds the two sub-configs,
    the weights are named ``encoder.*`` / ``decoder.*`` / ``enc_to_dec_proj.*``; ``wrap(encoder, decoder)`` builds the container"""
    path = resolve(proto)
    with open(os.path.join(path, "config.json")) as f:
        raw = json.load(f)
    if raw.get("model_type") != "vision-encoder-decoder":
        raise NotImplementedError(f"{path}: model_type {raw.get('model_type')!r} is not a VisionEncoderDecoderModel checkpoint")
    encoder = _vision_from_config(clean_config(raw["encoder"]), vit_cls)
    decoder, _ = _causal_lm_from_config(clean_config(raw["decoder"]))
    model = wrap(encoder, decoder)
    load_into(model, read_state(path), "\0no-base-prefix")
    for k in ("decoder_start_token_id", "pad_token_id", "eos_token_id"):
        if raw.get(k) is not None:
            model.config[k] = raw[k]
    return model
