"""BERT / RoBERTa towers on the HIP path: what the reference obtains from ``AutoModel.from_pretrained(proto)`` (text towers of
ConVIRT / GLoRIA, ref:vilmedic/blocks/huggingface/encoder/encoder_model.py:19-22) and from ``AutoModelForCausalLM.from_pretrained(proto,
is_decoder, add_cross_attention)`` (the pretrained report decoder, ref:vilmedic/blocks/huggingface/decoder/decoder_model.py:17-21).

Module trees and parameter names follow HF 4.55.3 (the reference's pin):
  BertModel / RobertaModel            hf:models/bert/modeling_bert.py, hf:models/roberta/modeling_roberta.py
      embeddings.{word,position,token_type}_embeddings, embeddings.LayerNorm, encoder.layer.{i}.*, pooler.dense
  BertLMHeadModel                     bert.* + cls.predictions.{bias, transform.dense, transform.LayerNorm, decoder}
  RobertaForCausalLM                  roberta.* + lm_head.{bias, dense, layer_norm, decoder}
The layer stack, attention, MLP and LayerNorm kernels are those of the BertGeneration path (vilmedic_amd.nn); what differs is the
embedding (token-type row, RoBERTa position ids: nn.BertFullEmbeddings) and the LM head's dense -> GELU -> LayerNorm transform.
"""
import torch
import torch.nn as nn

from ... import ops
from ...arena import arena_of
from ...nn import (BERT_DEFAULTS, ROBERTA_DEFAULTS, Affine, BertFullEmbeddings, BertPooler, BertStack, _Holder, _ln,
                   make_config, to_key_mask)
from ... import nn as _nn
from .decoder.bert_generation import ModelOutput, _rows


class BertModel(nn.Module):
    """HF BertModel / RobertaModel (``roberta=True``): embeddings + encoder (+ pooler); decoder stack when the config says
    ``is_decoder`` / ``add_cross_attention``."""

    def __init__(self, config, add_pooling_layer=True, roberta=False):
        super().__init__()
        self.config = config
        self.embeddings = BertFullEmbeddings(config, roberta=roberta)
        self.encoder = BertStack(config, cross=bool(config.add_cross_attention), enc_dim=config.get("encoder_hidden_size"))
        if add_pooling_layer:
            self.pooler = BertPooler(config)

    def forward(self, input_ids=None, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                output_hidden_states=False, **kw):
        arena = arena_of(self)
        arena.refresh()
        cfg = self.config
        input_ids = input_ids.cuda()
        x = self.embeddings(input_ids, arena)
        hs = [x] if output_hidden_states else None
        self_mask = to_key_mask(attention_mask.cuda() if attention_mask is not None else None)
        enc_mask = to_key_mask(encoder_attention_mask)
        enc = encoder_hidden_states
        if enc is not None:
            enc = (enc if enc.dtype == torch.bfloat16 else enc.to(torch.bfloat16)).contiguous()
        xr = None
        kvs, slots = self.encoder.cross_kv_all(enc, arena) if (enc is not None and self.encoder.cross and _nn.KV_ALL) else (None, None)
        for i, layer in enumerate(self.encoder.layer):
            x, xr = layer(x, arena, self_mask, bool(cfg.is_decoder), enc, enc_mask, xr=xr,
                          kv=kvs[i] if kvs else None, dkv_slot=slots[i] if slots else None)
            if hs is not None:
                hs.append(x)
        pooled = self.pooler(x, arena) if hasattr(self, "pooler") else None
        return ModelOutput(last_hidden_state=x, pooler_output=pooled, hidden_states=tuple(hs) if hs is not None else None,
                           past_key_values=None, attentions=None, cross_attentions=None)


class RobertaModel(BertModel):
    def __init__(self, config, add_pooling_layer=True):
        super().__init__(config, add_pooling_layer=add_pooling_layer, roberta=True)


class _CausalLM(nn.Module):
    """shared forward of BertLMHeadModel / RobertaForCausalLM: base model -> dense -> GELU -> LayerNorm -> tied decoder -> shifted CE
    (labels = input_ids, pads included: ref:decoder_model.py:46; hf:loss/loss_utils.py:49-72)."""

    # -- what vilmedic_amd.generation reads from a decoder: .bert (the base model), .lm_bias, .head_dense / .head_ln, .padded_vocab

    @property
    def padded_vocab(self):
        return (self.config.vocab_size + 7) // 8 * 8

    def generate(self, input_ids=None, **kwargs):
        from ...generation import generate
        return generate(self, input_ids=input_ids, **kwargs)

    def _transform(self, h, arena):
        dense, ln = self.head_dense, self.head_ln
        t = ops.linear_gelu(h, arena.shadow(dense.weight), dense.bias, wgrad_buf=arena.grad(dense.weight), bgrad_buf=arena.grad(dense.bias),
                            anchor=dense.weight)
        return _ln(arena, t, ln, self.config.layer_norm_eps)

    def forward(self, input_ids=None, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                labels=None, return_logits=True, row_weight=None, banned=None, top_k=None, **kw):
        arena = arena_of(self)
        arena.refresh()
        out = self.bert(input_ids, attention_mask=attention_mask, encoder_hidden_states=encoder_hidden_states,
                        encoder_attention_mask=encoder_attention_mask)
        h = self._transform(out.last_hidden_state, arena)
        emb = self.bert.embeddings.word_embeddings.weight
        V = self.config.vocab_size
        emb_sh = arena.shadow_rows(emb, self.padded_vocab)
        bias = self.lm_bias
        if labels is not None:
            if labels is not input_ids and not torch.equal(labels, input_ids):
                raise NotImplementedError("the HIP LM-head loss implements the reference's labels=input_ids contract "
                                          "(ref: decoder_model.py:46)")
            loss, logits, row_logp = ops.lm_head_loss(h, emb_sh, bias, input_ids.contiguous(), V, g_emb=_rows(arena.grad(emb), V),
                                                      g_bias=arena.grad(bias), want_logits=return_logits, row_weight=row_weight,
                                                      banned=banned, top_k=top_k)
            return ModelOutput(loss=loss, logits=logits, past_key_values=None, hidden_states=None, attentions=None,
                               cross_attentions=None, row_logp=row_logp)
        B, L, D = h.shape
        logits = ops.lm_logits_f32(h.reshape(B * L, D), emb_sh, bias, V).view(B, L, V)
        return ModelOutput(loss=None, logits=logits, past_key_values=None, hidden_states=None, attentions=None, cross_attentions=None)


class RobertaForCausalLM(_CausalLM):
    """hf:models/roberta/modeling_roberta.py RobertaForCausalLM: ``roberta`` (no pooler) + RobertaLMHead."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        d, s = config.hidden_size, config.initializer_range
        self.roberta = RobertaModel(config, add_pooling_layer=False)
        self.lm_head = _Holder()
        self.lm_head.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.lm_head.dense = Affine(d, d, std=s)
        self.lm_head.layer_norm = Affine(d, init="ones")
        self.lm_head.decoder = _Holder()                      # tied: the same Parameter objects under the HF key names
        self.lm_head.decoder.weight = self.roberta.embeddings.word_embeddings.weight
        self.lm_head.decoder.bias = self.lm_head.bias

    bert = property(lambda self: self.roberta)
    head_dense = property(lambda self: self.lm_head.dense)
    head_ln = property(lambda self: self.lm_head.layer_norm)
    lm_bias = property(lambda self: self.lm_head.bias)


class BertLMHeadModel(_CausalLM):
    """hf:models/bert/modeling_bert.py BertLMHeadModel: ``bert`` (no pooler) + BertOnlyMLMHead (``cls.predictions``)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        d, s = config.hidden_size, config.initializer_range
        # registered under the HF name "bert" through __setattr__ of nn.Module would collide with the ``bert`` property the
        # generation code reads; the property below simply returns the registered sub-module
        self._modules["bert"] = BertModel(config, add_pooling_layer=False)
        self.cls = _Holder()
        self.cls.predictions = _Holder()
        p = self.cls.predictions
        p.bias = nn.Parameter(torch.zeros(config.vocab_size))
        p.transform = _Holder()
        p.transform.dense = Affine(d, d, std=s)
        p.transform.LayerNorm = Affine(d, init="ones")
        p.decoder = _Holder()
        p.decoder.weight = self._modules["bert"].embeddings.word_embeddings.weight
        p.decoder.bias = p.bias

    @property
    def bert(self):
        return self._modules["bert"]

    head_dense = property(lambda self: self.cls.predictions.transform.dense)
    head_ln = property(lambda self: self.cls.predictions.transform.LayerNorm)
    lm_bias = property(lambda self: self.cls.predictions.bias)


def text_config(model_type, kwargs):
    """HF config defaults of the two architectures + the values of a checkpoint's config.json"""
    if model_type in ("roberta", "bert"):
        cfg = make_config(ROBERTA_DEFAULTS if model_type == "roberta" else BERT_DEFAULTS, kwargs)
        if (cfg.get("position_embedding_type") or "absolute") != "absolute":
            raise NotImplementedError(f"position_embedding_type={cfg.position_embedding_type!r}: the HIP path implements absolute position embeddings "
                                      "(what every checkpoint the reference's YAMLs name uses)")
        if int(cfg.type_vocab_size) < 1:
            raise ValueError("type_vocab_size must be >= 1 (token-type row 0 is added to every token)")
        return cfg
    raise NotImplementedError(f"model_type {model_type!r}: the HIP path builds 'bert', 'roberta' and 'bert-generation' text models")
