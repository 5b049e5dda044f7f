"""BertGenerationDecoder / BertGenerationEncoder on the HIP path.

Same module tree, parameter names and forward contract as HF's classes the reference instantiates
(ref: vilmedic/blocks/huggingface/decoder/decoder_model.py:23-26, encoder/encoder_model.py:24-26;
 hf: models/bert_generation/modeling_bert_generation.py:495-703).
"""
import torch
import torch.nn as nn

from .... import ops
from ....arena import arena_of
from .... import nn as _nn
from ....nn import BERT_GEN_DEFAULTS, BertEmbeddings, BertStack, Config, _Holder, make_config, to_key_mask


class ModelOutput:
    """attribute bag; ``vars(out)`` gives the dict the reference's DecoderModel.forward returns (decoder_model.py:48)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __getitem__(self, k):
        return self.__dict__[k]

    def __contains__(self, k):
        return k in self.__dict__

    def keys(self):
        return self.__dict__.keys()


class BertGenerationEncoder(nn.Module):
    """embeddings + layer stack (``bert`` inside the decoder; the text tower of ConVIRT/GLoRIA when bidirectional)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertStack(config, cross=bool(config.add_cross_attention), enc_dim=config.get("encoder_hidden_size"))

    def forward(self, input_ids, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                output_hidden_states=False, **kw):
        arena = arena_of(self)
        arena.refresh()
        cfg = self.config
        x = self.embeddings(input_ids, arena)
        hs = [x] if output_hidden_states else None
        self_mask = to_key_mask(attention_mask)
        enc_mask = to_key_mask(encoder_attention_mask)
        enc = encoder_hidden_states
        if enc is not None and enc.dtype != torch.bfloat16:
            enc = enc.to(torch.bfloat16)
        xr = None          # alias of x for the next residual (see nn.BertLayer: fuses the fork's gradient sum into LN backward)
        enc = enc.contiguous() if enc is not None else None
        # cross-attention K|V of all layers in one GEMM (and one dgrad / wgrad in backward): nn.BertStack.cross_kv_all
        kvs, slots = self.encoder.cross_kv_all(enc, arena) if (enc is not None and self.encoder.cross and _nn.KV_ALL) else (None, None)
        for i, layer in enumerate(self.encoder.layer):
            x, xr = layer(x, arena, self_mask, bool(cfg.is_decoder), enc, enc_mask, xr=xr,
                          kv=kvs[i] if kvs else None, dkv_slot=slots[i] if slots else None)
            if hs is not None:
                hs.append(x)
        return ModelOutput(last_hidden_state=x, hidden_states=tuple(hs) if hs is not None else None,
                           past_key_values=None, attentions=None, cross_attentions=None)


class BertGenerationDecoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bert = BertGenerationEncoder(config)
        self.lm_head = _Holder()
        self.lm_head.bias = nn.Parameter(torch.zeros(config.vocab_size))
        # tied head: same Parameter objects under the HF key names (lm_head.decoder.weight / .bias)
        self.lm_head.decoder = _Holder()
        self.lm_head.decoder.weight = self.bert.embeddings.word_embeddings.weight
        self.lm_head.decoder.bias = self.lm_head.bias

    def generate(self, input_ids=None, **kwargs):
        """greedy / sampling / beam search with a KV cache (vilmedic_amd.generation); same keyword surface as the HF call
        the reference makes (ref: blocks/huggingface/decoder/evaluation.py:73-78, blocks/rl/SCST.py:115-126,142-157)."""
        from ....generation import generate
        return generate(self, input_ids=input_ids, **kwargs)

    @property
    def padded_vocab(self):
        return (self.config.vocab_size + 7) // 8 * 8

    # (the names vilmedic_amd.generation reads; BertLMHeadModel / RobertaForCausalLM have a dense -> GELU -> LayerNorm transform here)
    lm_bias = property(lambda self: self.lm_head.bias)
    head_dense = None
    head_ln = None

    def forward(self, input_ids=None, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                labels=None, return_logits=True, row_weight=None, banned=None, top_k=None, **kw):
        arena = arena_of(self)   # root the arena HERE (before the sub-module forward) so it covers lm_head.bias too
        arena.refresh()
        out = self.bert(input_ids, attention_mask=attention_mask, encoder_hidden_states=encoder_hidden_states,
                        encoder_attention_mask=encoder_attention_mask)
        h = out.last_hidden_state
        emb = self.bert.embeddings.word_embeddings.weight
        V = self.config.vocab_size
        emb_sh = arena.shadow_rows(emb, self.padded_vocab)
        loss, logits = None, None
        if labels is not None:
            if labels is not input_ids and not torch.equal(labels, input_ids):
                raise NotImplementedError("the HIP LM-head loss implements the reference's labels=input_ids contract "
                                          "(ref: decoder_model.py:46)")
            loss, logits, row_logp = ops.lm_head_loss(h, emb_sh, self.lm_head.bias, input_ids.contiguous(), V,
                                                      g_emb=_rows(arena.grad(emb), V), g_bias=arena.grad(self.lm_head.bias),
                                                      want_logits=return_logits, row_weight=row_weight, banned=banned, top_k=top_k)
            return ModelOutput(loss=loss, logits=logits, past_key_values=None, hidden_states=None, attentions=None,
                               cross_attentions=None, row_logp=row_logp)
        else:
            B, L, D = h.shape
            logits = ops.lm_logits_f32(h.reshape(B * L, D), emb_sh, self.lm_head.bias, V).view(B, L, V)
        return ModelOutput(loss=loss, logits=logits, past_key_values=None, hidden_states=None, attentions=None,
                           cross_attentions=None)


def _rows(g, V):
    return g[:V] if g is not None else None


def decoder_config(kwargs):
    cfg = make_config(BERT_GEN_DEFAULTS, kwargs)
    cfg.is_decoder = True
    cfg.add_cross_attention = True
    return cfg
