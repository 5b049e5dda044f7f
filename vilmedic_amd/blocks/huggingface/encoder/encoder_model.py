"""EncoderModel -- ref: vilmedic/blocks/huggingface/encoder/encoder_model.py:10-66."""
import torch.nn as nn

from ....arena import arena_of
from ....nn import BERT_GEN_DEFAULTS, BertPooler, Config, make_config
from ..decoder.bert_generation import BertGenerationEncoder


class EncoderModel(nn.Module):
    """``proto``: a pretrained checkpoint -- the reference's ``AutoModel.from_pretrained(proto)`` (encoder_model.py:19-22) -- built as a
    BertModel / RobertaModel / BertGenerationEncoder on the HIP path from a local checkpoint directory or the local HF cache
    (blocks/huggingface/pretrained.py; a hub name that is not cached raises: this package never downloads);
    ``proto: null`` builds a random BertGenerationEncoder from the dict (bidirectional, no cross-attention).  ``add_pooling_layer`` puts
    an extra BertPooler on top in both cases (encoder_model.py:28-29; "roberta already has a pooler layer").  The output always carries
    ``pooler_output`` and supports both attribute and ``[...]`` access (the reference sets it by ``setattr`` and reads it by key:
    SURVEY §2.1)."""

    def __init__(self, encoder, **kwargs):
        super().__init__()
        encoder = dict(encoder)
        proto = encoder.pop("proto", None)
        add_pool = bool(encoder.pop("add_pooling_layer", False))
        encoder.pop("last_n_layers", None)
        if proto is not None:
            from ..pretrained import auto_model
            self.encoder = auto_model(proto)
            cfg = self.encoder.config
        else:
            cfg = make_config(BERT_GEN_DEFAULTS, encoder)
            cfg.is_decoder = False
            cfg.add_cross_attention = False
            self.encoder = BertGenerationEncoder(cfg)
        if add_pool:
            self.pooler = BertPooler(cfg)

    def forward(self, input_ids=None, attention_mask=None, output_hidden_states=None, **kwargs):
        arena = arena_of(self)
        arena.refresh()
        out = self.encoder(input_ids=input_ids, attention_mask=attention_mask, output_hidden_states=bool(output_hidden_states))
        if hasattr(self, "pooler"):
            out.pooler_output = self.pooler(out.last_hidden_state, arena)
        elif getattr(out, "pooler_output", None) is None:
            out.pooler_output = out.last_hidden_state[:, 0].float()
        return out

    def __repr__(self):
        return str(type(self.encoder).__name__) + "(" + str(dict(self.encoder.config)) + ")\n"
