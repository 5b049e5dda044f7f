"""DecoderModel -- same constructor / forward contract as ref:vilmedic/blocks/huggingface/decoder/decoder_model.py:8-53."""
import torch.nn as nn

from .bert_generation import BertGenerationDecoder, decoder_config


class DecoderModel(nn.Module):
    """If ``proto`` is set: the reference's ``AutoModelForCausalLM.from_pretrained(proto, config=<is_decoder, add_cross_attention>)``
    (decoder_model.py:17-21) -- a RobertaForCausalLM / BertLMHeadModel / BertGenerationDecoder on the HIP path, read from a local
    checkpoint directory or the local HF cache (blocks/huggingface/pretrained.py; never downloads); the cross-attention blocks an
    encoder checkpoint lacks are freshly initialised, as HF does.  Otherwise builds a BertGenerationDecoder from the YAML dict with
    is_decoder + add_cross_attention forced on (decoder_model.py:23-26)."""

    def __init__(self, decoder, **kwargs):
        super().__init__()
        decoder = dict(decoder)
        proto = decoder.pop("proto", None)
        if proto is not None:
            from ..pretrained import auto_causal_lm
            self.decoder = auto_causal_lm(proto)
        else:
            self.decoder = BertGenerationDecoder(decoder_config(decoder))
        self.generate = self.decoder.generate if hasattr(self.decoder, "generate") else None
        self.config = self.decoder.config

    def forward(self, input_ids, attention_mask, encoder_outputs=None, encoder_attention_mask=None, **kwargs):
        input_ids = input_ids.cuda()
        attention_mask = attention_mask.cuda()
        out = self.decoder(input_ids=input_ids, attention_mask=attention_mask, encoder_hidden_states=encoder_outputs,
                           encoder_attention_mask=encoder_attention_mask, labels=input_ids, **kwargs)
        return vars(out)

    def __repr__(self):
        return str(type(self.decoder).__name__) + "(" + str(dict(self.decoder.config)) + ")\n"
