"""RRS (report summarisation) -- ref:vilmedic/models/rrs/RRS.py:11-60: a bidirectional text encoder (EncoderModel) feeding the
cross-attending causal decoder (DecoderModel), both on the HIP path and both in one parameter arena."""
import torch
import torch.nn as nn

from ...arena import arena_of
from ...blocks.huggingface.decoder.decoder_model import DecoderModel
from ...blocks.huggingface.decoder.evaluation import evaluation
from ...blocks.huggingface.encoder.encoder_model import EncoderModel
from ..utils import get_n_params


class RRS(nn.Module):
    def __init__(self, encoder, decoder, dl=None, **kwargs):
        super().__init__()
        encoder, decoder = dict(encoder), dict(decoder)
        if dl:
            encoder["vocab_size"] = dl.dataset.src.tokenizer.vocab_size
            decoder["vocab_size"] = dl.dataset.tgt.tokenizer.vocab_size
        self.dec = DecoderModel(decoder)         # decoder first: its gradients are complete first and all-reduce while the
        self.enc = EncoderModel(encoder)         # encoder's backward still runs (parallel.ArenaDDP.split_at)
        self.eval_func = evaluation
        self.tokenizer = dl.dataset.tgt_tokenizer if dl else None
        self.split_backward = False      # set by ArenaDDP: detach the encoder output so backward can run in two phases
        self._split = None

    def forward(self, input_ids, attention_mask, decoder_input_ids, decoder_attention_mask, encoder_outputs=None,
                encoder_attention_mask=None, epoch=None, iteration=None, **kwargs):
        arena_of(self).refresh()          # one arena for encoder + decoder parameters
        if encoder_outputs is None:
            encoder_outputs, encoder_attention_mask = self.encode(input_ids, attention_mask, **kwargs)
            if self.split_backward and self.training and torch.is_grad_enabled() and encoder_outputs.requires_grad:
                leaf = encoder_outputs.detach().requires_grad_(True)
                self._split = (encoder_outputs, leaf)
                encoder_outputs = leaf
        return self.dec(input_ids=decoder_input_ids, attention_mask=decoder_attention_mask, encoder_outputs=encoder_outputs,
                        encoder_attention_mask=encoder_attention_mask, **kwargs)

    def encode(self, input_ids, attention_mask, **kwargs):
        input_ids, attention_mask = input_ids.cuda(), attention_mask.cuda()
        return self.enc(input_ids, attention_mask).last_hidden_state, attention_mask

    def __repr__(self):
        s = "model: RRS\n"
        s += "(enc):" + str(self.enc) + "\n"
        s += "(dec):" + str(self.dec) + "\n"
        s += "{}\n".format(get_n_params(self))
        return s
