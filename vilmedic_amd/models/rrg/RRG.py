"""RRG -- ref:vilmedic/models/rrg/RRG.py:10-52 (VisualEncoder + DecoderModel), on the HIP path."""
import torch
import torch.nn as nn

from ...arena import arena_of
from ...blocks.huggingface.decoder.decoder_model import DecoderModel
from ...blocks.huggingface.decoder.evaluation import evaluation
from ...blocks.vision import *  # noqa: F401,F403  (eval(proto) namespace, RRG.py:20)
from ..utils import get_n_params


class RRG(nn.Module):
    def __init__(self, decoder, cnn, dl=None, **kwargs):
        super().__init__()
        decoder = dict(decoder)
        if dl:
            decoder["vocab_size"] = dl.dataset.seq.tokenizer.vocab_size
        self.dec = DecoderModel(decoder)
        cnn = dict(cnn)
        self.enc = eval(cnn.pop("proto"))(**cnn)
        self.eval_func = evaluation
        self.split_backward = False      # set by ArenaDDP: detach the features so backward can run in two phases
        self._split = None

    def forward(self, input_ids, attention_mask, images, images_mask=None, encoder_outputs=None,
                encoder_attention_mask=None, epoch=None, iteration=None, **kwargs):
        input_ids = input_ids.cuda()
        attention_mask = attention_mask.cuda()
        arena_of(self).refresh()          # one arena for encoder + decoder parameters
        if encoder_outputs is None:
            encoder_outputs, encoder_attention_mask = self.encode(images, images_mask, **kwargs)
            if self.split_backward and self.training and torch.is_grad_enabled() and encoder_outputs.requires_grad:
                leaf = encoder_outputs.detach().requires_grad_(True)
                self._split = (encoder_outputs, leaf)
                encoder_outputs = leaf
        return self.dec(input_ids=input_ids, attention_mask=attention_mask, encoder_outputs=encoder_outputs,
                        encoder_attention_mask=encoder_attention_mask, **kwargs)

    def encode(self, images, images_mask=None, **kwargs):
        return self.enc.encode(images, images_mask, **kwargs)

    def __repr__(self):
        s = "model: RRG\n"
        s += "(enc):" + str(self.enc) + "\n"
        s += "(dec):" + str(self.dec) + "\n"
        s += "{}\n".format(get_n_params(self))
        return s
