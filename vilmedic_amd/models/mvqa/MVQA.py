"""MVQA -- ref: vilmedic/models/mvqa/MVQA.py:14-59 (CNN features -> Linear+LayerNorm adapter -> bidirectional BERT
stack without embeddings -> tanh pooler -> classifier -> loss)."""
import torch
import torch.nn as nn

from ...arena import arena_of
from ...blocks.classifier import *  # noqa: F401,F403
from ...blocks.classifier.evaluation import evaluation
from ...blocks.losses import *  # noqa: F401,F403
from ...blocks.vision import *  # noqa: F401,F403
from ...nn import BERT_GEN_DEFAULTS, BertPooler, BertStack, make_config
from ..utils import get_n_params


class MVQA(nn.Module):
    def __init__(self, cnn, classifier, adapter, transformer, loss, **kwargs):
        super().__init__()
        cnn, loss, classifier, adapter = dict(cnn), dict(loss), dict(classifier), dict(adapter)
        cnn_func, loss_func, classifier_func = cnn.pop("proto"), loss.pop("proto"), classifier.pop("proto")
        self.cnn = eval(cnn_func)(**cnn)
        cfg = make_config(BERT_GEN_DEFAULTS, transformer)
        self.adapter = nn.Sequential(nn.Linear(adapter.pop("input_size"), adapter.pop("output_size")),
                                     nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps))
        self.transformer = BertStack(cfg)            # state-dict keys ``transformer.layer.{i}...`` as HF BertEncoder
        self.pooler = BertPooler(cfg)
        self.classifier = eval(classifier_func)(**classifier)
        self.loss_func = eval(loss_func)(**loss)
        self.eval_func = evaluation

    def forward(self, images, labels=None, from_training=True, iteration=None, epoch=None, **kwargs):
        arena = arena_of(self)
        arena.refresh()
        out = self.cnn(images.cuda())                                   # [B, 49, 1664] (MIOpen)
        out = self.adapter(out.float()).to(torch.bfloat16).contiguous()  # [B, 49, 768]
        out = self.transformer(out, arena)                              # no mask, no embeddings (MVQA.py:43)
        out = self.pooler(out, arena)                                   # fp32 [B, 768]
        out = self.classifier(out)
        loss = torch.tensor(0.)
        if from_training:
            loss = self.loss_func(out, labels.cuda(), **kwargs)
        return {"loss": loss, "output": out, "answer": torch.argmax(out, dim=-1)}

    def __repr__(self):
        return super().__repr__() + "\n{}\n".format(get_n_params(self))
