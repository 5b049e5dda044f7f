"""RRG_SCST -- ref: vilmedic/models/rrg/RRG_SCST.py:13-91 (exported here; the reference leaves it commented out)."""
import copy
import glob
import os

import torch
import torch.nn as nn

from ...blocks.huggingface.decoder.evaluation import evaluation as evaluation_
from ...blocks.rl.SCST import SCST
from ..utils import get_n_params
from .RRG import RRG


def evaluation(models, config, dl, **kwargs):
    models = [m.model if not isinstance(m, nn.DataParallel) else m.module.model for m in models]
    return evaluation_(models, config, dl, **kwargs)


def load_checkpoint(ckpt):
    from ...executors.utils import vilmedic_state_dict_versioning
    if os.path.isdir(ckpt):
        found = glob.glob(os.path.join(ckpt, "*.pth"))
        assert len(found) == 1
        ckpt = found[0]
    state_dict = torch.load(ckpt, map_location="cpu")
    return vilmedic_state_dict_versioning(state_dict["model"], state_dict.get("__version__", None))


class RRG_SCST(nn.Module):
    def __init__(self, decoder, cnn, dl, scores="ROUGEL", ckpt=None, scores_args=None, scores_weights=None, top_k=None,
                 use_nll=False, **kwargs):
        super().__init__()
        self.model = RRG(copy.deepcopy(dict(decoder)), copy.deepcopy(dict(cnn)), dl=dl, **kwargs)
        if ckpt:
            self.model.load_state_dict(load_checkpoint(ckpt), strict=True)
        self.scst = SCST(decoder=self.model.dec.decoder, dl=dl, scores=scores, scores_args=scores_args,
                         scores_weights=scores_weights, use_nll=use_nll, top_k=top_k)
        self.eval_func = evaluation

    def forward(self, input_ids, attention_mask, images, images_mask=None, encoder_outputs=None, **kwargs):
        with torch.no_grad():                                   # 1. the greedy baseline's encoder pass (eval mode, as the reference)
            self.model.eval()
            enc_g = self.model.encode(images.cuda(), images_mask, **kwargs)
        self.model.train()                                      # 2. the sampling rollout's encoder pass (train mode, differentiated)
        enc, enc_mask = self.model.encode(images.cuda(), images_mask, **kwargs)
        # 3. both rollouts in one decode loop (greedy rows on the eval features, sampled rows on the train features), rewards, policy
        #    gradient -- the reference's forward_greedy + forward_sampling (RRG_SCST.py:53-75) without a second 128-step decode
        (loss, delta_reward, _, reward_sampling, _), _ = self.scst.forward_rollouts(
            input_ids=input_ids, attention_mask=attention_mask, greedy_encoder=enc_g, sampling_encoder=(enc, enc_mask))
        return {"loss": loss,
                "custom_print": "reward_sampling {}, delta_reward: {}".format(torch.mean(torch.tensor(reward_sampling)),
                                                                              float(delta_reward))}

    def __repr__(self):
        return "RRG_SCST\n" + str(self.scst) + "\n{}\n".format(get_n_params(self))
