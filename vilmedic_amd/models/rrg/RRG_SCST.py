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
        # data parallelism (ArenaDDP): the same two-phase backward as RRG -- the policy-gradient loss consumes a DETACHED copy of the train-mode
        # image features, so the decoder's gradient range is reduced while the encoder's backward still runs
        self.split_backward = False
        self._split = None

    @property
    def enc(self):
        return self.model.enc

    @property
    def dec(self):
        return self.model.dec

    def forward(self, input_ids, attention_mask, images, images_mask=None, encoder_outputs=None, **kwargs):
        with torch.no_grad():                                   # 1. the greedy baseline's encoder pass (eval mode, as the reference)
            self.model.eval()
            enc_g = self.model.encode(images.cuda(), images_mask, **kwargs)
        self.model.train()                                      # 2. the sampling rollout's encoder pass (train mode, differentiated)
        enc, enc_mask = self.model.encode(images.cuda(), images_mask, **kwargs)
        if self.split_backward and torch.is_grad_enabled() and enc.requires_grad:
            leaf = enc.detach().requires_grad_(True)
            self._split = (enc, leaf)
            enc = leaf
        # 3. both rollouts in one decode loop (greedy rows on the eval features, sampled rows on the train features), rewards, policy
        #    gradient -- the reference's forward_greedy + forward_sampling (RRG_SCST.py:53-75) without a second 128-step decode
        (loss, delta_reward, _, reward_sampling, _), _ = self.scst.forward_rollouts(
            input_ids=input_ids, attention_mask=attention_mask, greedy_encoder=enc_g, sampling_encoder=(enc, enc_mask))
        return {"loss": loss,
                "custom_print": "reward_sampling {}, delta_reward: {}".format(torch.mean(torch.tensor(reward_sampling)),
                                                                              float(delta_reward))}

    # ---- BASELINE configs[4] "HIP-graph-captured step": the differentiated half of the SCST step -- train-mode encoder pass,
    # teacher-forced decoder pass over the sampled rollout, policy-gradient loss, backward, fused Adam -- replayed from ONE captured graph
    # (vilmedic_amd.graph.GraphedTrainStep); the rollouts (data-dependent length, host-side rewards) run in front of it on the fused decode
    # step.  Static shapes: the sampled rollout is padded to max_length with weight-0 rows (SCST.pg_weights).  ref: RRG_SCST.py:59-85.
    def graphed_step(self, optimizer, input_ids, attention_mask, images, images_mask=None, **kwargs):
        """one whole training step (rollouts + captured update); returns what forward() returns, ``loss`` being a static device scalar.
        Falls back to forward() + backward() + optimizer.step() when the step cannot be replayed (use_nll, a different batch shape,
        an encoder whose train-mode features are random)."""
        images = images.cuda()
        key = tuple(images.shape)            # (input_ids only feed the host-side reward: their padded length is not part of the captured graph)
        g = getattr(self, "_graphed", None)
        # the graph recomputes the train-mode encoder pass the rollouts already ran once under no_grad: an encoder with batch statistics
        # (BatchNorm running means) or with random train-mode features (dropout) would be updated twice / disagree with the rollout
        if getattr(self, "_enc_replayable", None) is None:
            self._enc_replayable = not any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in self.model.enc.modules()) and \
                not float(getattr(getattr(self.model.enc, "dropout_out", None), "p", 0.0) or 0.0) > 0
        if self.scst.use_nll or images_mask is not None or not self._enc_replayable or (g is not None and g[0] != key):
            out = self(input_ids=input_ids, attention_mask=attention_mask, images=images, images_mask=images_mask, **kwargs)
            optimizer.zero_grad()
            if hasattr(optimizer, "gate"):
                optimizer.gate = out["loss"].detach()        # (never the static loss of an earlier replay)
            out["loss"].backward()
            optimizer.step()
            return out
        with torch.no_grad():
            self.model.eval()
            enc_g = self.model.encode(images, None)
            self.model.train()
            enc_s = self.model.encode(images, None)              # (= the features the captured pass recomputes: no dropout in this encoder)
            greedy, sampled = self.scst.forward_rollouts(input_ids=input_ids, attention_mask=attention_mask, greedy_encoder=enc_g,
                                                         sampling_encoder=enc_s, rollouts_only=True)
        reward_greedy, _, refs = self.scst.get_reward(greedy.detach(), input_ids)
        seq, row_w, (delta_reward, _, reward_sampling, _) = self.scst.pg_weights(sampled, input_ids, reward_greedy, pad_to=self.scst.max_length, ref_list=refs)
        if g is None:
            from ...graph import GraphedTrainStep

            def pg_step(images, seq, row_w):
                enc, enc_mask = self.model.encode(images, None)
                loss = self.scst.pg_loss(seq, row_w, enc, enc_mask)
                optimizer.zero_grad()
                optimizer.gate = loss.detach()
                loss.backward()
                optimizer.step()
                return loss
            g = self._graphed = (key, GraphedTrainStep(pg_step, dict(images=images, seq=seq, row_w=row_w), optimizer=optimizer, warmup=2))
        loss = g[1](images=images, seq=seq, row_w=row_w)
        return {"loss": loss, "launch_mode": "hip-graph replay" if g[1].graph is not None else "eager (graph warm-up)",
                "custom_print": "reward_sampling {}, delta_reward: {}".format(torch.mean(torch.tensor(reward_sampling)), float(delta_reward))}

    def __repr__(self):
        return "RRG_SCST\n" + str(self.scst) + "\n{}\n".format(get_n_params(self))
