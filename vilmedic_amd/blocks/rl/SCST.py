"""SCST -- self-critical sequence training on the HIP path.  ref: vilmedic/blocks/rl/SCST.py:14-195.

MI355X-first redesign of ``forward_sampling``: the reference un-wraps HF ``generate`` so autograd records L sequential
decode steps (SCST.py:142-157).  Here the multinomial rollout runs WITHOUT grad on the KV-cached decode path, and the
log-probabilities of the sampled tokens are then recomputed WITH grad by ONE teacher-forced batched forward whose
LM-head kernel applies the same logit filters (bad words, top-k) and the policy-gradient weights
``mask * (r_sample - r_greedy) / sum(mask)`` -- the same loss and gradient, L-fold fewer launches."""
import json
import os

import torch
import torch.nn as nn

from ..scorers import REWARD_COMPLIANT


def scst_loss(input, seq, reward_sampling, reward_greedy, scores_weights, pad_token_id):
    """ref: SCST.py:14-45 (kept for API parity; the training path fuses it into the LM-head loss kernel)."""
    input = input.clone()
    input[input == -float("Inf")] = 0.
    mask = (seq > pad_token_id).float()
    input = input.squeeze(-1) * mask
    input = input / torch.sum(mask)
    dev = input.device
    delta_rewards = [torch.tensor(rs, device=dev) - torch.tensor(rg, device=dev) for rs, rg in zip(reward_sampling, reward_greedy)]
    loss = sum(torch.sum(scores_weights[i] * (-input * r.unsqueeze(-1).expand_as(input))) for i, r in enumerate(delta_rewards))
    delta_reward = torch.mean(torch.stack(delta_rewards))
    delta_reward_per_metric = torch.mean(torch.stack(delta_rewards), dim=-1)
    return loss, delta_reward, delta_reward_per_metric


class SCST(nn.Module):
    def __init__(self, decoder, dl, scores, scores_args=None, scores_weights=None, top_k=None, use_nll=False):
        super().__init__()
        dataset = dl.dataset
        if hasattr(dataset, "tokenizer"):
            self.tokenizer, self.max_length = dataset.tokenizer, dataset.tokenizer_max_len
        elif hasattr(dataset, "tgt_tokenizer"):
            self.tokenizer, self.max_length = dataset.tgt_tokenizer, dataset.tgt_tokenizer_max_len
        else:
            raise NotImplementedError("Where is tokenizer in dataset?")
        self.__dict__["decoder"] = decoder            # not registered: the parameters belong to RRG
        self.top_k, self.use_nll = top_k, use_nll
        # rollouts run on the training-precision decode step (the sampled one is stochastic anyway; the reference's are under the
        # trainer's autocast when use_amp is set); VM_SCST_DECODE_DTYPE=fp32 makes them exact-fp32
        self.decode_dtype = os.environ.get("VM_SCST_DECODE_DTYPE", "bf16")
        cfg = decoder.config
        self.bos_token_id, self.eos_token_id, self.pad_token_id = cfg.bos_token_id, cfg.eos_token_id, cfg.pad_token_id
        assert scores is not None
        if not isinstance(scores, (list, tuple)):
            scores = [scores]
        scores = [s if callable(s) else s.lower() for s in scores]
        self.scores = scores
        if len(scores) > 1 or use_nll:
            assert scores_weights is not None, "You need to mention scores_weights"
            assert len(scores_weights) == len(scores) + (1 if use_nll else 0)
            self.scores_weights = list(scores_weights)
        else:
            self.scores_weights = [1.0]
        scores_args = scores_args if scores_args is not None else [None] * len(scores)
        if not isinstance(scores_args, (list, tuple)):
            scores_args = [scores_args]
        self.scores_args = scores_args
        self.scorers, self.scorers_index = [], []
        for index, score in enumerate(scores):
            if callable(score):
                self.scorers.append(score)
                self.scorers_index.append(1)
                continue
            assert score in REWARD_COMPLIANT, "{} not in {}".format(score, list(REWARD_COMPLIANT))
            scorer, scorer_index = REWARD_COMPLIANT[score]
            self.scorers.append(scorer(**scores_args[index]) if scores_args[index] else scorer())
            self.scorers_index.append(scorer_index)

    def forward_greedy(self, input_ids, encoder_hidden_states, encoder_attention_mask):
        assert not torch.is_grad_enabled(), "Please add torch.no_grad() decorator"
        out = self.decoder.generate(input_ids=torch.full((input_ids.shape[0], 1), self.bos_token_id, dtype=torch.long,
                                                         device=encoder_hidden_states.device),
                                    max_length=self.max_length, num_beams=1, return_dict_in_generate=True, decode_dtype=self.decode_dtype,
                                    encoder_hidden_states=encoder_hidden_states.detach(),
                                    encoder_attention_mask=encoder_attention_mask.detach())
        return self.get_reward(out.sequences.detach(), input_ids)

    def forward_sampling(self, input_ids, attention_mask, encoder_hidden_states, encoder_attention_mask, reward_greedy):
        assert torch.is_grad_enabled()
        dev = encoder_hidden_states.device
        banned = [self.pad_token_id, self.bos_token_id]
        with torch.no_grad():
            out = self.decoder.generate(input_ids=torch.full((input_ids.shape[0], 1), self.bos_token_id, dtype=torch.long, device=dev),
                                        max_length=self.max_length, num_beams=1, do_sample=True, top_k=self.top_k,
                                        bad_words_ids=[[b] for b in banned], return_dict_in_generate=True, decode_dtype=self.decode_dtype,
                                        encoder_hidden_states=encoder_hidden_states.detach(),
                                        encoder_attention_mask=encoder_attention_mask.detach())
        return self._policy_gradient(out.sequences, input_ids, attention_mask, encoder_hidden_states, encoder_attention_mask, reward_greedy)

    def forward_rollouts(self, input_ids, attention_mask, greedy_encoder, sampling_encoder, rollouts_only=False):
        """Both rollouts of a step in ONE decode loop (a decode step is launch-latency-bound: 2B rows cost about what B rows cost, so
        the greedy baseline rides along with the sampled rollout): rows [0, B) decode greedily on ``greedy_encoder`` = (features,
        mask) of the eval-mode encoder pass, rows [B, 2B) sample on ``sampling_encoder`` = those of the train-mode pass -- exactly the
        inputs forward_greedy / forward_sampling get (ref:vilmedic/models/rrg/RRG_SCST.py:53-75) -- then the rewards and the
        policy-gradient loss.  -> (forward_sampling's return tuple, reward_greedy)"""
        assert torch.is_grad_enabled() or rollouts_only
        from ...generation import trim_to_last_eos
        (enc_g, mask_g), (enc_s, mask_s) = greedy_encoder, sampling_encoder
        dev, B = enc_s.device, input_ids.shape[0]
        banned = [self.pad_token_id, self.bos_token_id]
        with torch.no_grad():
            out = self.decoder.generate(input_ids=torch.full((2 * B, 1), self.bos_token_id, dtype=torch.long, device=dev),
                                        max_length=self.max_length, num_beams=1, do_sample=True, greedy_rows=B, top_k=self.top_k,
                                        bad_words_ids=[[b] for b in banned], return_dict_in_generate=True, decode_dtype=self.decode_dtype,
                                        encoder_hidden_states=torch.cat([enc_g.detach(), enc_s.detach()]),
                                        encoder_attention_mask=torch.cat([mask_g.detach(), mask_s.detach()]))
        greedy = trim_to_last_eos(out.sequences[:B], self.eos_token_id)
        sampled = trim_to_last_eos(out.sequences[B:], self.eos_token_id)
        if rollouts_only:
            return greedy, sampled
        reward_greedy, _, refs = self.get_reward(greedy.detach(), input_ids)
        return self._policy_gradient(sampled, input_ids, attention_mask, enc_s, mask_s, reward_greedy, ref_list=refs), reward_greedy

    def pg_weights(self, seq, input_ids, reward_greedy, pad_to=None, ref_list=None):
        """rewards of the sampled rollout (host: tokenizer + scorers) -> (seq [B, T], row weights [B, T], bookkeeping).  Row (b, t) predicts
        seq[b, t + 1] with weight mask * (r_sample - r_greedy) / sum(mask) (ref:...SCST.py:14-45).  ``pad_to``: T is padded to this length
        with pad tokens of weight 0 -- the static shape of the graph-captured step; the loss does not change (weight-0 rows contribute 0,
        the causal mask keeps the pad positions out of every weighted row)."""
        dev = seq.device
        sampled_ids = seq[:, 1:].contiguous()
        reward_sampling, hyp_list, _ = self.get_reward(sampled_ids, input_ids, ref_list=ref_list)
        weights = self.scores_weights[-len(self.scorers):]
        delta = [torch.tensor(rs, device=dev, dtype=torch.float32) - torch.tensor(rg, device=dev, dtype=torch.float32)
                 for rs, rg in zip(reward_sampling, reward_greedy)]
        coef = sum(w * d for w, d in zip(weights, delta))                                   # [B]
        mask = (sampled_ids > self.pad_token_id).float()                                    # [B, T-1]
        T = seq.shape[1] if pad_to is None else max(int(pad_to), seq.shape[1])
        row_w = torch.zeros(seq.shape[0], T, dtype=torch.float32, device=dev)
        row_w[:, :seq.shape[1] - 1] = mask * coef[:, None] / mask.sum()
        if T > seq.shape[1]:
            seq = torch.cat([seq, torch.full((seq.shape[0], T - seq.shape[1]), self.pad_token_id, dtype=seq.dtype, device=dev)], 1)
        delta_reward = torch.mean(torch.stack(delta))
        delta_reward_per_metric = torch.mean(torch.stack(delta), dim=-1)
        return seq.contiguous(), row_w.contiguous(), (delta_reward, delta_reward_per_metric, reward_sampling, hyp_list)

    def pg_loss(self, seq, row_w, encoder_hidden_states, encoder_attention_mask):
        """the teacher-forced pass over the sampled rollout: sum of row_w * (-log p(seq[b, t + 1])) under the bad-word + top-k filtered
        distribution the tokens were drawn from (ref:...SCST.py:159-185)"""
        banned = [self.pad_token_id, self.bos_token_id]
        return self.decoder(input_ids=seq, attention_mask=None, encoder_hidden_states=encoder_hidden_states,
                            encoder_attention_mask=encoder_attention_mask, labels=seq, return_logits=False,
                            row_weight=row_w, banned=banned, top_k=self.top_k)["loss"]

    def _policy_gradient(self, seq, input_ids, attention_mask, encoder_hidden_states, encoder_attention_mask, reward_greedy, ref_list=None):
        """seq [B, T] with bos at 0 = the sampled rollout -> SCST loss through one teacher-forced pass (ref:...SCST.py:14-45,159-185)"""
        dev = encoder_hidden_states.device
        nll_loss = None
        if self.use_nll:
            nll_loss = self.decoder(input_ids=input_ids.to(dev), attention_mask=attention_mask.to(dev),
                                    encoder_hidden_states=encoder_hidden_states, encoder_attention_mask=encoder_attention_mask,
                                    labels=input_ids.to(dev), return_logits=False)["loss"]
        seq, row_w, (delta_reward, delta_reward_per_metric, reward_sampling, hyp_list) = self.pg_weights(seq, input_ids, reward_greedy, ref_list=ref_list)
        loss = self.pg_loss(seq, row_w, encoder_hidden_states, encoder_attention_mask)
        if self.use_nll:
            loss = loss + self.scores_weights[0] * nll_loss
        return loss, delta_reward, delta_reward_per_metric, reward_sampling, hyp_list

    def _decode_texts(self, ids):
        """token ids [B, T] -> texts: ONE device-to-host copy for the batch (iterating a device tensor row by row synchronises once per
        row), the tokenizer's batch decode when it has one"""
        rows = ids.tolist() if isinstance(ids, torch.Tensor) else [list(r) for r in ids]
        kw = dict(skip_special_tokens=True, clean_up_tokenization_spaces=False)
        batch = getattr(self.tokenizer, "batch_decode", None)
        return list(batch(rows, **kw)) if batch is not None else [self.tokenizer.decode(r, **kw) for r in rows]

    def get_reward(self, rollout_input_ids, input_ids, ref_list=None):
        """``ref_list``: the decoded references when the caller already has them (the greedy and the sampled rollout of a step score against
        the same ones)"""
        hyp_list = self._decode_texts(rollout_input_ids)
        if ref_list is None:
            ref_list = self._decode_texts(input_ids)
        reward = [scorer(ref_list, hyp_list)[idx] for scorer, idx in zip(self.scorers, self.scorers_index)]
        return reward, hyp_list, ref_list

    def __repr__(self):
        return "SCST\n" + json.dumps({"Scores": str(self.scores), "scores_args": str(self.scores_args),
                                      "scores_weights": str(self.scores_weights), "Generate": {"top_k": self.top_k}}, indent=4)
