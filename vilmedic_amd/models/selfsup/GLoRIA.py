"""GLoRIA -- ref: vilmedic/models/selfsup/GLoRIA.py:46-181 (global + local image-report contrastive pre-training).

Same constructor / forward contract and parameter names (``linguistic.*``, ``visual.*``, ``global_embedder.*``,
``local_embedder.weight`` [D, C, 1, 1]).  Differences that are deliberate:
  * the word-piece merge (``aggregate_tokens``, GLoRIA.py:123-177 -- a Python triple loop with one ``.item()`` device
    sync per token) runs as ONE segment-sum on the device: the ids are copied to the host once per batch, the
    token -> word map is built there, and ``index_add_`` does the sums (differentiable);
  * the 1x1-conv local embedder and the global embedder run on the bf16 MFMA GEMM of the HIP path;
  * the reference reads ``self.visual.cnn[6]`` -- an attribute its VisualEncoder does not have at this snapshot
    (it is ``.model``, visual_encoder.py:98); the hook is placed on ``self.visual.model[6]`` (ResNet layer3)."""
import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ...arena import arena_of
from ...blocks.huggingface.encoder.encoder_model import EncoderModel
from ...blocks.losses import GLoRIALoss, cosine_similarity, gloria_attention_fn  # noqa: F401
from ...blocks.vision import *  # noqa: F401,F403
from ...blocks.vision.micro_bn import micro_batches, use_micro_batch_norm
from ...nn import Affine
from ..utils import get_n_params


def evaluation(models, config, dl, from_training, **kwargs):
    """ref: GLoRIA.py:14-37."""
    model = models[0]
    losses, linguistics, visuals = [], [], []
    for batch in dl:
        batch = {k: v.cuda() if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
        out = model(**batch)
        losses.append(out["loss"].mean().cpu().data.numpy())
        if not from_training:
            linguistics.append(out["sent_embeddings"].cpu().data)
            visuals.append(out["global_features"].cpu().data)
    if from_training:
        return {"loss": np.ndarray.mean(np.array(losses))}
    return {"loss": np.ndarray.mean(np.array(losses)), "linguistic": torch.cat(linguistics), "visual": torch.cat(visuals)}


def chunks(lst, n):
    for i in range(0, len(lst), n):
        yield lst[i:i + n]


def word_segments(words):
    """token strings of ONE caption -> (word index of every token or -1 if the token is dropped, merged word strings).
    Follows the reference loop exactly (GLoRIA.py:139-166): a token not starting with ``##`` opens a new word and
    closes the previous one; ``##`` pieces join the open word; ``[SEP]`` closes the open word, becomes a word itself and
    ends the caption (everything after it is dropped); a caption without ``[SEP]`` loses its last open word."""
    seg = [-1] * len(words)
    out, bank, bank_tokens = [], [], []
    for t, word in enumerate(words):
        if word == "[SEP]":
            for u in bank_tokens:
                seg[u] = len(out)
            out.append("".join(bank))
            seg[t] = len(out)
            out.append(word)
            break
        if not word.startswith("##"):
            if bank:
                for u in bank_tokens:
                    seg[u] = len(out)
                out.append("".join(bank))
                bank, bank_tokens = [], []
            bank.append(word)
            bank_tokens.append(t)
        else:
            bank.append(word[2:])
            bank_tokens.append(t)
    return seg, out


class GLoRIA(nn.Module):
    def __init__(self, encoder, cnn, visual_embedder, loss, dl=None, forward_batch_size=12, **kwargs):
        super().__init__()
        encoder = dict(encoder)
        self.last_n_layers = int(encoder.get("last_n_layers", 4))
        self.linguistic = EncoderModel(encoder)
        vocab = dl.dataset.tokenizer.get_vocab() if dl is not None else {}
        self.idxtoword = {v: k for k, v in vocab.items()}
        cnn = dict(cnn)
        self.visual = eval(cnn.pop("proto"))(**cnn)
        ve = dict(visual_embedder)
        hidden = self.linguistic.encoder.config.hidden_size
        self.global_embedder = Affine(hidden, ve["feature_dim"], std=(1.0 / ve["feature_dim"]) ** 0.5)
        self.local_embedder = Affine(hidden, ve["interm_feature_dim"], 1, 1, std=(1.0 / ve["interm_feature_dim"]) ** 0.5, bias=False)
        self.up_sample = nn.Upsample(size=(299, 299), mode="bilinear", align_corners=True)
        self.loss_fn = GLoRIALoss(**dict(loss))
        self.activation = {}

        def hook(module, inp, out):
            self.activation["local_features"] = out
        self.visual.model[6].register_forward_hook(hook)          # output of layer3 (GLoRIA.py:77)
        use_micro_batch_norm(self.visual)
        self.eval_func = evaluation
        self.fbs = forward_batch_size

    # ------------------------------------------------------------------ embedders on the HIP GEMM
    def _embed_global(self, feats, arena):
        g = self.global_embedder
        return ops.linear(feats.to(torch.bfloat16).contiguous(), arena.shadow(g.weight), g.bias, wgrad_buf=arena.grad(g.weight),
                          bgrad_buf=arena.grad(g.bias), anchor=g.weight).float()

    def _embed_local(self, fmap, arena):
        """1x1 convolution, bias-free == a GEMM over the channel dim: [b,C,h,w] -> [b,D,h,w]"""
        b, C, h, w = fmap.shape
        le = self.local_embedder
        x = fmap.permute(0, 2, 3, 1).reshape(b * h * w, C).to(torch.bfloat16).contiguous()
        y = ops.linear(x, arena.shadow(le.weight).view(-1, C), None, wgrad_buf=arena.grad(le.weight).view(-1, C), anchor=le.weight)
        return y.view(b, h, w, -1).permute(0, 3, 1, 2).float()

    def forward(self, input_ids, attention_mask, images, **kwargs):
        arena = arena_of(self)
        arena.refresh()
        bs = images.shape[0]
        # The reference chunks both towers by forward_batch_size (GLoRIA.py:92-105); only the CNN's BatchNorm statistics depend on
        # that.  Here every tower runs ONCE over the batch and the BatchNorm layers group their statistics per micro-batch
        # (blocks/vision/micro_bn.py) -- the same values from batch / fbs times fewer launches.
        images_ = images.cuda()
        with micro_batches(self.fbs if (self.training and self.fbs < bs) else 0):
            global_features = self._embed_global(self.visual(self.up_sample(images_)), arena)
        local_features = self._embed_local(self.activation["local_features"], arena)
        out = self.linguistic(input_ids.cuda(), attention_mask.cuda(), output_hidden_states=True)
        hidden_states = torch.stack([h.float() for h in out["hidden_states"]])
        embeddings, sents = self.aggregate_tokens(hidden_states[-self.last_n_layers:], input_ids)
        sent_embeddings = torch.sum(torch.mean(embeddings, dim=2), dim=1)
        word_embeddings = torch.sum(embeddings, dim=1).permute(0, 2, 1)
        loss, _ = self.loss_fn(global_features, local_features, word_embeddings, sent_embeddings, sents)
        return {"loss": loss, "global_features": global_features, "local_features": local_features,
                "word_embeddings": word_embeddings, "sent_embeddings": sent_embeddings}

    def aggregate_tokens(self, embeddings, input_ids):
        """embeddings [layers, B, L, D], input_ids [B, L] -> ([B, layers, L, D] word-level sums zero-padded to L words,
        list of word lists padded with "[PAD]")  -- ref GLoRIA.py:123-177."""
        nl, B, L, D = embeddings.shape
        ids = input_ids.detach().cpu().tolist()                   # ONE device->host copy per batch
        seg = torch.full((B, L), -1, dtype=torch.long)
        sentences = []
        for b, row in enumerate(ids):
            words = [self.idxtoword.get(t, str(t)) for t in row]
            s, merged = word_segments(words)
            seg[b] = torch.tensor(s, dtype=torch.long)
            sentences.append(merged + ["[PAD]"] * (L - len(merged)))
        seg = seg.to(embeddings.device)
        keep = (seg >= 0).reshape(-1)
        dst = (torch.arange(B, device=seg.device).unsqueeze(1) * L + seg.clamp(min=0)).reshape(-1)[keep]
        src = embeddings.permute(1, 2, 0, 3).reshape(B * L, nl * D)[keep]
        out = torch.zeros(B * L, nl * D, dtype=embeddings.dtype, device=embeddings.device).index_add_(0, dst, src)
        return out.view(B, L, nl, D).permute(0, 2, 1, 3), sentences

    def __repr__(self):
        return "GLoRIA\n" + str(self.visual) + "\n" + str(self.linguistic) + "\n{}\n".format(get_n_params(self))
