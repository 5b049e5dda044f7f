"""ConVIRT -- ref: vilmedic/models/selfsup/conVIRT.py:13-109."""
import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ...arena import arena_of
from ...blocks.huggingface.encoder.encoder_model import EncoderModel
from ...blocks.losses import ConVIRTLoss, InfoNCELoss  # noqa: F401  (eval(proto) namespace)
from ...blocks.vision import *  # noqa: F401,F403
from ...blocks.vision.micro_bn import micro_batches, use_micro_batch_norm
from ..utils import get_n_params


def evaluation(models, config, dl, from_training, **kwargs):
    model = models[0]
    losses, linguistics, visuals = [], [], []
    for batch in dl:
        batch = {k: v.cuda() if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
        out = model(**batch)
        losses.append(out["loss"].mean().cpu().data.numpy())
        if not from_training:
            linguistics.append(out["linguistic"].cpu().data)
            visuals.append(out["visual"].cpu().data)
    if from_training:
        return {"loss": np.ndarray.mean(np.array(losses))}
    return {"loss": np.ndarray.mean(np.array(losses)), "linguistic": torch.cat(linguistics), "visual": torch.cat(visuals)}


class ProjectionMLP(nn.Sequential):
    """Linear -> ReLU -> Linear (conVIRT.py:58-67), state-dict keys ``0.weight / 0.bias / 2.weight / 2.bias`` as the reference's
    nn.Sequential; both products run on the bf16 MFMA GEMM (parameters in the model's arena, weight / bias gradients through the
    grouped weight-gradient launch), the embedding leaves the second GEMM in fp32 straight from the accumulators."""

    def __init__(self, d_in, d_out):
        super().__init__(nn.Linear(d_in, d_out), nn.ReLU(), nn.Linear(d_out, d_out))

    def forward(self, x, arena):
        l0, l2 = self[0], self[2]
        x = x.to(torch.bfloat16).contiguous()
        h = ops.linear(x, arena.shadow(l0.weight), l0.bias, wgrad_buf=arena.grad(l0.weight), bgrad_buf=arena.grad(l0.bias), anchor=l0.weight)
        h = torch.relu(h)
        return ops.linear(h, arena.shadow(l2.weight), l2.bias, wgrad_buf=arena.grad(l2.weight), bgrad_buf=arena.grad(l2.bias),
                          anchor=l2.weight, out_f32=True)


class ConVIRT(nn.Module):
    def __init__(self, encoder, cnn, projection, loss, forward_batch_size=256, **kwargs):
        super().__init__()
        self.linguistic = EncoderModel(encoder)
        cnn = dict(cnn)
        self.visual = eval(cnn.pop("proto"))(**cnn)
        projection = dict(projection)
        pd = projection["projection_dim"]
        self.vis_proj = ProjectionMLP(projection["visual_embedding_dim"], pd)
        self.lin_proj = ProjectionMLP(projection["textual_embedding_dim"], pd)
        loss = dict(loss)
        self.loss_fn = eval(loss.pop("proto"))(**loss)
        self.fbs = forward_batch_size          # micro-batch of the reference's tower loop; only BatchNorm statistics depend on it
        self._visual_has_bn = any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in self.visual.modules())
        use_micro_batch_norm(self.visual)       # BatchNorm layers that can take their statistics per micro-batch at full batch
        self.eval_func = evaluation

    def forward(self, input_ids, attention_mask, images, **kwargs):
        images, input_ids, attention_mask = images.cuda(), input_ids.cuda(), attention_mask.cuda()
        arena = arena_of(self)
        arena.refresh()
        # The reference runs both towers in forward_batch_size micro-batches inside ONE autograd graph (conVIRT.py:83-95).  For the
        # text tower (LayerNorm only) that equals a single pass over the batch, which is what runs here.  A CNN image tower in
        # training mode normalises with the statistics of each MICRO-batch: it also runs ONCE over the batch, with its BatchNorm
        # layers grouping the statistics per micro-batch (blocks/vision/micro_bn.py: same values, 1 pass instead of batch / fbs).
        text = self.linguistic(input_ids=input_ids, attention_mask=attention_mask)
        linguistics = self.lin_proj(text["pooler_output"], arena)
        bs = images.shape[0]
        with micro_batches(self.fbs if (self.training and self._visual_has_bn and self.fbs < bs) else 0):
            vis = self.visual(images)
        visuals = self.vis_proj(vis if vis.dim() == 2 else vis[:, 0], arena)
        loss, loss_l, loss_v = self.loss_fn(linguistics, visuals)
        return {"loss": loss, "loss_l": loss_l, "loss_v": loss_v, "linguistic": linguistics, "visual": visuals}

    def __repr__(self):
        return "ConVIRT\n" + str(self.visual) + "\n" + str(self.linguistic) + "\n" + str(self.loss_fn) + "\n{}\n".format(get_n_params(self))
