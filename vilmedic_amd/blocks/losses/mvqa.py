"""LabelSmoothingCrossEntropy on the HIP path -- ref: vilmedic/blocks/losses/mvqa/LabelSmoothingCrossEntropyLoss.py:31-48."""
import torch
import torch.nn as nn

from ..._lib import check, lib, ptr, stream


class _SmoothCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, smoothing, reduction):
        R, Cc = logits.shape
        x = logits.detach().float().contiguous()
        loss_sum = torch.zeros(1, dtype=torch.float32, device=x.device)
        dl = torch.empty_like(x)
        scale = 1.0 / R if reduction == "mean" else 1.0
        check(lib().vm_ce_smooth_fwd_bwd(ptr(x), ptr(target.contiguous()), R, Cc, smoothing, ptr(loss_sum), ptr(dl), scale, stream()),
              "vm_ce_smooth_fwd_bwd")
        ctx.save_for_backward(dl)
        return (loss_sum * scale).squeeze(0)

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None


class LabelSmoothingCrossEntropy(nn.Module):
    def __init__(self, smoothing=0.1, reduction="mean", **kwargs):
        super().__init__()
        if reduction not in ("mean", "sum"):
            raise NotImplementedError("reduction must be 'mean' or 'sum'")
        self.smoothing = smoothing
        self.reduction = reduction

    def forward(self, output, target):
        return _SmoothCEFn.apply(output, target, self.smoothing, self.reduction)
