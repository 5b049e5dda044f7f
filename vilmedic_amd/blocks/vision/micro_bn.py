"""BatchNorm with MICRO-BATCH statistics at full batch.

The reference's contrastive models run their towers in ``forward_batch_size`` micro-batches inside one autograd graph
(ref: vilmedic/models/selfsup/conVIRT.py:83-95 with forward_batch_size 4 in config/SELFSUP/convirt-mimic.yml:22; GLoRIA.py:92-105), so in
training mode every BatchNorm of the CNN normalises each micro-batch with ITS OWN statistics and updates the running statistics
micro-batch by micro-batch.  Executing that literally costs ``batch / forward_batch_size`` passes over a ~160-kernel CNN (64 passes of
4 images at the reference's ConVIRT setting: the step is launch-bound).  Here the CNN runs ONCE over the whole batch and only the
normalisation is grouped: ``x.view(G, g, C, H, W)`` -> per-(group, channel) mean / biased variance -> normalise, which is what the
G sequential passes compute; the running statistics receive the same G exponential-moving-average updates in closed form
(r_G = (1 - m)^G r_0 + m * sum_c (1 - m)^(G - 1 - c) s_c).  Convolutions, pooling and activations are per-sample, so nothing else
changes.  A trailing partial micro-batch (batch % g) goes through the ordinary batch-norm path.
"""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

_micro = {"size": 0}


@contextlib.contextmanager
def micro_batches(size):
    """inside: every MicroBatchNorm2d in training mode normalises groups of ``size`` consecutive samples separately"""
    old = _micro["size"]
    _micro["size"] = int(size or 0)
    try:
        yield
    finally:
        _micro["size"] = old


class MicroBatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        g = _micro["size"]
        B = x.shape[0]
        if not self.training or g <= 0 or g >= B or not self.track_running_stats:
            return super().forward(x)
        G = B // g
        main, rest = x[:G * g], x[G * g:]
        C = x.shape[1]
        xg = main.reshape(G, g, C, *x.shape[2:])
        red = (1, 3, 4)
        var, mean = torch.var_mean(xg.float(), dim=red, unbiased=False, keepdim=True)        # [G,1,C,1,1]
        y = (xg - mean.to(x.dtype)) * torch.rsqrt(var + self.eps).to(x.dtype)
        if self.affine:
            y = y * self.weight.view(1, 1, C, 1, 1) + self.bias.view(1, 1, C, 1, 1)
        y = y.reshape(main.shape)
        with torch.no_grad():
            mom = self.momentum if self.momentum is not None else 0.1      # (momentum=None = cumulative average is not used by these models)
            n = g * main[0, 0].numel()
            w = mom * (1.0 - mom) ** torch.arange(G - 1, -1, -1, device=x.device, dtype=torch.float32)     # weight of micro-batch c
            keep = (1.0 - mom) ** G
            m_c = mean.view(G, C)
            v_c = var.view(G, C) * (n / max(n - 1, 1))                     # running_var takes the unbiased estimate
            self.running_mean.mul_(keep).add_((w[:, None] * m_c).sum(0).to(self.running_mean.dtype))
            self.running_var.mul_(keep).add_((w[:, None] * v_c).sum(0).to(self.running_var.dtype))
            self.num_batches_tracked += G
        if rest.shape[0]:
            y = torch.cat([y, super().forward(rest)])
        return y


def use_micro_batch_norm(module):
    """swap every nn.BatchNorm2d of ``module`` for a MicroBatchNorm2d sharing its parameters / buffers (state-dict keys unchanged)"""
    for name, child in list(module.named_children()):
        if type(child) is nn.BatchNorm2d:
            new = MicroBatchNorm2d(child.num_features, eps=child.eps, momentum=child.momentum, affine=child.affine,
                                   track_running_stats=child.track_running_stats)
            new.weight, new.bias = child.weight, child.bias
            new.running_mean, new.running_var, new.num_batches_tracked = child.running_mean, child.running_var, child.num_batches_tracked
            new.train(child.training)
            setattr(module, name, new)
        else:
            use_micro_batch_norm(child)
    return module
