"""LR schedulers the reference adds to torch's (ref: vilmedic/blocks/schedulers/*.py), nameable from YAML (``trainor.lr_decay``).

* LinearWarmupCosineAnnealingLR -- linear ramp from ``warmup_start_lr`` to the base rate over ``warmup_epochs`` steps, then half a
  cosine down to ``eta_min`` at ``max_epochs`` (after lightning-bolts 0.5.0).  Implemented in closed form; the reference's chainable
  recurrence produces the same sequence (fixture G17), including its mirror-image continuation past ``max_epochs``.
* DecreasingCosineAnnealingWarmRestarts -- torch's CosineAnnealingWarmRestarts whose current rates are multiplied by ``factor``
  (floored at ``min_lr``) at every step taken while the count of completed restarts is in ``epochs`` (the reference class cannot
  be constructed on current torch -- it reads an attribute its base constructor's initial step() needs before assigning it --
  so this one is defined by its known-answer test, not by a fixture).
"""
import math

from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts, LRScheduler


class LinearWarmupCosineAnnealingLR(LRScheduler):
    def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=0.0, last_epoch=-1):
        self.warmup_epochs, self.max_epochs = warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        super().__init__(optimizer, last_epoch)

    def _at(self, base_lr, e):
        if e < self.warmup_epochs:
            return self.warmup_start_lr + e * (base_lr - self.warmup_start_lr) / max(1, self.warmup_epochs - 1)
        span = self.max_epochs - self.warmup_epochs
        return self.eta_min + 0.5 * (base_lr - self.eta_min) * (1 + math.cos(math.pi * (e - self.warmup_epochs) / span))

    def get_lr(self):
        return [self._at(b, self.last_epoch) for b in self.base_lrs]


class DecreasingCosineAnnealingWarmRestarts(CosineAnnealingWarmRestarts):
    def __init__(self, factor, epochs, min_lr=0, eps=1e-8, **kwargs):
        # set before the base constructor runs: it takes the initial step() itself (the reference assigns these afterwards and
        # fails with AttributeError on current torch)
        self.factor, self.epochs, self.eps, self.min_lr = factor, epochs, eps, min_lr
        self.current_epoch = -1                      # the constructor's own step() is restart 0
        super().__init__(**kwargs)

    def step(self, epoch=None):
        super().step(epoch)
        if self.T_cur == 0:
            self.current_epoch += 1
        if self.current_epoch in self.epochs:
            for i, group in enumerate(self.optimizer.param_groups):
                old = float(group["lr"])
                new = max(old * self.factor, self.min_lr)
                if old - new > self.eps:
                    group["lr"] = new


def linear_warmup_decay(warmup_steps, total_steps, cosine=True, linear=False):
    """multiplier for LambdaLR: linear warm-up, then cosine / linear decay to 0 at ``total_steps`` (or constant)"""
    assert not (linear and cosine)

    def fn(step):
        if step < warmup_steps:
            return float(step) / float(max(1, warmup_steps))
        if not (cosine or linear):
            return 1.0
        progress = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
        return 0.5 * (1.0 + math.cos(math.pi * progress)) if cosine else 1.0 - progress
    return fn
