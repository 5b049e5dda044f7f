"""Fused Adam/AdamW over the parameter arena (one HIP launch per step; also refreshes the bf16 shadows)."""
import torch

from ._lib import check, lib, ptr, stream
from .arena import arena_of


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW arithmetic (``decoupled_weight_decay`` selects AdamW) executed by ``vm_adam_step`` on
    the flat arena.  Accepts a model (preferred) -- the arena is rooted there."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled_weight_decay=False):
        self.arena = arena_of(model)
        params = [p for p, _ in self.arena._layout]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      decoupled_weight_decay=decoupled_weight_decay))
        a = self.arena
        self.m = torch.zeros_like(a.flat)
        self.v = torch.zeros_like(a.flat)
        self.steps = 0
        self.grad_scale = 1.0
        # per-step scalars mirrored in device memory, so that step() can be a node of a captured HIP graph (graph.py): the 1-based
        # step count (bias corrections are computed in the kernel), the learning rate, and an optional gate (a device scalar, normally
        # the loss: the update is skipped while it is NaN / Inf -- the reference's NaN guard without a host read)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=a.flat.device)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=a.flat.device)
        self._lr_host = float(lr)
        self.gate = None
        # set by ArenaDDP (attach_optimizer) after a reducing backward: the averaged bf16 wire buffer the next step() reads INSTEAD of the
        # fp32 gradient arena (consumed once)
        self.grad_wire = None
        # frozen parameters: their gradient slices stay zero and m=v=0 -> update is exactly 0 (no weight decay applied
        # would still move them, so decay is rejected when something is frozen)
        if weight_decay and any(not p.requires_grad for p in params):
            raise ValueError("FusedAdam: weight_decay with frozen parameters is not supported")

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        from . import ops
        ops.join_side()              # queued / side-stream parameter gradients land before the update (no-op when nothing is pending)
        g = self.param_groups[0]
        self.steps += 1
        b1, b2 = g["betas"]
        a = self.arena
        if float(g["lr"]) != self._lr_host:          # a scheduler moved the learning rate: refresh the device copy (outside any capture)
            self.sync_lr()
        if self.gate is not None:                    # a skipped (NaN / Inf) step does not count, as in the reference's loop (optimizer.step() not called)
            self.step_dev.add_(torch.isfinite(self.gate).reshape(-1)[:1].to(torch.int64))
        else:
            self.step_dev.add_(1)
        wire, self.grad_wire = self.grad_wire, None
        fn, grads = (lib().vm_adam_step_dev, a.gflat) if wire is None else (lib().vm_adam_step_wire, wire)
        check(fn(ptr(a.flat), ptr(grads), ptr(self.m), ptr(self.v), ptr(a.shadow_flat), a.numel,
                 g["lr"], b1, b2, g["eps"], g["weight_decay"], int(g["decoupled_weight_decay"]),
                 1 - b1 ** self.steps, 1 - b2 ** self.steps, self.grad_scale,
                 ptr(self.lr_dev), ptr(self.step_dev), ptr(self.gate) if self.gate is not None else None, stream()),
              "vm_adam_step")
        a.mark_shadow_fresh()

    def sync_lr(self):
        self._lr_host = float(self.param_groups[0]["lr"])
        self.lr_dev.fill_(self._lr_host)

    def state_dict(self):
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        # the step count that the bias corrections use is the DEVICE counter: gated (NaN / Inf) steps and skipped graph replays do not
        # advance it, while the host-side ``steps`` counts calls.  One host read at checkpoint time.
        self.steps = int(self.step_dev.item())
        return {"m": self.m, "v": self.v, "steps": self.steps, "param_groups": groups}      # (step_dev / lr_dev are rebuilt from these)

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.steps = sd["steps"]
        for g, saved in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in saved.items() if k != "params"})
        self.step_dev.fill_(int(self.steps))
        self.sync_lr()
