"""The training step as ONE captured HIP graph (MI355X-first: HIP streams and graphs instead of a tracing compiler).

``GraphedTrainStep(step_fn, example_batch)`` runs ``step_fn(**batch)`` -- forward, autograd backward, gradient all-reduce if any,
optimizer step -- eagerly a few times (lazy kernel attributes, allocator, workspaces), captures one more execution with
``torch.cuda.graph`` and from then on REPLAYS it: the host enqueues one graph launch per step instead of ~800 kernel launches.
What makes the captured step correct on replay (everything that used to be a launch-time scalar now lives in device memory):
  * dropout masks: launch-time seeds are frozen into the graph; the kernels add the device counter ``ops.seed_dev`` that the captured
    step itself advances, so every replay draws fresh masks and forward / backward of one replay agree (vm_gemm_epilogue.dropout_seed_dev);
  * Adam: step count, bias corrections and learning rate are read from device scalars (vm_adam_step_dev); ``optimizer.sync_lr()``
    refreshes the rate after a scheduler moved it;
  * NaN / Inf guard (ref: vilmedic/executors/trainor.py:109-112): the optimizer is gated on the loss ON THE DEVICE -- a non-finite
    loss skips the update without a host read;
  * inputs: the batch is copied into static tensors before each replay; the loss comes back in a static scalar.
Measured on one MI355X (profiles/r03_e_schedule_experiments.txt): replay removes the host from the step and keeps the side-stream branch
concurrent (three hardware queues); its rate is within +-1.5 % of eager launches, which is why bench.py reports both modes.
"""
import torch

from . import ops


class GraphedTrainStep:
    def __init__(self, step_fn, example_batch, optimizer=None, warmup=3):
        """``step_fn(**batch) -> loss tensor`` must do the whole step (zero_grad, backward, optimizer.step).  ``example_batch``: dict of
        device tensors (static shapes); ``optimizer``: a FusedAdam (its update is gated on the loss inside the graph)."""
        self.step_fn = step_fn
        self.optimizer = optimizer
        self.static = {k: v.clone() if isinstance(v, torch.Tensor) else v for k, v in example_batch.items()}
        dev = next(v.device for v in self.static.values() if isinstance(v, torch.Tensor))
        self.device = dev
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.graph = None
        self._warm = warmup

    def _one(self):
        loss = self.step_fn(**self.static)
        ops.advance_seed_dev(self.device)
        self.loss.copy_(loss.detach().float())
        return loss

    def __call__(self, **batch):
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                self.static[k].copy_(v, non_blocking=True)
        if self.graph is None:
            if self._warm > 0:
                self._warm -= 1
                self._one()
                return self.loss
            if self.optimizer is not None and hasattr(self.optimizer, "sync_lr"):
                self.optimizer.sync_lr()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            steps0 = getattr(self.optimizer, "steps", None)
            try:
                with ops.capture(g):
                    self._one()
            except BaseException:
                # an aborted capture leaves host-side state behind that points into the dead capture: queued weight-gradient / LayerNorm
                # reduce items (tensors of the aborted graph's pool), a pending side-stream join, gradient buffers already marked clean.
                # Drop all of it, so that the eager iteration that follows starts from a clean slate (its first weight gradients then
                # accumulate into freshly zeroed buffers instead of flushing stale queue entries with overwrite = 1)
                ops.reset_host_state()
                if steps0 is not None:
                    self.optimizer.steps = steps0
                raise
            if steps0 is not None:
                self.optimizer.steps = steps0      # capturing is not a step
            self.graph = g                # (the capture itself executed nothing: this call's step is the replay below)
        if self.optimizer is not None and hasattr(self.optimizer, "_lr_host") and float(self.optimizer.param_groups[0]["lr"]) != self.optimizer._lr_host:
            self.optimizer.sync_lr()
        self.graph.replay()
        if self.optimizer is not None and hasattr(self.optimizer, "steps"):
            self.optimizer.steps += 1      # host mirror of the device step counter (checkpoints)
        return self.loss
