"""Trainor -- ref: vilmedic/executors/trainor.py:14-203 and trainor_accelerate.py:24-156 (merged: one process per GPU).

Loop semantics kept: forward -> NaN/Inf guard -> backward (loss / grad_accu) -> every grad_accu iterations optional
clip_grad_norm_, optimizer step, zero_grad, scheduler iteration step; end of epoch: epoch_step, evaluation, early-stop
score = mean of ``early_stop_metric`` over splits, eval_step, one-best checkpoint
{model, training_scheduler, optimizer, config, __version__}.
Differences (MI355X-first): bf16 activations with fp32 master weights instead of fp16 autocast + GradScaler; data
parallelism = ArenaDDP over RCCL (flat-gradient all-reduce), the NaN-skip decision is taken COLLECTIVELY (the
reference's per-rank skip would desynchronise DDP collectives); no stray ``break`` after the first
iteration (reference defect, trainor_accelerate.py:155)."""
import os

import numpy as np
import torch

from .. import ops
from ..arena import arena_of
from ..config import to_container
from .utils import (CheckpointSaver, __version__, create_data_loader, create_model, create_optimizer,
                    create_training_scheduler, get_logger)
from .validator import Validator


class Trainor(object):
    def __init__(self, config, seed, logger=None):
        self.config, self.seed = config, seed
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(self.local_rank)
        self.dist = None
        self.logger = logger or get_logger()
        if self.rank != 0:
            self.logger.setLevel("WARNING")
        torch.manual_seed(seed + self.rank)
        np.random.seed(seed + self.rank)
        ops.manual_seed(seed + self.rank)
        if config.get("use_amp"):             # the reference's fp16-autocast switch (trainor.py:49,104): here the transformer path is bf16 anyway,
            from ..blocks.vision import visual_encoder      # and the MIOpen-backed CNN towers run bf16 channels-last convolutions under autocast
            visual_encoder.CNN_AMP = True
        self.state = None
        if config.get("ckpt") is not None:
            self.state = torch.load(config.ckpt, map_location="cpu")
        self.ckpt_dir = config.get("ckpt_dir") or "ckpt"
        os.makedirs(self.ckpt_dir, exist_ok=True)
        self.dl = create_data_loader(config, "train", self.logger, rank=self.rank, world=self.world)
        self.model = create_model(config, self.dl, self.logger, from_training=True, state_dict=self.state)
        self.optimizer = create_optimizer(config, self.logger, self.model, state_dict=self.state)
        # The RCCL communicator is created AFTER the model, its arena and the optimizer state exist: device memory allocated
        # after init_process_group was measurably slower on this stack (measured in round 1: +5.6 ms per RRG step).
        self.ddp = None
        from ..parallel import force_collectives
        if self.world > 1 or force_collectives():       # (VM_FORCE_DDP: the RCCL path with a 1-rank group, tests/test_ddp_nccl_gpu.py)
            import torch.distributed as dist
            from ..arena import arena_of
            from ..parallel import ArenaDDP
            from ..parallel import default_bf16_wire
            # trainor.ddp_wire: "fp32" (default: the exact mean, all-reduced in place on the gradient arena) or "bf16" (half the bytes over
            # xGMI, one rounding per gradient; also VM_DDP_WIRE=bf16)
            bf16_wire = (str(config.get("ddp_wire")) == "bf16") if config.get("ddp_wire") else default_bf16_wire()
            wire = torch.empty(arena_of(self.model).numel, dtype=torch.bfloat16, device=torch.device("cuda", self.local_rank)) if bf16_wire else None
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29534")
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))
            self.dist = dist
            self.ddp = ArenaDDP(self.model, self.dist, wire=wire, bf16_wire=bf16_wire)
        self.training_scheduler = create_training_scheduler(config, self.optimizer, self.logger, state_dict=self.state)
        self.saver = CheckpointSaver(self.ckpt_dir, self.logger, seed, ckpt=config.get("ckpt"))
        self.grad_accu = int(config.get("grad_accu") or 1)
        # device-side NaN guard: fused optimizer and one micro-batch per step (with accumulation a bad micro-batch has to be dropped
        # before it is summed into the others, which needs the host decision)
        self.device_gate = hasattr(self.optimizer, "gate") and self.grad_accu == 1
        self.clip = config.get("clip_grad_norm")
        if self.ddp is not None and self.clip is None:
            self.ddp.attach_optimizer(self.optimizer)      # the optimizer reads the averaged bf16 wire buffer itself (no cast pass back)
        # trainor.graph_step: true -- models that can replay their whole update (rollouts aside) from one captured graph do so
        # (BASELINE configs[4]: RRG + SCST with a HIP-graph-captured step); single process, one micro-batch per step, no clipping
        # (under data parallelism the captured step holds the RCCL collectives too -- ArenaDDP.backward and the MIN all-reduce of the
        # finiteness flag are stream work like everything else)
        graphable = bool(config.get("graph_step")) and self.grad_accu == 1 and self.clip is None and hasattr(self.optimizer, "gate")
        self.graph_step = graphable and self.ddp is None and hasattr(self.model, "graphed_step")
        # ... and every other model has its whole iteration (forward, backward, fused Adam with the device-side NaN gate) captured per batch
        # shape by vilmedic_amd.graph.GraphedTrainStep: the host enqueues one graph launch per iteration, the rate no longer depends on it
        # (a model that brings its own captured step -- RRG_SCST: host-side rewards inside forward -- cannot be captured whole; under data
        # parallelism, where its graphed_step is not used, it runs eagerly instead of wasting two warm-up iterations per batch signature
        # on a capture that must fail)
        self.graph_any = graphable and not self.graph_step and not hasattr(self.model, "graphed_step")
        self._graphs = {}
        self.max_graphs = int(config.get("graph_cache") or 8)     # captured batch signatures kept (each owns its activation pool)
        self.eval_start = int(config.get("eval_start") or 0)
        self.evaluator = Validator(config.validator_view, [self.model], self.dl, seed, True, self.logger, self.rank, self.world) \
            if config.get("validator_view") is not None else None

    def _all_finite(self, loss):
        flag = torch.isfinite(loss.detach()).float().reshape(1)
        if self.dist is not None:
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(flag.item())

    def _gate(self, loss):
        """NaN / Inf guard WITHOUT a host read (the reference tests the loss on the host every iteration, trainor.py:109-112, which
        here would make the host wait for the forward pass before it may enqueue the backward pass): the fused optimizer takes a device
        scalar and skips its update while that scalar is not finite (FusedAdam.gate); under data parallelism the decision is taken
        collectively with a device-side MIN all-reduce of the finiteness flag, so every rank skips the same step."""
        gate = loss.detach().float().reshape(1)
        if self.dist is not None:
            flag = torch.isfinite(gate).float()
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
            gate = torch.where(flag > 0, gate, torch.full_like(gate, float("nan")))
        self.optimizer.gate = gate

    def _zero_grad(self):
        """gradients of EVERY parameter stay inside the arena's flat buffer (that buffer is what ArenaDDP all-reduces): a
        torch.optim optimizer's zero_grad(set_to_none=True) would let autograd allocate fresh .grad tensors outside it for the
        parameters of native torch modules (CNN backbones, adapters), which would then never be averaged across ranks"""
        arena_of(self.model).zero_grad()

    def _optimizer_step(self, epoch, iteration):
        if self.clip is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.optimizer.step()
        self._zero_grad()
        self.training_scheduler.iteration_step(epoch + float(iteration) / max(len(self.dl), 1))      # frac_epoch, trainor.py:125-126,152-153

    def _graphed_iteration(self, batch):
        """one iteration replayed from the HIP graph captured for this batch's tensor shapes (two eager iterations, then the capture); None
        when the batch carries something that is neither a tensor (a static input of the graph) nor a None / scalar constant (part of the graph's
        key) -- the caller then runs it eagerly.  The captured forward is called WITHOUT ``epoch`` / ``iteration`` (the eager path passes them): a
        model whose forward depends on them must not set ``graph_step``."""
        from ..graph import GraphedTrainStep
        tensors = {k: v.cuda() for k, v in batch.items() if isinstance(v, torch.Tensor)}
        consts = {k: v for k, v in batch.items() if not isinstance(v, torch.Tensor)}
        if not all(v is None or isinstance(v, (bool, int, float, str)) for v in consts.values()):
            return None
        key = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(tensors.items())) + tuple(sorted((k, repr(v)) for k, v in consts.items()))
        g = self._graphs.get(key)
        if g is False:                                       # this signature failed to capture once (a forward that reads the device): eager
            return None
        if g is None and len(self._graphs) >= self.max_graphs:
            # every captured signature owns a private pool of activations: variable-length batches must not grow memory without bound.  The
            # least recently replayed graph goes (its pool is released with it); under data parallelism every rank sees the same sequence of
            # signatures only if the loaders pad alike, so there the cache is simply not grown past the cap
            if self.ddp is not None:
                return None
            old = next(iter(self._graphs))
            del self._graphs[old]
        if g is not None:
            self._graphs[key] = self._graphs.pop(key)        # most recently used last
        if g is None:
            def step_fn(**tb):
                loss = self.model(**tb, **consts)["loss"].mean()
                self._zero_grad()
                self._gate(loss)                             # NaN / Inf loss: the update is skipped on the device (trainor.py:109-112), on every rank
                if self.ddp is not None:
                    self.ddp.backward(loss)
                else:
                    loss.backward()
                self.optimizer.step()
                return loss
            g = self._graphs[key] = GraphedTrainStep(step_fn, tensors, optimizer=self.optimizer, warmup=2)
        try:
            return g(**tensors)
        except RuntimeError as e:                            # capture failed (nothing of this batch has run yet): this signature stays eager
            if g.graph is not None:
                raise                                        # a replay error is a real error
            self.logger.warning("graph_step: capture failed for this batch signature ({}); running it eagerly".format(str(e).splitlines()[0]))
            self._graphs[key] = False
            return None

    def start(self):
        cfg = self.config
        early_stop_start = int(cfg.get("early_stop_start") or 0)
        decay_metric_start = int(cfg.get("decay_metric_start") or 0)
        for epoch in range(int(self.training_scheduler.epoch), int(cfg.epochs) + 1):
            self.model.train()
            losses = []
            iteration, pending, out = 0, False, {}
            for iteration, batch in enumerate(self.dl, start=1):
                if self.graph_step:                      # the model replays its own update from a captured HIP graph (RRG_SCST.graphed_step)
                    out = self.model.graphed_step(self.optimizer, **batch, epoch=epoch, iteration=iteration)
                    self.training_scheduler.iteration_step(epoch + float(iteration) / max(len(self.dl), 1))
                    losses.append(out["loss"].detach().clone())
                    continue
                if self.graph_any:
                    loss = self._graphed_iteration(batch)
                    if loss is None:
                        self._zero_grad()                # (a graphed iteration leaves its gradients behind: it zeroes them before its backward)
                    else:
                        self.training_scheduler.iteration_step(epoch + float(iteration) / max(len(self.dl), 1))
                        losses.append(loss.detach().clone())
                        out = {"loss": loss}
                        if iteration % 50 == 0 and self.rank == 0:
                            self.logger.info("Epoch {}, iter {}, lr {:.2e}, loss {:.4f} (hip-graph replay)".format(
                                epoch + 1, iteration, self.optimizer.param_groups[0]["lr"], float(torch.nanmean(torch.stack(losses[-50:]).float()))))
                        continue
                out = self.model(**batch, epoch=epoch, iteration=iteration)
                if "loss" not in out:
                    continue
                loss = out["loss"].mean()
                if self.device_gate:                      # the skip happens inside the optimizer kernel; the gradients of a bad batch are
                    self._gate(loss)                      # discarded by the zero_grad that follows every step
                elif not self._all_finite(loss):          # trainor.py:109-112, decided collectively (host read: torch.optim / grad_accu > 1)
                    self.logger.warning("NaN/Inf loss: batch skipped on all ranks")
                    self._zero_grad()
                    pending = False
                    continue
                step_now = iteration % self.grad_accu == 0
                # micro-batch gradients are SUMMED, not averaged (the reference calls loss.backward() on the unscaled loss,
                # trainor.py:114); the cross-rank mean is taken once, on the iteration that steps
                if self.ddp is not None:
                    self.ddp.backward(loss, sync=step_now)       # two-phase backward; all-reduce overlapped with the encoder's
                else:
                    loss.backward()
                pending = True
                if step_now:
                    self._optimizer_step(epoch, iteration)
                    pending = False
                losses.append(loss.detach())
                if iteration % 50 == 0 and self.rank == 0:
                    self.logger.info("Epoch {}, iter {}, lr {:.2e}, loss {:.4f} {}".format(
                        epoch + 1, iteration, self.optimizer.param_groups[0]["lr"], float(torch.nanmean(torch.stack(losses[-50:]).float())),
                        out.get("custom_print", "")))
            # last update of the epoch when len(dl) is not a multiple of grad_accu (trainor.py:139-150)
            if iteration % self.grad_accu != 0 and "loss" in out and pending:
                if self.ddp is not None:
                    self.ddp.finish()
                self._optimizer_step(epoch, iteration)
            if losses:                                   # skipped (non-finite) batches do not enter the epoch's mean, as in the reference
                t = torch.stack(losses).float()
                ok = torch.isfinite(t)
                n_bad = int((~ok).sum())
                if n_bad:
                    self.logger.warning("{} batch(es) with a NaN/Inf loss were skipped this epoch".format(n_bad))
                training_loss = float(t[ok].mean()) if n_bad < t.numel() else float("inf")
            else:
                training_loss = float("inf")
            if self.dist is not None:       # the same number on every rank: it can drive early stopping / lr decay (a per-rank value
                from ..parallel import mean_over_ranks      # would let ranks leave the loop at different epochs)
                training_loss = mean_over_ranks(training_loss, self.dist, weight=max(1, len(losses)))
            self.logger.info("Epoch {} done: training_loss {:.4f}".format(epoch + 1, training_loss))
            self.training_scheduler.epoch_step()
            # evaluation / early stopping / lr decay, with the reference's epoch + 1 bookkeeping (trainor.py:157-203)
            early_stop_score, decay_metric = None, None
            do_early_stop = epoch + 1 >= early_stop_start
            do_lr_decay = epoch + 1 >= decay_metric_start
            do_eval = epoch + 1 >= self.eval_start
            metric = cfg.get("early_stop_metric")
            if metric == "training_loss" and do_early_stop:
                early_stop_score = training_loss
            if do_eval and self.evaluator is not None:
                self.evaluator.epoch = epoch
                scores = self.evaluator.start()
                if metric != "training_loss" and metric is not None and do_early_stop:
                    missing = [i for i, sc in enumerate(scores) if metric not in sc]
                    if missing:              # the reference raises KeyError here (np.mean over s[early_stop_metric]); never save silently nothing
                        raise KeyError(f"early_stop_metric {metric!r} is not among the validator scores {sorted(scores[missing[0]])}")
                    early_stop_score = float(np.mean([sc[metric] for sc in scores]))
            if do_lr_decay:
                decay_metric = training_loss if self.training_scheduler.decay_on_training_loss else early_stop_score
            ret = self.training_scheduler.eval_step(decay_metric=decay_metric, early_stop_score=early_stop_score)
            if ret["done_training"]:
                self.logger.info("Early stopped reached")
                break
            if ret["save_state"] and self.rank == 0:
                self.saver.save({"model": self.model.state_dict(), "training_scheduler": self.training_scheduler.state_dict(),
                                 "optimizer": self.optimizer.state_dict(), "config": to_container(cfg), "__version__": __version__},
                                tag=early_stop_score, current_epoch=epoch + 1)
