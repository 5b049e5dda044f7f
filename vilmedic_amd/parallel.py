"""Data parallelism over the parameter arena: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Replaces the reference's two strategies: ``nn.DataParallel`` (vilmedic/executors/utils.py:128-133)
and Accelerate/DDP (vilmedic/executors/trainor_accelerate.py:91-93,132).  Semantics reproduced: identical replicas,
per-rank batches, gradients AVERAGED across ranks every optimizer step (mean of per-rank mean losses == loss on the
concatenated batch because every rank contributes B*(L-1) tokens).

Because all gradients live in one flat fp32 buffer, the exchange is a handful of LARGE collectives (xGMI rings are
per-link bound: fewer, bigger messages) instead of DDP's 25 MB buckets; optionally compressed to bf16 on the wire
(446 MB instead of 892 MB at the RRG ViT-B + 12-layer-decoder size).
"""
import os

import torch


def force_collectives():
    """VM_FORCE_DDP=1: take every collective path even with a ONE-rank group, so that a single-GPU box executes the RCCL calls
    (tests/test_ddp_nccl_gpu.py, bench.py).  Production runs never set it: a 1-rank group then skips the communication."""
    return bool(os.environ.get("VM_FORCE_DDP"))


def active(dist=None):
    """the process group to communicate over, or None: initialised AND (more than one rank, or collectives forced)"""
    if dist is None:
        import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist if (dist.get_world_size() > 1 or force_collectives()) else None


def _avg_allreduce(t, dist, world, async_op=False):
    backend = dist.get_backend()
    if backend == "nccl":
        return dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=async_op)
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)   # gloo has no AVG
    if async_op:
        class _W:
            def wait(self_inner):
                w.wait()
                t.div_(world)
        return _W()
    t.div_(world)
    return None


def chunk_ranges(numel, chunks, align=64):
    per = (numel + chunks - 1) // chunks
    per = (per + align - 1) // align * align
    return [(s, min(numel, s + per)) for s in range(0, numel, per)]


def allreduce_mean_(flat, dist, chunks=4, wire_dtype=None, to_wire=None, from_wire=None):
    """In-place mean over ranks of a flat tensor, as ``chunks`` pipelined collectives.
    wire_dtype: None (same dtype) or torch.bfloat16 (compress on the wire; to_wire/from_wire do the casts)."""
    world = dist.get_world_size()
    if world == 1 and not force_collectives():
        return flat
    ranges = chunk_ranges(flat.numel(), chunks)
    if wire_dtype is None:
        works = [_avg_allreduce(flat[s:e], dist, world, async_op=True) for s, e in ranges]
        for w in works:
            w.wait()
        return flat
    wire = torch.empty(flat.numel(), dtype=wire_dtype, device=flat.device)
    works = []
    for s, e in ranges:
        to_wire(flat[s:e], wire[s:e])
        works.append(_avg_allreduce(wire[s:e], dist, world, async_op=True))
    for (s, e), w in zip(ranges, works):
        w.wait()
        from_wire(wire[s:e], flat[s:e])
    return flat


def broadcast_(flat, dist, src=0):
    if dist.get_world_size() > 1 or force_collectives():
        dist.broadcast(flat, src=src)
    return flat


def default_bf16_wire():
    import os
    return os.environ.get("VM_DDP_WIRE", "fp32").lower() == "bf16"


class ArenaDDP:
    """Wraps a model whose parameters live in a ParamArena: broadcasts rank 0's parameters at construction and averages
    the flat gradient buffer across ranks.

    ``backward(loss)`` overlaps communication with compute when the model exposes an encoder/decoder split (RRG): the
    decoder consumes a DETACHED copy of the image features, so ``loss.backward()`` runs exactly the decoder's graph; the
    decoder's arena range (parameters are laid out [dec | enc]) is then all-reduced asynchronously on RCCL's stream
    while the encoder's backward (``features.backward(d_features)``) runs on the compute stream.  The encoder range itself
    is reduced in ``enc_buckets`` pieces, back to front: the encoder's layers carry backward marks (ops.backward_mark), and when
    the backward pass crosses the mark in front of layer i the gradients of layers >= i (which lie BEHIND it in the arena) are
    complete and their all-reduce starts while layers < i are still being differentiated; only the front bucket is exposed.
    Every rank issues the same collectives in the same order (the marks fire in program order).
    ``finish()`` (no overlap) remains for callers that ran ``loss.backward()`` themselves."""

    def __init__(self, model, dist, chunks=4, bf16_wire=None, wire=None, enc_buckets=4):
        from . import ops
        from .arena import arena_of
        self.dist = dist
        self.model = model
        self.arena = arena_of(model)
        self.chunks = chunks
        # the wire format of the gradient all-reduce.  Default fp32: the averaged gradients equal the single-process gradients of the
        # concatenated batch (SURVEY §8e's parity statement) -- the collective runs in place on the fp32 gradient arena, no cast passes.
        # bf16 (``bf16_wire=True``, trainor.ddp_wire: bf16, VM_DDP_WIRE=bf16) halves the bytes over xGMI at one rounding per gradient
        # (measured 1.7e-3 rel-L2 on the C2 model): an explicit opt-in, reported by bench.py as config.ddp_wire.
        if bf16_wire is None:
            bf16_wire = default_bf16_wire()
        self.bf16_wire = bool(bf16_wire)
        self._ops = ops
        self.world = dist.get_world_size()
        broadcast_(self.arena.flat, dist)
        self.arena.refresh(force=True)
        self.split_at = None
        if hasattr(model, "enc") and hasattr(model, "dec") and hasattr(model, "split_backward"):
            enc_offs = [p._vm_off for p in model.enc.parameters()]
            dec_offs = [p._vm_off + p.numel() for p in model.dec.parameters()]
            if enc_offs and dec_offs and min(enc_offs) >= max(dec_offs):
                self.split_at = min(enc_offs)
                model.split_backward = True
        # encoder buckets: {mark tag: arena offset where that layer's parameters start}; models may provide ``ddp_marks`` themselves
        self._marks, self._live, self.mark_starts, self._opt = {}, None, 0, None
        if self.split_at is not None and enc_buckets > 1:
            marks = getattr(model, "ddp_marks", None) or self._layer_marks(model)
            offs = sorted(o for o in marks.values() if self.split_at < o < self.arena.numel)
            if offs:
                want = [self.split_at + (self.arena.numel - self.split_at) * k // enc_buckets for k in range(1, enc_buckets)]
                chosen = {min(offs, key=lambda o: abs(o - w)) for w in want}
                self._marks = {t: o for t, o in marks.items() if o in chosen}
                ops._bwd_mark["cb"] = self._on_mark
        # ``wire``: optional pre-allocated bf16 staging buffer (callers allocate it before init_process_group, see bench.py)
        self._wire = (wire if wire is not None else
                      torch.empty(self.arena.numel, dtype=torch.bfloat16, device=self.arena.flat.device)) if bf16_wire else None

    @staticmethod
    def _layer_marks(model):
        """("enc_layer", i) -> arena offset of layer i's first parameter, for encoders with an ``encoder.layer`` list whose layers lie
        in the arena in order (nn.ViTModel); {} otherwise"""
        enc = getattr(getattr(model, "enc", None), "model", None)
        layers = getattr(getattr(enc, "encoder", None), "layer", None)
        if layers is None:
            return {}
        spans = []
        for layer in layers:
            offs = [(p._vm_off, p._vm_off + p.numel()) for p in layer.parameters()]
            if not offs:
                return {}
            spans.append((min(o for o, _ in offs), max(e for _, e in offs)))
        if any(spans[i][1] > spans[i + 1][0] for i in range(len(spans) - 1)):      # not laid out layer after layer
            return {}
        return {("enc_layer", i): spans[i][0] for i in range(1, len(spans))}

    def _on_mark(self, tag):
        """backward has just crossed the mark in front of an encoder layer: reduce everything behind it that is not reduced yet"""
        st = self._live
        off = self._marks.get(tag)
        if st is None or off is None or off >= st["hi"]:
            return
        ops = self._ops
        ops.flush_param_grads()                          # the queued weight / LayerNorm gradients of the layers behind the mark
        with ops.collective_context(self.arena.flat.device):
            st["pending"] += self._start(off, st["hi"], 1)
        st["hi"] = off
        self.mark_starts += 1

    # ---- asynchronous all-reduce of gflat[s:e] in `chunks` pieces; returns the pending work items
    def _start(self, s, e, chunks):
        g, works = self.arena.gflat, []
        for cs, ce in [(s + a, s + b) for a, b in chunk_ranges(e - s, chunks)]:
            if self.bf16_wire:
                self._ops.cast_to_bf16(g[cs:ce], self._wire[cs:ce])
                works.append((cs, ce, _avg_allreduce(self._wire[cs:ce], self.dist, self.world, async_op=True)))
            else:
                works.append((cs, ce, _avg_allreduce(g[cs:ce], self.dist, self.world, async_op=True)))
        return works

    def attach_optimizer(self, optimizer):
        """a FusedAdam that reads the averaged bf16 wire buffer itself (vm_adam_step_wire): the cast pass back into the fp32 gradient arena
        is skipped, so after a reducing backward ``p.grad`` holds this rank's LOCAL gradients -- only for loops that do nothing with the
        gradients between backward and step (no clipping, no inspection).  No-op for an fp32 wire or another optimizer."""
        if self.bf16_wire and hasattr(optimizer, "grad_wire") and getattr(optimizer, "arena", None) is self.arena:
            self._opt = optimizer

    def _wait(self, works):
        fused = self.bf16_wire and self._opt is not None and self._covered(works)
        for cs, ce, w in works:
            w.wait()
            if self.bf16_wire and not fused:
                self._ops.cast_to_f32(self._wire[cs:ce], self.arena.gflat[cs:ce])
        if fused:
            self._opt.grad_wire = self._wire

    def _covered(self, works):
        """the pending pieces tile the whole arena (a reducing backward / finish() always does; checked, not assumed)"""
        at = 0
        for cs, ce in sorted((cs, ce) for cs, ce, _ in works):
            if cs > at:
                return False
            at = max(at, ce)
        return at >= self.arena.numel

    def backward(self, loss, sync=True):
        """loss.backward() + gradient averaging, communication overlapped with the encoder's backward when possible.
        ``sync=False`` (a gradient-accumulation micro-batch that does not step): both backward phases run, nothing is reduced."""
        self._check_grads_in_arena()
        n = self.arena.numel
        split = getattr(self.model, "_split", None) if self.split_at is not None else None
        if split is None:
            loss.backward()
            if sync:
                self._wait(self._start(0, n, self.chunks))
            return
        feats, leaf = split
        if not sync:
            loss.backward()                              # decoder graph (the features were detached) ...
            if leaf.grad is not None:
                feats.backward(leaf.grad)                # ... then ALWAYS the encoder graph: the split must never swallow it
            self._ops.join_side()
            self.model._split = None
            return
        ops, dev = self._ops, self.arena.flat.device
        # The collectives are enqueued from the SIDE stream (where the weight-gradient GEMMs run): they are ordered after
        # the gradients they reduce without the main stream ever waiting for the side stream between the two phases.
        ops._side["defer"] = True
        try:
            loss.backward()                              # decoder graph only (features were detached)
            ops.flush_param_grads()                      # the decoder's queued weight gradients, before their range is reduced
            with ops.collective_context(dev):
                pending = self._start(0, self.split_at, max(1, self.chunks // 2))
            self._live = {"hi": n, "pending": pending}   # the encoder's backward marks start the buckets behind them (_on_mark)
            if leaf.grad is not None:
                feats.backward(leaf.grad)                # encoder graph, overlapping the decoder's (and its own rear buckets') all-reduce
            hi, self._live = self._live["hi"], None
            ops.flush_param_grads()
            with ops.collective_context(dev):
                pending += self._start(self.split_at, hi, 1 if hi < n else max(1, self.chunks // 2))
                self._wait(pending)
        finally:
            self._live = None
            ops._side["defer"] = False
        ops.join_side()                                  # the optimizer (main stream) waits for the averaged gradients
        self.model._split = None

    def finish(self):
        self._check_grads_in_arena()
        self._wait(self._start(0, self.arena.numel, self.chunks))

    def _check_grads_in_arena(self):
        """only ``arena.gflat`` is all-reduced: a parameter whose .grad was re-created outside it (optimizer.zero_grad(
        set_to_none=True) followed by a native-autograd backward) would silently keep per-rank gradients -> fold it back in"""
        for p, _ in self.arena._layout:
            g = p.grad
            if not p.requires_grad or g is p._vm_grad_view:
                continue
            if g is None:
                p.grad = p._vm_grad_view
            elif g.data_ptr() != p._vm_grad_view.data_ptr():
                p._vm_grad_view.add_(g.to(p._vm_grad_view.dtype))
                p.grad = p._vm_grad_view


def all_gather_with_grad(x, dist):
    """all-gather along dim 0 whose backward returns this rank's slice of the (summed) upstream gradient -- the
    contrastive-negatives exchange (all-gather fwd, reduce-scatter bwd)."""
    return _AllGather.apply(x, dist)


class _AllGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dist):
        world, rank = dist.get_world_size(), dist.get_rank()
        ctx.meta = (dist, world, rank, x.shape[0])
        outs = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(outs, x.contiguous())
        return torch.cat(outs, 0)

    @staticmethod
    def backward(ctx, g):
        dist, world, rank, n = ctx.meta
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)      # reduce-scatter expressed as all-reduce + slice (small payload)
        return g[rank * n:(rank + 1) * n], None


# ---- rank-consistent evaluation: every rank must take the same early-stop / checkpoint decision, otherwise one rank leaves the
# training loop while the others wait in the next gradient all-reduce
def gather_interleaved(items, dist):
    """per-rank lists of a round-robin sharded dataset (rank r holds samples r, r + world, ...) -> the full list in dataset
    order, on every rank"""
    world = dist.get_world_size()
    if world == 1 and not force_collectives():
        return list(items)
    parts = [None] * world
    dist.all_gather_object(parts, list(items))
    out = []
    for i in range(max(len(p) for p in parts)):
        for p in parts:
            if i < len(p):
                out.append(p[i])
    return out


def mean_over_ranks(value, dist, weight=1.0, device=None):
    """weighted mean of a python scalar over ranks (weight = the number of samples it averages), identical on every rank"""
    if dist.get_world_size() == 1 and not force_collectives():
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value) * float(weight), float(weight)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0] / t[1]) if float(t[1]) > 0 else float("nan")
