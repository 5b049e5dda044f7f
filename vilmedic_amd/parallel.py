"""Data parallelism over the parameter arena: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Replaces the reference's two strategies: ``nn.DataParallel`` (vilmedic/executors/utils.py:128-133)
and Accelerate/DDP (vilmedic/executors/trainor_accelerate.py:91-93,132).  Semantics reproduced: identical replicas,
per-rank batches, gradients AVERAGED across ranks every optimizer step (mean of per-rank mean losses == loss on the
concatenated batch because every rank contributes B*(L-1) tokens).

Because all gradients live in one flat fp32 buffer, the exchange is a handful of LARGE collectives (xGMI rings are
per-link bound: fewer, bigger messages) instead of DDP's 25 MB buckets; optionally compressed to bf16 on the wire
(446 MB instead of 892 MB at the RRG ViT-B + 12-layer-decoder size).
"""
import torch


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
    if world == 1:
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
    if dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


class ArenaDDP:
    """Wraps a model whose parameters live in a ParamArena: broadcasts rank 0's parameters at construction and averages
    the flat gradient buffer across ranks in ``finish()`` (call between ``backward()`` and ``optimizer.step()``)."""

    def __init__(self, model, dist, chunks=4, bf16_wire=True):
        from . import ops
        from .arena import arena_of
        self.dist = dist
        self.arena = arena_of(model)
        self.chunks = chunks
        self.bf16_wire = bf16_wire
        self._ops = ops
        broadcast_(self.arena.flat, dist)
        self.arena.refresh(force=True)

    def finish(self):
        ops = self._ops
        if self.bf16_wire:
            allreduce_mean_(self.arena.gflat, self.dist, self.chunks, torch.bfloat16,
                            to_wire=lambda src, dst: ops.cast_to_bf16(src, dst), from_wire=lambda src, dst: ops.cast_to_f32(src, dst))
        else:
            allreduce_mean_(self.arena.gflat, self.dist, self.chunks)


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
