"""gloo tests (CPU) of the data-parallel host logic at world sizes 2 AND 8 (the node size BASELINE.json's metric is quoted on; no
multi-GPU node was available to any round, so this is the only evidence at the target world size): chunked flat-gradient averaging equals
the single-process gradient on the concatenated batch; all-gather-with-grad equals the single-process contrastive loss; two-phase backward,
mark-started encoder buckets, gradient accumulation, the row-sharded contrastive loss, the GLoRIA gather and the validation merge."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)                       # 8 ranks on the build container's 8 cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


WORLDS = (2, 8)


def _run(fn, world=2):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return dict(ret)


def _grad_avg(rank, world):
    from vilmedic_amd.parallel import allreduce_mean_, broadcast_
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 4))
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    if rank:
        flat += float(rank)   # replicas start different: broadcast must fix it
    broadcast_(flat, dist)
    off = 0
    for p in model.parameters():
        p.data.copy_(flat[off:off + p.numel()].view_as(p)); off += p.numel()
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(4 * world, 16, generator=g), torch.randn(4 * world, 4, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    torch.nn.functional.mse_loss(model(xs), ys).backward()
    gflat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    out = {}
    for name, kw in {"fp32": {}, "bf16": dict(wire_dtype=torch.bfloat16, to_wire=lambda s, d: d.copy_(s), from_wire=lambda s, d: d.copy_(s))}.items():
        gf = gflat.clone()
        allreduce_mean_(gf, dist, chunks=3, **kw)
        out[name] = gf
    # single-process reference on the concatenated batch
    ref = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 4))
    ref.load_state_dict(model.state_dict())
    torch.nn.functional.mse_loss(ref(X), Y).backward()
    out["ref"] = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    return out


@pytest.mark.parametrize("world", WORLDS)
def test_flat_gradient_allreduce_equals_single_process_gradient(world):
    r = _run(_grad_avg, world)
    for rank in range(world):
        torch.testing.assert_close(r[rank]["fp32"], r[rank]["ref"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r[rank]["bf16"], r[rank]["ref"], rtol=2e-2, atol=2e-3)
        torch.testing.assert_close(r[0]["fp32"], r[rank]["fp32"], rtol=0, atol=0)      # every replica holds the same averaged gradient


def _contrastive(rank, world):
    from oracle import torch_ref as O
    from vilmedic_amd.parallel import all_gather_with_grad
    g = torch.Generator().manual_seed(9)
    T, V = torch.randn(4 * world, 32, generator=g), torch.randn(4 * world, 32, generator=g)
    t = T[rank * 4:(rank + 1) * 4].clone().requires_grad_(True)
    v = V[rank * 4:(rank + 1) * 4].clone().requires_grad_(True)
    tg, vg = all_gather_with_grad(t, dist), all_gather_with_grad(v, dist)
    # each rank evaluates the global loss; DDP-style averaging of identical losses keeps the scale
    loss = O.convirt_loss(tg, vg, 0.1, 0.75)[0]
    (loss / world).backward()
    Tr, Vr = T.clone().requires_grad_(True), V.clone().requires_grad_(True)
    lref = O.convirt_loss(Tr, Vr, 0.1, 0.75)[0]
    lref.backward()
    return dict(loss=loss.detach(), lref=lref.detach(), gt=t.grad, gv=v.grad,
                rt=Tr.grad[rank * 4:(rank + 1) * 4], rv=Vr.grad[rank * 4:(rank + 1) * 4])


@pytest.mark.parametrize("world", WORLDS)
def test_allgather_negatives_equals_single_process_contrastive_loss(world):
    r = _run(_contrastive, world)
    for rank in range(world):
        torch.testing.assert_close(r[rank]["loss"], r[rank]["lref"])
        torch.testing.assert_close(r[rank]["gt"], r[rank]["rt"], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(r[rank]["gv"], r[rank]["rv"], rtol=1e-4, atol=1e-6)


def _eval_merge(rank, world):
    from vilmedic_amd.parallel import gather_interleaved, mean_over_ranks
    full = [f"sample {i}" for i in range(3 * world + 1)]         # ragged: rank 0 holds one sample more than the others
    mine = full[rank::world]                                     # create_data_loader's round-robin shard
    merged = gather_interleaved(mine, dist)
    loss = mean_over_ranks(2.0 + 3.0 * rank, dist, weight=len(mine))
    return merged, loss


@pytest.mark.parametrize("world", WORLDS)
def test_validation_outputs_and_losses_are_identical_on_every_rank(world):
    """every rank must see the same merged refs / hyps (dataset order) and the same sample-weighted loss, so that early stopping and
    checkpointing decide identically on all ranks"""
    r = _run(_eval_merge, world)
    n = 3 * world + 1
    want = sum((2.0 + 3.0 * k) * len(range(k, n, world)) for k in range(world)) / n
    for rank in range(world):
        assert r[rank][0] == [f"sample {i}" for i in range(n)]
        assert r[rank][1] == r[0][1] == pytest.approx(want)


class _FakeArena:
    """the slice of ParamArena that ArenaDDP uses (flat parameters, flat gradients with ``p.grad`` views, offsets), on the CPU: the
    real arena refuses CPU tensors by design, the orchestration under test does not care where the buffers live"""

    def __init__(self, model):
        params = list(model.parameters())
        self.numel = sum(p.numel() for p in params)
        self.flat, self.gflat = torch.zeros(self.numel), torch.zeros(self.numel)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p._vm_grad_view = self.gflat[off:off + n].view(p.shape)
                p.grad = p._vm_grad_view
                p._vm_off = off
                off += n
        self._layout = [(p, p._vm_off) for p in params]

    def valid(self):
        return True

    def refresh(self, force=False):
        pass


class _TwoPhase(torch.nn.Module):
    """decoder parameters first, encoder parameters last (the arena order ArenaDDP's split needs), RRG's detach protocol"""

    def __init__(self):
        super().__init__()
        self.dec = torch.nn.Sequential(torch.nn.Linear(12, 24), torch.nn.Tanh(), torch.nn.Linear(24, 3))
        self.enc = torch.nn.Sequential(torch.nn.Linear(10, 12), torch.nn.Tanh())
        self.split_backward, self._split = False, None

    def forward(self, x, y):
        feats = self.enc(x)
        if self.split_backward and torch.is_grad_enabled():
            leaf = feats.detach().requires_grad_(True)
            self._split = (feats, leaf)
            feats = leaf
        return torch.nn.functional.mse_loss(self.dec(feats), y)


def _arena_ddp(rank, world):
    from vilmedic_amd.parallel import ArenaDDP
    torch.manual_seed(3 + rank)                      # replicas start DIFFERENT: the constructor's broadcast must align them
    model = _TwoPhase()
    model.__dict__["_vm_arena_cache"] = _FakeArena(model)
    ddp = ArenaDDP(model, dist, chunks=4, bf16_wire=False)
    assert ddp.split_at == sum(p.numel() for p in model.dec.parameters()) and model.split_backward
    g = torch.Generator().manual_seed(11)
    X, Y = torch.randn(4 * world, 10, generator=g), torch.randn(4 * world, 3, generator=g)
    loss = model(X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4])
    ddp.backward(loss)                               # decoder range reduced first, then the encoder range
    assert model._split is None
    ref = _TwoPhase()
    ref.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    ref(X, Y).backward()
    return (ddp.arena.flat.clone(), ddp.arena.gflat.clone(),
            torch.cat([p.grad.reshape(-1) for p in ref.parameters()]))


@pytest.mark.parametrize("world", WORLDS)
def test_arena_ddp_two_phase_backward_equals_single_process_gradient(world):
    """ArenaDDP end to end on 2 / 8 gloo ranks: parameter broadcast, decoder / encoder split point, two-phase backward with the chunked
    all-reduce of each range, result = the gradient of the loss on the concatenated batch"""
    r = _run(_arena_ddp, world)
    for rank in range(world):
        torch.testing.assert_close(r[0][0], r[rank][0], rtol=0, atol=0)          # identical replicas after the broadcast
        torch.testing.assert_close(r[rank][1], r[rank][2], rtol=1e-5, atol=1e-6)


class _TwoPhaseLayers(_TwoPhase):
    """as _TwoPhase with a three-layer encoder that carries backward marks in front of layers 1 and 2 (what nn.ViTModel does)"""

    def __init__(self):
        super().__init__()
        self.enc = torch.nn.ModuleList([torch.nn.Linear(10, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 12)])

    def forward(self, x, y):
        from vilmedic_amd import ops
        h = x
        for i, layer in enumerate(self.enc):
            if i:
                h = ops.backward_mark(h, ("enc_layer", i))
            h = torch.tanh(layer(h))
        feats = h
        if self.split_backward and torch.is_grad_enabled():
            leaf = feats.detach().requires_grad_(True)
            self._split = (feats, leaf)
            feats = leaf
        return torch.nn.functional.mse_loss(self.dec(feats), y)


def _arena_ddp_buckets(rank, world):
    """the encoder range reduced in three buckets, back to front, each started from the backward mark in front of its first layer while
    the layers before it are still being differentiated: two rear buckets from marks + the front bucket at the end"""
    from vilmedic_amd import ops
    from vilmedic_amd.parallel import ArenaDDP
    torch.manual_seed(7)
    model = _TwoPhaseLayers()
    arena = _FakeArena(model)
    model.__dict__["_vm_arena_cache"] = arena
    model.ddp_marks = {("enc_layer", i): model.enc[i].weight._vm_off for i in (1, 2)}
    try:
        ddp = ArenaDDP(model, dist, chunks=2, bf16_wire=False, enc_buckets=3)
        assert len(ddp._marks) == 2 and ops._bwd_mark["cb"] is not None
        g = torch.Generator().manual_seed(13)
        X, Y = torch.randn(4 * world, 10, generator=g), torch.randn(4 * world, 3, generator=g)
        outs = []
        for step in range(2):                              # twice: the bucket state must reset between steps
            arena.gflat.zero_()
            before = ddp.mark_starts
            ddp.backward(model(X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]))
            outs.append((ddp.mark_starts - before, arena.gflat.clone()))
        ref = _TwoPhaseLayers()
        ref.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        cb, ops._bwd_mark["cb"] = ops._bwd_mark["cb"], None     # the single-process reference runs without marks
        ref(X, Y).backward()
        ops._bwd_mark["cb"] = cb
        return outs, torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    finally:
        ops._bwd_mark["cb"] = None


@pytest.mark.parametrize("world", WORLDS)
def test_arena_ddp_encoder_buckets_start_from_backward_marks(world):
    r = _run(_arena_ddp_buckets, world)
    for rank in range(world):
        outs, ref = r[rank]
        for fired, gflat in outs:
            assert fired == 2, fired
            torch.testing.assert_close(gflat, ref, rtol=1e-5, atol=1e-6)


def _arena_ddp_grad_accu(rank, world):
    """two micro-batches per optimizer step: the first backward must NOT reduce but MUST run both phases (with the split
    enabled and a plain loss.backward() the encoder received no gradient at all); between them an optimizer drops .grad
    (zero_grad(set_to_none=True) style) on one native parameter, which must be folded back into the arena before the all-reduce"""
    from vilmedic_amd.parallel import ArenaDDP
    torch.manual_seed(5)
    model = _TwoPhase()
    model.__dict__["_vm_arena_cache"] = _FakeArena(model)
    ddp = ArenaDDP(model, dist, chunks=2, bf16_wire=False)
    g = torch.Generator().manual_seed(12)
    X, Y = torch.randn(8 * world, 10, generator=g), torch.randn(8 * world, 3, generator=g)
    xs, ys = X[rank * 8:(rank + 1) * 8], Y[rank * 8:(rank + 1) * 8]
    ddp.backward(model(xs[:4], ys[:4]), sync=False)
    enc_after_first = ddp.arena.gflat[ddp.split_at:].abs().sum().item()
    stray = model.enc[0].bias
    saved = stray.grad.clone()
    stray.grad = None                                   # what torch.optim's zero_grad(set_to_none=True) / a foreign hook may leave behind
    stray.grad = saved.clone()                          # ... and autograd then accumulates into a tensor OUTSIDE the arena
    stray._vm_grad_view.zero_()
    ddp.backward(model(xs[4:], ys[4:]), sync=True)
    ref = _TwoPhase()
    ref.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    total = 0.0
    for r in range(world):                              # mean over ranks of the SUM over each rank's micro-batches
        for lo in (0, 4):
            total = total + ref(X[r * 8 + lo:r * 8 + lo + 4], Y[r * 8 + lo:r * 8 + lo + 4])
    (total / world).backward()
    return enc_after_first, ddp.arena.gflat.clone(), torch.cat([p.grad.reshape(-1) for p in ref.parameters()]), stray.grad is stray._vm_grad_view


@pytest.mark.parametrize("world", WORLDS)
def test_arena_ddp_gradient_accumulation_trains_the_encoder_and_folds_stray_grads(world):
    r = _run(_arena_ddp_grad_accu, world)
    for rank in range(world):
        assert r[rank][0] > 0, "encoder gradient missing after a non-stepping micro-batch"
        torch.testing.assert_close(r[rank][1], r[rank][2], rtol=1e-5, atol=1e-6)
        assert r[rank][3]


def _gloria_local_gather(rank, world):
    """GLoRIA's local loss over the GLOBAL batch: every rank holds 3 (image, caption) pairs, the local feature maps and
    word embeddings are all-gathered with gradient (the product's _maybe_gather, as GLoRIALoss.forward calls it), the loss is that of
    the 3 x world-pair batch and each rank's gradients are its slice of the single-process gradients (x world: ArenaDDP then averages parameter
    gradients over ranks).  The loss itself is evaluated by the oracle here (the product's local loss is HIP-only)."""
    from oracle import torch_ref as O
    from vilmedic_amd.blocks.losses.selfsup import _maybe_gather
    g = torch.Generator().manual_seed(21)
    b, D, T = 3, 16, 5
    B = b * world
    img = torch.randn(B, D, 3, 3, generator=g)
    words = torch.randn(B, D, T, generator=g)
    lens = [(5, 3, 4, 2, 5, 4)[i % 6] for i in range(B)]
    li = img[rank * b:(rank + 1) * b].clone().requires_grad_(True)
    lw = words[rank * b:(rank + 1) * b].clone().requires_grad_(True)
    (gi, gw), _, w = _maybe_gather(li, lw)
    assert w == world and gi.shape[0] == B
    l0, l1 = O.gloria_local_loss(gi, gw, lens, 4.0, 5.0, 10.0)
    (l0 + l1).backward()
    fi, fw = img.clone().requires_grad_(True), words.clone().requires_grad_(True)
    r0, r1 = O.gloria_local_loss(fi, fw, lens, 4.0, 5.0, 10.0)
    (r0 + r1).backward()
    return (float(l0 + l1), float(r0 + r1), li.grad, world * fi.grad[rank * b:(rank + 1) * b], lw.grad, world * fw.grad[rank * b:(rank + 1) * b])


@pytest.mark.parametrize("world", WORLDS)
def test_gloria_local_loss_contrasts_the_global_batch(world):
    r = _run(_gloria_local_gather, world)
    for rank in range(world):
        assert r[rank][0] == pytest.approx(r[rank][1], rel=1e-6)
        torch.testing.assert_close(r[rank][2], r[rank][3], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r[rank][4], r[rank][5], rtol=1e-5, atol=1e-6)


def _torch_similarity(a, b, normalize, inv_tau, eps, off=0):
    """CPU stand-in for the HIP _SimilarityLossFn (same signature, plain autograd): row / column losses of the pairs (i, i + off)"""
    ah = a / a.norm(dim=1, keepdim=True).clamp_min(eps) if normalize else a
    bh = b / b.norm(dim=1, keepdim=True).clamp_min(eps) if normalize else b
    S = ah @ bh.t() * inv_tau
    lo, hi = max(0, -off), min(a.shape[0], b.shape[0] - off)
    idx = torch.arange(lo, hi)
    d = S[idx, idx + off]
    return torch.logsumexp(S, 1)[lo:hi] - d, torch.logsumexp(S, 0)[lo + off:hi + off] - d


def _row_sharded_contrastive(rank, world):
    """ConVIRTLoss under data parallelism: each rank evaluates only ITS rows of the global similarity (two [b, B] problems with the paired
    column at rank * b + i, blocks/losses/selfsup._paired_losses) -- the per-sample losses must be the rank's slice of the single-process
    losses on the concatenated batch and the gradients its slice x world (ArenaDDP then averages over ranks)."""
    from oracle import torch_ref as O
    from vilmedic_amd.blocks.losses import selfsup

    class _Fn:
        apply = staticmethod(_torch_similarity)
    selfsup._SimilarityLossFn = _Fn                    # the HIP kernels need a GPU; the host logic under test is everything around them
    g = torch.Generator().manual_seed(13)
    T, V = torch.randn(4 * world, 32, generator=g), torch.randn(4 * world, 32, generator=g)
    t = T[rank * 4:(rank + 1) * 4].clone().requires_grad_(True)
    v = V[rank * 4:(rank + 1) * 4].clone().requires_grad_(True)
    loss, loss_l, loss_v = selfsup.ConVIRTLoss(tau=0.1, lambda_=0.75)(t, v)
    loss.backward()
    Tr, Vr = T.clone().requires_grad_(True), V.clone().requires_grad_(True)
    lref, ref_l, ref_v = O.convirt_loss(Tr, Vr, 0.1, 0.75)
    lref.backward()
    sl = slice(rank * 4, (rank + 1) * 4)
    return dict(ll=loss_l.detach(), rl=ref_l.detach()[sl], lv=loss_v.detach(), rv=ref_v.detach()[sl], gt=t.grad, rt=world * Tr.grad[sl], gv=v.grad, rvg=world * Vr.grad[sl],
                loss=loss.detach(), local_mean=(0.75 * ref_v.detach()[sl] + 0.25 * ref_l.detach()[sl]).mean())


@pytest.mark.parametrize("world", WORLDS)
def test_row_sharded_contrastive_losses_equal_the_global_batch(world):
    r = _run(_row_sharded_contrastive, world)
    for rank in range(world):
        torch.testing.assert_close(r[rank]["ll"], r[rank]["rl"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r[rank]["lv"], r[rank]["rv"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r[rank]["loss"], r[rank]["local_mean"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r[rank]["gt"], r[rank]["rt"], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(r[rank]["gv"], r[rank]["rvg"], rtol=1e-4, atol=1e-6)
