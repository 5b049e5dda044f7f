"""ArenaDDP on RCCL (torch.distributed backend "nccl") with a ONE-rank group on the GPU box: the collective path itself -- parameter
broadcast, two-phase backward with the all-reduce enqueued from the side stream, bf16 and fp32 wire formats, the gradient-
accumulation (no-sync) micro-batch, the grouped weight-gradient flush in front of each reduced range -- executes on the real
communicator and must reproduce the single-process step.  (World sizes > 1 are covered by the gloo tests on CPU and by the driver's
scaling run; one GPU is all a test box has.)  ref: vilmedic/executors/trainor_accelerate.py:91-93,126-142."""
import os

import pytest
import torch

import golden_recipes as R

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def nccl():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this interpreter")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29571", rank=0, world_size=1, device_id=dev())
    yield dist
    dist.destroy_process_group()


def _rrg(vit_cfg, dec_cfg, seed=0):
    from vilmedic_amd.models.rrg.RRG import RRG
    torch.manual_seed(seed)
    return RRG(decoder=dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **dec_cfg),
               cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute", dropout_out=0.0, **vit_cfg)).to(dev())


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


@pytest.mark.parametrize("bf16_wire", [False, True])
def test_arena_ddp_step_equals_single_process_step(nccl, bf16_wire):
    from vilmedic_amd.arena import arena_of
    from vilmedic_amd.parallel import ArenaDDP
    m1, m2 = _rrg(R.VIT_TINY, R.DEC_TINY), _rrg(R.VIT_TINY, R.DEC_TINY)
    m2.load_state_dict(m1.state_dict())
    images = R.make_images(8, R.VIT_TINY["image_size"], seed=3).to(dev())
    ids, am = R.make_reports(8, 64, R.DEC_TINY["vocab_size"], seed=3)          # 512 token rows: the grouped weight-gradient path
    ids, am = ids.to(dev()), am.to(dev())
    m1.train(), m2.train()
    m1(input_ids=ids, attention_mask=am, images=images)["loss"].backward()
    ddp = ArenaDDP(m2, nccl, bf16_wire=bf16_wire)
    assert ddp.split_at is not None and m2.split_backward
    ddp.backward(m2(input_ids=ids, attention_mask=am, images=images)["loss"])
    torch.cuda.synchronize()
    assert ddp.mark_starts == len(ddp._marks) >= 1, (ddp.mark_starts, ddp._marks)      # the encoder's rear buckets started from backward marks
    g1, g2 = arena_of(m1).gflat, arena_of(m2).gflat
    assert g1.abs().sum().item() > 0
    err = _rel(g2, g1)
    print(f"[parity] ArenaDDP 1-rank nccl, bf16_wire={bf16_wire}: rel L2 error of the flat gradient {err:.3e}", flush=True)
    assert err <= (4e-3 if bf16_wire else 1e-6)
    # gradient accumulation: a non-stepping micro-batch (no collective) followed by a stepping one == the summed gradients
    arena_of(m1).zero_grad(), arena_of(m2).zero_grad()
    for lo in (0, 4):
        m1(input_ids=ids[lo:lo + 4], attention_mask=am[lo:lo + 4], images=images[lo:lo + 4])["loss"].backward()
    ddp.backward(m2(input_ids=ids[:4], attention_mask=am[:4], images=images[:4])["loss"], sync=False)
    assert arena_of(m2).gflat[ddp.split_at:].abs().sum().item() > 0          # the encoder received its gradient
    ddp.backward(m2(input_ids=ids[4:], attention_mask=am[4:], images=images[4:])["loss"], sync=True)
    torch.cuda.synchronize()
    assert _rel(arena_of(m2).gflat, arena_of(m1).gflat) <= (4e-3 if bf16_wire else 1e-5)


def test_optimizer_reads_the_wire_buffer_and_takes_the_same_step(nccl):
    """ArenaDDP.attach_optimizer: FusedAdam consumes the averaged bf16 wire buffer (vm_adam_step_wire) instead of gradients cast back to
    fp32 first.  bf16 -> fp32 is exact, so parameters, moments and bf16 shadows after two steps are BIT-identical to the unattached
    path; the fp32 gradient arena keeps the local gradients (documented side effect)."""
    from vilmedic_amd.arena import arena_of
    from vilmedic_amd.optim import FusedAdam
    from vilmedic_amd.parallel import ArenaDDP
    m1, m2 = _rrg(R.VIT_TINY, R.DEC_TINY), _rrg(R.VIT_TINY, R.DEC_TINY)
    m2.load_state_dict(m1.state_dict())
    images = R.make_images(8, R.VIT_TINY["image_size"], seed=5).to(dev())
    ids, am = R.make_reports(8, 64, R.DEC_TINY["vocab_size"], seed=5)
    ids, am = ids.to(dev()), am.to(dev())
    m1.train(), m2.train()
    o1, o2 = FusedAdam(m1, lr=1e-3), FusedAdam(m2, lr=1e-3)
    d1, d2 = ArenaDDP(m1, nccl, bf16_wire=True), ArenaDDP(m2, nccl, bf16_wire=True)
    d2.attach_optimizer(o2)
    for step in range(2):
        for m, o, d in ((m1, o1, d1), (m2, o2, d2)):
            out = m(input_ids=ids, attention_mask=am, images=images)
            o.zero_grad()
            d.backward(out["loss"])
            if d is d2:
                assert o.grad_wire is d._wire
            o.step()
            assert o.grad_wire is None
    torch.cuda.synchronize()
    a1, a2 = arena_of(m1), arena_of(m2)
    assert torch.equal(a1.flat, a2.flat) and torch.equal(o1.m, o2.m) and torch.equal(o1.v, o2.v) and torch.equal(a1.shadow_flat, a2.shadow_flat)
    print("[parity] FusedAdam on the bf16 wire buffer == FusedAdam on the cast-back gradients after 2 steps: bit-identical", flush=True)
    # an fp32 wire, or an optimizer of another arena, is left alone
    d3 = ArenaDDP(m1, nccl, bf16_wire=False)
    d3.attach_optimizer(o1)
    d1.attach_optimizer(o2)
    assert d3._opt is None and d1._opt is None


def test_fp32_wire_default_at_the_benched_batch(nccl, monkeypatch):
    """"averaged grads == single-process grads".  The DEFAULT wire (fp32, all-reduced in place on the gradient arena) on the
    full ViT-B/16 + 12-layer model at the benched batch (B = 64, L = 128): ArenaDDP's two-phase backward with the mark-started encoder
    buckets reproduces the single-process gradients to float-accumulation noise (the weight-gradient flushes regroup)."""
    import bench
    from vilmedic_amd.arena import arena_of
    from vilmedic_amd.parallel import ArenaDDP
    monkeypatch.delenv("VM_DDP_WIRE", raising=False)
    model = bench.build_model(dev())
    model.train()
    for mod in model.modules():
        if hasattr(mod, "cfg") and hasattr(mod.cfg, "hidden_dropout_prob"):
            mod.cfg.hidden_dropout_prob = mod.cfg.attention_probs_dropout_prob = 0.0
    images, ids, am = bench.synthetic_batch(64, 128, bench.DEC_12L["vocab_size"], dev(), seed=0)
    arena_of(model).zero_grad()
    model(input_ids=ids, attention_mask=am, images=images, return_logits=False)["loss"].backward()
    torch.cuda.synchronize()
    ref = arena_of(model).gflat.clone()
    arena_of(model).zero_grad()
    ddp = ArenaDDP(model, nccl)
    assert ddp.bf16_wire is False and ddp._wire is None
    ddp.backward(model(input_ids=ids, attention_mask=am, images=images, return_logits=False)["loss"])
    torch.cuda.synchronize()
    got = arena_of(model).gflat
    err, worst = _rel(got, ref), ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"[parity] default (fp32) gradient wire on the C2 model at B=64: rel L2 {err:.3e}, max abs / max |g| {worst:.3e}", flush=True)
    assert err <= 1e-5 and worst <= 1e-5


def test_bf16_wire_error_on_the_c2_model(nccl):
    """the opt-in bf16 wire format rounds every gradient to bf16 once before the all-reduce (446 MB instead of 892 MB per step at the
    BASELINE configs[1] size): its error against the fp32 gradients of the same backward pass, on the full ViT-B/16 + 12-layer model"""
    import bench
    from vilmedic_amd.arena import arena_of
    from vilmedic_amd.parallel import ArenaDDP
    model = bench.build_model(dev())
    model.train()
    for mod in model.modules():                      # no dropout: both backward passes see the same graph
        if hasattr(mod, "cfg") and hasattr(mod.cfg, "hidden_dropout_prob"):
            mod.cfg.hidden_dropout_prob = mod.cfg.attention_probs_dropout_prob = 0.0
    images, ids, am = bench.synthetic_batch(16, 128, bench.DEC_12L["vocab_size"], dev(), seed=0)
    model(input_ids=ids, attention_mask=am, images=images, return_logits=False)["loss"].backward()
    torch.cuda.synchronize()
    ref = arena_of(model).gflat.clone()
    arena_of(model).zero_grad()
    ddp = ArenaDDP(model, nccl, bf16_wire=True)
    ddp.backward(model(input_ids=ids, attention_mask=am, images=images, return_logits=False)["loss"])
    torch.cuda.synchronize()
    got = arena_of(model).gflat
    err, worst = _rel(got, ref), ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"[parity] bf16 gradient wire on the C2 model ({got.numel() / 1e6:.1f} M parameters): rel L2 {err:.3e}, max abs / max |g| {worst:.3e}", flush=True)
    assert err <= 4e-3 and worst <= 4e-3           # one round-to-nearest bf16: 2^-9 = 1.95e-3 relative per element


# ---------------------------------------------------------------------------------------------------------------------------------
# VM_FORCE_DDP=1 makes a ONE-rank group take every collective path (vilmedic_amd/parallel.py: active / force_collectives), so the RCCL
# calls of the training loop, of the contrastive-negatives all-gather and of the GLoRIA gather execute on the GPU box's communicator.
# With one rank every collective is the identity, so the results must equal the run without a process group.
def _tiny_trainor(tmp_path, tag, extra=(), yml="rrg-vit-synthetic.yml", prepare=None):
    from vilmedic_amd.config import executor_view, get_config
    from vilmedic_amd.executors import Trainor
    os.makedirs(tmp_path / tag, exist_ok=True)
    cfg = get_config(os.path.join(os.path.dirname(__file__), "..", "config", "RRG", yml),
                     ["dataset.num_samples=16", "dataset.image_size=32", "dataset.vocab_size=97", "dataset.tokenizer_max_len=16",
                      "model.decoder.hidden_size=128", "model.decoder.num_attention_heads=2", "model.decoder.intermediate_size=256",
                      "model.decoder.num_hidden_layers=2", "model.decoder.max_position_embeddings=64",
                      "model.cnn.image_size=32", "model.cnn.patch_size=8", "model.cnn.hidden_size=128", "model.cnn.num_attention_heads=2",
                      "model.cnn.intermediate_size=256", "model.cnn.num_hidden_layers=2",
                      "trainor.batch_size=4", "trainor.epochs=1", "trainor.eval_start=0", "validator.batch_size=4", "validator.beam_width=2",
                      f"ckpt_dir={tmp_path / tag}", "trainor.optim_params.lr=0.003"] + list(extra))
    t = executor_view(cfg, "trainor")
    t["validator_view"] = executor_view(cfg, "validator")
    tr = Trainor(t, seed=0)
    if prepare is not None:
        prepare(tr)
    losses, fwd = [], tr.model.forward

    def recording(**batch):
        out = fwd(**batch)
        if "loss" in out and tr.model.training:
            losses.append(out["loss"].detach().float().reshape(()).clone())
        return out
    if getattr(tr, "graph_any", False):            # replayed iterations do not call forward(): record what the graphed iteration returns
        gi = tr._graphed_iteration

        def graphed(batch):
            loss = gi(batch)
            if loss is not None:                   # (None: this batch signature runs eagerly -- capture refused or failed)
                losses.append(loss.detach().float().reshape(()).clone())
            return loss
        tr._graphed_iteration = graphed
    else:
        tr.model.forward = recording
    tr.start()
    torch.cuda.synchronize()
    from vilmedic_amd.arena import arena_of
    return tr, torch.stack(losses).cpu(), arena_of(tr.model).flat.clone(), tr.evaluator.scores


def test_trainor_graph_step_replays_the_eager_iterations(tmp_path):
    """trainor.graph_step for a model without its own graphed_step: every iteration of ``Trainor.start()`` (forward, backward, fused Adam
    with the device-side NaN gate) is replayed from the HIP graph captured for the batch's shapes after two eager iterations.  Without
    dropout the loss trajectory and the final parameters are those of the eager loop; validation and checkpointing run as before."""
    extra = ["model.decoder.hidden_dropout_prob=0.0", "model.decoder.attention_probs_dropout_prob=0.0", "trainor.epochs=2"]
    t0, l0, p0, s0 = _tiny_trainor(tmp_path, "eager", extra)
    t1, l1, p1, s1 = _tiny_trainor(tmp_path, "graph", extra + ["trainor.graph_step=true"])
    assert t1.graph_any and not t0.graph_any and len(t1._graphs) == 1
    g = next(iter(t1._graphs.values()))
    assert g.graph is not None                                    # 12 iterations: 2 eager, then captured and replayed
    assert l0.shape == l1.shape and l0.numel() == 12            # (epochs 0 .. 2 of 4 iterations each, as the reference counts them)
    dl, dp = (l0 - l1).abs().max().item(), _rel(p1, p0)
    print(f"[parity] Trainor.start() with graph_step: max |loss_t - loss_t(eager)| {dl:.3e}, rel L2 of the final parameters {dp:.3e}; "
          f"validation {s1[0]} vs {s0[0]}", flush=True)
    assert dl <= 1e-5 and dp <= 1e-5
    assert s1[0] == s0[0]
    assert int(t1.optimizer.state_dict()["steps"]) == 12


def test_trainor_graph_step_falls_back_to_eager_when_the_capture_fails(tmp_path):
    """a forward that reads the device (legal eagerly, illegal while a stream is capturing) makes the capture of its batch signature fail:
    the signature is marked eager, the host-side queues of the aborted capture are dropped (ops.reset_host_state), the optimizer's step count
    is restored, and the run continues eagerly to the same parameters as a run that never tried; a graph cache of one signature evicts"""
    extra = ["model.decoder.hidden_dropout_prob=0.0", "model.decoder.attention_probs_dropout_prob=0.0", "trainor.epochs=2"]

    def spoil(tr):
        fwd = tr.model.forward

        def reading(**batch):
            out = fwd(**batch)
            if "loss" in out:
                float(out["loss"].detach().float().sum())              # the host read
            return out
        tr.model.forward = reading
    t0, l0, p0, s0 = _tiny_trainor(tmp_path, "eager_fb", extra)
    t1, l1, p1, s1 = _tiny_trainor(tmp_path, "graph_fb", extra + ["trainor.graph_step=true"], prepare=spoil)
    assert t1.graph_any and list(t1._graphs.values()) == [False]          # tried once, failed, stays eager
    assert int(t1.optimizer.state_dict()["steps"]) == int(t0.optimizer.state_dict()["steps"]) == 12
    dp = _rel(p1, p0)
    print(f"[parity] Trainor.start() with graph_step whose capture fails: rel L2 of the final parameters vs the eager run {dp:.3e}; "
          f"validation {s1[0]} vs {s0[0]}", flush=True)
    assert dp <= 1e-5 and s1[0] == s0[0]
    # LRU eviction: with room for ONE signature a second batch shape replaces the first (single process)
    t2, _, _, _ = _tiny_trainor(tmp_path, "graph_lru", extra + ["trainor.graph_step=true", "trainor.graph_cache=1"])
    assert t2.max_graphs == 1 and len(t2._graphs) == 1
    key = next(iter(t2._graphs))
    batch = {k: (v[:2] if isinstance(v, torch.Tensor) else v) for k, v in next(iter(t2.dl)).items()}       # another batch size = another signature
    t2._graphed_iteration(batch)
    assert len(t2._graphs) == 1 and next(iter(t2._graphs)) != key


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_trainor_start_on_rccl_equals_single_process(nccl, tmp_path, monkeypatch, wire):
    """ref: trainor_accelerate.py:91-93,111-156.  Trainor.start() (two epochs of 4 iterations, validation with beam search, checkpoint)
    once without a process group and once with VM_FORCE_DDP on the 1-rank RCCL group: ArenaDDP's two-phase backward, the collective
    NaN flag (MIN all-reduce), mean_over_ranks and gather_interleaved of the validator all run on the communicator.  fp32 wire: the
    loss trajectory and the final parameters agree to float-atomic noise; bf16 wire: one rounding of each gradient per step."""
    monkeypatch.delenv("VM_FORCE_DDP", raising=False)
    tr0, l0, p0, s0 = _tiny_trainor(tmp_path, "plain")
    assert tr0.ddp is None
    monkeypatch.setenv("VM_FORCE_DDP", "1")
    tr1, l1, p1, s1 = _tiny_trainor(tmp_path, "ddp_" + wire, [f"trainor.ddp_wire={wire}"])
    assert tr1.ddp is not None and tr1.dist is nccl and tr1.ddp.bf16_wire == (wire == "bf16")
    assert tr1.ddp.split_at is not None and tr1.ddp.mark_starts >= 1
    assert l0.numel() == l1.numel() == 8
    err_l, err_p = (l1 - l0).abs().max().item(), _rel(p1, p0)
    print(f"[parity] Trainor.start() on 1-rank RCCL, wire={wire}: max |loss_t - loss_t(single)| {err_l:.3e}, rel L2 of the final parameters {err_p:.3e}; "
          f"validation {s1[0]} vs {s0[0]}", flush=True)
    if wire == "fp32":          # (not bit-for-bit: the loss is a float atomicAdd over rows, and the two-phase backward regroups the flushes)
        assert err_l <= 2e-4 and err_p <= 2e-3
        assert s1[0] == s0[0]                       # the validator's scores (beam-search reports gathered over the group)
    else:
        assert err_l <= 5e-2 and err_p <= 2e-2
    assert len([f for f in os.listdir(tmp_path / ("ddp_" + wire)) if f.endswith(".pth")]) == 1


def test_rrg_scst_trainor_on_rccl_equals_single_process(nccl, tmp_path, monkeypatch):
    """BASELINE configs[4] as it is worded -- RRG + SCST under data parallelism (ref: vilmedic/models/rrg/RRG_SCST.py:59-85 driven by
    trainor_accelerate.py:111-156): ``Trainor.start()`` on config/RRG/rrg-scst-synthetic.yml (tiny sizes) once without a process group and once
    with VM_FORCE_DDP on the 1-rank RCCL group.  RRG_SCST forwards enc / dec / split_backward to ArenaDDP, so the policy-gradient step takes
    the two-phase backward (decoder range reduced while the ViT backward runs, encoder buckets started from the backward marks); same seeds ->
    same rollouts -> the same loss trajectory and parameters (fp32 wire)."""
    extra = ["model.decoder.hidden_dropout_prob=0.0", "model.decoder.attention_probs_dropout_prob=0.0", "trainor.optim_params.lr=0.0003",
             "model.top_k=8"]
    monkeypatch.delenv("VM_FORCE_DDP", raising=False)
    tr0, l0, p0, _ = _tiny_trainor(tmp_path, "scst_plain", extra, yml="rrg-scst-synthetic.yml")
    assert tr0.ddp is None
    monkeypatch.setenv("VM_FORCE_DDP", "1")
    tr1, l1, p1, _ = _tiny_trainor(tmp_path, "scst_ddp", extra + ["trainor.ddp_wire=fp32"], yml="rrg-scst-synthetic.yml")
    assert tr1.ddp is not None and tr1.ddp.split_at is not None and tr1.model.split_backward and tr1.ddp.mark_starts >= 1
    assert l0.numel() == l1.numel() and l0.numel() >= 4
    err_l, err_p = (l1 - l0).abs().max().item(), _rel(p1, p0)
    print(f"[parity] RRG_SCST Trainor.start() on 1-rank RCCL (two-phase backward): max |loss_t - loss_t(single)| {err_l:.3e} over {l0.numel()} steps, "
          f"rel L2 of the final parameters {err_p:.3e}", flush=True)
    assert err_l <= 2e-4 and err_p <= 2e-3


def test_convirt_forward_all_gathers_negatives_on_rccl(nccl, monkeypatch):
    """ConVIRT.forward in training mode with the contrastive-negatives exchange (parallel.all_gather_with_grad: all_gather forward,
    all-reduce + own slice backward; ref: SURVEY §8e row 2) on the 1-rank RCCL group == the same forward without a group."""
    from vilmedic_amd.models import ConVIRT
    from vilmedic_amd import parallel
    TXT = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=211, max_position_embeddings=40,
               layer_norm_eps=1e-5, bos_token_id=0, pad_token_id=1, eos_token_id=2)
    RESNET = dict(num_channels=3, embedding_size=16, hidden_sizes=[16, 32], depths=[1, 1], layer_type="basic", hidden_act="relu")
    torch.manual_seed(7)
    model = ConVIRT(encoder=dict(proto=None, add_pooling_layer=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **TXT),
                    cnn=dict(proto="VisualEncoder", backbone="hfresnet", permute="batch_first", dropout_out=0.0, **RESNET),
                    projection=dict(visual_embedding_dim=32, textual_embedding_dim=128, projection_dim=64),
                    loss=dict(proto="ConVIRTLoss", tau=0.1, lambda_=0.75), forward_batch_size=8).to(dev())
    images = R.make_images(8, 8, seed=3).to(dev())
    ids, am = R.make_reports(8, 16, TXT["vocab_size"], seed=3)
    ids, am = ids.to(dev()), am.to(dev())
    model.train()
    calls = {"n": 0}
    real = parallel._AllGather.forward

    def counting(ctx, x, dist):
        calls["n"] += 1
        return real(ctx, x, dist)
    monkeypatch.setattr(parallel._AllGather, "forward", staticmethod(counting))
    res = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("VM_FORCE_DDP", "1")
        else:
            monkeypatch.delenv("VM_FORCE_DDP", raising=False)
        for p in model.parameters():
            p.grad = None
        from vilmedic_amd.arena import arena_of
        arena_of(model).zero_grad()
        out = model(input_ids=ids, attention_mask=am, images=images)
        out["loss"].backward()
        torch.cuda.synchronize()
        res.append((out["loss"].item(), out["loss_l"].float().clone(), arena_of(model).gflat.clone()))
    assert calls["n"] == 2, calls                      # linguistic + visual embeddings went through the RCCL all-gather exactly once
    (la, ra, ga), (lb, rb, gb) = res
    print(f"[parity] ConVIRT all-gather on 1-rank RCCL: loss {lb:.6f} vs {la:.6f}, gradient rel L2 {_rel(gb, ga):.3e}", flush=True)
    assert abs(la - lb) <= 1e-5 and (ra - rb).abs().max().item() <= 1e-4 and _rel(gb, ga) <= 3e-2      # (one [B,B] problem vs two row-block problems: G is rounded to bf16 in two parts)


def test_gloria_loss_gathers_local_features_on_rccl(nccl, monkeypatch):
    """GLoRIALoss under (forced) data parallelism: global embeddings, the [b, D, h, w] region features, the word embeddings and the word
    lists are gathered -- and with gather_local=False only the two global embeddings.  One rank: all three equal."""
    from vilmedic_amd.blocks.losses import GLoRIALoss
    B, D, T, hw = 6, 32, 9, 5
    gen = torch.Generator().manual_seed(5)
    base = [torch.randn(B, D, generator=gen), torch.randn(B, D, hw, hw, generator=gen), torch.randn(B, D, T, generator=gen),
            torch.randn(B, D, generator=gen)]
    lens = [3, 8, 5, 2, 7, 4]
    sents = [["[CLS]"] + ["w"] * (n - 1) + ["[SEP]"] + ["[PAD]"] * (T - n - 1) for n in lens]
    res = []
    for force, gather_local in ((False, True), (True, True), (True, False)):
        if force:
            monkeypatch.setenv("VM_FORCE_DDP", "1")
        else:
            monkeypatch.delenv("VM_FORCE_DDP", raising=False)
        xs = [t.clone().to(dev()).requires_grad_(True) for t in base]
        loss, attn = GLoRIALoss(1.0, 1.0, 4.0, 5.0, 10.0, gather_local=gather_local)(xs[0], xs[1], xs[2], xs[3], sents)
        loss.backward()
        torch.cuda.synchronize()
        res.append((loss.item(), [x.grad.clone() for x in xs], attn[0].clone()))
    for k in (1, 2):
        assert abs(res[k][0] - res[0][0]) <= 1e-6, (res[k][0], res[0][0])
        assert all(_rel(a, b) <= 1e-5 for a, b in zip(res[k][1], res[0][1])) and _rel(res[k][2], res[0][2]) <= 1e-6
    print(f"[parity] GLoRIALoss gathers on 1-rank RCCL: loss {res[1][0]:.6f} == {res[0][0]:.6f} (local features gathered / global only)", flush=True)
