"""GPU parity of the task-level pieces: contrastive / CE losses, MVQA core, text encoder, SCST loss, Trainor loop."""
import os

import pytest
import torch

import golden_recipes as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("B", [8, 64])
def test_convirt_infonce_lsce_vs_golden(golden, B):
    """loss within 2e-3 abs (bf16 similarity GEMM, fp32 log-sum-exp); gradients within 3e-2 relative L2."""
    from vilmedic_amd.blocks.losses import ConVIRTLoss, InfoNCELoss, LabelSmoothingCrossEntropy
    g = golden("g6_losses")
    gen = torch.Generator().manual_seed(1234 + B)
    l = torch.randn(B, 96, generator=gen)
    v = torch.randn(B, 96, generator=gen)
    ld, vd = l.to(dev()).requires_grad_(True), v.to(dev()).requires_grad_(True)
    ref = g[f"convirt_{B}"]
    loss, ll, lv = ConVIRTLoss(tau=0.1, lambda_=0.75)(ld, vd)
    loss.backward()
    assert abs(loss.item() - ref["loss"].item()) <= 2e-3
    torch.testing.assert_close(ll.cpu(), ref["loss_l"], rtol=0, atol=2e-2)
    torch.testing.assert_close(lv.cpu(), ref["loss_v"], rtol=0, atol=2e-2)
    assert rel(ld.grad.cpu(), ref["gl"]) <= 3e-2 and rel(vd.grad.cpu(), ref["gv"]) <= 3e-2
    l2, v2 = (0.2 * l).to(dev()).requires_grad_(True), (0.2 * v).to(dev()).requires_grad_(True)
    ref = g[f"infonce_{B}"]
    loss, lt, li = InfoNCELoss(tau=0.1)(l2, v2)
    loss.backward()
    assert abs(loss.item() - ref["loss"].item()) <= 3e-3
    torch.testing.assert_close(lt.cpu(), ref["loss_t"], rtol=0, atol=3e-2)
    assert rel(l2.grad.cpu(), ref["gl"]) <= 3e-2 and rel(v2.grad.cpu(), ref["gv"]) <= 3e-2
    logits = torch.randn(B, 33, generator=gen)
    ref = g[f"lsce_{B}"]
    x = logits.to(dev()).requires_grad_(True)
    loss = LabelSmoothingCrossEntropy(smoothing=0.1)(x, ref["target"].to(dev()))
    loss.backward()
    torch.testing.assert_close(loss.cpu(), ref["loss"], rtol=1e-5, atol=1e-5)      # fp32 kernel
    torch.testing.assert_close(x.grad.cpu(), ref["g"], rtol=1e-4, atol=1e-6)


def test_gloria_loss_vs_golden(golden):
    from vilmedic_amd.blocks.losses import GLoRIALoss
    g = golden("g6_losses")["gloria"]
    B, D, T, hw = g["B"], g["D"], g["T"], g["hw"]
    gen = torch.Generator().manual_seed(99)
    glob = torch.randn(B, D, generator=gen).to(dev()).requires_grad_(True)
    loc = torch.randn(B, D, hw, hw, generator=gen).to(dev()).requires_grad_(True)
    words = torch.randn(B, D, T, generator=gen).to(dev()).requires_grad_(True)
    sent = torch.randn(B, D, generator=gen).to(dev()).requires_grad_(True)
    sents = [["[CLS]"] + ["w"] * (n - 1) + ["[SEP]"] + ["[PAD]"] * (T - n - 1) for n in g["cap_lens"]]
    loss, attn = GLoRIALoss(1.0, 1.0, 4.0, 5.0, 10.0)(glob, loc, words, sent, sents)
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) <= 2e-2
    assert rel(loc.grad.cpu(), g["g_loc"]) <= 1e-3 and rel(words.grad.cpu(), g["g_words"]) <= 1e-3     # local part: fp32 HIP kernels (csrc/gloria.hip)
    assert rel(glob.grad.cpu(), g["g_glob"]) <= 3e-2 and rel(sent.grad.cpu(), g["g_sent"]) <= 3e-2     # global part: bf16 GEMM
    torch.testing.assert_close(attn[0].cpu(), g["attn0"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,D,T,hw,seed", [(6, 32, 9, 5, 99), (5, 40, 21, 7, 3), (8, 768, 33, 19, 4)])
def test_gloria_local_loss_kernels_vs_oracle(golden, B, D, T, hw, seed):
    """csrc/gloria.hip (every caption x image pair at once, fp32, ragged caption lengths masked in the kernels) against the oracle's
    restatement of the reference's per-caption loop (ref:GLoRIALoss.py:78-129), forward and backward: the G6 fixture shape (whose
    gradients / first attention map are the reference's own), a shape with D, T, P all off the 16-multiples the GEMMs are padded to, and
    the production feature size (D = 768, 19 x 19 regions).  Caption lengths include 1 and T."""
    from oracle import torch_ref as O
    from vilmedic_amd.blocks.losses import GLoRIALoss
    gen = torch.Generator().manual_seed(seed)
    if seed == 99:
        torch.randn(B, D, generator=gen)                       # the fixture draws the global features first
    loc = torch.randn(B, D, hw, hw, generator=gen)
    words = torch.randn(B, D, T, generator=gen)
    if D >= 256:                                               # unit-variance features of width 768 saturate both softmaxes: scale to O(1) scores
        loc, words = loc * D ** -0.25, words * D ** -0.25
    lens = golden("g6_losses")["gloria"]["cap_lens"] if seed == 99 else ([T, 1, 2, max(1, T - 3)] + [max(1, T // 2)] * B)[:B]
    crit = GLoRIALoss(1.0, 1.0, 4.0, 5.0, 10.0)
    dl, dw = loc.to(dev()).requires_grad_(True), words.to(dev()).requires_grad_(True)
    l0, l1, maps = crit._local(dl, dw, lens)
    (l0 + 2.0 * l1).backward()
    rl, rw = loc.clone().requires_grad_(True), words.clone().requires_grad_(True)
    r0, r1 = O.gloria_local_loss(rl, rw, lens, 4.0, 5.0, 10.0)
    (r0 + 2.0 * r1).backward()
    torch.testing.assert_close(torch.stack([l0, l1]).detach().cpu(), torch.stack([r0, r1]).detach(), rtol=2e-5, atol=2e-5)
    e_loc, e_w = rel(dl.grad.cpu(), rl.grad), rel(dw.grad.cpu(), rw.grad)
    print(f"[parity] gloria local B={B} D={D} T={T} P={hw * hw}: loss {l0.item():.6f}/{l1.item():.6f} vs {r0.item():.6f}/{r1.item():.6f}  "
          f"grad rel-l2 loc {e_loc:.2e} words {e_w:.2e}")
    assert e_loc <= 1e-4 and e_w <= 1e-4
    assert [m.shape for m in maps] == [(1, n, hw, hw) for n in lens]
    assert torch.count_nonzero(dw.grad[1, :, lens[1]:]) == 0             # words past the caption's length get no gradient
    if seed == 99:
        g = golden("g6_losses")["gloria"]
        torch.testing.assert_close(maps[0].cpu(), g["attn0"], rtol=1e-4, atol=1e-6)


def test_mvqa_core_and_text_encoder_vs_golden(golden):
    from vilmedic_amd.arena import arena_of
    from vilmedic_amd.blocks.huggingface.encoder.encoder_model import EncoderModel
    from vilmedic_amd.nn import BERT_GEN_DEFAULTS, BertPooler, BertStack, make_config
    g = golden("g9_mvqa_text")
    m = g["mvqa"]
    cfg = make_config(BERT_GEN_DEFAULTS, dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **m["cfg"]))

    class Core(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer, self.pooler = BertStack(cfg), BertPooler(cfg)
    core = Core().to(dev())
    st = {"transformer." + k: v for k, v in R.rand_state(R.bert_stack_shapes(m["cfg"]), m["seed"]).items()}
    st["pooler.dense.weight"], st["pooler.dense.bias"] = m["pw"], m["pb"]
    core.load_state_dict(st, strict=True)
    core.eval()
    arena = arena_of(core)
    with torch.no_grad():
        h = core.transformer(m["x"].to(dev()).to(BF), arena)
        pooled = core.pooler(h, arena)
    err = (h.float().cpu() - m["hidden"]).abs()
    assert bool((err <= 3e-2 + 3e-2 * m["hidden"].abs()).all())
    torch.testing.assert_close(pooled.cpu(), m["pooled"], rtol=0, atol=2e-2)
    t = g["text"]
    enc = EncoderModel(dict(proto=None, add_pooling_layer=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **t["cfg"])).to(dev())
    sd = R.rand_state(R.text_encoder_shapes(t["cfg"]), t["seed"])
    enc.encoder.load_state_dict(sd, strict=True)
    enc.pooler.load_state_dict({"dense.weight": t["pw"], "dense.bias": t["pb"]})
    enc.eval()
    ids, am = R.make_reports(t["B"], t["L"], t["cfg"]["vocab_size"], seed=t["seed"])
    with torch.no_grad():
        out = enc(input_ids=ids.to(dev()), attention_mask=am.to(dev()))
    err = (out.last_hidden_state.float().cpu() - t["last_hidden_state"]).abs()
    assert bool((err <= 3e-2 + 3e-2 * t["last_hidden_state"].abs()).all())
    torch.testing.assert_close(out["pooler_output"].cpu(), t["pooler_output"], rtol=0, atol=2e-2)


def test_scst_weighted_lm_head_loss_matches_scst_loss_oracle():
    """The fused policy-gradient loss (row weights + banned columns inside the LM-head CE kernel) equals the reference's
    scst_loss applied to log-softmax of the bad-word-masked logits (ref: SCST.py:14-45,150-170)."""
    from oracle import torch_ref as O
    from test_hip_models_gpu import build_decoder
    cfg = R.DEC_TINY
    dec, st = build_decoder(cfg, 5)
    dec.train()
    B, T, S = 4, 12, 7
    gen = torch.Generator().manual_seed(3)
    seq = torch.randint(3, cfg["vocab_size"], (B, T), generator=gen)
    seq[:, 0] = 0
    seq[1, 6], seq[1, 7:] = 2, 1
    seq[3, 3], seq[3, 4:] = 2, 1
    enc = torch.randn(B, S, cfg["hidden_size"], generator=gen)
    rs = [torch.rand(B, generator=gen).tolist()]
    rg = [torch.rand(B, generator=gen).tolist()]
    sampled = seq[:, 1:]
    mask = (sampled > 1).float()
    coef = torch.tensor(rs[0]) - torch.tensor(rg[0])
    row_w = torch.zeros(B, T)
    row_w[:, :-1] = mask * coef[:, None] / mask.sum()
    enc_d = enc.to(dev()).to(BF).requires_grad_(True)
    out = dec.decoder(input_ids=seq.to(dev()), attention_mask=None, encoder_hidden_states=enc_d, encoder_attention_mask=None,
                      labels=seq.to(dev()), return_logits=False, row_weight=row_w.to(dev()), banned=[1, 0])
    out["loss"].backward()
    # oracle
    sto = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    enc_o = enc.clone().requires_grad_(True)
    h = O.decoder_hidden(seq, None, enc_o, None, sto, cfg)
    logits = O.lm_logits(h, sto).float()[:, :-1]
    logits[:, :, [1, 0]] = -float("inf")
    logp = torch.log_softmax(logits, -1).gather(2, sampled.unsqueeze(-1))
    ref = O.scst_loss(logp, sampled, rs, rg, [1.0], 1)
    ref.backward()
    assert abs(out["loss"].item() - ref.item()) <= 2e-3 * max(1.0, abs(ref.item())), (out["loss"].item(), ref.item())
    lp_hip = out["row_logp"][:, :-1].cpu()
    assert (lp_hip - logp.squeeze(-1).detach()).abs()[mask.bool()].max() <= 5e-2
    assert rel(enc_d.grad.float().cpu(), enc_o.grad) <= 5e-2
    gname = "bert.encoder.layer.1.output.dense.weight"
    assert rel(dict(dec.decoder.named_parameters())[gname].grad.cpu(), sto[gname].grad) <= 5e-2


def test_rrg_scst_step_and_validator_decode_run():
    """RRG_SCST forward (greedy baseline + sampling rollout + fused policy-gradient loss) and the decode driver run end to
    end on synthetic data; the loss is finite and gradients reach the encoder."""
    from vilmedic_amd.blocks.huggingface.decoder.evaluation import evaluation
    from vilmedic_amd.config import Cfg
    from vilmedic_amd.datasets import SyntheticImSeq
    from vilmedic_amd.models import RRG_SCST
    ds = SyntheticImSeq(num_samples=8, image_size=32, vocab_size=97, tokenizer_max_len=16)
    dl = torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=ds.get_collate_fn())
    model = RRG_SCST(decoder=dict(proto=None, **R.DEC_TINY), cnn=dict(proto="VisualEncoder", backbone="vit", permute="no_permute",
                                                                       **R.VIT_TINY), dl=dl, scores="ROUGEL", top_k=20).to(dev())
    batch = next(iter(dl))
    out = model(**batch)
    assert torch.isfinite(out["loss"])
    out["loss"].backward()
    p = model.model.enc.model.encoder.layer[0].intermediate.dense.weight
    assert p.grad.abs().sum() > 0
    res = evaluation([model.model.eval()], Cfg(beam_width=2, length_penalty=None), dl)
    assert len(res["hyps"]) == 8 and len(res["refs"]) == 8


def test_trainor_runs_and_checkpoints(tmp_path):
    from vilmedic_amd.config import executor_view, get_config
    from vilmedic_amd.executors import Trainor
    import os
    cfg = get_config(os.path.join(os.path.dirname(__file__), "..", "config", "RRG", "rrg-vit-synthetic.yml"),
                     ["dataset.num_samples=16", "dataset.image_size=32", "dataset.vocab_size=97", "dataset.tokenizer_max_len=16",
                      "model.decoder.hidden_size=128", "model.decoder.num_attention_heads=2", "model.decoder.intermediate_size=256",
                      "model.decoder.num_hidden_layers=2", "model.decoder.max_position_embeddings=64",
                      "model.cnn.image_size=32", "model.cnn.patch_size=8", "model.cnn.hidden_size=128", "model.cnn.num_attention_heads=2",
                      "model.cnn.intermediate_size=256", "model.cnn.num_hidden_layers=2",
                      "trainor.batch_size=4", "trainor.epochs=2", "trainor.eval_start=0", "validator.batch_size=4",
                      "validator.beam_width=2", f"ckpt_dir={tmp_path}", "trainor.optim_params.lr=0.003"])
    t = executor_view(cfg, "trainor")
    t["validator_view"] = executor_view(cfg, "validator")
    tr = Trainor(t, seed=0)
    tr.start()
    ckpts = [f for f in os.listdir(tmp_path) if f.endswith(".pth")]
    assert len(ckpts) == 1
    sd = torch.load(os.path.join(tmp_path, ckpts[0]), map_location="cpu")
    assert {"model", "training_scheduler", "optimizer", "config", "__version__"} <= set(sd)


def _tiny_rrg_cfg(tmp_path, extra):
    from vilmedic_amd.config import get_config
    return get_config(os.path.join(os.path.dirname(__file__), "..", "config", "RRG", "rrg-vit-synthetic.yml"),
                      ["dataset.num_samples=16", "dataset.image_size=32", "dataset.vocab_size=97", "dataset.tokenizer_max_len=16",
                       "model.decoder.hidden_size=128", "model.decoder.num_attention_heads=2", "model.decoder.intermediate_size=256",
                       "model.decoder.num_hidden_layers=1", "model.decoder.max_position_embeddings=64",
                       "model.cnn.image_size=32", "model.cnn.patch_size=8", "model.cnn.hidden_size=128", "model.cnn.num_attention_heads=2",
                       "model.cnn.intermediate_size=256", "model.cnn.num_hidden_layers=1",
                       "trainor.batch_size=4", "validator.batch_size=4", "validator.beam_width=1", f"ckpt_dir={tmp_path}",
                       "trainor.optim_params.lr=0.003"] + extra)


def test_trainor_keeps_the_reference_loop_semantics(tmp_path):
    """ref:vilmedic/executors/trainor.py:95-203: (a) with grad_accu = 3 and 4 iterations per epoch the optimizer steps at
    iteration 3 AND once more on the left-over micro-batch at the end of the epoch (:139-150) -> 2 steps per epoch; (b) evaluation starts
    when epoch + 1 >= eval_start and the checkpoint carries epoch + 1 in its name; (c) an early_stop_metric the validator does not
    produce raises KeyError instead of silently never saving."""
    from vilmedic_amd.config import executor_view
    from vilmedic_amd.executors import Trainor
    cfg = _tiny_rrg_cfg(tmp_path / "a", ["trainor.epochs=1", "trainor.grad_accu=3", "trainor.eval_start=2", "trainor.early_stop_metric=ROUGEL"])
    os.makedirs(tmp_path / "a")
    t = executor_view(cfg, "trainor")
    t["validator_view"] = executor_view(cfg, "validator")
    tr = Trainor(t, seed=0)
    assert len(tr.dl) == 4
    tr.start()                                          # epochs 0 and 1 (range(0, epochs + 1) as in the reference)
    assert tr.optimizer.steps == 2 * 2, tr.optimizer.steps             # (a): iteration 3 + the left-over update, per epoch
    ckpts = sorted(f for f in os.listdir(tmp_path / "a") if f.endswith(".pth"))
    assert len(ckpts) == 1 and ckpts[0].split("_")[-2] == "2", ckpts   # (b): only epoch index 1 was evaluated, named epoch + 1 = 2
    os.makedirs(tmp_path / "b")
    cfg = _tiny_rrg_cfg(tmp_path / "b", ["trainor.epochs=0", "trainor.eval_start=0", "trainor.early_stop_metric=no_such_metric"])
    t = executor_view(cfg, "trainor")
    t["validator_view"] = executor_view(cfg, "validator")
    with pytest.raises(KeyError, match="no_such_metric"):
        Trainor(t, seed=0).start()                      # (c)


def test_trainor_skips_a_nan_batch_on_the_device(tmp_path):
    """the reference drops a batch whose loss is NaN / Inf (trainor.py:109-112).  Here the decision is taken inside the fused optimizer
    kernel from a device scalar (no host read of the loss per iteration): the poisoned iteration must leave every parameter untouched,
    must not count as an optimizer step for Adam's bias correction, and the epoch's mean loss must ignore it."""
    from vilmedic_amd.arena import arena_of
    from vilmedic_amd.config import executor_view
    from vilmedic_amd.executors import Trainor
    os.makedirs(tmp_path / "n")
    cfg = _tiny_rrg_cfg(tmp_path / "n", ["trainor.epochs=0", "trainor.eval_start=99"])
    tr = Trainor(executor_view(cfg, "trainor"), seed=0)
    assert tr.device_gate
    arena, seen = arena_of(tr.model), []
    fwd = tr.model.forward

    def poisoned(**batch):
        out = fwd(**batch)
        seen.append(arena.flat.clone())
        if len(seen) == 2:                                   # second iteration: NaN loss (and NaN gradients)
            out["loss"] = out["loss"] * float("nan")
        return out
    tr.model.forward = poisoned
    tr.start()
    assert len(seen) == 4
    assert not torch.equal(seen[1], seen[0])                # iteration 1 updated the weights
    assert torch.equal(seen[2], seen[1])                    # iteration 2 (NaN) did not
    assert not torch.equal(seen[3], seen[2]) and torch.isfinite(arena.flat).all()
    assert int(tr.optimizer.step_dev.item()) == 3           # three real steps out of four iterations


def test_gloria_model_vs_oracle():
    """GLoRIA: CNN tower (MIOpen) + text tower, HIP embedders, on-device word-piece merge, GLoRIALoss --
    against the oracle composition on the CPU (same CNN module class in fp32)."""
    import types
    from oracle import torch_ref as O
    from vilmedic_amd.blocks.vision import VisualEncoder
    from vilmedic_amd.models import GLoRIA
    vocab = ["[CLS]", "[PAD]", "[SEP]"] + [f"w{i}" for i in range(40)] + [f"##p{i}" for i in range(20)]
    V = len(vocab)
    txt = dict(R.TXT_TINY, vocab_size=V, pad_token_id=1)
    tok = types.SimpleNamespace(get_vocab=lambda: {w: i for i, w in enumerate(vocab)})
    dl = types.SimpleNamespace(dataset=types.SimpleNamespace(tokenizer=tok))
    torch.manual_seed(3)
    cnn = dict(proto="VisualEncoder", backbone="resnet50", output_layer="avgpool", dropout_out=0.0, permute="batch_first", freeze=False)
    model = GLoRIA(encoder=dict(proto=None, last_n_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **txt),
                   cnn=dict(cnn), visual_embedder=dict(interm_feature_dim=1024, feature_dim=2048), loss=dict(), dl=dl,
                   forward_batch_size=3).to(dev())
    model.train()
    B, L = 5, 14
    g = torch.Generator().manual_seed(9)
    ids = torch.full((B, L), 1, dtype=torch.long)
    am = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(5, L - 1, (1,), generator=g))
        ids[b, 0] = 0
        ids[b, 1:n] = torch.randint(3, V, (n - 1,), generator=g)
        ids[b, 1] = 3 + b                                    # the word after [CLS] is never a ## piece
        ids[b, n] = 2
        am[b, :n + 1] = 1
    images = torch.randn(B, 3, 64, 64, generator=g)
    out = model(input_ids=ids.to(dev()), attention_mask=am.to(dev()), images=images.to(dev()))
    st = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    vis = VisualEncoder(**{k: v for k, v in cnn.items() if k != "proto"})
    vis.load_state_dict({k[len("visual."):]: v for k, v in st.items() if k.startswith("visual.")})
    vis.train()
    ref_loss, gf, lf, word, sent = O.gloria_forward(images, ids, am, st, txt, vis.model, dict(enumerate(vocab)), 2, 3)
    assert rel(out["global_features"].float().cpu(), gf) <= 3e-2
    assert rel(out["local_features"].float().cpu(), lf) <= 3e-2
    assert rel(out["word_embeddings"].float().cpu(), word) <= 3e-2
    assert rel(out["sent_embeddings"].float().cpu(), sent) <= 3e-2
    assert abs(out["loss"].item() - ref_loss.item()) <= 3e-2 * max(1.0, abs(ref_loss.item())), (out["loss"].item(), ref_loss.item())
    out["loss"].backward()
    for name in ["global_embedder.weight", "local_embedder.weight", "visual.model.0.weight",
                 "linguistic.encoder.encoder.layer.0.attention.self.query.weight", "linguistic.encoder.embeddings.word_embeddings.weight"]:
        p = dict(model.named_parameters())[name]
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum().item() > 0, name


@pytest.mark.parametrize("rel,small", [
    ("RRG/rrg-hf-synthetic.yml", ["dataset.image_size=32", "dataset.vocab_size=97", "dataset.tokenizer_max_len=16",
                                  "model.vision.proto_config_args.image_size=32", "model.vision.proto_config_args.patch_size=8",
                                  "model.vision.proto_config_args.hidden_size=128", "model.vision.proto_config_args.num_attention_heads=2",
                                  "model.vision.proto_config_args.intermediate_size=256", "model.vision.proto_config_args.num_hidden_layers=2",
                                  "model.decoder.proto_config_args.hidden_size=128", "model.decoder.proto_config_args.num_attention_heads=2",
                                  "model.decoder.proto_config_args.intermediate_size=256", "model.decoder.proto_config_args.num_hidden_layers=2",
                                  "model.decoder.proto_config_args.max_position_embeddings=64", "validator.beam_width=2"]),
    ("SELFSUP/convirt-synthetic.yml", ["dataset.image_size=64", "dataset.vocab_size=97", "dataset.tokenizer_max_len=16",
                                       "model.encoder.hidden_size=128", "model.encoder.num_attention_heads=2", "model.encoder.intermediate_size=256",
                                       "model.encoder.num_hidden_layers=2", "model.encoder.max_position_embeddings=64", "model.encoder.vocab_size=97",
                                       "model.projection.textual_embedding_dim=128", "model.projection.projection_dim=128"]),
    ("SELFSUP/gloria-synthetic.yml", ["dataset.image_size=64", "dataset.vocab_size=97", "dataset.tokenizer_max_len=16",
                                      "model.encoder.hidden_size=128", "model.encoder.num_attention_heads=2", "model.encoder.intermediate_size=256",
                                      "model.encoder.num_hidden_layers=2", "model.encoder.max_position_embeddings=64", "model.encoder.vocab_size=97",
                                      "model.encoder.last_n_layers=2", "model.forward_batch_size=4"]),
    ("MVQA/vqa-synthetic.yml", ["dataset.image_size=64", "model.transformer.num_hidden_layers=2"]),
    ("RRS/rrs-synthetic.yml", ["dataset.vocab_size=97", "dataset.src_max_len=24", "dataset.tgt_max_len=12"] + [
        f"model.{side}.{k}" for side in ("encoder", "decoder") for k in ("hidden_size=128", "num_attention_heads=2", "intermediate_size=256",
                                                                          "num_hidden_layers=2", "max_position_embeddings=64")]),
])
def test_trainor_runs_every_task_config(tmp_path, rel, small):
    """bin/train.py's path (Trainor + Validator) on a reduced version of each shipped task config: loss finite and decreasing
    machinery intact (checkpoint written)."""
    import os
    from vilmedic_amd.config import executor_view, get_config
    from vilmedic_amd.executors import Trainor
    cfg = get_config(os.path.join(os.path.dirname(__file__), "..", "config", rel),
                     small + ["dataset.num_samples=16", "trainor.batch_size=8", "trainor.epochs=1", "trainor.eval_start=0",
                              "validator.batch_size=8", f"ckpt_dir={tmp_path}"])
    t = executor_view(cfg, "trainor")
    t["validator_view"] = executor_view(cfg, "validator")
    tr = Trainor(t, seed=0)
    tr.start()
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".pth")]) == 1


def test_vicreg_loss_vs_golden(golden):
    """VICREGLoss (exported by the reference's losses package): invariance / variance terms in fp32, both [D,D] covariance products
    and their gradients on the bf16 MFMA GEMM.  Loss within 2e-3 relative, covariance term within 5e-3, gradients within 1e-2."""
    from vilmedic_amd.blocks.losses import VICREGLoss
    for case in golden("g14_vicreg").values():
        N, D = case["N"], case["D"]
        gen = torch.Generator().manual_seed(4321 + N)
        z1 = 0.7 * torch.randn(N, D, generator=gen) + 0.1
        z2 = z1 + 0.3 * torch.randn(N, D, generator=gen)
        a, b = z1.to(dev()).requires_grad_(True), z2.to(dev()).requires_grad_(True)
        crit = VICREGLoss(sim_loss_weight=25.0, var_loss_weight=25.0, cov_loss_weight=1.0)
        loss = crit(a, b)
        loss.backward()
        assert abs(loss.item() - case["loss"].item()) <= 2e-3 * abs(case["loss"].item()), (loss.item(), case["loss"].item())
        cov = VICREGLoss.covariance_loss(a.detach(), b.detach()).item()
        assert abs(cov - case["cov"].item()) <= 5e-3 * case["cov"].item(), (cov, case["cov"].item())
        assert rel(a.grad.cpu(), case["g1"]) <= 1e-2 and rel(b.grad.cpu(), case["g2"]) <= 1e-2


def test_zoo_checkpoint_round_trip_generates_the_same_reports(tmp_path):
    """SURVEY §8(f) rank 4 on the GPU: a model trained here -> the reference's checkpoint wire format (old DataParallel prefix and
    pre-1.3.2 encoder names, ref:vilmedic/executors/utils.py:26-34) -> a zoo directory -> AutoModel.from_pretrained -> the loaded model
    gives the same loss and the same greedy / beam-2 reports on raw inputs (image FILES through the device pipeline, raw sentences through
    the tokenizer) as the model it was saved from.  (Published zoo checkpoints cannot be fetched here.)"""
    import copy
    import types
    import yaml
    from test_datasets import _make_corpus
    from test_zoo import _model_cfg
    from vilmedic_amd import models as M
    from vilmedic_amd.datasets import ImSeq
    from vilmedic_amd.optim import FusedAdam
    from vilmedic_amd.zoo import AutoModel
    root, zoo = str(tmp_path / "data"), str(tmp_path / "zoo" / "rrg-tiny")
    os.makedirs(root), os.makedirs(zoo)
    _make_corpus(root)
    seq = dict(root=root, file="report.tok", tokenizer=None, tokenizer_max_len=12, processing="r2gen_clean_report", source="tgt")
    image = dict(root=root, file="image.tok", image_path=root, resize=40, crop=32, ext=".png")
    train = ImSeq(seq=seq, image=image, split="train", ckpt_dir=zoo)
    cfg = _model_cfg()
    mc = copy.deepcopy(cfg)
    torch.manual_seed(0)
    model = getattr(M, mc.pop("proto"))(**mc, dl=types.SimpleNamespace(dataset=train)).to(dev())
    files = [os.path.join(root, "img", f"validate_{i}.png") for i in range(3)]
    sents = ["The heart is normal. 2. No effusion.", "clear lungs", "small effusion the heart is enlarged"]
    # a few optimizer steps so that the weights are not the initialisation
    model.train()
    opt = FusedAdam(model, lr=1e-3)
    for _ in range(3):
        batch = {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in train.inference(seq=sents, image=files).items()}
        opt.zero_grad()
        model(**batch)["loss"].backward()
        opt.step()
    model.eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    old = {"module." + k.replace("enc.model.", "enc.0.cnn."): v.clone() for k, v in sd.items()}
    torch.save({"model": old, "__version__": "1.2.9"}, os.path.join(zoo, "0.5_3_0.pth"))
    yaml.safe_dump({"name": "rrg_tiny", "model": cfg,
                    "dataset": {"proto": "ImSeq",
                                "seq": {"vocab_file": "vocab.tgt", "tokenizer_max_len": 12, "processing": "r2gen_clean_report", "source": "tgt"},
                                "image": {"resize": 40, "crop": 32, "ext": ".png"}}}, open(os.path.join(zoo, "config.yml"), "w"))
    loaded, dataset = AutoModel.from_pretrained(zoo)
    loaded = loaded.to(dev()).eval()
    # raw inputs through the LOADED dataset (evaluation transform: resize to the crop size, no random crop / flip; the training dataset
    # above draws augmentations) -- token ids must be those of the training tokenizer
    b2 = {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in dataset.inference(seq=sents, image=files).items()}
    assert torch.equal(b2["input_ids"].cpu(), train.seq.inference(sents)["input_ids"]) and b2["images"].shape == (3, 3, 32, 32)
    b1 = b2
    with torch.no_grad():
        l1, l2 = model(**b1)["loss"].item(), loaded(**b2)["loss"].item()
        assert abs(l1 - l2) <= 1e-5 * abs(l1), (l1, l2)           # (the CE kernel sums the rows' losses with atomics: order-dependent last bits)
        for beams in (1, 2):                       # greedy and beam-2 reports through the models' own encode + generate
            outs = []
            for m, b in ((model, b1), (loaded, b2)):
                hs, hmask = m.encode(images=b["images"])
                outs.append(m.dec.generate(encoder_hidden_states=hs, encoder_attention_mask=hmask, num_beams=beams, max_length=12).cpu())
            assert torch.equal(outs[0], outs[1]), (beams, outs)
    assert outs[0].shape[0] == 3 and outs[0].shape[1] <= 12
