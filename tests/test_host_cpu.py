"""CPU tests (no GPU needed): C-ABI library loads and exports every symbol include/vmhip.h declares, module trees carry
the reference's parameter names, config loader semantics, scorers, fail-loud behaviour without a device."""
import ctypes
import os
import re

import pytest
import torch

import golden_recipes as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vilmedic_amd import _lib, build
    path = build.build(verbose=False)
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "vmhip.h")).read()
    declared = set(re.findall(r"\b(vm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vmhip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib.vm_sizeof_gemm_epilogue.restype = ctypes.c_int
    assert lib.vm_sizeof_gemm_epilogue() == ctypes.sizeof(_lib.GemmEpilogue)      # struct layout agreed with the C side


def test_ops_fail_loudly_on_cpu_tensors():
    from vilmedic_amd import ops
    from vilmedic_amd._lib import VmHipError
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(VmHipError):
        ops.gemm(a, 0, a, 0, torch.zeros(8, 8, dtype=torch.bfloat16), 8, 8, 8)


def test_rrs_parameter_names_match_reference_checkpoints():
    """RRS = ``enc`` (EncoderModel.encoder = BertGenerationEncoder) + ``dec`` (DecoderModel.decoder), ref models/rrs/RRS.py:15-23"""
    from vilmedic_amd.models import RRS
    m = RRS(encoder=dict(proto=None, **R.TXT_TINY), decoder=dict(proto=None, **R.DEC_TINY))
    want = {"enc.encoder." + k for k in R.text_encoder_shapes(R.TXT_TINY)} | {"dec.decoder." + k for k in R.decoder_shapes(R.DEC_TINY)}
    assert set(m.state_dict()) == want | {"dec.decoder.lm_head.decoder.weight", "dec.decoder.lm_head.decoder.bias"}


def test_module_parameter_names_match_reference_checkpoints():
    from vilmedic_amd.blocks.huggingface.decoder.decoder_model import DecoderModel
    from vilmedic_amd.blocks.vision import VisualEncoder
    enc = VisualEncoder(backbone="vit", permute="no_permute", visual_projection=dict(in_features=128, out_features=64), **R.VIT_TINY)
    assert set(enc.state_dict()) == {"model." + k for k in R.vit_shapes(R.VIT_TINY)} | {"visual_projection.weight", "visual_projection.bias"}
    dec = DecoderModel(dict(proto=None, **R.DEC_TINY))
    names = set(dec.decoder.state_dict())
    assert names == set(R.decoder_shapes(R.DEC_TINY)) | {"lm_head.decoder.weight", "lm_head.decoder.bias"}
    # 12-layer decoder: 317 distinct tensors
    big = DecoderModel(dict(proto=None, hidden_size=64, num_attention_heads=1, intermediate_size=64, num_hidden_layers=12,
                            vocab_size=50, max_position_embeddings=16))
    assert len(list(big.parameters())) == 317
    assert dec.decoder.lm_head.decoder.weight is dec.decoder.bert.embeddings.word_embeddings.weight      # tied head
    for k, shape in R.decoder_shapes(R.DEC_TINY).items():
        assert tuple(dec.decoder.state_dict()[k].shape) == tuple(shape), k


def test_rrg_hf_parameter_names_match_hf_vision_encoder_decoder():
    """RRG_HF: same state-dict keys and shapes as transformers' VisionEncoderDecoderModel built from the
    same two configs (ref: models/rrg/RRG_HF.py:27-92), including the default ViT pooler and enc_to_dec_proj."""
    transformers = pytest.importorskip("transformers")
    from vilmedic_amd.models import RRG_HF
    venc = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=96, image_size=32, patch_size=16)
    vdec = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=160, vocab_size=70,
                max_position_embeddings=24, bos_token_id=2, eos_token_id=1, pad_token_id=0)
    mine = RRG_HF(vision=dict(proto_model="vit", proto_config="vit", proto_config_args=venc),
                  decoder=dict(proto_model="bert-generation", proto_config="bert-generation", proto_config_args=dict(vdec)))
    enc = transformers.ViTModel(transformers.ViTConfig(**venc))
    dec = transformers.BertGenerationDecoder(transformers.BertGenerationConfig(is_decoder=True, add_cross_attention=True, **vdec))
    ref = transformers.VisionEncoderDecoderModel(encoder=enc, decoder=dec)
    ref_sd = {k: tuple(v.shape) for k, v in ref.state_dict().items()}

    def to_installed_hf(name):      # this build keeps the pinned 4.55.3 ViT names; transformers >= 5 renamed them
        if not name.startswith("encoder.") or "encoder.layers.0.attention.q_proj.weight" not in ref_sd:
            return name
        for a, b in (("encoder.encoder.layer.", "encoder.layers."), ("attention.attention.query", "attention.q_proj"),
                     ("attention.attention.key", "attention.k_proj"), ("attention.attention.value", "attention.v_proj"),
                     ("attention.output.dense", "attention.o_proj"), ("intermediate.dense", "mlp.fc1"), ("output.dense", "mlp.fc2")):
            name = name.replace(a, b)
        return name

    my_sd = {to_installed_hf(k): tuple(v.shape) for k, v in mine.model.state_dict().items()}
    assert set(my_sd) == set(ref_sd), (sorted(set(my_sd) ^ set(ref_sd))[:8])
    for k in ref_sd:
        assert my_sd[k] == ref_sd[k], k
    assert "enc_to_dec_proj.weight" in my_sd and "encoder.pooler.dense.weight" in my_sd
    with pytest.raises(NotImplementedError):
        RRG_HF(encoderdecoder="some/pretrained-name")


def test_rrg_hf_builds_from_local_checkpoint_directories(tmp_path):
    """RRG_HF's pretrained arguments (ref:models/rrg/RRG_HF.py:24-25, 48-49, 86-87) on local directories: ``encoderdecoder=<dir>`` rebuilds
    both towers from the nested config and loads every tensor (tied LM head included); ``vision`` / ``decoder`` strings go through the
    AutoModel / AutoModelForCausalLM loaders; a name that is not on disk raises (the package never downloads)."""
    import golden_recipes as R
    from vilmedic_amd.models import RRG_HF
    dcfg = R.DEC_TINY
    vcfg = dict(R.VIT_TINY, hidden_size=64, intermediate_size=128, num_attention_heads=1)
    vst = R.rand_state(R.vit_pooled_shapes(vcfg), 1)
    dst = R.rand_state(R.decoder_shapes(dcfg), 2)
    state = {"encoder." + k: v for k, v in vst.items()}
    state.update({"decoder." + k: v for k, v in dst.items()})
    state["enc_to_dec_proj.weight"] = torch.randn(dcfg["hidden_size"], vcfg["hidden_size"])
    state["enc_to_dec_proj.bias"] = torch.randn(dcfg["hidden_size"])
    d = R.write_ved_dir(str(tmp_path / "ved"), "vit", vcfg, dcfg, state)
    m = RRG_HF(encoderdecoder=d)
    sd = m.model.state_dict()
    for k, v in state.items():
        assert torch.equal(sd[k].float().cpu(), v), k
    assert torch.equal(sd["decoder.lm_head.decoder.weight"].float().cpu(), dst["bert.embeddings.word_embeddings.weight"])
    assert m.model._vm_missing_keys == [] and m.model._vm_unexpected_keys == []
    assert m.model.decoder.config.is_decoder and m.model.decoder.config.add_cross_attention
    assert m.model.config.decoder_start_token_id == dcfg["bos_token_id"] and m.model.config.pad_token_id == dcfg["pad_token_id"]
    # DeiT encoder inside the container
    dvst = R.rand_state(R.deit_shapes(R.DEIT_TINY), 3)
    dstate = {"encoder." + k: v for k, v in dvst.items()}
    dstate.update({"decoder." + k: v for k, v in dst.items()})
    dd = R.write_ved_dir(str(tmp_path / "ved_deit"), "deit", R.DEIT_TINY, dcfg, dstate)
    md = RRG_HF(encoderdecoder=dd)
    assert md.model.encoder.n_special == 2 and "encoder.embeddings.distillation_token" in md.model.state_dict()
    assert md.model._vm_missing_keys == ["encoder.pooler.dense.bias", "encoder.pooler.dense.weight"]      # fresh pooler, as HF does
    # strings
    vst2 = R.rand_state(R.vit_pooled_shapes(R.VIT_TINY), 4)
    dv = R.write_proto_dir(str(tmp_path / "vit"), "vit", dict(R.VIT_TINY), vst2)
    dc = R.write_proto_dir(str(tmp_path / "dec"), "bert-generation", dict(dcfg, is_decoder=True, add_cross_attention=True), dst)
    m2 = RRG_HF(vision=dv, decoder=dc)
    sd2 = m2.model.state_dict()
    for k, v in vst2.items():
        assert torch.equal(sd2["encoder." + k].float().cpu(), v), k
    for k, v in dst.items():
        assert torch.equal(sd2["decoder." + k].float().cpu(), v), k
    assert not hasattr(m2.model, "enc_to_dec_proj")
    with pytest.raises(NotImplementedError):
        RRG_HF(vision="google/vit-not-on-disk", decoder=dc)


def test_cnn_backbones_have_torchvision_names_and_shapes():
    from vilmedic_amd.blocks.vision import VisualEncoder
    enc = VisualEncoder(backbone="resnet18", permute="batch_first", output_layer="layer4", pretrained=False)
    keys = set(enc.state_dict())
    assert "model.0.weight" in keys and "model.4.0.conv1.weight" in keys          # nn.Sequential truncation, as the reference
    with torch.no_grad():
        assert enc.model(torch.zeros(1, 3, 64, 64)).shape == (1, 512, 2, 2)
    dn = VisualEncoder(backbone="densenet169", permute="batch_first", output_layer="features", pretrained=False)
    with torch.no_grad():
        assert dn.model(torch.zeros(1, 3, 64, 64)).shape == (1, 1664, 2, 2)
    rn = VisualEncoder(backbone="resnet50", permute="batch_first", output_layer="avgpool", pretrained=False)
    with torch.no_grad():
        assert rn.model(torch.zeros(1, 3, 64, 64)).flatten(1).shape == (1, 2048)


def test_config_loader_includes_dotlist_and_coercion(tmp_path):
    from vilmedic_amd.config import executor_view, get_config
    (tmp_path / "base.yml").write_text("model:\n  proto: RRG\n  decoder:\n    layer_norm_eps: 1e-05\n    hidden_size: 768\ntrainor:\n  batch_size: 16\n")
    (tmp_path / "child.yml").write_text("includes:\n  - base.yml\nmodel:\n  decoder:\n    hidden_size: 128\ntrainor:\n  optim_params:\n    lr: '5e-5'\n")
    c = get_config(str(tmp_path / "child.yml"), ["trainor.batch_size=4", "model.decoder.proto=null"])
    assert c.model.proto == "RRG" and c.model.decoder.hidden_size == 128 and c.model.decoder.layer_norm_eps == 1e-5
    assert c.trainor.batch_size == 4 and c.trainor.optim_params.lr == 5e-5 and c.model.decoder.proto is None
    v = executor_view(c, "trainor")
    assert v.batch_size == 4 and v.model.proto == "RRG"
    d = c.model.decoder
    d2 = dict(d)
    assert d2.pop("proto") is None and "proto" in d          # sub-trees support ** / pop like a DictConfig


def test_rouge_l_and_reward_table():
    from vilmedic_amd.blocks.scorers import REWARD_COMPLIANT, RougeL
    mean, per = RougeL()(["the heart is normal", "no pleural effusion"], ["the heart is normal", "effusion pleural no"])
    assert per[0] == 1.0 and 0 < per[1] < 1 and abs(mean - sum(per) / 2) < 1e-12
    assert REWARD_COMPLIANT["rougel"][1] == 1


def test_bit_parallel_lcs_equals_the_table():
    """ROUGE-L's longest common subsequence runs bit-parallel on the host (the SCST reward of 2 x batch rollouts per step): same length
    as the quadratic table on random token lists, empty and one-sided inputs included"""
    import random
    from vilmedic_amd.blocks.scorers import _lcs

    def table(a, b):
        prev = [0] * (len(b) + 1)
        for x in a:
            cur = [0]
            for j, y in enumerate(b):
                cur.append(prev[j] + 1 if x == y else max(prev[j + 1], cur[j]))
            prev = cur
        return prev[-1]
    rng = random.Random(0)
    for _ in range(400):
        n, m, V = rng.randint(0, 140), rng.randint(0, 140), rng.choice([2, 3, 12, 60])
        a, b = [str(rng.randrange(V)) for _ in range(n)], [str(rng.randrange(V)) for _ in range(m)]
        assert _lcs(a, b) == table(a, b)
    assert _lcs([], ["a"]) == 0 and _lcs(["a"], []) == 0 and _lcs(["a"] * 70, ["a"] * 65) == 65


def test_out_of_scope_models_raise():
    import vilmedic_amd.models as M
    with pytest.raises(NotImplementedError):
        M.RRS_HF()


@pytest.mark.parametrize("rel", ["RRG/rrg-vit-synthetic.yml", "RRG/rrg-hf-synthetic.yml", "SELFSUP/convirt-synthetic.yml",
                                 "SELFSUP/gloria-synthetic.yml", "MVQA/vqa-synthetic.yml", "RRS/rrs-synthetic.yml",
                                 "RRG/rrg-scst-synthetic.yml", "RRG/rrg-resnet18-synthetic.yml"])
def test_every_shipped_yaml_parses_and_constructs(rel):
    """plugin-surface test: every YAML under config/ goes through the loader, ``eval(proto)`` resolves the
    dataset and the model class and the model constructs (reduced depth / width so it stays a CPU-second test)."""
    import copy
    import types
    from vilmedic_amd import datasets as D, models as M
    from vilmedic_amd.config import executor_view, get_config
    small = ["dataset.num_samples=4"]
    if "rrg-resnet18" in rel:
        small += ["dataset.image_size=64"]
    if "RRG/rrg-vit" in rel or "rrg-scst" in rel:
        small += ["model.decoder.num_hidden_layers=1", "model.cnn.num_hidden_layers=1"]
    if "rrg-hf" in rel:
        small += ["model.vision.proto_config_args.num_hidden_layers=1", "model.decoder.proto_config_args.num_hidden_layers=1"]
    if "SELFSUP" in rel:
        small += ["model.encoder.num_hidden_layers=1", "dataset.image_size=32"]
    if "MVQA" in rel:
        small += ["model.transformer.num_hidden_layers=1", "dataset.image_size=32"]
    if "RRS" in rel:
        small += ["model.encoder.num_hidden_layers=1", "model.decoder.num_hidden_layers=1"]
    cfg = get_config(os.path.join(os.path.dirname(__file__), "..", "config", rel), small)
    t = executor_view(cfg, "trainor")
    dcfg = copy.deepcopy(t.dataset)
    ds = getattr(D, dcfg.pop("proto"))(split="train", **dcfg)
    batch = ds.get_collate_fn()([ds[0], ds[1]])
    assert batch["input_ids" if "RRS" in rel else "images"].shape[0] == 2
    dl = types.SimpleNamespace(dataset=ds)
    mcfg = copy.deepcopy(t.model)
    model = getattr(M, mcfg.pop("proto"))(**mcfg, dl=dl)
    assert callable(model.eval_func) and sum(p.numel() for p in model.parameters()) > 1000
    assert executor_view(cfg, "validator").batch_size > 0


def test_data_parallel_training_shards_have_equal_batch_counts():
    """create_data_loader under WORLD_SIZE > 1: every rank iterates the same number of training batches (an extra batch on one rank
    would hang its gradient all-reduce); validation keeps every sample"""
    import logging
    from vilmedic_amd.config import wrap
    from vilmedic_amd.executors.utils import create_data_loader
    logger = logging.getLogger("t"); logger.settings = logger.info
    for n, world, bs in [(17, 2, 4), (16, 2, 8), (23, 3, 4), (9, 4, 2)]:
        cfg = wrap({"dataset": {"proto": "SyntheticImSeq", "num_samples": n, "image_size": 8, "vocab_size": 50, "tokenizer_max_len": 8},
                    "batch_size": bs, "num_workers": 0})
        counts = [len(create_data_loader(cfg, "train", logger, rank=r, world=world)) for r in range(world)]
        assert len(set(counts)) == 1, (n, world, bs, counts)
        val = [len(create_data_loader(cfg, "validate", logger, called_by_validator=True, rank=r, world=world).dataset) for r in range(world)]
        assert sum(val) == n


def test_bleu_matches_reference_scorer_and_compute_scores_dumps(golden, tmp_path):
    """BLEU against the reference's vendored COCO-caption scorer (fixture G15, exact); ROUGE-N known answers; compute_scores: the
    config's metric list, dumps next to the checkpoints, unknown metrics skipped"""
    import json
    import logging
    import numpy as np
    from vilmedic_amd.blocks.scorers import Bleu, Rouge1, Rouge2, compute_scores
    g = golden("g15_bleu")
    for n in (4, 2):
        corpus, per = Bleu(n)(g["refs"], g["hyps"])
        assert corpus == pytest.approx(g[f"n{n}"]["corpus"], rel=1e-12, abs=1e-15)
        assert per == pytest.approx(g[f"n{n}"]["per_sentence"], rel=1e-12, abs=1e-15)
    assert Rouge1()(["the cat sat"], ["the cat"])[0] == pytest.approx(0.8) and Rouge2()(["a b c d"], ["a b x d"])[0] == pytest.approx(1 / 3)
    logger = logging.getLogger("scores")
    s = compute_scores(["BLEU", "ROUGEL", "ROUGE2", {"chexbert": {}}], g["refs"], g["hyps"], "validate", 7, str(tmp_path), 3, logger)
    assert set(s) == {"BLEU", "ROUGEL", "ROUGE2"} and s["BLEU"] == pytest.approx(g["n4"]["corpus"])
    assert open(tmp_path / "validate_7_hyps.txt").read().split("\n") == g["hyps"]
    assert json.loads(open(tmp_path / "validate_7_metrics.txt").read())["epoch"] == 3
    c = compute_scores(["accuracy"], np.array([1, 0, 2, 2]), np.eye(3)[[1, 0, 2, 0]], "test", 0, str(tmp_path), 0, logger)
    assert c["accuracy"] == 75.0


@pytest.mark.parametrize("layer_type,hidden_sizes,depths,extra", [
    ("basic", [16, 32, 48, 64], [2, 1, 2, 1], {}),
    ("bottleneck", [32, 64, 96, 128], [1, 2, 1, 1], {"downsample_in_bottleneck": True, "downsample_in_first_stage": True}),
])
def test_hfresnet_backbone_matches_transformers_resnet(layer_type, hidden_sizes, depths, extra):
    """``backbone: hfresnet``: state-dict names, feature map and input gradient against the installed transformers ResNetModel
    (fp32, CPU, train-mode BatchNorm)"""
    tr = pytest.importorskip("transformers")
    from vilmedic_amd.blocks.vision import VisualEncoder
    kw = dict(num_channels=3, embedding_size=8, hidden_sizes=hidden_sizes, depths=depths, layer_type=layer_type, hidden_act="relu", **extra)
    enc = VisualEncoder(backbone="hfresnet", permute="batch_first", dropout_out=0.0, **kw)
    ref = tr.ResNetModel(tr.ResNetConfig(**kw))
    sd = ref.state_dict()
    assert set(enc.model.state_dict()) == set(sd)
    enc.model.load_state_dict(sd, strict=True)
    x = torch.randn(2, 3, 64, 64)
    a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    enc.model.train(); ref.train()
    got, want = enc.model(a), ref(b).last_hidden_state
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    got.square().mean().backward(); want.square().mean().backward()
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(enc.model.state_dict()["embedder.embedder.normalization.running_mean"], ref.state_dict()["embedder.embedder.normalization.running_mean"])


def test_c1_model_state_dict_drives_the_oracle():
    """BASELINE configs[0] (hfresnet-18 + projection + 2-layer decoder): the product model's own state dict (HF / reference names)
    is exactly what oracle.rrg_cnn_forward consumes -- the CPU half of the C1 parity test (the HIP half needs a GPU)"""
    from oracle import torch_ref as O
    from vilmedic_amd.models import RRG
    cnn = dict(proto="VisualEncoder", backbone="hfresnet", permute="batch_first", dropout_out=0.0, layer_type="basic", embedding_size=8,
               hidden_sizes=[8, 16, 24, 32], depths=[1, 1, 1, 1], hidden_act="relu", visual_projection=dict(in_features=32, out_features=128))
    model = RRG(decoder=dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **R.DEC_TINY), cnn=dict(cnn))
    st = {k: v.detach().clone() for k, v in model.state_dict().items() if "lm_head.decoder" not in k}
    images = R.make_images(2, 64, seed=3)
    ids, am = R.make_reports(2, 12, R.DEC_TINY["vocab_size"], seed=3)
    cnn_cfg = {k: cnn[k] for k in ("layer_type", "hidden_sizes", "depths", "hidden_act")}
    loss, logits = O.rrg_cnn_forward(images, ids, am, st, cnn_cfg, R.DEC_TINY, training=True)
    assert torch.isfinite(loss) and logits.shape == (2, 12, R.DEC_TINY["vocab_size"])
    # the product's CNN half (plain torch modules) gives the feature map the oracle computed
    model.train()
    fmap = model.enc.model(images)
    torch.testing.assert_close(fmap, O.hf_resnet_forward(images, st, cnn_cfg, prefix="enc.model.", training=True), rtol=1e-5, atol=1e-5)
    assert fmap.shape == (2, 32, 2, 2)


def test_reference_lr_schedulers(golden):
    """LinearWarmupCosineAnnealingLR (closed form) reproduces the reference's chainable recurrence step for step, also past
    max_epochs (fixture G17); the YAML name resolves in the training scheduler; DecreasingCosineAnnealingWarmRestarts known answer"""
    import types
    from vilmedic_amd.blocks import schedulers as S
    from vilmedic_amd.executors.utils import TrainingScheduler
    g = golden("g17_schedulers")

    def run(make, steps):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=0.02)
        sch = make(opt)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(steps):
            opt.step(); sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        return lrs
    assert run(lambda o: S.LinearWarmupCosineAnnealingLR(o, 10, 40), 55) == pytest.approx(g["lwca_10_40"], abs=1e-12)
    assert run(lambda o: S.LinearWarmupCosineAnnealingLR(o, 5, 20, warmup_start_lr=0.001, eta_min=0.002), 24) == pytest.approx(g["lwca_5_20_start_eta"], abs=1e-12)
    fn = S.linear_warmup_decay(3, 10, cosine=True)
    assert [fn(i) for i in range(12)] == pytest.approx(g["lambda_cosine"], abs=1e-15)
    lrs = run(lambda o: S.DecreasingCosineAnnealingWarmRestarts(factor=0.5, epochs=[2, 3], min_lr=1e-4, optimizer=o, T_0=4, T_mult=1, eta_min=0.0), 17)
    assert lrs[0] == lrs[4] == lrs[16] == pytest.approx(0.02) and lrs[8] == lrs[12] == pytest.approx(0.01) and lrs[10] == pytest.approx(0.005)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.02)
    ts = TrainingScheduler("LinearWarmupCosineAnnealingLR", opt, "training_loss", 5, {"warmup_epochs": 2, "max_epochs": 6})
    seen = []
    for _ in range(4):
        opt.step(); ts.epoch_step()
        seen.append(opt.param_groups[0]["lr"])
    assert seen[0] == pytest.approx(0.02) and seen[1] == pytest.approx(0.02) and seen[-1] < seen[1]


def test_train_cli_run_directory_conventions(tmp_path):
    """bin/train.py: <ckpt_dir>/<name>/ run directory, relative ``ckpt=`` resolved inside it, seed taken from the checkpoint's name"""
    import importlib.util
    from vilmedic_amd.config import wrap
    spec = importlib.util.spec_from_file_location("train_cli", os.path.join(os.path.dirname(__file__), "..", "bin", "train.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    cfg, seed = cli.prepare(wrap({"name": "exp1", "ckpt_dir": str(tmp_path), "seed": 7}))
    assert cfg["ckpt_dir"] == str(tmp_path / "exp1") and os.path.isdir(cfg["ckpt_dir"]) and seed == 7
    open(tmp_path / "exp1" / "1.68_10_560435.pth", "w").close()
    cfg, seed = cli.prepare(wrap({"name": "exp1", "ckpt_dir": str(tmp_path), "seed": 7, "ckpt": "1.68_10_560435.pth"}))
    assert cfg["ckpt"] == str(tmp_path / "exp1" / "1.68_10_560435.pth") and seed == 560435


def test_micro_batch_norm_equals_the_chunked_tower_loop():
    """blocks/vision/micro_bn.py: a CNN run ONCE over the batch with per-micro-batch BatchNorm statistics == the reference's loop over
    forward_batch_size chunks (conVIRT.py:83-95) -- outputs, input / parameter gradients and the running statistics after the pass,
    including a trailing partial micro-batch; state-dict keys are unchanged by the module swap"""
    import copy
    import torch.nn as nn
    from vilmedic_amd.blocks.vision.micro_bn import micro_batches, use_micro_batch_norm
    torch.manual_seed(0)
    ref = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 4, 3, padding=1), nn.BatchNorm2d(4))
    net = use_micro_batch_norm(copy.deepcopy(ref))
    assert list(net.state_dict()) == list(ref.state_dict())
    x = torch.randn(14, 3, 6, 6)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    net.train(), ref.train()
    with micro_batches(4):
        ya = net(xa)
    yb = torch.cat([ref(xb[i:i + 4]) for i in range(0, 14, 4)])
    torch.testing.assert_close(ya, yb, rtol=1e-5, atol=1e-5)
    ya.square().sum().backward()
    yb.square().sum().backward()
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-4, atol=1e-4)
    for (n, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=2e-4, msg=n)
    for (n, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
        torch.testing.assert_close(a.float(), b.float(), rtol=1e-6, atol=1e-6, msg=n)
    net.eval(), ref.eval()
    torch.testing.assert_close(net(x), ref(x), rtol=1e-6, atol=1e-6)


def test_bench_spawns_or_refuses_with_a_clear_message():
    """`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run on 127.0.0.1); on a node with
    fewer GPUs it says so and exits 2 instead of asking for a launcher (VERDICT r2 missing #1)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-300:])
    assert "--gpus 2 requested" in r.stderr and "GPU(s)" in r.stderr and "launch with" not in r.stderr


def test_scst_policy_gradient_weights_pad_to_a_static_shape():
    """SCST.pg_weights (the host half of the policy-gradient loss, ref:vilmedic/blocks/rl/SCST.py:14-45): with pad_to the sampled rollout
    is extended with pad tokens whose rows weigh 0 -- the weighted rows are unchanged, which is what lets the graph-captured step
    (RRG_SCST.graphed_step) run on one static shape"""
    from vilmedic_amd.config import executor_view, get_config
    from vilmedic_amd import datasets as D, models as M
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = get_config(os.path.join(root, "config", "RRG", "rrg-scst-synthetic.yml"),
                     ["dataset.num_samples=4", "dataset.image_size=32", "dataset.vocab_size=97", "dataset.tokenizer_max_len=12",
                      "model.decoder.hidden_size=128", "model.decoder.num_attention_heads=2", "model.decoder.intermediate_size=256",
                      "model.decoder.num_hidden_layers=1", "model.decoder.max_position_embeddings=64", "model.cnn.image_size=32", "model.cnn.patch_size=8",
                      "model.cnn.hidden_size=128", "model.cnn.num_attention_heads=2", "model.cnn.intermediate_size=256", "model.cnn.num_hidden_layers=1"])
    t = executor_view(cfg, "trainor")
    dcfg = dict(t.dataset)
    ds = getattr(D, dcfg.pop("proto"))(split="train", **dcfg)
    dl = torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=ds.get_collate_fn())
    mcfg = dict(t.model)
    model = getattr(M, mcfg.pop("proto"))(**mcfg, dl=dl)
    scst = model.scst
    batch = next(iter(dl))
    g = torch.Generator().manual_seed(1)
    seq = torch.randint(3, 97, (4, 7), generator=g)
    seq[:, 0] = scst.bos_token_id
    seq[2, 4] = scst.eos_token_id
    seq[2, 5:] = scst.pad_token_id
    reward_greedy = [[0.1, 0.2, 0.3, 0.4]]
    s0, w0, aux0 = scst.pg_weights(seq, batch["input_ids"], reward_greedy)
    s1, w1, aux1 = scst.pg_weights(seq, batch["input_ids"], reward_greedy, pad_to=scst.max_length)
    T = scst.max_length
    assert s0.shape == (4, 7) and torch.equal(s0, seq) and s1.shape == (4, T) and w1.shape == (4, T)
    assert torch.equal(s1[:, :7], seq) and bool((s1[:, 7:] == scst.pad_token_id).all())
    assert torch.equal(w1[:, :7], w0) and bool((w1[:, 6:] == 0).all()) and bool((w0[2, 4:] == 0).all())
    assert float(aux0[0]) == float(aux1[0])


# ----------------------------------------------------------------------------- pretrained `proto` towers, DeiT (names / loading; CPU)
@pytest.mark.parametrize("mt,cfg", [("roberta", R.ROBERTA_TINY), ("bert", R.BERT_TINY)])
def test_proto_directory_builds_towers_with_hf_names(tmp_path, mt, cfg):
    """EncoderModel / DecoderModel with ``proto: <checkpoint dir>`` (ref:encoder_model.py:19-22, decoder_model.py:17-21): the module trees
    carry exactly the recipe's (= HF 4.55.3's) parameter names and the loaded values; nothing is left freshly initialised"""
    from vilmedic_amd.blocks.huggingface.decoder.decoder_model import DecoderModel
    from vilmedic_amd.blocks.huggingface.encoder.encoder_model import EncoderModel
    st = R.rand_state(R.text_model_shapes(cfg), 5)
    enc = EncoderModel(dict(proto=R.write_proto_dir(str(tmp_path / "enc"), mt, cfg, st)))
    assert type(enc.encoder).__name__ == ("RobertaModel" if mt == "roberta" else "BertModel")
    sd = enc.encoder.state_dict()
    assert set(sd) == set(st) and all(torch.equal(sd[k], st[k]) for k in st)
    assert enc.encoder._vm_missing_keys == [] and enc.encoder._vm_unexpected_keys == []
    assert enc.encoder.config.hidden_size == cfg["hidden_size"] and not enc.encoder.config.is_decoder
    dst = R.rand_state(R.causal_lm_shapes(cfg, mt), 6)
    dec = DecoderModel(dict(proto=R.write_proto_dir(str(tmp_path / "dec"), mt, cfg, dst)))
    d = dec.decoder
    assert type(d).__name__ == ("RobertaForCausalLM" if mt == "roberta" else "BertLMHeadModel")
    assert d.config.is_decoder and d.config.add_cross_attention and dec.config is d.config and callable(dec.generate)
    tied = {"lm_head.decoder.weight", "lm_head.decoder.bias"} if mt == "roberta" else {"cls.predictions.decoder.weight", "cls.predictions.decoder.bias"}
    sd = d.state_dict()
    assert set(sd) == set(dst) | tied and all(torch.equal(sd[k], dst[k]) for k in dst)
    assert d._vm_missing_keys == []
    assert d.bert.embeddings.word_embeddings.weight is (d.lm_head.decoder.weight if mt == "roberta" else d.cls.predictions.decoder.weight)


def test_proto_from_a_masked_lm_checkpoint_follows_hf_loading_rules(tmp_path):
    """what ``allenai/biomed_roberta_base`` is: a RobertaForMaskedLM checkpoint.  AutoModel strips the ``roberta.`` prefix, drops the MLM head and
    initialises the pooler afresh; AutoModelForCausalLM keeps the head and initialises the cross-attention blocks afresh -- and the key sets
    equal those of the HF classes themselves (installed transformers)."""
    transformers = pytest.importorskip("transformers")
    from vilmedic_amd.blocks.huggingface.decoder.decoder_model import DecoderModel
    from vilmedic_amd.blocks.huggingface.encoder.encoder_model import EncoderModel
    hcfg = transformers.RobertaConfig(**R.ROBERTA_TINY)
    torch.manual_seed(0)
    mlm = transformers.RobertaForMaskedLM(hcfg)
    mlm.save_pretrained(str(tmp_path / "mlm"))
    ref = mlm.state_dict()
    enc = EncoderModel(dict(proto=str(tmp_path / "mlm"))).encoder
    assert set(enc._vm_missing_keys) == {"pooler.dense.weight", "pooler.dense.bias"}
    assert all(k.startswith("lm_head.") for k in enc._vm_unexpected_keys) and enc._vm_unexpected_keys
    assert torch.equal(enc.state_dict()["encoder.layer.1.output.dense.weight"], ref["roberta.encoder.layer.1.output.dense.weight"])
    assert set(enc.state_dict()) == set(transformers.RobertaModel(hcfg).state_dict())
    dec = DecoderModel(dict(proto=str(tmp_path / "mlm"))).decoder
    assert dec._vm_missing_keys and all("crossattention" in k for k in dec._vm_missing_keys)
    assert torch.equal(dec.state_dict()["lm_head.dense.weight"], ref["lm_head.dense.weight"])
    hcfg.is_decoder, hcfg.add_cross_attention = True, True
    assert set(dec.state_dict()) == set(transformers.RobertaForCausalLM(hcfg).state_dict())
    bcfg = transformers.BertConfig(**R.BERT_TINY, is_decoder=True, add_cross_attention=True)
    from vilmedic_amd.blocks.huggingface.bert_models import BertLMHeadModel, text_config
    assert set(BertLMHeadModel(text_config("bert", dict(R.BERT_TINY, is_decoder=True, add_cross_attention=True))).state_dict()) \
        == set(transformers.BertLMHeadModel(bcfg).state_dict())
    # a parameter of the stack itself missing from the checkpoint is an error, never a silent random tower
    from safetensors.torch import load_file, save_file
    broken = {k: v for k, v in load_file(str(tmp_path / "mlm" / "model.safetensors")).items() if "layer.0.output.dense.weight" not in k}
    save_file(broken, str(tmp_path / "mlm" / "model.safetensors"))
    with pytest.raises(RuntimeError, match="lacks parameters"):
        EncoderModel(dict(proto=str(tmp_path / "mlm")))


def test_hub_proto_that_is_not_on_disk_raises_instead_of_downloading(monkeypatch, tmp_path):
    from vilmedic_amd.blocks.huggingface.encoder.encoder_model import EncoderModel
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with pytest.raises(NotImplementedError, match="never downloads"):
        EncoderModel(dict(proto="allenai/biomed_roberta_base"))


def test_deit_parameter_names():
    """backbone: deit (ref:visual_encoder.py:59-61) and RRG_HF proto_model: deit (ref:config/RRG/baseline-HF.yml:22)"""
    from vilmedic_amd.blocks.vision import VisualEncoder
    from vilmedic_amd.models import RRG_HF
    enc = VisualEncoder(backbone="deit", permute="no_permute", **R.DEIT_TINY)
    assert set(enc.state_dict()) == {"model." + k for k in R.deit_shapes(R.DEIT_TINY)}
    for k, shape in R.deit_shapes(R.DEIT_TINY).items():
        assert tuple(enc.state_dict()["model." + k].shape) == tuple(shape), k
    m = RRG_HF(vision=dict(proto_model="deit", proto_config="deit", proto_config_args=dict(R.DEIT_TINY)),
               decoder=dict(proto_model="bert-generation", proto_config="bert-generation", proto_config_args=dict(R.DEC_TINY)))
    names = set(m.state_dict())
    assert {"model.encoder." + k for k in R.deit_shapes(R.DEIT_TINY)} <= names
    assert "model.encoder.pooler.dense.weight" in names and "model.encoder.embeddings.distillation_token" in names


def test_dense_block_takes_the_concatenation_path_off_the_gpu_and_equals_torchvision_form():
    """blocks/vision/cnn._DenseBlock: the single-feature-buffer path (micro_bn.dense_block_forward) is for channels-last DEVICE tensors only -- a CPU tensor,
    an NCHW tensor, or a block whose norm1 is not a MicroBatchNorm2d runs the running torch.cat form, which equals torchvision's cat(list-of-features) form;
    the eligibility test itself never touches the library"""
    import torch.nn as nn
    from vilmedic_amd.blocks.vision import cnn, micro_bn
    torch.manual_seed(0)
    blk = cnn._DenseBlock(3, 16, 2, 8)
    x = torch.randn(2, 16, 5, 5)
    layers = list(blk.values())
    assert not micro_bn.dense_block_ok(layers, x)                                            # stock BatchNorm2d, CPU
    micro_bn.use_micro_batch_norm(blk)
    assert not micro_bn.dense_block_ok(layers, x)                                            # CPU tensor
    assert not micro_bn.dense_block_ok(layers, x.contiguous(memory_format=torch.channels_last))
    assert micro_bn._grad_target(layers[0].norm1.weight) is None and micro_bn._grad_target(None) is None
    assert layers[0].norm1._running_args(x.device) is None or layers[0].norm1.running_mean.device.type == "cpu"
    blk.eval()
    feats = [x]
    for layer in layers:                                                                     # torchvision: every layer reads cat(all earlier features)
        feats.append(layer(torch.cat(feats, 1)))
    assert torch.allclose(blk(x), torch.cat(feats, 1), atol=1e-6)
    assert isinstance(layers[0].relu1, nn.ReLU)


@pytest.mark.parametrize("extra", [dict(), dict(use_layer_scale=False, hidden_act="relu"), dict(drop_path_rate=0.3, layer_scale_init_value=0.5)])
def test_hfpoolformer_backbone_matches_transformers_poolformer(extra):
    """``backbone: hfpoolformer`` (ref:vilmedic/blocks/vision/visual_encoder.py:67-69,192-194): state-dict names, initial layer scales, feature map and
    input gradient against the installed transformers PoolFormerModel (fp32, CPU); with stochastic depth the two draw the same gates from the same seed"""
    tr = pytest.importorskip("transformers")
    from vilmedic_amd.blocks.vision import VisualEncoder
    kw = dict(num_channels=3, depths=[1, 2, 1, 1], hidden_sizes=[8, 16, 24, 32], num_encoder_blocks=4, mlp_ratio=2.0, pool_size=3, **extra)
    enc = VisualEncoder(backbone="hfpoolformer", permute="batch_first", dropout_out=0.0, **kw)
    ref = tr.PoolFormerModel(tr.PoolFormerConfig(**kw))
    sd = ref.state_dict()
    assert set(enc.model.state_dict()) == set(sd)
    if extra.get("use_layer_scale", True):
        torch.testing.assert_close(enc.model.state_dict()["encoder.block.1.1.layer_scale_2"], sd["encoder.block.1.1.layer_scale_2"])
    enc.model.load_state_dict(sd, strict=True)
    x = torch.randn(2, 3, 64, 64)
    a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    train = bool(extra.get("drop_path_rate"))
    enc.model.train(train); ref.train(train)
    torch.manual_seed(5)
    got = enc.model(a)
    torch.manual_seed(5)
    want = ref(b).last_hidden_state
    assert got.shape == (2, 32, 2, 2)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    got.square().mean().backward(); want.square().mean().backward()
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-7)


def test_nd_densenet_equals_the_torchvision_named_densenet_in_two_dimensions_and_runs_volumes():
    """``backbone: _3d_densenet121`` (ref:vilmedic/blocks/vision/visual_encoder.py:8-13,71: MONAI's N-d DenseNet; MONAI is not installed, parity with it is
    UNPINNED): with spatial_dims=2 the restated network computes what the torchvision-named DenseNet-121 of blocks/vision/cnn.py computes (same weights under
    the key mapping ``denselayerK.layers.X`` <-> ``denselayerK.X``, ``class_layers.out`` <-> ``classifier``); in 3-D the ``features`` cut returns a volume
    feature map and the constructor checks of the reference hold"""
    from vilmedic_amd.blocks.vision import cnn, densenet3d
    torch.manual_seed(2)
    nd = densenet3d.build("_3d_densenet121", None, False, spatial_dims=2, in_channels=3, out_channels=10).eval()
    tv = cnn._FACTORY["densenet121"]()
    tv.classifier = torch.nn.Linear(tv.classifier.in_features, 10)
    tv.eval()
    sd = {k.replace(".layers.", ".").replace("class_layers.out.", "classifier."): v for k, v in nd.state_dict().items()}
    assert set(sd) == set(tv.state_dict())
    tv.load_state_dict(sd, strict=True)
    x = torch.randn(2, 3, 64, 64)
    torch.testing.assert_close(nd(x), tv(x), rtol=1e-5, atol=1e-6)
    vol = densenet3d.build("_3d_densenet121", "features", False, spatial_dims=3, in_channels=1, out_channels=2, block_config=(2, 2), init_features=8, growth_rate=4)
    out = vol.eval()(torch.randn(1, 1, 32, 32, 32))
    assert out.shape == (1, 16, 4, 4, 4), out.shape        # 8 -> 16 -> (transition) 8 -> 16 channels; 32 / 2 / 2 / 2 voxels
    names = list(densenet3d.DenseNetND(3, 1, 2, block_config=(1, 1), init_features=8, growth_rate=4).state_dict())
    assert "features.denseblock1.denselayer1.layers.conv2.weight" in names and "features.transition1.conv.weight" in names and "class_layers.out.bias" in names
    with pytest.raises(ValueError):
        densenet3d.build("_3d_densenet999", None, False, spatial_dims=3, in_channels=1, out_channels=2)
