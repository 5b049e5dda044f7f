"""Zoo loader: checkpoint wire format + (model, dataset) rebuilt from a checkpoint directory -- CPU."""
import os

import pytest
import torch
import yaml

from test_datasets import _make_corpus


def _model_cfg():
    return {"proto": "RRG",
            "decoder": {"proto": None, "hidden_size": 64, "num_attention_heads": 2, "intermediate_size": 128, "num_hidden_layers": 1,
                        "max_position_embeddings": 32, "bos_token_id": 0, "pad_token_id": 1, "eos_token_id": 2},
            "cnn": {"proto": "VisualEncoder", "backbone": "vit", "permute": "no_permute", "dropout_out": 0.0, "image_size": 32, "patch_size": 16,
                    "hidden_size": 64, "num_attention_heads": 2, "intermediate_size": 128, "num_hidden_layers": 1}}


def test_auto_model_from_checkpoint_directory(tmp_path):
    import copy
    import types
    from vilmedic_amd import models as M
    from vilmedic_amd.datasets import ImSeq
    from vilmedic_amd.zoo import AutoModel
    root, zoo = str(tmp_path / "data"), str(tmp_path / "zoo" / "rrg-tiny")
    os.makedirs(root), os.makedirs(zoo)
    _make_corpus(root)
    seq = dict(root=root, file="report.tok", tokenizer=None, tokenizer_max_len=12, processing="r2gen_clean_report", source="tgt")
    image = dict(root=root, file="image.tok", image_path=root, resize=40, crop=32, ext=".png")
    train = ImSeq(seq=seq, image=image, split="train", ckpt_dir=zoo)                 # writes zoo/vocab.tgt
    cfg = _model_cfg()
    torch.manual_seed(0)
    mc = copy.deepcopy(cfg)
    model = getattr(M, mc.pop("proto"))(**mc, dl=types.SimpleNamespace(dataset=train))
    sd = model.state_dict()
    # the wire format of an old multi-GPU checkpoint: DataParallel prefix + pre-1.3.2 encoder names (executors/utils.py:26-34)
    old = {"module." + k.replace("enc.model.", "enc.0.cnn."): v.clone() for k, v in sd.items()}
    torch.save({"model": old, "__version__": "1.2.9"}, os.path.join(zoo, "0.5_3_0.pth"))
    yaml.safe_dump({"name": "rrg_tiny", "model": cfg,
                    "dataset": {"proto": "ImSeq",
                                "seq": {"vocab_file": "vocab.tgt", "tokenizer_max_len": 12, "processing": "r2gen_clean_report", "source": "tgt"},
                                "image": {"resize": 40, "crop": 32, "ext": ".png"}}}, open(os.path.join(zoo, "config.yml"), "w"))
    loaded, dataset = AutoModel.from_pretrained(zoo)
    assert not loaded.training and len(dataset) == 0 and dataset.tokenizer.vocab_size == train.tokenizer.vocab_size
    got = loaded.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k].cpu(), sd[k]) for k in sd)
    batch = dataset.inference(seq=["The heart is normal. 2. No effusion.", "clear lungs"])
    ref = train.seq.get_collate_fn()([{"tgt_seq": " ".join(train.seq.processing(s).split())} for s in ["The heart is normal. 2. No effusion.", "clear lungs"]])
    assert torch.equal(batch["input_ids"], ref["input_ids"]) and batch["input_ids"].shape == (2, 12)
    # zoo names resolve under VILMEDIC_ZOO_DIR; a missing one says why it cannot be fetched
    os.environ["VILMEDIC_ZOO_DIR"] = str(tmp_path / "empty")
    try:
        with pytest.raises(FileNotFoundError, match="no network"):
            AutoModel.from_pretrained("rrg/baseline-mimic")
        with pytest.raises(KeyError):
            AutoModel.from_pretrained("rrg/does-not-exist")
    finally:
        del os.environ["VILMEDIC_ZOO_DIR"]
    with pytest.raises(EnvironmentError):
        AutoModel()


def test_export_training_checkpoint_to_zoo_directory(tmp_path):
    """a Trainor-format checkpoint ({"model", "config", ...}) -> zoo directory -> AutoModel: data files dropped from the dataset
    section, vocabulary / label files copied and re-rooted, weights identical"""
    import copy
    import types
    from vilmedic_amd import models as M
    from vilmedic_amd.datasets import ImSeq
    from vilmedic_amd.zoo import AutoModel
    from vilmedic_amd.zoo.export import export
    root, ck, out = str(tmp_path / "data"), str(tmp_path / "ckpt"), str(tmp_path / "zoo")
    os.makedirs(root)
    _make_corpus(root)
    dcfg = {"proto": "ImSeq",
            "seq": dict(root=root, file="report.tok", tokenizer=None, tokenizer_max_len=12, processing="r2gen_clean_report", source="tgt"),
            "image": dict(root=root, file="image.tok", image_path=root, resize=40, crop=32, ext=".png")}
    train = ImSeq(seq=dcfg["seq"], image=dcfg["image"], split="train", ckpt_dir=ck)
    cfg = _model_cfg()
    mc = copy.deepcopy(cfg)
    model = getattr(M, mc.pop("proto"))(**mc, dl=types.SimpleNamespace(dataset=train))
    path = os.path.join(ck, "0.25_2_0.pth")
    torch.save({"model": model.state_dict(), "optimizer": {}, "training_scheduler": {}, "__version__": "1.3.6",
                "config": {"name": "rrg_files", "ckpt_dir": ck, "model": cfg, "dataset": dcfg, "batch_size": 8}}, path)
    export(path, out)
    written = yaml.safe_load(open(os.path.join(out, "config.yml")))
    assert written["dataset"]["seq"] == {"tokenizer": None, "tokenizer_max_len": 12, "processing": "r2gen_clean_report", "source": "tgt",
                                         "vocab_file": "vocab.tgt"}
    assert written["dataset"]["image"] == {"resize": 40, "crop": 32, "ext": ".png"}
    assert sorted(os.listdir(out)) == ["0.25_2_0.pth", "config.yml", "vocab.tgt"]
    loaded, dataset = AutoModel.from_pretrained(out)
    sd = model.state_dict()
    assert all(torch.equal(loaded.state_dict()[k].cpu(), sd[k]) for k in sd)
    assert dataset.tokenizer.vocab_size == train.tokenizer.vocab_size and dataset.image.resize == 40
