"""Turn a training checkpoint (executors/trainor.py: {"model", "config", "__version__", ...}) into a zoo directory that
``AutoModel.from_pretrained`` loads: ``config.yml`` whose dataset section keeps only what inference needs (data files dropped,
``vocab_file`` / ``label_file`` pointing at copies inside the directory), the vocabulary / label files, and the weights.

    python -m vilmedic_amd.zoo.export ckpt/0.31_4_0.pth out_dir
"""
import copy
import os
import shutil
import sys

import torch
import yaml

_DATA_KEYS = ("root", "file", "image_path", "hf_dataset", "hf_field", "hf_local", "hf_filter")


def _inference_dataset_config(dcfg, ckpt_dir, out_dir):
    dcfg = copy.deepcopy(dcfg)
    for name, sub in list(dcfg.items()):
        if not isinstance(sub, dict):
            continue
        had_file = sub.get("file") is not None
        for k in _DATA_KEYS:
            sub.pop(k, None)
        if name in ("seq", "src", "tgt") and sub.get("tokenizer") is None and sub.get("vocab_file") is None and had_file:
            source = {"src": "src", "tgt": "tgt"}.get(name) or sub.get("source") or "src"     # ImSeq's ``seq`` names its own source
            vocab = "vocab.{}".format(source)
            shutil.copy(os.path.join(ckpt_dir, vocab), os.path.join(out_dir, vocab))
            sub["vocab_file"] = vocab
        if name == "label" and had_file:
            shutil.copy(os.path.join(ckpt_dir, "labels.tok"), os.path.join(out_dir, "labels.tok"))
            sub["label_file"] = "labels.tok"
    return dcfg


def export(checkpoint, out_dir):
    state = torch.load(checkpoint, map_location="cpu")
    config = state.get("config")
    if not config or "model" not in config or "dataset" not in config:
        raise ValueError("checkpoint carries no training config with model / dataset sections")
    os.makedirs(out_dir, exist_ok=True)
    ckpt_dir = config.get("ckpt_dir") or os.path.dirname(os.path.abspath(checkpoint))
    out = {"name": config.get("name"), "model": config["model"],
           "dataset": _inference_dataset_config(config["dataset"], ckpt_dir, out_dir)}
    with open(os.path.join(out_dir, "config.yml"), "w") as f:
        yaml.safe_dump(out, f, sort_keys=False)
    torch.save({"model": state["model"], "__version__": state.get("__version__")},
               os.path.join(out_dir, os.path.basename(checkpoint)))
    return out_dir


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    print(export(sys.argv[1], sys.argv[2]))
