#!/usr/bin/env python3
"""BASELINE configs[2] with the text tower the reference's YAML names (config/SELFSUP/convirt-mimic.yml:24: ``proto: allenai/biomed_roberta_base``):
a RoBERTa-base-shaped checkpoint directory (random weights: nothing can be downloaded) is written to a temporary directory, the ConVIRT model is
built through ``EncoderModel(proto=<dir>)`` -> RobertaModel on the HIP path, and a training step at B = 256, L = 128 is timed next to the same
model with the config-dict BertGeneration tower of config/SELFSUP/convirt-synthetic.yml.
    python tools/convirt_roberta_bench.py [--steps 10] [--amp 1] [--check 1]"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--amp", type=int, default=0)
    ap.add_argument("--check", type=int, default=0, help="1: VM_WGRAD_CHECK (assert every first-touch store targets a zero buffer) on two extra steps")
    a = ap.parse_args()
    import golden_recipes as R
    from vilmedic_amd import datasets as D, models as M, ops
    from vilmedic_amd.blocks.vision import visual_encoder
    from vilmedic_amd.config import executor_view, get_config
    from vilmedic_amd.optim import FusedAdam
    if a.amp:
        visual_encoder.CNN_AMP = True
    cfg_r = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, vocab_size=50265, max_position_embeddings=514,
                 type_vocab_size=1, layer_norm_eps=1e-5, bos_token_id=0, pad_token_id=1, eos_token_id=2, hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1)
    d = tempfile.mkdtemp(prefix="vm_roberta_base_")
    st = R.rand_state(R.text_model_shapes(cfg_r), 1, std=0.02)
    R.write_proto_dir(d, "roberta", cfg_r, st)
    del st
    cfg = get_config(os.path.join(ROOT, "config", "SELFSUP", "convirt-synthetic.yml"),
                     ["dataset.num_samples=256", "trainor.batch_size=256", "dataset.vocab_size=50265", f"model.encoder.proto={d}"])
    t = executor_view(cfg, "trainor")
    dcfg = dict(t.dataset)
    ds = getattr(D, dcfg.pop("proto"))(split="train", **dcfg)
    dl = torch.utils.data.DataLoader(ds, batch_size=256, collate_fn=ds.get_collate_fn())
    mcfg = dict(t.model)
    enc = dict(mcfg["encoder"])
    mcfg["encoder"] = {"proto": d}                      # what the reference's YAML holds: the checkpoint name only
    model = getattr(M, mcfg.pop("proto"))(**mcfg, dl=dl).cuda()
    assert type(model.linguistic.encoder).__name__ == "RobertaModel", type(model.linguistic.encoder).__name__
    batch = {k: v.cuda() if isinstance(v, torch.Tensor) else v for k, v in next(iter(dl)).items()}
    opt = FusedAdam(model, lr=5e-5)
    model.train()

    def step():
        out = model(**batch, epoch=1, iteration=1)
        out["loss"].mean().backward()
        opt.step()
        opt.zero_grad()
        return out["loss"]
    for _ in range(4):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    res = {"task": "convirt with EncoderModel(proto=<RoBERTa-base-shaped dir>)", "ms_per_step": round(dt * 1e3, 2), "pairs_per_s": round(256 / dt, 1), "batch": 256,
           "text_tower": type(model.linguistic.encoder).__name__, "cnn_tower": "bf16 autocast" if a.amp else "fp32", "loss": round(float(loss.detach()), 4),
           "params": sum(p.numel() for p in model.parameters())}
    if a.check:
        ops.WGRAD_CHECK = True
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.WGRAD_CHECK = False
        res["first_touch_check"] = "2 steps with VM_WGRAD_CHECK: every overwriting weight-gradient launch found its buffers zero"
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
