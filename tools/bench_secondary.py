#!/usr/bin/env python3
"""Secondary metrics of SURVEY §8(d), one GPU, through the shipped task configs (so the plugin surface is what is timed):

    python tools/bench_secondary.py [--steps 20] [--warmup 5] [--only convirt,gloria,mvqa,rrs,scst,decode] [--dry]

  convirt / gloria  contrastive training step, pairs/s          (config/SELFSUP/*-synthetic.yml, per-GPU batch 256 / 48)
  mvqa              training images/s and inference images/s    (config/MVQA/vqa-synthetic.yml, batch 256)
  rrs               summarisation training step, pairs/s        (config/RRS/rrs-synthetic.yml, batch 64)
  scst              RRG + SCST step (two rollouts + policy gradient), pairs/s   (config/RRG/rrg-scst-synthetic.yml, batch 32)
  decode            greedy and beam-4 decode of the RRG decoder, tokens/s       (B = 64 or VM_DECODE_BENCH_BATCH, 64 new tokens)

Prints one JSON object per measurement.  ``--dry`` builds every dataset and model on the CPU and stops (what can be checked without
a GPU)."""
import argparse
import json
import logging
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TASKS = {
    "convirt": ("SELFSUP/convirt-synthetic.yml", 256, ["dataset.num_samples=256"]),
    "gloria": ("SELFSUP/gloria-synthetic.yml", 48, ["dataset.num_samples=48"]),
    "mvqa": ("MVQA/vqa-synthetic.yml", 256, ["dataset.num_samples=256"]),
    "rrs": ("RRS/rrs-synthetic.yml", 64, ["dataset.num_samples=64"]),
    "scst": ("RRG/rrg-scst-synthetic.yml", 32, ["dataset.num_samples=32"]),
}


def build(task, dry):
    from vilmedic_amd import datasets as D, models as M
    from vilmedic_amd.config import executor_view, get_config
    from vilmedic_amd.executors.utils import create_optimizer
    rel, batch, extra = TASKS[task]
    cfg = get_config(os.path.join(ROOT, "config", rel), extra + [f"trainor.batch_size={batch}"])
    t = executor_view(cfg, "trainor")
    dcfg = dict(t.dataset)
    ds = getattr(D, dcfg.pop("proto"))(split="train", **dcfg)
    dl = torch.utils.data.DataLoader(ds, batch_size=batch, collate_fn=ds.get_collate_fn())
    mcfg = dict(t.model)
    model = getattr(M, mcfg.pop("proto"))(**mcfg, dl=dl)
    batch_dict = next(iter(dl))
    if dry:
        return model, None, batch_dict, batch
    model = model.cuda()
    logger = logging.getLogger("bench_secondary")
    logger.settings = logger.info
    opt = create_optimizer(t, logger, model)
    batch_dict = {k: v.cuda() if isinstance(v, torch.Tensor) else v for k, v in batch_dict.items()}
    return model, opt, batch_dict, batch


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    t1 = time.perf_counter()                          # the host has ENQUEUED every step (unless something in a step waits for the device)
    torch.cuda.synchronize()
    timed.host_ms = (t1 - t0) / steps * 1e3
    return (time.perf_counter() - t0) / steps


def run_task(task, args):
    model, opt, batch, B = build(task, args.dry)
    n_params = sum(p.numel() for p in model.parameters())
    if args.dry:
        print(json.dumps({"task": task, "dry": True, "model": type(model).__name__, "params": n_params, "batch_keys": sorted(batch)}), flush=True)
        return
    model.train()

    mode = {"launch_mode": "eager"}

    def train_step():
        if task == "scst" and args.scst_graph:       # rollouts + the update replayed from one captured HIP graph (RRG_SCST.graphed_step)
            mode["launch_mode"] = model.graphed_step(opt, **batch).get("launch_mode", "eager")
            return
        out = model(**batch, epoch=1, iteration=1)
        out["loss"].mean().backward()
        opt.step()
        opt.zero_grad()

    steps = max(2, args.steps // 4) if task == "scst" else args.steps            # an SCST step is ~2 x 128 decode steps long
    dt = timed(train_step, steps, max(3, min(args.warmup, steps)) if task == "scst" else min(args.warmup, steps))      # (scst: 2 eager warm-ups + the capture)
    unit = "images/s" if task == "mvqa" else "pairs/s"
    print(json.dumps({"task": task, "metric": f"{task} training step", "value": round(B / dt, 1), "unit": unit, "ms_per_step": round(dt * 1e3, 2),
                      "host_enqueue_ms_per_step": round(timed.host_ms, 2), "batch": B, "params": n_params, "steps": steps, "cnn_tower": "bf16 autocast" if args.amp else "fp32", **(mode if task == "scst" else {})}), flush=True)
    if task == "mvqa":
        model.eval()

        def infer():
            with torch.no_grad():
                model(**batch)
        dt = timed(infer, args.steps, args.warmup)
        print(json.dumps({"task": task, "metric": "mvqa inference", "value": round(B / dt, 1), "unit": "images/s", "ms_per_step": round(dt * 1e3, 2),
                          "host_enqueue_ms_per_step": round(timed.host_ms, 2), "batch": B, "cnn_tower": "bf16 autocast" if args.amp else "fp32"}), flush=True)


def run_decode(args):
    import bench
    if args.dry:
        print(json.dumps({"task": "decode", "dry": True}), flush=True)
        return
    dev = torch.device("cuda")
    dec = bench.build_model(dev).eval().dec.decoder
    B, S, T = int(os.environ.get("VM_DECODE_BENCH_BATCH", "64")), 197, 65
    enc = torch.randn(B, S, 768, device=dev).bfloat16()
    mask = torch.ones(B, S, dtype=torch.bool, device=dev)
    start = torch.zeros(B, 1, dtype=torch.long, device=dev)
    for beams, dtype in ((1, "bf16"), (4, "bf16"), (1, "fp32"), (4, "fp32")):
        kw = dict(bos_token_id=0, eos_token_id=2, pad_token_id=1, max_length=T, decode_dtype=dtype)      # rows that emit eos early still occupy their batch slot
        if beams > 1:
            kw["num_beams"] = beams
        with torch.no_grad():
            dec.generate(input_ids=start, encoder_hidden_states=enc, encoder_attention_mask=mask, **kw)      # captures the graphs
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = dec.generate(input_ids=start, encoder_hidden_states=enc, encoder_attention_mask=mask, **kw)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = out.shape[1] - 1
        print(json.dumps({"task": "decode", "metric": f"decode beams={beams} {dtype}", "value": round(B * steps / dt, 1), "unit": "tokens/s",
                          "ms_per_step": round(dt / max(1, steps) * 1e3, 3), "batch": B, "steps": steps}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--only", default="convirt,gloria,mvqa,rrs,scst,decode")
    ap.add_argument("--dry", action="store_true")
    ap.add_argument("--amp", type=int, default=0, help="1 = trainor.use_amp: the CNN towers run their convolutions in bf16 under autocast (default: fp32, "
                                                       "the reference's default); both modes take channels-last images and the HIP BatchNorm")
    ap.add_argument("--scst-graph", type=int, default=0, help="scst: 1 = the update replayed from one captured HIP graph (RRG_SCST.graphed_step; measured 231 vs 200 ms: an extra encoder pass and the rollout padded to max_length), 0 = eager")
    ap.add_argument("--cudnn-benchmark", type=int, default=-1, help="1 / 0: force torch.backends.cudnn.benchmark (MIOpen's exhaustive find per convolution shape) "
                                                                     "on / off for an A/B; default: leave what the package set")
    args = ap.parse_args()
    if args.cudnn_benchmark >= 0:
        import torch
        torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    if args.amp:
        from vilmedic_amd.blocks.vision import visual_encoder
        visual_encoder.CNN_AMP = True
    for task in args.only.split(","):
        if task == "decode":
            run_decode(args)
        else:
            run_task(task, args)


if __name__ == "__main__":
    main()
