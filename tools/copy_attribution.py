#!/usr/bin/env python3
"""Which lines of the package issue the device-to-device copies / fills of one RRG training step (bench.py's step)?  torch.profiler with stacks over two
steps; aten::copy_ / fill_ / zero_ calls grouped by the innermost vilmedic_amd (or bench.py) frame.    python tools/copy_attribution.py [--steps 2]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    import bench
    from vilmedic_amd import ops
    from vilmedic_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    model = bench.build_model(dev).train()
    ops.manual_seed(1234)
    opt = FusedAdam(model, lr=1e-4)
    images, ids, am = bench.synthetic_batch(64, 128, bench.DEC_12L["vocab_size"], dev, seed=0)

    def step():
        out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
        opt.zero_grad()
        opt.gate = out["loss"].detach()
        out["loss"].backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    agg = collections.Counter()
    for e in prof.events():
        if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::cat"):
            where = "?"
            for fr in e.stack or []:
                if "vilmedic_amd" in fr or "bench.py" in fr or "copy_attribution" in fr:
                    where = fr.strip()
                    break
            agg[(e.name, where)] += 1
    for (name, where), n in agg.most_common(40):
        print(f"{n / a.steps:7.1f} per step  {name:18s} {where}")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))


if __name__ == "__main__":
    main()
