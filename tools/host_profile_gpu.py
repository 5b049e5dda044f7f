#!/usr/bin/env python3
"""cProfile of the HOST side of bench.py's eager training step on a GPU box (no synchronisation inside the profiled steps: what is measured is
the Python / ctypes / autograd time to enqueue them -- bench.py's ``host_enqueue_ms_per_step`` broken down by function).
    python tools/host_profile_gpu.py [--steps 20] [--top 40]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vilmedic_amd.optim import FusedAdam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    model.train()
    opt = FusedAdam(model, lr=1e-4)
    V = bench.DEC_12L["vocab_size"]
    images, ids, am = bench.synthetic_batch(64, 128, V, dev, seed=0)

    def step():
        out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
        opt.zero_grad()
        opt.gate = out["loss"].detach()
        out["loss"].backward()
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    import gc
    gc.collect(); gc.freeze()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"unprofiled: enqueue {1e3 * (t1 - t0) / args.steps:.2f} ms per step, wall {1e3 * (t2 - t0) / args.steps:.2f} ms per step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(args.top)


if __name__ == "__main__":
    main()
