#!/usr/bin/env python3
"""Feasibility probe: capture bench.py's whole training step (forward + autograd backward with the side stream + fused Adam)
into one HIP graph with torch.cuda.graph and replay it.  Prints eager vs replay ms/step and the loss trajectory.
(Seeds / Adam bias corrections are frozen at capture in this probe: it only answers "does capture work, what does replay cost".)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    from vilmedic_amd import ops
    from vilmedic_amd.optim import FusedAdam
    model = bench.build_model(dev)
    model.train()
    ops.manual_seed(1234)
    opt = FusedAdam(model, lr=1e-4)
    B, L, V = 64, 128, bench.DEC_12L["vocab_size"]
    images, ids, am = bench.synthetic_batch(B, L, V, dev, seed=0)
    static_loss = torch.zeros((), device=dev)

    def step():
        out = model(input_ids=ids, attention_mask=am, images=images, return_logits=False)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        return out["loss"]

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            l = step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        l = step()
    torch.cuda.synchronize()
    print(f"eager: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step, loss {l.item():.4f}", flush=True)
    g = torch.cuda.CUDAGraph()
    try:
        t0 = time.perf_counter()
        with torch.cuda.graph(g):
            l = step()
            static_loss.copy_(l.detach())
        torch.cuda.synchronize()
        print(f"capture ok in {time.perf_counter() - t0:.2f} s", flush=True)
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        print("CAPTURE FAILED:", repr(e)[:500], flush=True)
        return
    losses = []
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        g.replay()
        if i % 5 == 0:
            losses.append(static_loss.item())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"replay: {dt * 1e3:.2f} ms/step ({B / dt:.0f} pairs/s), losses {losses}", flush=True)
    t0 = time.perf_counter()
    for i in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"replay (no host reads): {dt * 1e3:.2f} ms/step ({B / dt:.0f} pairs/s)", flush=True)
    print(f"mem allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
