#!/usr/bin/env python3
"""Timing of the BatchNorm passes alone (csrc/batchnorm.hip) at the layer shapes of MVQA's DenseNet-169 step (B = 256):
    python tools/bn_bench.py [--iters 20] [--dtype bf16,f32]
per shape: the statistics call (statistics + finalize), the normalisation pass, the backward call (reductions + finalize + gradient, accumulating
onto a gradient buffer for the strided dense-block layers); microseconds per call (HIP events) and algorithmic TB/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilmedic_amd._lib import VM_BF16, VM_F32, check, lib, ptr, stream  # noqa: E402

# (rows, channels normalised, row stride of x / dx): norm2 of block 1, then norm1 of a mid-block layer of blocks 1-4
SHAPES = [(256 * 56 * 56, 128, 128), (256 * 56 * 56, 160, 256), (256 * 28 * 28, 384, 512), (256 * 14 * 14, 768, 1280), (256 * 7 * 7, 1024, 1664),
          (256 * 14 * 14, 128, 128), (256 * 7 * 7, 128, 128)]          # ... and norm2 of blocks 3 / 4 (64 of the 169 layers)


def timed(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16,f32")
    a = ap.parse_args()
    dev = torch.device("cuda")
    for name in a.dtype.split(","):
        td, dt, esz = (torch.bfloat16, VM_BF16, 2) if name == "bf16" else (torch.float32, VM_F32, 4)
        for R, C, ld in SHAPES:
            sets = []
            for k in range(2):
                x = torch.randn(R, ld, device=dev, dtype=td)
                dy = torch.randn(R, C, device=dev, dtype=td)
                y = torch.empty(R, C, device=dev, dtype=td)
                dx = torch.zeros(R, ld, device=dev, dtype=td)
                sets.append((x, dy, y, dx))
            gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
            mean, rstd, var = (torch.empty(1, C, device=dev) for _ in range(3))
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            ws = torch.empty(lib().vm_batchnorm_nhwc_ws(1, R, C), dtype=torch.uint8, device=dev)
            k = [0]
            acc = int(ld != C)

            def stats():
                k[0] ^= 1
                x = sets[k[0]][0]
                check(lib().vm_batchnorm_nhwc_stats(ptr(x), ld, None, 0, ptr(mean), ptr(rstd), ptr(var), C, None, 1, R, C, 1e-5, dt, ptr(ws), ws.numel(), stream()), "s")

            def apply():
                k[0] ^= 1
                x, _, y, _ = sets[k[0]]
                check(lib().vm_batchnorm_nhwc_apply(ptr(x), ld, None, ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(var), C, None, None, None, 0.0, 1e-5,
                                                    1, R, C, dt, 1, stream()), "a")

            def bwd():
                k[0] ^= 1
                x, dy, _, dx = sets[k[0]]
                check(lib().vm_batchnorm_nhwc_bwd_ex(ptr(dy), ptr(x), ld, None, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), C, ptr(dx), ld, acc, None, ptr(dg), ptr(db),
                                                     1, R, C, dt, 1, 1, ptr(ws), ws.numel(), stream()), "b")
            stats()
            ts, ta, tb = timed(stats, a.iters), timed(apply, a.iters), timed(bwd, a.iters)
            n = float(R) * C * esz
            print(f"{name} rows {R} C {C} ld {ld}: stats {ts:7.1f} us ({n / ts * 1e-6:4.2f} TB/s)  apply {ta:7.1f} us ({2 * n / ta * 1e-6:4.2f} TB/s)  "
                  f"bwd{'+acc' if acc else ''} {tb:7.1f} us ({(5 + acc) * n / tb * 1e-6:4.2f} TB/s)", flush=True)
            del sets


if __name__ == "__main__":
    main()
