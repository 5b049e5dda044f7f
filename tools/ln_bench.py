#!/usr/bin/env python3
"""Timing of the LayerNorm kernels alone at the training step's shapes (ViT: 12608 x 768, decoder: 8192 x 768):
    python tools/ln_bench.py [--iters 50]
forward, backward partial (with both residual-fork operands, as most of the step's calls have) and the batched dgamma / dbeta
reduce of ``--batch`` problems; microseconds per launch (HIP events on the launch stream) and algorithmic bytes against 8 TB/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilmedic_amd import _lib  # noqa: E402
from vilmedic_amd._lib import check, lib, ptr, stream  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
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
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=62)
    args = ap.parse_args()
    dev = torch.device("cuda")
    for rows, cols in ((12608, 768), (8192, 768)):
        g = torch.Generator(device="cuda").manual_seed(0)
        x, dy, dy2, dres = ((torch.randn(rows, cols, device=dev, generator=g)).bfloat16() for _ in range(4))
        gamma, beta = torch.rand(cols, device=dev) + 0.5, torch.randn(cols, device=dev)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        nws = lib().vm_layernorm_bwd_ws(rows, cols) // 4
        wss = [torch.empty(nws, device=dev) for _ in range(args.batch)]
        dg, db = torch.zeros(args.batch, cols, device=dev), torch.zeros(args.batch, cols, device=dev)
        fwd = lambda: check(lib().vm_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, cols, 1e-12, stream()), "f")
        k = [0]

        def bwd():
            k[0] = (k[0] + 1) % args.batch
            check(lib().vm_layernorm_bwd_partial(ptr(dy), ptr(dy2), ptr(dres), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), rows, cols,
                                                 ptr(wss[k[0]]), stream()), "b")
        arr = (_lib.LnReduceProblem * args.batch)()
        for i, q in enumerate(arr):
            q.ws, q.dgamma, q.dbeta, q.rows, q.cols = wss[i].data_ptr(), dg[i].data_ptr(), db[i].data_ptr(), rows, cols
        red = lambda: check(lib().vm_layernorm_bwd_reduce_batched(arr, args.batch, stream()), "r")
        one = lambda: check(lib().vm_layernorm_bwd_reduce(ptr(wss[0]), ptr(dg[0]), ptr(db[0]), rows, cols, stream()), "r1")
        tf, tb, tr, t1 = timed(fwd, args.iters), timed(bwd, args.iters), timed(red, 10), timed(one, args.iters)
        bf, bb = 2.0 * rows * cols * 2, 2.0 * rows * cols * 5
        print(f"rows {rows} cols {cols} cap {os.environ.get('VM_LN_BWD_CAP', 'default')}: fwd {tf:6.1f} us ({bf / tf * 1e-6:4.2f} TB/s)   "
              f"bwd partial {tb:6.1f} us ({bb / tb * 1e-6:4.2f} TB/s, slabs {nws // (2 * cols)})   reduce x{args.batch} {tr:6.1f} us "
              f"({tr / args.batch:5.2f} us each; alone {t1:5.1f} us)", flush=True)


if __name__ == "__main__":
    main()
