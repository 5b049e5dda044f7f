#!/usr/bin/env python3
"""Timing of the ConVIRT / InfoNCE / GLoRIA-global similarity loss alone at SURVEY §8(d)'s size (global batch 2048, 768 features):
    python tools/contrastive_bench.py [B] [D]
forward + backward of _SimilarityLossFn (row / column cross-entropy of S = a_hat b_hat^T / tau); kernel time per family from the
library's HIP-event profiler.  Algorithmic work (S counted once): 2 B^2 D FLOP forward, 4 B^2 D backward (two gradient GEMMs) ->
the MFMA fraction printed is (6 B^2 D) / time / 2.5 PFLOP/s; minimum HBM traffic: the two [B, D] operands and their gradients."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilmedic_amd._lib import lib  # noqa: E402
from vilmedic_amd.blocks.losses.selfsup import _SimilarityLossFn  # noqa: E402


def main():
    B, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (2048, 768)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    a = torch.randn(B, D, generator=g).to(dev).requires_grad_(True)
    b = torch.randn(B, D, generator=g).to(dev).requires_grad_(True)

    def step():
        a.grad = b.grad = None
        r, c = _SimilarityLossFn.apply(a, b, True, 10.0, 1e-8)
        (r.mean() + c.mean()).backward()
        return r
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    L = lib()
    L.vm_prof_reset(); L.vm_prof_enable(1)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    L.vm_prof_enable(0)
    tot, fam = 0.0, {}
    for f, name in enumerate(["gemm", "attention", "layernorm", "loss", "elementwise", "optimizer", "decode"]):
        t, w, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        if L.vm_prof_read(f, ctypes.byref(t), ctypes.byref(w), ctypes.byref(n)) == 0 and n.value:
            fam[name] = (round(t.value / 5, 4), n.value // 5)
            tot += t.value / 5
    import tempfile
    dump = os.path.join(tempfile.gettempdir(), "contr_prof.txt")
    L.vm_prof_dump(dump.encode())
    for line in open(dump):
        f = line.split()
        if len(f) >= 4 and f[1].startswith("contrastive"):
            print(f"    {f[1]:40s} {float(f[3]) / float(f[2]) * 1e3:8.1f} us per launch ({int(float(f[2]))} launches)")
    flop = 6.0 * B * B * D
    print(f"contrastive B={B} D={D}: wall {wall:.3f} ms fwd+bwd (host-bound launches included); vmhip kernels {tot:.3f} ms {fam}; "
          f"algorithmic {flop * 1e-9:.1f} GFLOP -> {flop / tot * 1e-9:.1f} TFLOP/s = {flop / tot * 1e-9 / 2500:.3f} of the bf16 MFMA peak")


if __name__ == "__main__":
    main()
