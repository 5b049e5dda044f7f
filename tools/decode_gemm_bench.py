#!/usr/bin/env python3
"""vm_decode_gemm at the decode step's shapes (M = 64 rows, bf16), weights WARM (one buffer, launch after launch: L2 / MALL hits) against COLD
(rotating over > 256 MB of distinct weight buffers: every launch streams its weights from HBM) -- what a side-stream prefetch of the next
layer's weights could buy per launch.  HIP events around graph replays of 200 dependent launches; microseconds per launch.
    python tools/decode_gemm_bench.py"""
import ctypes as C_
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilmedic_amd._lib import VM_BF16, DecodeGemmArgs, check, lib, stream  # noqa: E402


def timed(fn, n):
    """n launches replayed from ONE HIP graph (the decode step's own launch mode: eager launches from Python are host-bound at this size)"""
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


def main():
    dev = torch.device("cuda")
    M = int(os.environ.get("VM_DECODE_BENCH_BATCH", "64"))
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
        nbuf = max(2, int(320e6 / (N * K * 2)))
        Ws = [torch.randn(N, K, device=dev).bfloat16() for _ in range(nbuf)]
        A = torch.randn(M, K, device=dev).bfloat16()
        Cout = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        bias = torch.zeros(N, device=dev)

        def launch(W):
            g = DecodeGemmArgs()
            g.dtype = VM_BF16
            g.A, g.lda, g.W, g.ldw, g.C, g.ldc = A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), Cout.data_ptr(), Cout.stride(0)
            g.M, g.N, g.K, g.act = M, N, K, 0
            g.bias = bias.data_ptr()
            check(lib().vm_decode_gemm(C_.byref(g), stream()), "vm_decode_gemm")
        warm = timed(lambda i: launch(Ws[0]), 200)
        cold = timed(lambda i: launch(Ws[i % nbuf]), 200)
        mb = N * K * 2 / 1e6
        print(f"M {M} N {N} K {K} ({mb:.1f} MB of weights): warm {warm:6.2f} us   cold {cold:6.2f} us   ({mb / cold * 1e-6 * 1e6:.2f} TB/s cold)   "
              f"HBM floor at 6 TB/s {mb / 6.0:.2f} us", flush=True)


if __name__ == "__main__":
    main()
