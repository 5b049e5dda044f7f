#!/usr/bin/env python3
"""Timing of vm_attention_fwd / vm_attention_bwd alone at the training step's shapes (B = 64, 12 heads of 64):
    python tools/attn_bench.py [--iters 20] [--only vit,self,cross] [--dropout 0.1]
Prints one line per shape: forward / backward microseconds (HIP events on the launch stream), the algorithmic FLOP rate against the
2.5 PFLOP/s bf16 MFMA peak and the algorithmic bytes against 8 TB/s.  Under ``rocprofv3 --pmc`` it is the workload of
tools/pmc_kernels.sh."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilmedic_amd import ops  # noqa: E402

SHAPES = {"vit": (64, 12, 197, 197, False, False), "self": (64, 12, 128, 128, True, True), "cross": (64, 12, 128, 197, False, True)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="vit,self,cross")
    ap.add_argument("--dropout", type=float, default=0.1)
    args = ap.parse_args()
    dev = torch.device("cuda")
    for kind in args.only.split(","):
        B, H, Lq, Lk, causal, masked = SHAPES[kind]
        D = H * 64
        g = torch.Generator(device="cuda").manual_seed(1)
        km = torch.ones(B, Lk, dtype=torch.uint8, device=dev) if masked else None
        if kind == "cross":
            q = (torch.randn(B, Lq, D, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
            kv = (torch.randn(B, Lk, 2 * D, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
            f = lambda: ops.cross_attention(q, kv, km, H, args.dropout)
        else:
            qkv = (torch.randn(B, Lq, 3 * D, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
            f = lambda: ops.self_attention(qkv, km, H, causal, args.dropout)
        do = torch.randn(B, Lq, D, device=dev, generator=g).bfloat16()
        for _ in range(3):
            f().backward(do)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        outs = []
        ev[0].record()
        for _ in range(args.iters):
            outs.append(f())
        ev[1].record()
        for x in outs:
            x.backward(do)
        ev[2].record()
        torch.cuda.synchronize()
        fw, bw = ev[0].elapsed_time(ev[1]) / args.iters * 1e3, ev[1].elapsed_time(ev[2]) / args.iters * 1e3
        frac = 0.5 if causal else 1.0
        flop_f = 4.0 * B * H * Lq * Lk * 64 * frac
        bytes_f = 2.0 * (B * Lq * D * 2 + 2 * B * Lk * D)          # q, o + k, v (bf16)
        bytes_b = 2.0 * (3 * B * Lq * D + 2 * B * Lk * D + B * Lq * D + 2 * B * Lk * D)   # q, o, do, k, v read + dq, dk, dv written
        print(f"{kind:5s} B{B} H{H} Lq{Lq} Lk{Lk} causal={int(causal)} p={args.dropout}: fwd {fw:7.1f} us ({flop_f / fw * 1e-6:6.1f} TFLOP/s = "
              f"{flop_f / fw * 1e-6 / 2500:.3f} of MFMA peak, {bytes_f / fw * 1e-6:5.2f} TB/s algorithmic)   bwd {bw:7.1f} us "
              f"({2.5 * flop_f / bw * 1e-6:6.1f} TFLOP/s = {2.5 * flop_f / bw * 1e-6 / 2500:.3f}, {bytes_b / bw * 1e-6:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
