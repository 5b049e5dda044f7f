#!/usr/bin/env python3
"""Timing of GLoRIA's local loss alone (csrc/gloria.hip + its six GEMMs), forward + backward, at the reference configuration's size
(config/SELFSUP/gloria-mimic.yml: batch 48, 768 features; 19 x 19 regions, captions of T/2..T words):
    python tools/gloria_bench.py [B] [D] [T] [hw]
Prints milliseconds per forward + backward, the six contractions' algorithmic FLOP (2 x 3 x M N K each: bf16 x 3 operands) and the rate."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilmedic_amd.blocks.losses import GLoRIALoss  # noqa: E402


def main():
    B, D, T, hw = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (48, 768, 64, 19)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    img = (torch.randn(B, D, hw, hw, generator=g) * D ** -0.25).to(dev).requires_grad_(True)
    words = (torch.randn(B, D, T, generator=g) * D ** -0.25).to(dev).requires_grad_(True)
    lens = [int(x) for x in torch.randint(T // 2, T + 1, (B,), generator=g)]
    crit = GLoRIALoss()
    for it in range(6):
        if it == 2:
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            t0 = time.perf_counter()
        img.grad = words.grad = None
        l0, l1, _ = crit._local(img, words, lens)
        (l0 + l1).backward()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    # per-family kernel time of one more forward + backward (the library's HIP-event profiler)
    import ctypes
    from vilmedic_amd._lib import lib
    L = lib()
    L.vm_prof_reset(); L.vm_prof_enable(1)
    img.grad = words.grad = None
    l0, l1, _ = crit._local(img, words, lens)
    (l0 + l1).backward()
    torch.cuda.synchronize()
    L.vm_prof_enable(0)
    fam = {}
    for f, name in enumerate(["gemm", "attention", "layernorm", "loss", "elementwise", "optimizer", "decode"]):
        t, w, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        if L.vm_prof_read(f, ctypes.byref(t), ctypes.byref(w), ctypes.byref(n)) == 0 and n.value:
            fam[name] = (round(t.value, 3), n.value)
    print("kernel ms (launches) per family:", fam)
    Tp, Pp = (max(lens) + 15) // 16 * 16, (hw * hw + 15) // 16 * 16
    flop = 6 * 2.0 * (B * Tp) * (B * Pp) * D            # six contractions of (B Tp) x (B Pp) x D multiply-adds
    print(f"gloria local loss B={B} D={D} T<={T} P={hw * hw}: {ms:.2f} ms fwd+bwd, loss {l0.item():.4f}/{l1.item():.4f}, "
          f"{flop * 1e-9:.0f} GFLOP algorithmic (x3 on the bf16 MFMA) -> {flop / ms * 1e-9:.1f} TFLOP/s, peak memory "
          f"{torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB")


if __name__ == "__main__":
    main()
