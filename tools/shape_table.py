#!/usr/bin/env python3
"""Per-shape table of a VM_PROF_DUMP file (the per-launch HIP-event records of bench.py's two roofline steps, each kernel alone on the GPU):
    VM_PROF_DUMP=gpurun_out/prof_dump.txt python bench.py ... ; python tools/shape_table.py gpurun_out/prof_dump.txt [steps=2] > profiles/rNN_shape_table.txt
One line per (family, tag): launches per step, ms per step, us per launch, TFLOP/s and the fraction of the 2.5 PFLOP/s dense bf16 MFMA peak (GEMM /
attention families; the other families record bytes).  Regressions by shape show up here, not in the family totals."""
import sys

FAM = {0: "gemm", 1: "attention", 2: "layernorm", 3: "loss", 4: "elementwise", 5: "optimizer", 6: "decode"}


def main(path, steps=2):
    rows = []
    for line in open(path):
        f = line.split()
        if len(f) < 5:
            continue
        fam, tag, n, ms, work = int(f[0]), f[1], int(f[2]), float(f[3]), float(f[4])
        rows.append((fam, tag, n, ms, work))
    tot = {}
    print(f"# {path}: {steps} recorded steps")
    print(f"{'family':10s} {'tag':58s} {'n/step':>6s} {'ms/step':>8s} {'us/launch':>9s} {'TFLOP/s | TB/s':>14s} {'frac':>6s}")
    for fam, tag, n, ms, work in sorted(rows, key=lambda r: (r[0], -r[3])):
        rate = work / ms / 1e9 if ms > 0 else 0.0           # work per ms -> 1e12 units per second
        frac = rate / 2500.0 if fam in (0, 1) else rate / 8.0
        print(f"{FAM.get(fam, str(fam)):10s} {tag:58s} {n / steps:6.1f} {ms / steps:8.3f} {ms / n * 1e3:9.1f} {rate:14.1f} {frac:6.3f}")
        t = tot.setdefault(fam, [0.0, 0.0])
        t[0] += ms / steps
        t[1] += work / steps
    print("# family totals per step")
    for fam, (ms, work) in sorted(tot.items()):
        rate = work / ms / 1e9 if ms > 0 else 0.0
        print(f"# {FAM.get(fam, str(fam)):10s} {ms:8.3f} ms  {rate:10.1f} {'TFLOP/s' if fam in (0, 1) else 'TB/s'}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
