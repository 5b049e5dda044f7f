#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c X.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2> res.txt
    python tools/resource_report.py res.txt"""
import re
import subprocess
import sys

PATS = [("V", r"VGPRs: (\d+)"), ("A", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
        ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")]


def main(path):
    for blk in open(path).read().split("Function Name: ")[1:]:
        name = blk.split()[0]
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        vals = []
        for key, pat in PATS:
            m = re.search(pat, blk)
            vals.append(f"{key}={m.group(1) if m else '?'}")
        print(f"{name[:84]:84s} " + " ".join(vals))


if __name__ == "__main__":
    main(sys.argv[1])
