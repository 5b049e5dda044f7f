#!/usr/bin/env python3
"""Per-kernel statistics of the STEADY-STATE steps of a `rocprofv3 --kernel-trace --output-format csv` run: kernels are grouped into steps by a
marker kernel that runs once per step (default: the fused optimizer, `adam_kernel`), and only the last N steps are aggregated -- the first
steps of a task hold MIOpen's solver search (hundreds of `naive_conv_*` launches of 100+ ms each) and lazy initialisation, which a whole-run
`--stats` summary mixes into every average.
    python tools/steady_stats.py <kernel_trace.csv> [--marker adam_kernel] [--last 3] [--top 40]"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="adam_kernel")
    ap.add_argument("--last", type=int, default=3)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < a.last + 1:
        raise SystemExit(f"only {len(marks)} launches of {a.marker!r}: cannot cut {a.last} steady steps")
    lo, hi = marks[-a.last - 1] + 1, marks[-1] + 1
    sel = rows[lo:hi]
    wall = (sel[-1][1] - sel[0][0]) / 1e6 / a.last
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        agg[n][0] += 1
        agg[n][1] += e - s
    tot = sum(v[1] for v in agg.values()) / 1e6 / a.last
    print(f"# {a.trace}: last {a.last} steps between launches of {a.marker}: {len(sel) / a.last:.0f} kernels per step, "
          f"{tot:.2f} ms of kernel time per step, {wall:.2f} ms wall per step (trace timestamps)")
    print("kernel,calls_per_step,ms_per_step,avg_us,pct")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"\"{n[:140]}\",{c / a.last:.1f},{t / 1e6 / a.last:.3f},{t / 1e3 / c:.1f},{100.0 * t / 1e6 / a.last / tot:.2f}")


if __name__ == "__main__":
    main()
