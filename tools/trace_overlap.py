#!/usr/bin/env python3
"""How much of a rocprofv3 --kernel-trace CSV ran concurrently: sum of kernel durations vs the union of their intervals, per window of the
last ``--last`` seconds of the trace (the timed steps), and the same split per queue / stream column.
    python tools/trace_overlap.py <kernel_trace.csv> [--last-ms 300]"""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--last-ms", type=float, default=300.0)
args = ap.parse_args()
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"]) for r in csv.DictReader(open(args.csv))]
rows.sort()
t_end = max(r[1] for r in rows)
win = [r for r in rows if r[0] >= t_end - args.last_ms * 1e6]
total = sum(e - s for s, e, *_ in win)
union, cur_s, cur_e = 0, None, None
for s, e, *_ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = max(r[1] for r in win) - win[0][0]
print(f"{len(win)} kernels in the last {args.last_ms:.0f} ms: sum of durations {total / 1e6:.2f} ms, union {union / 1e6:.2f} ms, span {span / 1e6:.2f} ms "
      f"(concurrent {100.0 * (total - union) / total:.1f} % of kernel time, idle {100.0 * (span - union) / span:.1f} % of the span)")
per = collections.Counter()
for s, e, q, st, _ in win:
    per[(q, st)] += e - s
for k, v in per.most_common():
    print(f"  queue {k[0]} stream {k[1]}: {v / 1e6:.2f} ms")
