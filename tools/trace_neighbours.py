#!/usr/bin/env python3
"""Which kernels surround the launches of one kernel in a rocprofv3 --kernel-trace CSV (diagnostic: who causes the runtime's blit copies).
    python tools/trace_neighbours.py <kernel_trace.csv> <name substring>"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
short = lambda n: n.split("(")[0][-60:]
ctx = collections.Counter()
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        ctx[(prev, nxt, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Stream_Id"))] += 1
for k, v in ctx.most_common(30):
    print(v, k)
