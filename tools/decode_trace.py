#!/usr/bin/env python3
"""Decode positions out of a rocprofv3 --kernel-trace CSV of ``tools/bench_secondary.py --only decode``: the launches between two consecutive
token-selection kernels (``select_tokens_kernel``: greedy, ``beam_topk`` / ``beam_rows``: beam search) are one position.  Positions are grouped by
(selection kernel, precision of the projection kernels, launch count); for each group: wall per position, and per kernel launches / time / gap
in front; the median position of the bf16 greedy group is listed launch by launch.
    python tools/decode_trace.py <kernel_trace.csv>"""
import collections
import csv
import statistics
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
names = [short(r["Kernel_Name"]) for r in rows]
t0 = [int(r["Start_Timestamp"]) for r in rows]
t1 = [int(r["End_Timestamp"]) for r in rows]
is_sel = lambda n: n.startswith("select_tokens_kernel") or n.startswith("beam_topk") or n.startswith("beam_merge")
ends = [i for i, n in enumerate(names) if is_sel(n) and not (i + 1 < len(names) and is_sel(names[i + 1]))]      # last selection launch of a position
groups = collections.defaultdict(list)
for a, b in zip(ends, ends[1:]):
    span = range(a + 1, b + 1)
    if len(span) > 400 or (t0[a + 1] - t1[a]) > 200000:      # a pause of the host between two generate() calls
        continue
    f32 = any("<true" in names[i] and "decode_gemm" in names[i] for i in span) or any("f32" in names[i] for i in span if "attn" in names[i])
    sel = "greedy" if names[b].startswith("select_tokens") else "beam"
    groups[(sel, "fp32" if f32 else "bf16", len(span))].append((a + 1, b + 1))
for key, spans in sorted(groups.items(), key=lambda kv: -len(kv[1])):
    if len(spans) < 8:
        continue
    walls = [(t1[b - 1] - t1[a - 1]) / 1e3 for a, b in spans]
    print(f"== {key[0]} {key[1]}: {len(spans)} positions of {key[2]} launches; wall per position (us): min {min(walls):.1f} median {statistics.median(walls):.1f}")
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for a, b in spans:
        for i in range(a, b):
            e = agg[names[i]]; e[0] += 1; e[1] += (t1[i] - t0[i]) / 1e3; e[2] += (t0[i] - t1[i - 1]) / 1e3
    n = len(spans)
    tk = sum(e[1] for e in agg.values()) / n
    tg = sum(e[2] for e in agg.values()) / n
    print(f"   per position: kernels {tk:.1f} us + gaps {tg:.1f} us")
    print(f"   {'kernel':60s} {'n/pos':>6s} {'us/pos':>8s} {'mean us':>8s} {'mean gap':>8s}")
    for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:60s} {c / n:6.1f} {d / n:8.1f} {d / c:8.2f} {g / c:8.2f}")
    if key[0] == "greedy" and key[1] == "bf16":
        order = sorted(range(len(spans)), key=lambda i: walls[i])
        a, b = spans[order[len(order) // 2]]
        print("   -- median position, launch by launch: kernel, us, gap in front, grid")
        for i in range(a, b):
            r = rows[i]
            print(f"      {names[i]:60s} {(t1[i] - t0[i]) / 1e3:7.2f} {(t0[i] - t1[i - 1]) / 1e3:7.2f} {r.get('Grid_Size_X') or r.get('Grid_Size') or '':>8}")
