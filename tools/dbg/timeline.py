"""per-stream timeline summary of the last training step in a rocprofv3 kernel trace csv"""
import csv, collections, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "copyBuffer" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
step = rows[adam[-2] + 1:adam[-1] + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
print("step span ms %.2f kernels %d" % ((t1 - t0) / 1e6, len(step)))
by = collections.defaultdict(list)
for r in step: by[r["Stream_Id"]].append(r)
def fam(n):
    return ("gemm" if "gemm_fast" in n else "attn" if "attn" in n else "ln" if n.startswith("ln_") or " ln_" in n[:20] else "splitk" if "splitk" in n
            else "colsum" if "colsum" in n else "torch" if "at::native" in n else n[:18])
for sid, rs in by.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    f = collections.Counter()
    for r in rs: f[fam(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("stream", sid, "n", len(rs), "busy %.2f ms" % (busy / 1e6), {k: round(v / 1e6, 2) for k, v in f.most_common(9)})
ce = [r for r in step if "ce_shift" in r["Kernel_Name"]][0]
print("fwd %.2f ms, bwd+opt %.2f ms" % ((int(ce["Start_Timestamp"]) - t0) / 1e6, (t1 - int(ce["End_Timestamp"])) / 1e6))
main = max(by.values(), key=len)
gap, prev, big = 0, None, []
for r in main:
    if prev is not None:
        g = int(r["Start_Timestamp"]) - prev
        if g > 0:
            gap += g
            if g > 20000: big.append((g / 1e3, r["Kernel_Name"][:40]))
    prev = max(prev or 0, int(r["End_Timestamp"]))
print("main-stream gaps %.2f ms; gaps > 20us:" % (gap / 1e6), sorted(big, reverse=True)[:12])
# union busy time of all streams (any kernel running)
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in step] + [(int(r["End_Timestamp"]), -1) for r in step])
act, last, idle = 0, t0, 0
for t, d in ev:
    if act == 0: idle += t - last
    act += d; last = t
print("GPU fully idle %.2f ms of the step" % (idle / 1e6))
