#!/usr/bin/env python3
"""Skeleton of the hottest loop of one kernel in a hipcc -S listing (MFMA / LDS read / DMA / wait / barrier order):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only X.hip -o X.s ; python tools/isa_loop.py X.s '<mangled-name substring>'"""
import re
import sys


def main(path, pat):
    s = open(path).read()
    names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", s, re.M) if pat in m.group(1)]
    for name in names:
        a = s.index(name + ":")
        b = s.index(".Lfunc_end", a)
        body = s[a:b].split("\n")
        labels = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(body)]
        best, bi = -1, 0
        for k in range(len(labels) - 1):
            n = sum("v_mfma" in l for l in body[labels[k]:labels[k + 1]])
            if n > best:
                best, bi = n, k
        scratch = sum(l.strip().startswith("scratch_") for l in body)
        print(f"== {name}: hottest block {body[labels[bi]].split(':')[0]} with {best} MFMAs; scratch ops in kernel: {scratch}")
        res, cnt = [], 0
        for l in body[labels[bi]:labels[bi + 1]]:
            t = l.strip().split(";")[0].strip()
            if not t or t.endswith(":"):
                continue
            op = t.split()[0]
            if op.startswith(("v_mfma", "ds_read", "ds_write", "s_barrier", "global_load_lds", "s_cbranch", "scratch_", "global_", "buffer_")):
                tok = {"v_mfma": "M", "ds_read": "r", "global_load_lds": "D"}.get(next((k for k in ("v_mfma", "ds_read", "global_load_lds") if op.startswith(k)), ""), op)
            elif op.startswith("s_waitcnt"):
                tok = "<" + t.replace("s_waitcnt ", "") + ">"
            else:
                cnt += 1
                continue
            if cnt:
                res.append(f"{cnt}")
                cnt = 0
            res.append(tok)
        print(" ".join(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
