#!/usr/bin/env python3
"""HBM-side traffic of the GEMM family of one training step, from separate rocprofv3 --pmc passes over bench.py
(MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE in their own passes, in KiB-like units of 1 KB; FETCH_SIZE x 2 on gfx950).

    python tools/pmc_family.py run  <outdir>      # on the GPU box: two passes (FETCH_SIZE, WRITE_SIZE) over a 4-step bench, every kernel alone
    python tools/pmc_family.py sum  <outdir> <json_out>   # aggregate -> {"build_digest", "steps", "gemm": {...}, "kernels": {...}}

The JSON is committed under profiles/ and read by bench.py at run time (roofline.traffic): it carries the digest of the library it was
measured on, so a bench line can say whether the figure belongs to the build that printed it."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARMUP = 3, 1
GEMM = ("gemm_fast_kernel", "gemm_p8w_kernel", "gemm_p8_kernel", "gemm_grouped_kernel", "gemm_pair_kernel", "gemm_kernel", "gemm_skinny")


def run(out):
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", VM_SIDE_STREAM="0")
    for i, ctr in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", f"p{i}", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(STEPS), "--warmup", str(WARMUP), "--no-cpu-baseline", "--no-roofline",
               "--no-secondary"]
        with open(os.path.join(out, f"p{i}.log"), "w") as f:
            subprocess.run(cmd, cwd=ROOT, env=env, stdout=f, stderr=subprocess.STDOUT, check=False)


def summarise(out, dst):
    sys.path.insert(0, ROOT)
    from vilmedic_amd import build
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            kn = kn[:kn.rfind("(")] if kn.endswith(")") else kn          # drop the argument list only ("(anonymous namespace)::" stays)
            k = per[kn[:90]][r["Counter_Name"]]
            k[0] += float(r["Counter_Value"])
            k[1] += 1
    steps = STEPS + WARMUP
    kernels, fam = {}, {"fetch_bytes": 0.0, "write_bytes": 0.0, "launches": 0}
    for name, ctrs in per.items():
        fetch = 2.0 * ctrs["FETCH_SIZE"][0] * 1024 if "FETCH_SIZE" in ctrs else 0.0          # x2: the gfx950 correction of the guide
        write = ctrs["WRITE_SIZE"][0] * 1024 if "WRITE_SIZE" in ctrs else 0.0
        n = max(ctrs["FETCH_SIZE"][1] if "FETCH_SIZE" in ctrs else 0, ctrs["WRITE_SIZE"][1] if "WRITE_SIZE" in ctrs else 0)
        kernels[name] = {"launches_per_step": round(n / steps, 2), "fetch_MB_per_step": round(fetch / steps / 1e6, 2),
                         "write_MB_per_step": round(write / steps / 1e6, 2)}
        if any(g in name for g in GEMM):
            fam["fetch_bytes"] += fetch / steps
            fam["write_bytes"] += write / steps
            fam["launches"] += n / steps
    res = {"build_digest": build._lib_digest(), "steps_profiled": steps, "command": "VM_SIDE_STREAM=0 rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace "
           "-- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary  (one pass per counter; FETCH_SIZE x 2 on gfx950)",
           "gemm_family": {"fetch_MB_per_step": round(fam["fetch_bytes"] / 1e6, 1), "write_MB_per_step": round(fam["write_bytes"] / 1e6, 1),
                           "traffic_MB_per_step": round((fam["fetch_bytes"] + fam["write_bytes"]) / 1e6, 1), "launches_per_step": round(fam["launches"], 1)},
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -(kv[1]["fetch_MB_per_step"] + kv[1]["write_MB_per_step"]))[:40])}
    with open(dst, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["gemm_family"]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        summarise(sys.argv[2], sys.argv[3])
