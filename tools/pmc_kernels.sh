#!/bin/bash
# rocprofv3 PMC passes (separate passes: gfx950 has 8 SQ / 4 TCC slots; FETCH_SIZE takes 3, WRITE_SIZE 2) over any command, averaged per
# kernel name.  usage: tools/pmc_kernels.sh <tag> <kernel-name substring> -- <command...>
# FETCH_SIZE is reported x2 (the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md "HBM").
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; pat=$2; shift 3
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $R && TMPDIR=/tmp rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o p$i -- "$@" > $out/p$i.log 2>&1)
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$pat" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, vals in sorted(v.items()):
        m = sum(vals) / len(vals)
        if c == "FETCH_SIZE":
            print("   %-28s n=%d mean=%.4g KB  (x2 gfx950 correction: %.4g MB)" % (c, len(vals), m, 2 * m / 1024))
        elif c == "WRITE_SIZE":
            print("   %-28s n=%d mean=%.4g KB  (%.4g MB)" % (c, len(vals), m, m / 1024))
        else:
            print("   %-28s n=%d mean=%.6g" % (c, len(vals), m))
PY
