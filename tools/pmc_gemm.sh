#!/bin/bash
# PMC passes for one GEMM shape (separate passes: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2).  usage: pmc_gemm.sh tag M N K la lb split
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
export VM_GEMM_VARIANT=${VM_GEMM_VARIANT:-0}
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o p$i -- $R/tools/gpu_probe.bin bench "$@" > $out/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print('   %-28s n=%d mean=%.4g' % (c, len(vals), sum(vals)/len(vals)))
PY
