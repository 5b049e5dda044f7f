#!/bin/bash
export TMPDIR=/tmp
for b in gpu_probe gpu_probe_nt; do
  for s in "12608 2304 768 0 0 1" "12608 768 3072 0 0 1"; do
    echo -n "$b $s: "; VM_GEMM_VARIANT=4 timeout 60 tools/$b.bin bench $s | head -1 | cut -c50-
  done
  cd /tmp; VM_GEMM_VARIANT=4 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/nt_$b -o p -- /root/repo/tools/$b.bin bench 12608 2304 768 0 0 1 > /dev/null 2>&1; cd /root/repo
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("/tmp/nt_$b/p_counter_collection.csv")) if 'gemm' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
rows.sort(key=lambda r:int(r['Dispatch_Id']))
v=[float(r['Counter_Value']) for r in rows][3:23]
print("$b FETCH_SIZE KB (full kernel) mean", sum(v)/len(v))
PY
done
