#!/bin/bash
# round 6, session k: 128 x 256 tiles (64 x 128 per wave, 32-wide K-tiles, 3-slot ring, two workgroups per CU: VM_GEMM_VARIANT=6) against production
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export LD_LIBRARY_PATH=$R/vilmedic_amd/csrc:$LD_LIBRARY_PATH
timeout 300 tools/gpu_probe.bin ab VM_GEMM_VARIANT -1 6 > gpurun_out/r06k_ab.txt 2>&1
grep "^ab\|fails\|FAIL" gpurun_out/r06k_ab.txt
O=gpurun_out/r06k_onef.txt; : > $O
for shape in "12608 2304 768 0 0 1" "12608 3072 768 0 0 7" "8192 2304 768 0 0 1" "8192 3072 768 0 0 7" "8192 30528 768 0 0 1" "12608 18432 768 0 0 1"; do
  for v in -1 6; do
    for d in 0 1; do
      if [ $v == -1 ] && [ $d != 0 ]; then continue; fi
      echo -n "variant=$v dbg=$d " >> $O
      VM_GEMM_VARIANT=$v VM_GEMM_DEBUG=$d timeout 60 tools/gpu_probe.bin onef $shape >> $O 2>&1
    done
  done
done
cat $O
