#!/bin/bash
# round 6, session d: plain (L2-allocating) stores and the split-DMA K-tile (VM_GEMM_VARIANT=11) of the wide-tile kernel
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export LD_LIBRARY_PATH=$R/vilmedic_amd/csrc:$LD_LIBRARY_PATH
O=gpurun_out/r06d_stores.txt
: > $O
for shape in "12608 2304 768 0 0 1" "8192 2048 768 0 0 1" "8192 30528 768 0 0 1" "12608 3072 768 0 0 7" "8192 2304 768 0 0 1" "12608 3072 768 0 1 0"; do
  for v in -1 10 11; do
    for d in 0 1 5; do
      if [ $v == -1 ] && [ $d != 0 ]; then continue; fi
      echo -n "variant=$v dbg=$d " >> $O
      VM_GEMM_VARIANT=$v VM_GEMM_DEBUG=$d timeout 60 tools/gpu_probe.bin onef $shape >> $O 2>&1
    done
  done
done
cat $O
timeout 200 tools/gpu_probe.bin ab VM_GEMM_VARIANT 10 11 > gpurun_out/r06d_ab.txt 2>&1
grep "^ab\|fails" gpurun_out/r06d_ab.txt
