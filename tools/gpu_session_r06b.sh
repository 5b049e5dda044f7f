#!/bin/bash
# round 6, session b: store flavour / segment-size experiments of the wave-private epilogue (VM_GEMM_DEBUG 5: plain stores, 6: 256-B segments nt, 7: 256-B plain)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export LD_LIBRARY_PATH=$R/vilmedic_amd/csrc:$LD_LIBRARY_PATH
O=gpurun_out/r06b_stores.txt
: > $O
for shape in "12608 2304 768 0 0 1" "8192 2048 768 0 0 1" "8192 30528 768 0 0 1" "12608 3072 768 0 0 7" "12608 3072 768 0 1 0"; do
  for v in -1 9 10; do
    for d in 0 1 3 5 6 7; do
      if [ $v != 10 ] && [ $d != 0 ]; then continue; fi
      echo -n "variant=$v dbg=$d " >> $O
      VM_GEMM_VARIANT=$v VM_GEMM_DEBUG=$d timeout 60 tools/gpu_probe.bin onef $shape >> $O 2>&1
    done
  done
done
cat $O
