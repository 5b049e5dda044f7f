#!/bin/bash
# round 6, session n: are the wide tile's exposed stores a lockstep effect?  VM_GEMM_DEBUG 8 / 9 / 10 delay groups of workgroups at kernel start
# (2 phases x 4 sleeps, 4 phases x 2 sleeps, 4 phases x 1 sleep of ~3.5 us) so their epilogues fall into the other groups' K loops
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export LD_LIBRARY_PATH=$R/vilmedic_amd/csrc:$LD_LIBRARY_PATH
O=gpurun_out/r06n_skew.txt
: > $O
for shape in "8192 30528 768 0 0 1" "12608 2304 768 0 0 1" "12608 3072 768 0 0 7" "12608 3072 768 0 1 8" "8192 3072 768 0 0 7" "8192 2304 768 0 0 1"; do
  for v in -1 10; do
    for d in 0 3 8 9 10 11; do
      if [ $v != 10 ] && [ $d != 0 ]; then continue; fi
      echo -n "variant=$v dbg=$d " >> $O
      VM_GEMM_VARIANT=$v VM_GEMM_DEBUG=$d timeout 60 tools/gpu_probe.bin onef $shape >> $O 2>&1
    done
  done
done
cat $O
