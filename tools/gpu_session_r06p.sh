#!/bin/bash
# round 6, session p: BatchNorm streaming loops with packed rows in flight + the block-count rule; bench of the kernels, the tests, the CNN configs
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06p_bn.txt
: > $O
timeout 600 python -m pytest tests/test_hip_batchnorm_gpu.py -q -x 2>&1 | tail -2 | tee -a $O
for bl in 4096 512 256; do
  echo "== VM_BN_BLOCKS=$bl" | tee -a $O
  VM_BN_BLOCKS=$bl python tools/bn_bench.py 2>&1 | grep rows >> $O
  for amp in 0 1; do
    VM_BN_BLOCKS=$bl timeout 900 python tools/bench_secondary.py --only mvqa,convirt,gloria --amp $amp --steps 10 --warmup 3 2>&1 | grep '"task"' | cut -c1-200 | tee -a $O
  done
done
