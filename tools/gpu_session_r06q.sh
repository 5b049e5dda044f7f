#!/bin/bash
# round 6, session q: torch.backends.cudnn.benchmark (what the reference's bin/utils.py:158 get_seed switches on: MIOpen's find per convolution shape) off / on
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06q_cudnn_benchmark.txt
: > $O
for cb in 0 1; do
  for amp in 0 1; do
    echo "== cudnn.benchmark=$cb amp=$amp" | tee -a $O
    timeout 1500 python tools/bench_secondary.py --only mvqa,convirt,gloria --amp $amp --cudnn-benchmark $cb --steps 10 --warmup 4 2>&1 | grep '"task"' | cut -c1-190 | tee -a $O
  done
done
