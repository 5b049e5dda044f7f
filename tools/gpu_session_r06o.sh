#!/bin/bash
# round 6, session o: DenseNet block on one feature buffer + running statistics inside the BatchNorm kernel: tests, then MVQA / ConVIRT / GLoRIA steps
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_hip_batchnorm_gpu.py -q -x -s 2>&1 | grep -a "parity\|passed\|failed\|Error" | cut -c1-200
timeout 1200 python -m pytest tests -q -x -m gpu -k "mvqa or densenet or convirt or gloria or cnn or tower" 2>&1 | tail -4
rm -f gpurun_out/r06o_secondary.jsonl
for amp in 0 1; do
  timeout 600 python tools/bench_secondary.py --only mvqa,convirt,gloria --amp $amp --steps 10 --warmup 3 2>&1 | grep '"task"' | tee -a gpurun_out/r06o_secondary.jsonl
done
bash tools/profile_task.sh r06r mvqa > /dev/null 2>&1
head -3 gpurun_out/r06r_steady_kernel_stats_mvqa.csv | cut -c1-200
