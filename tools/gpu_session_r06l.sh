#!/bin/bash
# round 6, session l: dK/dV dropout pair index in 32-bit arithmetic (csrc/libvmhip.so) against the 64-bit form (tools/ab_lib0), same box, interleaved;
# then the attention tests and the MVQA / decode secondary metrics with the running-concatenation DenseNet blocks
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp vilmedic_amd/csrc/libvmhip.so /tmp/lib_new.so
for rep in 1 2; do
  cp tools/ab_lib0/libvmhip.so vilmedic_amd/csrc/libvmhip.so
  python tools/attn_bench.py --iters 50 2>&1 | grep -v amdgpu.ids | sed "s/^/old /" >> gpurun_out/r06l_attn_ab.txt
  cp /tmp/lib_new.so vilmedic_amd/csrc/libvmhip.so
  python tools/attn_bench.py --iters 50 2>&1 | grep -v amdgpu.ids | sed "s/^/new /" >> gpurun_out/r06l_attn_ab.txt
done
cat gpurun_out/r06l_attn_ab.txt
python -m pytest tests -m gpu -x -q -k "attention or dropout" 2>&1 | tail -3 > gpurun_out/r06l_pytest_attention.txt; cat gpurun_out/r06l_pytest_attention.txt
python tools/bench_secondary.py --only mvqa --steps 10 --warmup 4 2>&1 | grep '"task"' > gpurun_out/r06l_bench_mvqa.jsonl
python tools/bench_secondary.py --only mvqa --steps 10 --warmup 4 --amp 1 2>&1 | grep '"task"' >> gpurun_out/r06l_bench_mvqa.jsonl
cat gpurun_out/r06l_bench_mvqa.jsonl
