#!/bin/bash
# round 6, session i: decode step with / without the side-stream L2 prefetch of the next projections' weights (same box, interleaved), decode tests
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/decode_gemm_bench.py > gpurun_out/r06i_decode_gemm_warm_cold.txt 2>&1
for rep in 1 2; do
  for pf in 0 1; do
    VM_DECODE_PREFETCH=$pf python tools/bench_secondary.py --only decode 2>&1 | grep '"task"' | sed "s/^/prefetch=$pf /" >> gpurun_out/r06i_decode_ab.txt
  done
done
cat gpurun_out/r06i_decode_gemm_warm_cold.txt gpurun_out/r06i_decode_ab.txt
python -m pytest tests -m gpu -x -q -k "decode or greedy or beam or scst or ensemble or generate" 2>&1 | tail -5 > gpurun_out/r06i_pytest_decode.txt; cat gpurun_out/r06i_pytest_decode.txt
for pf in 0 1; do
  VM_DECODE_PREFETCH=$pf python tools/bench_secondary.py --only scst --steps 12 --warmup 4 2>&1 | grep '"task"' | sed "s/^/prefetch=$pf /" >> gpurun_out/r06i_scst_ab.txt
done
cat gpurun_out/r06i_scst_ab.txt
