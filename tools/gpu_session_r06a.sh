#!/bin/bash
# round 6, session a: the wave-private epilogue of the wide-tile kernel (VM_GEMM_VARIANT=10) against production (-1) and the staged epilogue (9)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export LD_LIBRARY_PATH=$R/vilmedic_amd/csrc:$LD_LIBRARY_PATH
bash tools/gpu_call.sh r06a \
  "timeout 200 tools/gpu_probe.bin ab VM_GEMM_VARIANT -1 10" \
  "timeout 200 tools/gpu_probe.bin ab VM_GEMM_VARIANT 9 10" \
  "VM_GEMM_VARIANT=10 timeout 60 tools/gpu_probe.bin bench 12608 2304 768 0 0 1" \
  "VM_GEMM_VARIANT=9 timeout 60 tools/gpu_probe.bin bench 12608 2304 768 0 0 1" \
  "VM_GEMM_VARIANT=10 timeout 60 tools/gpu_probe.bin bench 8192 2048 768 0 0 1" \
  "VM_GEMM_VARIANT=10 VM_GEMM_P8_MF=5 timeout 60 tools/gpu_probe.bin bench 12608 3072 768 0 0 1" \
  "VM_GEMM_VARIANT=10 timeout 60 tools/gpu_probe.bin bench 8192 30528 768 0 0 1" \
  "timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r06a_bench_default.json 2> gpurun_out/r06a_bench_default.err"
cat gpurun_out/r06a_1.log | tail -30
