#!/bin/bash
for s in "8192 30528 768 0 0 1" "8192 768 30528 0 1 1" "30528 768 8192 1 1 1" "12608 3072 768 0 0 1" "12608 2304 768 0 0 1"; do
  for v in 0 1 2 4; do
    echo -n "shape $s v=$v: "; VM_GEMM_VARIANT=$v timeout 60 tools/gpu_probe.bin bench $s 2>&1 | head -1 | cut -c50-
  done
done
