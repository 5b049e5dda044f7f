#!/bin/bash
timeout 100 tools/gpu_probe.bin 2>&1 | grep -A20 "VARIANT=5" | grep -c " ok"
for s in "12608 2304 768 0 0 1" "12608 3072 768 0 0 1" "12608 3072 768 0 1 1" "12608 768 3072 0 0 1" "8192 3072 768 0 0 1" "8192 30528 768 0 0 1" "8192 768 30528 0 1 1" "3072 768 12608 1 1 3" "30528 768 8192 1 1 1" "8192 8192 8192 0 0 1"; do
  for v in 0 4 5; do
    echo -n "shape $s v=$v: "; VM_GEMM_VARIANT=$v timeout 60 tools/gpu_probe.bin bench $s | head -1 | cut -c50-
  done
done
