#!/bin/bash
# sweep the column-group width of the tile order on the hot GEMM shapes (tools/gpu_probe.bin bench M N K la lb split)
for s in "12608 2304 768 0 0 1" "12608 3072 768 0 0 1" "12608 3072 768 0 1 1" "12608 768 3072 0 0 1" "12608 768 3072 0 1 1" "8192 3072 768 0 0 1" "8192 30528 768 0 0 1" "8192 768 30528 0 1 1" "8192 2304 768 0 0 1"; do
  for gw in 0 2 3 4 6 8 12; do
    echo -n "shape $s gw=$gw: "; VM_GEMM_GROUPW=$gw VM_GEMM_DEBUG_ONLY0=1 timeout 60 tools/gpu_probe.bin bench $s | head -1
  done
done
