#!/bin/bash
# one gpurun batch: every command under its own timeout, logs under gpurun_out/<tag>_*.log
tag=$1; shift
mkdir -p gpurun_out
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "=== [$tag:$i] $cmd" | tee -a gpurun_out/${tag}_index.log
  start=$(date +%s.%N)
  timeout 420 bash -c "$cmd" > gpurun_out/${tag}_$i.log 2>&1
  rc=$?
  end=$(date +%s.%N)
  echo "    rc=$rc  $(echo "$end - $start" | bc) s" | tee -a gpurun_out/${tag}_index.log
done
