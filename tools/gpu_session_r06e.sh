#!/bin/bash
# round 6, session e: rolled (csrc/libvmhip.so) against unrolled (tools/ab_lib0/libvmhip.so, -DVM_EPI_ROLL=0) item loop of the production GEMM epilogue
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for lib in ab_lib0 roll; do
    if [ $lib == roll ]; then export LD_LIBRARY_PATH=$R/vilmedic_amd/csrc; else export LD_LIBRARY_PATH=$R/tools/ab_lib0; fi
    timeout 200 tools/gpu_probe.bin ab VM_GEMM_VARIANT -1 -1 > gpurun_out/r06e_${lib}_$rep.txt 2>&1
  done
done
for f in gpurun_out/r06e_*.txt; do echo "== $f"; grep "^ab\|fails" $f; done
# whole step with each library
cp vilmedic_amd/csrc/libvmhip.so /tmp/lib_roll.so
for rep in 1 2; do
  cp tools/ab_lib0/libvmhip.so vilmedic_amd/csrc/libvmhip.so
  timeout 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r06e_bench_unrolled_$rep.json 2> gpurun_out/r06e_bench_unrolled_$rep.err
  cp /tmp/lib_roll.so vilmedic_amd/csrc/libvmhip.so
  timeout 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r06e_bench_rolled_$rep.json 2> gpurun_out/r06e_bench_rolled_$rep.err
done
for f in gpurun_out/r06e_bench_*.json; do echo "== $f"; head -c 400 $f; echo; done
