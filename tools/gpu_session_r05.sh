#!/bin/bash
# the round's measurement session on the GPU box (everything lands under gpurun_out/r05g_*): PMC traffic of the GEMM family and of the dominant
# shape, steady-state kernel statistics of the ConVIRT / MVQA steps, the bench line, the per-kernel summary of the bench step, secondary metrics
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
python tools/pmc_family.py run /tmp/pmcfam > $O/r05g_pmc_family.log 2>&1
python tools/pmc_family.py sum /tmp/pmcfam $O/r05g_pmc_gemm_family.json >> $O/r05g_pmc_family.log 2>&1
if [ -x tools/gpu_probe.bin ]; then
  (LD_LIBRARY_PATH=vilmedic_amd/csrc bash tools/pmc_kernels.sh r05g gemm -- tools/gpu_probe.bin one 12608 2304 768 0 0 1 > $O/r05g_pmc_gemm.txt 2>&1)
fi
bash tools/profile_task.sh r05g convirt
bash tools/profile_task.sh r05g mvqa
python tools/bench_secondary.py --only convirt,gloria,mvqa --steps 10 --warmup 4 --amp 1 2>&1 | grep -a '"task"' > $O/r05g_bench_secondary_amp.jsonl
python tools/bench_secondary.py --only convirt,gloria,mvqa,rrs,scst,decode --steps 10 --warmup 4 2>&1 | grep -a '"task"' > $O/r05g_bench_secondary.jsonl
python bench.py > $O/r05g_bench_default.json 2> $O/r05g_bench_default.err
out=/tmp/prof_bench; rm -rf $out
VM_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/r05g_prof_bench.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05g_kernel_stats_sidestream_off.csv
head -c 600 $O/r05g_bench_default.json; echo; cat $O/r05g_bench_secondary_amp.jsonl; cat $O/r05g_pmc_family.log | tail -3
