#!/bin/bash
# the round's measurement session on the GPU box (everything lands under gpurun_out/r06z_*): the GPU test suite with its measured errors, PMC traffic
# of the GEMM family and of the dominant shape on THIS build, the bench line (after the family summary is in place: it is read at run time), the per-kernel
# summary of the bench step, the trace overlap of eager vs graph replay, secondary metrics
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
(timeout 900 python -m pytest tests -q -m gpu -s 2>&1 | grep -a "parity\|passed\|failed\|FAILED\|Error") > $O/r06z_pytest_gpu.txt
tail -2 $O/r06z_pytest_gpu.txt
python tools/pmc_family.py run /tmp/pmcfam > $O/r06z_pmc_family.log 2>&1
python tools/pmc_family.py sum /tmp/pmcfam $O/r06z_pmc_gemm_family.json >> $O/r06z_pmc_family.log 2>&1
tail -2 $O/r06z_pmc_family.log
cp $O/r06z_pmc_gemm_family.json profiles/r06_pmc_gemm_family.json
if [ -x tools/gpu_probe.bin ]; then
  (LD_LIBRARY_PATH=vilmedic_amd/csrc bash tools/pmc_kernels.sh r06z gemm -- tools/gpu_probe.bin one 12608 2304 768 0 0 1 > $O/r06z_pmc_gemm.txt 2>&1)
  cp $O/r06z_pmc_gemm.txt profiles/r06_pmc_gemm.txt
  (LD_LIBRARY_PATH=vilmedic_amd/csrc VM_GEMM_VARIANT=10 bash tools/pmc_kernels.sh r06z10 gemm_p8 -- tools/gpu_probe.bin one 8192 30528 768 0 0 1 > $O/r06z_pmc_gemm_lmhead.txt 2>&1)
fi
python bench.py > $O/r06z_bench_default.json 2> $O/r06z_bench_default.err
head -c 500 $O/r06z_bench_default.json; echo
python bench.py --no-cpu-baseline > $O/r06z_bench_default_2.json 2> /dev/null
out=/tmp/prof_bench; rm -rf $out
VM_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/r06z_prof_bench.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06z_kernel_stats_sidestream_off.csv
out=/tmp/prof_bench2; rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/r06z_prof_bench2.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06z_kernel_stats_overlapped.csv
f=$(find $out -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_overlap.py $f --last-ms 130 > $O/r06z_trace_overlap_eager.txt 2>&1
out=/tmp/prof_bench3; rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python bench.py --graph 1 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/r06z_prof_bench3.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_overlap.py $f --last-ms 130 > $O/r06z_trace_overlap_graph.txt 2>&1
VM_PROF_DUMP=$O/r06z_prof_dump.txt python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/shape_table.py $O/r06z_prof_dump.txt > $O/r06z_shape_table.txt 2>&1
python tools/bench_secondary.py --only convirt,gloria,mvqa,rrs,scst,decode --steps 10 --warmup 4 2>&1 | grep -a '"task"' > $O/r06z_bench_secondary.jsonl
python tools/bench_secondary.py --only convirt,gloria,mvqa --steps 10 --warmup 4 --amp 1 2>&1 | grep -a '"task"' > $O/r06z_bench_secondary_amp.jsonl
cat $O/r06z_bench_secondary.jsonl $O/r06z_bench_secondary_amp.jsonl | cut -c1-200
cat $O/r06z_trace_overlap_eager.txt $O/r06z_trace_overlap_graph.txt | head -30
