#!/bin/bash
# round 6, session g: kernel trace of the decode benchmark (one greedy position: per-launch duration and gap)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=/tmp/dec_trace; rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python tools/bench_secondary.py --only decode > gpurun_out/r06g_decode_bench.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python tools/decode_trace.py $f > gpurun_out/r06g_decode_position.txt 2>&1; cp $f gpurun_out/r06g_kernel_trace.csv 2>/dev/null; gzip -f gpurun_out/r06g_kernel_trace.csv
grep '"task"' gpurun_out/r06g_decode_bench.log
head -150 gpurun_out/r06g_decode_position.txt
python tools/bench_secondary.py --only decode 2>&1 | grep '"task"' > gpurun_out/r06g_decode_untraced.jsonl; cat gpurun_out/r06g_decode_untraced.jsonl
