#!/bin/bash
# rocprofv3 --kernel-trace --stats of one secondary task's training step (tools/bench_secondary.py); the per-kernel summary CSV is copied to
# gpurun_out/<tag>_kernel_stats_<task>.csv.   usage: tools/profile_task.sh <tag> <task> [env assignments...]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; task=$2; shift 2
out=/tmp/prof_${tag}_${task}
rm -rf $out; mkdir -p $out $R/gpurun_out
(cd $R && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python tools/bench_secondary.py --only $task --steps 6 --warmup 3 ${BENCH_ARGS:-} > $R/gpurun_out/${tag}_${task}.log 2>&1)
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${tag}_kernel_stats_${task}.csv; else echo "no kernel_stats.csv under $out" >&2; find $out | head >&2; fi
t=$(find $out -name "*kernel_trace.csv" | head -1)
# steady-state steps only (the whole-run summary above is dominated by MIOpen's solver search in the first step)
[ -n "$t" ] && python3 $R/tools/steady_stats.py "$t" --marker "${MARKER:-adam_kernel}" --last 3 --top 45 > $R/gpurun_out/${tag}_steady_kernel_stats_${task}.csv
grep -a '"task"' $R/gpurun_out/${tag}_${task}.log | tail -2
