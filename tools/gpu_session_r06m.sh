#!/bin/bash
# round 6, session m: flush policy of the grouped weight-gradient queue -- "fill" (launches that fill their 256-CU rounds) against "threshold" (rounds 4-5), interleaved
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for mode in threshold fill hybrid; do
    VM_WGRAD_FLUSH=$mode timeout 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r06m_bench_${mode}_$rep.json 2> gpurun_out/r06m_bench_${mode}_$rep.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06m_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["family_ms_per_step"], d["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
