# one gpurun call: full GPU test suite, default bench line, rocprofv3 kernel stats of a short bench (outputs under gpurun_out/r01_e)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01_e
timeout 110 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r01_e/pytest.txt; cat gpurun_out/r01_e/pytest.txt
timeout 100 python bench.py > gpurun_out/r01_e/bench_default.json 2> gpurun_out/r01_e/bench.err; tail -c 1500 gpurun_out/r01_e/bench_default.json
timeout 60 rocprofv3 --kernel-trace --stats -d gpurun_out/r01_e/prof -o run --output-format csv -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r01_e/bench_prof.json 2> gpurun_out/r01_e/prof.err; tail -c 300 gpurun_out/r01_e/bench_prof.json; ls gpurun_out/r01_e/prof | head
