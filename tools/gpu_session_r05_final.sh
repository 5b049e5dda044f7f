export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
(timeout 800 python -m pytest tests -q -m gpu -s 2>&1 | grep -a "parity\|passed\|failed\|FAILED\|Error" ) > $O/r05h_pytest_gpu.txt
tail -3 $O/r05h_pytest_gpu.txt
python tools/pmc_family.py run /tmp/pmcfam > $O/r05h_pmc_family.log 2>&1
python tools/pmc_family.py sum /tmp/pmcfam $O/r05h_pmc_gemm_family.json >> $O/r05h_pmc_family.log 2>&1
tail -1 $O/r05h_pmc_family.log
cp $O/r05h_pmc_gemm_family.json profiles/r05_pmc_gemm_family.json
python bench.py > $O/r05h_bench_default.json 2> $O/r05h_bench_default.err
head -c 400 $O/r05h_bench_default.json; echo
python tools/bench_secondary.py --only mvqa,convirt --steps 10 --warmup 4 2>&1 | grep -a '"task"' > $O/r05h_bench_secondary.jsonl
python tools/bench_secondary.py --only mvqa,convirt --steps 10 --warmup 4 --amp 1 2>&1 | grep -a '"task"' >> $O/r05h_bench_secondary.jsonl
cat $O/r05h_bench_secondary.jsonl
