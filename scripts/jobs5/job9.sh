# round 5, job 9: defaults = pll_h3 + 64-lane deframer: the whole GPU suite, then the bench line
mkdir -p gpurun_out/r5
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r5/job9_pytest.txt
cat gpurun_out/r5/job9_pytest.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5/job9_bench.out 2> gpurun_out/r5/job9_bench.err
tail -2 gpurun_out/r5/job9_bench.err; tail -1 gpurun_out/r5/job9_bench.out
cp bench_detail.json gpurun_out/r5/job9_bench_detail.json
