# round 5, job 13: after the prune (12 objects): the whole GPU suite, fuzz, the bench line
mkdir -p gpurun_out/r5
( time timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r5/job13_pytest.txt 2>&1
cat gpurun_out/r5/job13_pytest.txt
( timeout 200 python scripts/fuzz_parity.py 45 9000 2>&1 | tail -2 ) > gpurun_out/r5/job13_fuzz.txt
( PIPE=1 timeout 200 python scripts/fuzz_parity.py 30 9500 2>&1 | tail -2 ) >> gpurun_out/r5/job13_fuzz.txt
cat gpurun_out/r5/job13_fuzz.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5/job13_bench.out 2> gpurun_out/r5/job13_bench.err
tail -3 gpurun_out/r5/job13_bench.err; tail -1 gpurun_out/r5/job13_bench.out
cp bench_detail.json gpurun_out/r5/job13_bench_detail.json
