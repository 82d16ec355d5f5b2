# round 5, job 23: final check -- the GPU suite, smoke, the bench line
mkdir -p gpurun_out/r5
( time timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r5/job23_pytest.txt 2>&1
cat gpurun_out/r5/job23_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5/job23_bench.out 2> gpurun_out/r5/job23_bench.err
tail -3 gpurun_out/r5/job23_bench.err; tail -1 gpurun_out/r5/job23_bench.out
cp bench_detail.json gpurun_out/r5/job23_bench_detail.json
