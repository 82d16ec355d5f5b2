# round 4, job 9: the full GPU suite, then the round's profiles (bench line, rocprofv3 kernel stats, PMC) and the node line
mkdir -p gpurun_out/r4
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r4/job9_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/job9_smoke.txt 2>&1
bash scripts/collect_profiles.sh r04 > gpurun_out/r4/job9_collect.log 2>&1
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 > gpurun_out/r4/job9_bench_node.json 2> gpurun_out/r4/job9_bench_node.err
cat gpurun_out/r4/job9_pytest.txt gpurun_out/r4/job9_smoke.txt; tail -5 gpurun_out/r4/job9_collect.log | cut -c1-800
