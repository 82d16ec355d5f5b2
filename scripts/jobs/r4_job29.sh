# round 4, job 29: the full GPU suite on the round's final kernels, smoke, then the round's profiles again (bench line with its
# own PMC passes, rocprofv3 kernel stats, PMC traffic / SQ counters for C3 / C2 / C5) and the node line
mkdir -p gpurun_out/r4
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r4/job29_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/job29_smoke.txt 2>&1
bash scripts/collect_profiles.sh r04 > gpurun_out/r4/job29_collect.log 2>&1
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 > gpurun_out/r4/job29_bench_node.json 2> gpurun_out/r4/job29_bench_node.err
cat gpurun_out/r4/job29_pytest.txt gpurun_out/r4/job29_smoke.txt; tail -5 gpurun_out/r4/job29_collect.log | cut -c1-800
