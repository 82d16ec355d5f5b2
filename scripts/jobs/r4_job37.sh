# round 4, job 37: the round's profiles on the final tree (deframer out of scratch), smoke, the node line
mkdir -p gpurun_out/r4
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/job37_smoke.txt 2>&1
bash scripts/collect_profiles.sh r04 > gpurun_out/r4/job37_collect.log 2>&1
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 > gpurun_out/r4/job37_bench_node.json 2> gpurun_out/r4/job37_bench_node.err
cat gpurun_out/r4/job37_smoke.txt; tail -3 gpurun_out/r4/job37_collect.log | cut -c1-400
