# round 4, job 31: the full GPU suite with twelve central taps as the default, the round's profiles once more, fuzz
mkdir -p gpurun_out/r4
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r4/job31_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/job31_smoke.txt 2>&1
bash scripts/collect_profiles.sh r04 > gpurun_out/r4/job31_collect.log 2>&1
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 > gpurun_out/r4/job31_bench_node.json 2> gpurun_out/r4/job31_bench_node.err
( timeout 400 python scripts/fuzz_parity.py 300 610000 2>&1 | tail -1 ) > gpurun_out/r4/job31_fuzz.txt
( TABLE=192k timeout 300 python scripts/fuzz_parity.py 200 620000 2>&1 | tail -1 ) >> gpurun_out/r4/job31_fuzz.txt
( DEFRAMER=1 timeout 300 python scripts/fuzz_parity.py 200 630000 2>&1 | tail -1 ) >> gpurun_out/r4/job31_fuzz.txt
cat gpurun_out/r4/job31_pytest.txt gpurun_out/r4/job31_smoke.txt gpurun_out/r4/job31_fuzz.txt; tail -3 gpurun_out/r4/job31_collect.log | cut -c1-400
