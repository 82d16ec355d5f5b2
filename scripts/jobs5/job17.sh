# round 5, job 17: the round's profiles (bench line, rocprofv3 kernel stats, PMC), wave budget of pll_h3, node lines
mkdir -p gpurun_out/r5
bash scripts/collect_profiles.sh r05 > gpurun_out/r5/job17_collect.log 2>&1
tail -3 gpurun_out/r5/job17_collect.log | cut -c1-600
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 20 --warmup 5 > gpurun_out/r5/job17_bench_node2.json 2> gpurun_out/r5/job17_bench_node2.err
timeout 600 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --channels 2048 --steps 20 --warmup 5 > gpurun_out/r5/job17_bench_node8.json 2> gpurun_out/r5/job17_bench_node8.err
tail -1 gpurun_out/r5/job17_bench_node8.json | cut -c1-900
rm -f gnuais_amd/csrc/build/pll_h3.o
make -s -C gnuais_amd/csrc EXTRA="-DPLLH3_BUDGET" 2>&1 | grep -iE " error"
timeout 600 python scripts/pllh3_wave_budget.py > gpurun_out/r5/job17_pllh3_budget.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5/job17_pllh3_budget.txt
