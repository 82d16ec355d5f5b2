# round 5, job 1: the new bench line (length, time), the PLL forms in today's pipeline, the recurrence wave's budget
mkdir -p gpurun_out/r5
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5/job1_bench.out 2> gpurun_out/r5/job1_bench.err
tail -1 gpurun_out/r5/job1_bench.out | wc -c
cp bench_detail.json gpurun_out/r5/job1_bench_detail.json
timeout 900 python scripts/time_pll_forms.py 3:0x02 6:0x02 51:0x02 3:0x1f 6:0x1f 51:0x1f 3:0x1e 6:0x1e 3:0x1f 6:0x1f > gpurun_out/r5/job1_forms.txt 2>&1
rm -f gnuais_amd/csrc/build/pll_nrzi3.o
make -s -C gnuais_amd/csrc EXTRA=-DPLL3_BUDGET 2>&1 | grep -iE "error" 
timeout 600 python scripts/pll_wave_budget.py > gpurun_out/r5/job1_pll_wave_budget.txt 2>&1
tail -3 gpurun_out/r5/job1_bench.err; tail -1 gpurun_out/r5/job1_bench.out | cut -c1-600; cat gpurun_out/r5/job1_forms.txt; cat gpurun_out/r5/job1_pll_wave_budget.txt
