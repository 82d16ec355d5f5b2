# round 5, job 2: row schedules (microbenchmark), the six-wave form's budget, togglers at priority 3 in the pipeline
mkdir -p gpurun_out/r5
( cd scripts/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/pll_rows_sched.bin pll_rows_sched.hip && /tmp/pll_rows_sched.bin ) > gpurun_out/r5/job2_rows_sched.txt 2>&1
rm -f gnuais_amd/csrc/build/pll_nrzi.o
make -s -C gnuais_amd/csrc EXTRA="-DPLL6_BUDGET" 2>&1 | grep -iE "error"
timeout 600 python scripts/pll6_wave_budget.py > gpurun_out/r5/job2_pll6_budget_prio0.txt 2>&1
rm -f gnuais_amd/csrc/build/pll_nrzi.o
make -s -C gnuais_amd/csrc EXTRA="-DPLL6_BUDGET -DPLL_TOG_PRIO=3" 2>&1 | grep -iE "error"
timeout 600 python scripts/pll6_wave_budget.py > gpurun_out/r5/job2_pll6_budget_prio3.txt 2>&1
timeout 900 python scripts/time_pll_forms.py 6:0x02 6:0x1f 6:0x1e 3:0x1f 6:0x1f > gpurun_out/r5/job2_forms_togprio3.txt 2>&1
cat gpurun_out/r5/job2_rows_sched.txt gpurun_out/r5/job2_pll6_budget_prio0.txt gpurun_out/r5/job2_pll6_budget_prio3.txt gpurun_out/r5/job2_forms_togprio3.txt
