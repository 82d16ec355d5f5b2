mkdir -p gpurun_out/r5
rm -f gnuais_amd/csrc/build/pll_h3.o
make -s -C gnuais_amd/csrc EXTRA="-DPLLH3_BUDGET" 2>&1 | grep -iE " error"
timeout 600 python scripts/pllh3_wave_budget.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/job24_clock.txt
cat gpurun_out/r5/job24_clock.txt
