# round 5, job 4: pll_h3 (variant 8): parity, fuzz, timing, wave budget
mkdir -p gpurun_out/r5
( timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ragged_chunks or noise_only" 2>&1 | tail -8 ) > gpurun_out/r5/job4_pytest.txt
cat gpurun_out/r5/job4_pytest.txt
( PLL_VARIANT=8 timeout 200 python scripts/fuzz_parity.py 60 5000 2>&1 | tail -4 ) > gpurun_out/r5/job4_fuzz.txt
cat gpurun_out/r5/job4_fuzz.txt
timeout 900 python scripts/time_pll_forms.py 8:0x02 3:0x02 8:0x1f 3:0x1f 8:0x1e 8:0x03 8:0x1f 3:0x1f > gpurun_out/r5/job4_forms.txt 2>&1
cat gpurun_out/r5/job4_forms.txt
rm -f gnuais_amd/csrc/build/pll_h3.o
make -s -C gnuais_amd/csrc EXTRA="-DPLLH3_BUDGET" 2>&1 | grep -iE "error"
timeout 600 python scripts/pllh3_wave_budget.py > gpurun_out/r5/job4_pllh3_budget.txt 2>&1
cat gpurun_out/r5/job4_pllh3_budget.txt
