mkdir -p gpurun_out/r4
( GNUAIS_PLL_VARIANT=7 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r4/job6_pytest_tp.txt
timeout 300 python scripts/time_pll_tp.py 256 > gpurun_out/r4/job6_tp.txt 2>&1
timeout 300 python scripts/time_pll_tp.py 1024 >> gpurun_out/r4/job6_tp.txt 2>&1
cat gpurun_out/r4/job6_pytest_tp.txt; grep -v amdgpu.ids gpurun_out/r4/job6_tp.txt
