# round 5, job 18: pll_h3 with the next block's header fetched beside the rows: parity, fuzz, timing
mkdir -p gpurun_out/r5
( timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/r5/job18_pytest.txt
cat gpurun_out/r5/job18_pytest.txt
( PLL_VARIANT=8 timeout 200 python scripts/fuzz_parity.py 40 11000 2>&1 | tail -2 ) > gpurun_out/r5/job18_fuzz.txt
cat gpurun_out/r5/job18_fuzz.txt
timeout 900 python scripts/time_pll_forms.py 0:0x02 0:0x1f 0:0x1f 0:0x02 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/job18_forms.txt
cat gpurun_out/r5/job18_forms.txt
