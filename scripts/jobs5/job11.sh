# round 5, job 11: the FIR's central sum on register pairs (fir_dpk): parity, fuzz, timing
mkdir -p gpurun_out/r5
( timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r5/job11_pytest.txt
cat gpurun_out/r5/job11_pytest.txt
( timeout 200 python scripts/fuzz_parity.py 60 7000 2>&1 | tail -3 ) > gpurun_out/r5/job11_fuzz.txt
cat gpurun_out/r5/job11_fuzz.txt
timeout 900 python scripts/time_pll_forms.py 0:0x01:fir_dpk=0 0:0x01:fir_dpk=1 0:0x1f:fir_dpk=0 0:0x1f:fir_dpk=1 0:0x1f:fir_dpk=0 0:0x1f:fir_dpk=1 0:0x03:fir_dpk=1 0:0x19:fir_dpk=1 > gpurun_out/r5/job11_dpk.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5/job11_dpk.txt
