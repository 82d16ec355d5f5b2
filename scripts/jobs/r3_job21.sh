timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q -k "192 or c5 or C5" 2>&1 | tail -2
TABLE=192k timeout 300 python scripts/fuzz_parity.py 40 2>&1 | tail -1
timeout 900 python scripts/time_fir_pk.py C5 2>&1 | grep -v amdgpu
