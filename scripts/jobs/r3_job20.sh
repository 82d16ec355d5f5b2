GNUAIS_FIR_PK=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
GNUAIS_FIR_PK=1 TABLE=192k timeout 300 python scripts/fuzz_parity.py 40 2>&1 | tail -1
GNUAIS_FIR_PK=1 timeout 300 python scripts/fuzz_parity.py 40 2>&1 | tail -1
timeout 900 python scripts/time_fir_pk.py 2>&1 | grep -v amdgpu
