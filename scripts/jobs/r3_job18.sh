GNUAIS_FIR_PK=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
GNUAIS_FIR_PK=1 timeout 900 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "c5 or c3_full or threshold or c2" 2>&1 | tail -3
GNUAIS_FIR_PK=1 TABLE=192k timeout 300 python scripts/fuzz_parity.py 60 2>&1 | tail -1
GNUAIS_FIR_PK=1 timeout 300 python scripts/fuzz_parity.py 60 2>&1 | tail -1
timeout 900 python scripts/time_fir_pk.py 2>&1 | grep -v amdgpu
python scripts/debug_pk.py 2>&1 | grep -v amdgpu | cut -c1-120
