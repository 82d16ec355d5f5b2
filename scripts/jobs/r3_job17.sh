mkdir -p gpurun_out/r3
GNUAIS_FIR_PK=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -6
GNUAIS_FIR_PK=1 timeout 900 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "c5 or c3_full or threshold or c2" 2>&1 | tail -4
timeout 900 python scripts/time_fir_pk.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3/time_fir_pk.txt
