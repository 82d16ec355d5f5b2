mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_hip_fullsize.py tests/test_hip_parity.py -m gpu -x -q -k "c5 or 192 or C5" 2>&1 | tail -4
TABLE=192k timeout 300 python scripts/fuzz_parity.py 120 2>&1 | tail -3
timeout 600 python scripts/time_c5_inloop.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3/c5_inloop.txt
