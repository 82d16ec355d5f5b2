mkdir -p gpurun_out/r3
echo "== prefetch 1"; timeout 600 python scripts/time_c5_inloop.py 2>&1 | grep "fir_inloop 1"
cp gnuais_amd/libgnuais_hip.so /tmp/keep.so; cp scripts/ab/libgnuais_hip_pf0.so gnuais_amd/libgnuais_hip.so
echo "== prefetch 0"; timeout 600 python scripts/time_c5_inloop.py 2>&1 | grep "fir_inloop 1"
cp /tmp/keep.so gnuais_amd/libgnuais_hip.so
echo "== prefetch 1 again"; timeout 600 python scripts/time_c5_inloop.py 2>&1 | grep "fir_inloop 1"
timeout 900 python -m pytest tests/test_hip_fullsize.py tests/test_hip_parity.py -m gpu -x -q -k "c5 or 192 or C5" 2>&1 | tail -2
TABLE=192k timeout 300 python scripts/fuzz_parity.py 60 2>&1 | tail -1
