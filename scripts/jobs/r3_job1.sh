mkdir -p gpurun_out/r3
./scripts/ubench/load_width.bin 16384 48000 512 5 > gpurun_out/r3/load_width_w5.txt 2>&1
./scripts/ubench/load_width.bin 16384 48000 512 8 > gpurun_out/r3/load_width_w8.txt 2>&1
cat gpurun_out/r3/load_width_w5.txt gpurun_out/r3/load_width_w8.txt
for spec in "2 0" "2 1" "2 3" "2 129" "2 130" "4 128" "4 129"; do set -- $spec; echo "== parity cpl $1 form $2"; GNUAIS_FIR_CPL=$1 GNUAIS_FIR_FORM=$2 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3; done > gpurun_out/r3/parity_wide.txt 2>&1
cat gpurun_out/r3/parity_wide.txt
GNUAIS_FIR_CPL=2 GNUAIS_FIR_FORM=1 timeout 900 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "threshold or c3 or silence" 2>&1 | tail -3
timeout 1200 python scripts/time_fir_wide.py all > gpurun_out/r3/time_fir_wide.txt 2>&1
cat gpurun_out/r3/time_fir_wide.txt
