mkdir -p gpurun_out/r3
timeout 900 python scripts/time_fir_elim.py > gpurun_out/r3/time_fir_elim.txt 2>&1
cat gpurun_out/r3/time_fir_elim.txt
