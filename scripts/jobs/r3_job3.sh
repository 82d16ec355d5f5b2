mkdir -p gpurun_out/r3
./scripts/ubench/load_width.bin 16384 48000 512 5 > gpurun_out/r3/load_width_jitter.txt 2>&1
cat gpurun_out/r3/load_width_jitter.txt
