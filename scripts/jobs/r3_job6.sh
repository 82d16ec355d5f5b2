mkdir -p gpurun_out/r3
( python scripts/fir_wave_timeline.py 1 0 512; python scripts/fir_wave_timeline.py 1 0 256; python scripts/fir_wave_timeline.py 1 0 1024; python scripts/fir_wave_timeline.py 2 1 512; python scripts/fir_wave_timeline.py 2 0x85 512 ) > gpurun_out/r3/fir_wave_timeline.txt 2>&1
cat gpurun_out/r3/fir_wave_timeline.txt
