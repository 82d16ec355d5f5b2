mkdir -p gpurun_out/r3
( for t in "fir_tail=0" "fir_tail=10 fir_T2=128" "fir_tail=12 fir_T2=128" "fir_tail=15 fir_T2=128" "fir_tail=12 fir_T2=256" "fir_tail=20 fir_T2=256"; do python scripts/fir_wave_timeline.py 1 0 512 $t | head -1; done
  python scripts/fir_wave_timeline.py 1 0 512 fir_tail=12 fir_T2=128
  for t in "fir_tail=0" "fir_tail=12 fir_T2=128"; do python scripts/fir_wave_timeline.py 2 1 512 $t | head -1;  python scripts/fir_wave_timeline.py 2 0x85 512 $t | head -1; done
) > gpurun_out/r3/fir_wave_timeline2.txt 2>&1
cat gpurun_out/r3/fir_wave_timeline2.txt
GNUAIS_FIR_CPL=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
python scripts/time_fir_wide.py all 1:0:512 2:1:512 2:0x85:512
