# round 5, job 10: the FIR sets the period now -- two FIR streams, tail segments, depth
mkdir -p gpurun_out/r5
timeout 1200 python scripts/time_pll_forms.py 0:0x1f 0:0x1f:fir_streams=2 0:0x1f:fir_streams=2:nbuf=4 0:0x1f:fir_T2=512 0:0x1f:fir_tail=1 0:0x01:fir_streams=2 0:0x1f 0:0x1f:fir_streams=2 > gpurun_out/r5/job10_fir_streams.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5/job10_fir_streams.txt
