# round 5, job 7: with the PLL stage shorter the FIR sets the period -- its cheaper forms again (packed, ten taps)
mkdir -p gpurun_out/r5
timeout 1200 python scripts/time_pll_forms.py 8:0x1f:hdlc_lpw=64 8:0x1f:hdlc_lpw=64:fir_pk=1 8:0x1f:hdlc_lpw=64:fir_nc=0 8:0x1f:hdlc_lpw=64:fir_pk=1:nbuf=4 8:0x01:fir_pk=1 8:0x01 8:0x1f:hdlc_lpw=64 8:0x1f:hdlc_lpw=64:fir_pk=1 > gpurun_out/r5/job7_fir_forms.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5/job7_fir_forms.txt
