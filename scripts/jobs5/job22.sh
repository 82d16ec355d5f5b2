# round 5, job 22: deframer -> K3 -> deframer is one serial loop (0.42 + 0.06 ms = the period): ring lag 2 opens it
mkdir -p gpurun_out/r5
{
echo "== lag 1 (as built)"; timeout 600 python scripts/time_pll_forms.py 0:0x1f 0:0x19 0:0x1f 2>&1 | grep -v amdgpu.ids
echo "== GNUAIS_K2B_LAG=2"; GNUAIS_K2B_LAG=2 timeout 600 python scripts/time_pll_forms.py 0:0x1f 0:0x19 0:0x1f 0:0x1f:nbuf=4 0:0x1f:hdlc_lpw=32 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r5/job22_lag.txt 2>&1
cat gpurun_out/r5/job22_lag.txt
