# round 5, job 8: hand-off depth with pll_h3 and the 64-lane deframer
mkdir -p gpurun_out/r5
timeout 1200 python scripts/time_pll_forms.py 8:0x1f:hdlc_lpw=64 8:0x1f:hdlc_lpw=64:nbuf=4 8:0x1f:hdlc_lpw=64:nbuf=5 8:0x1f:hdlc_lpw=64:nbuf=4:k3_same=0 8:0x1f:hdlc_lpw=32:nbuf=4 8:0x1f:hdlc_lpw=64 8:0x1f:hdlc_lpw=64:nbuf=4 > gpurun_out/r5/job8_depth.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5/job8_depth.txt
