# round 5, job 5: pll_h3 in the pipeline x deframer width / hand-off depth / K3's stream
mkdir -p gpurun_out/r5
timeout 1200 python scripts/time_pll_forms.py 8:0x1f 8:0x1f:hdlc_lpw=8 8:0x1f:hdlc_lpw=32 8:0x1f:hdlc_lpw=64 8:0x1f:nbuf=4 8:0x1f:k3_same=0 8:0x1f:nbuf=4:hdlc_lpw=32 8:0x0b 8:0x19 8:0x1b 3:0x1f 8:0x1f > gpurun_out/r5/job5_grid.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5/job5_grid.txt
