# round 5, job 19: which stage costs the FIR what (stage masks, steady state)
mkdir -p gpurun_out/r5
timeout 1200 python scripts/time_pll_forms.py 0:0x01 0:0x09 0:0x11 0:0x19 0:0x03 0:0x0b 0:0x13 0:0x1f 0:0x01 0:0x11 0:0x09 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/job19_masks.txt
cat gpurun_out/r5/job19_masks.txt
