mkdir -p gpurun_out/r5
( cd scripts/ubench && python gen_int_rate.py && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/int_rate.bin int_rate.hip && /tmp/int_rate.bin ) > gpurun_out/r5/job12_int_rate.txt 2>&1
cat gpurun_out/r5/job12_int_rate.txt
