# round 4, job 2: host-side scheduling knobs (nbuf, cold hold, two FIR streams, ring lag) on the C3 chain
mkdir -p gpurun_out/r4
( timeout 900 python scripts/time_sched.py 4,-1,1,1 4,0,1,1 4,50,1,1 5,0,1,1 6,0,1,1 8,0,1,1 4,0,2,1 6,0,2,1 6,0,2,2 6,0,1,2 5,0,2,2 6,0,2,2,6 6,0,1,2,6 4,-1,1,1 ) > gpurun_out/r4/job2_sched.txt 2>&1
( GNUAIS_FIR_STREAMS=2 GNUAIS_NBUF=6 GNUAIS_COLD_HOLD_US=0 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r4/job2_pytest_knobs.txt
cat gpurun_out/r4/job2_sched.txt | grep -v amdgpu.ids; cat gpurun_out/r4/job2_pytest_knobs.txt
