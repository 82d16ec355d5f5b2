mkdir -p gpurun_out/r4
( REPS=7 timeout 600 python scripts/time_sched.py 4,-1,1,1 3,-1,1,1 2,-1,1,1 3,-1,2,1 3,-1,1,1,0,32 4,-1,1,1 ) > gpurun_out/r4/job13_nbuf.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/job13_nbuf.txt
