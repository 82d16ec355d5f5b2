# round 4, job 8: the option space once more as a grid, deframer width included (nbuf,hold,firstreams,lag,pll,lpw)
mkdir -p gpurun_out/r4
( REPS=5 timeout 1200 python scripts/time_sched.py 4,-1,1,1,0,16 4,-1,1,1,0,32 5,-1,1,1,0,32 5,-1,1,2,0,32 6,-1,1,2,0,32 5,-1,1,1,0,16 5,-1,1,1,0,64 6,-1,1,2,0,64 4,-1,1,2,0,64 4,-1,1,2,0,32 4,-1,1,1,0,8 5,-1,2,2,0,32 4,-1,1,1,0,16 ) > gpurun_out/r4/job8_grid.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/job8_grid.txt
