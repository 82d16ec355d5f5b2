# round 4, job 25: hand-off depth x candidate-ring lag with the cheaper FIR of jobs 22/23 (was the K2b -> K3 -> K2b loop the bound at depth 4?)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job25.txt
rm -f $out
( REPS=5 timeout 600 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 4,-1,1,2 5,-1,1,2 3,-1,1,2 4,-1,2,2 4,-1,1,1,0,32 4,-1,1,2,0,32 4,-1,1,1,0,64 2>&1 | grep -v amdgpu ) >> $out
cat $out
