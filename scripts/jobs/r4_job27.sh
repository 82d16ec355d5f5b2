# round 4, job 27: the C3 pipeline with stages switched off -- which stages set the period?
mkdir -p gpurun_out/r4
( timeout 600 python scripts/time_masks.py 31,3 31,4 1,4 30,3 30,4 3,4 26,4 24,4 2,4 8,4 27,4 11,4 31,4 2>&1 | grep -v amdgpu ) > gpurun_out/r4/job27.txt
cat gpurun_out/r4/job27.txt
