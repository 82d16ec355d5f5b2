# round 4, job 17: the 12-tap FIR's register claim (waves per SIMD: 71 -> 7, 79 -> 6, 87 -> 5 (default), 95 -> 5, 103 -> 4) in the C3 pipeline
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/job17_claim.txt
cp gnuais_amd/libgnuais_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for v in 87 103 111 127 143; do
  cp scripts/ab/lib_claim$v.so gnuais_amd/libgnuais_hip.so
  echo "FIR claim v$v" >> gpurun_out/r4/job17_claim.txt
  ( REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> gpurun_out/r4/job17_claim.txt
done
done
cp /tmp/lib_keep.so gnuais_amd/libgnuais_hip.so
cat gpurun_out/r4/job17_claim.txt
