# round 4, job 21: (a) every step's batch as S time slices (scripts/time_slices.py), (b) the 12-tap K1s with its taps in VGPRs
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job21.txt
rm -f $out
( REPS=5 timeout 900 python scripts/time_slices.py 1,3 2,3 2,4 3,4 4,4 4,6 6,6 8,8 1,3 2,4 4,6 2>&1 | grep -v amdgpu ) >> $out
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for rep in 1 2; do
for lib in new vtaps; do
  if [ $lib = vtaps ]; then cp scripts/ab/lib_vtaps.so gnuais_amd/libgnuais_hip.so; else cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; fi
  echo "C3 lib $lib" >> $out
  ( REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
cat $out
