# round 4, job 28: a lower register claim for the 12-tap FIR (80 / 72 / 88 instead of 104) TOGETHER with an LDS claim per FIR wave that
# keeps it at four waves per SIMD beside a PLL workgroup: then 4 x 80 + 64 (PLL) + 120 (deframer) = 504 registers fit a SIMD and a
# deframer wave no longer has to wait for a FIR wave to retire
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job28.txt
rm -f $out
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
echo "base (claim v103, no LDS claim)" >> $out
( REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
for lib in v79 v71 v87; do
  cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so
  for lds in 0 2048 2304 2560 3072 4096; do
    echo "claim $lib fir_lds $lds" >> $out
    ( FIR_LDS=$lds REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
  done
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
for lds in 2048 3072; do
  echo "claim v103 fir_lds $lds" >> $out
  ( FIR_LDS=$lds REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
cat $out
