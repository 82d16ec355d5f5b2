# round 4, job 24: (a) CU split: deframer + K3 on R reserved CUs, the FIR on the others (GNUAIS_CU_SPLIT, two mask layouts),
# (b) the PLL scanner with 3 / 4 blocks of sign words in flight (PLL_AHEAD_N)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job24.txt
rm -f $out
echo "base" >> $out
( REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
for lay in 0 1; do
for R in 32 64 96; do
  echo "cu_split $R layout $lay" >> $out
  ( GNUAIS_CU_SPLIT=$R GNUAIS_CU_LAYOUT=$lay REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
done
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for lib in ahead3 ahead4 new; do
  if [ $lib = new ]; then cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; else cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so; fi
  echo "C3 lib $lib" >> $out
  ( REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
cat $out
