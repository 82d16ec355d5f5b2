# round 4, job 44: outputs per FIR wave (GNUAIS_FIR_T) once more with the memory-bound twelve-tap FL2 kernel
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job44.txt
rm -f $out
for T in 512 384 768 1024 512; do
  echo "fir_T $T" >> $out
  ( GNUAIS_FIR_T=$T REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
cat $out
