# round 4, job 42: the deframer fed segment by segment beside its PLL launch (k2b_dataflow) once more, now that it is out of scratch memory
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job42.txt
rm -f $out
for df in 0 1; do
  echo "k2b_dataflow $df" >> $out
  ( GNUAIS_K2B_DATAFLOW=$df REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 2,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
cat $out
