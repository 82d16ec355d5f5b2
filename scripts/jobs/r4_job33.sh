# round 4, job 33: K3 on the deframer's stream (GNUAIS_K3_SAME=1): at ring lag 1 the two never overlap, so the event hand-over
# deframer -> K3 -> next deframer (two cross-stream waits per call, in the loop that sets the period) becomes stream order
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job33.txt
rm -f $out
for rep in 1 2 3; do
for same in 0 1; do
  echo "k3_same $same" >> $out
  ( GNUAIS_K3_SAME=$same REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
done
( GNUAIS_K3_SAME=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) >> $out
cat $out
