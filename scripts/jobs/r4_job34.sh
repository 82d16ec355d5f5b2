# round 4, job 34: the device instead of the host waits for K3 of call i-nbuf in front of a call's FIR launch (GNUAIS_DEV_WAIT=1)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job34.txt
rm -f $out
for rep in 1 2 3; do
for dw in 0 1; do
  echo "dev_wait $dw" >> $out
  ( GNUAIS_DEV_WAIT=$dw REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
done
( GNUAIS_DEV_WAIT=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -3 ) >> $out
cat $out
