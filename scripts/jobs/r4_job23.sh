# round 4, job 23: K1s gathers sign and threshold bit with one v_alignbit per output (FL2: taps scaled by a power of two, the
# certified distance at |y'| = 2.0): the whole GPU suite, then C3 A/B inside one library (GNUAIS_FIR_FLAG2=0 is the kernel of
# job 22), then fuzz
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job23.txt
rm -f $out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) >> $out
for rep in 1 2; do
for fl in 0 1; do
  echo "C3 flag2 $fl" >> $out
  ( GNUAIS_FIR_FLAG2=$fl REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
done
( timeout 400 python scripts/fuzz_parity.py 300 550000 2>&1 | tail -1 ) >> $out
( PIPE=1 timeout 300 python scripts/fuzz_parity.py 200 560000 2>&1 | tail -1 ) >> $out
( timeout 200 python bench.py --no-cpu --no-others --no-e2e 2>/dev/null | tail -1 ) > gpurun_out/r4/job23_bench.json
cat $out
