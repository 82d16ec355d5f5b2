# round 4, job 36: the deframer without its two scratch-resident state words (nstartsign and bufferpos lived in private memory --
# a load / store through the vector memory pipeline inside every event turn) and with the hunting section at both ends of a turn:
# A/B against the library before (scripts/ab/lib_before.so) and against no-scratch alone; stage masks; suite; deframer fuzz
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job36.txt
rm -f $out
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for rep in 1 2; do
for lib in before noscratch_hunt_once new; do
  if [ $lib = new ]; then cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; else cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so; fi
  echo "lib $lib" >> $out
  ( REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
  if [ $rep = 1 ]; then ( timeout 300 python scripts/time_masks.py 8,4 24,4 11,4 2>&1 | grep -v amdgpu ) >> $out; fi
done
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) >> $out
( DEFRAMER=1 timeout 300 python scripts/fuzz_parity.py 200 800000 2>&1 | tail -1 ) >> $out
( timeout 300 python scripts/fuzz_parity.py 200 810000 2>&1 | tail -1 ) >> $out
cat $out
