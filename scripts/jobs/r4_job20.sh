# round 4, job 20: ten central taps in the 12-tap K1s (fir_nc) on top of the lane-parallel settling: the whole GPU suite,
# then C3 A/B -- the tree with both settle switches off (scripts/ab/lib_nodefer.so, 12 taps) / this tree held at 12 taps
# (GNUAIS_FIR_NC=12) / this tree as it is -- then fuzz
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/job20.txt
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) >> gpurun_out/r4/job20.txt
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for rep in 1 2; do
for lib in nodefer new12 new; do
  if [ $lib = nodefer ]; then cp scripts/ab/lib_nodefer.so gnuais_amd/libgnuais_hip.so; else cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; fi
  echo "C3 lib $lib" >> gpurun_out/r4/job20.txt
  if [ $lib = new12 ]; then export GNUAIS_FIR_NC=12; else unset GNUAIS_FIR_NC; fi
  ( REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> gpurun_out/r4/job20.txt
done
done
unset GNUAIS_FIR_NC
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
( timeout 700 python scripts/fuzz_parity.py 600 520000 2>&1 | tail -1 ) >> gpurun_out/r4/job20.txt
( timeout 200 python bench.py --no-cpu --no-others --no-e2e 2>/dev/null | tail -1 ) > gpurun_out/r4/job20_bench.json
cat gpurun_out/r4/job20.txt
