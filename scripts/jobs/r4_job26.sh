# round 4, job 26: K2b's bitmaps with 32-bit carries (45 instead of 118 VALU instructions per pack word) and immediate LDS offsets
# (kernel templated on the lanes per workgroup): the whole GPU suite, C3 A/B against the same tree with EV_BITMAPS32=0, fuzz
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job26.txt
rm -f $out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) >> $out
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for rep in 1 2; do
for lib in ev64 new; do
  if [ $lib = new ]; then cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; else cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so; fi
  echo "C3 lib $lib" >> $out
  ( REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
( DEFRAMER=1 timeout 400 python scripts/fuzz_parity.py 240 570000 2>&1 | tail -1 ) >> $out
( timeout 400 python scripts/fuzz_parity.py 240 580000 2>&1 | tail -1 ) >> $out
( GNUAIS_HDLC_LPW=64 timeout 300 python scripts/fuzz_parity.py 120 590000 2>&1 | tail -1 ) >> $out
( GNUAIS_HDLC_LPW=24 timeout 300 python scripts/fuzz_parity.py 100 600000 2>&1 | tail -1 ) >> $out
( timeout 200 python bench.py --no-cpu --no-others --no-e2e 2>/dev/null | tail -1 ) > gpurun_out/r4/job26_bench.json
cat $out
