# round 4, job 45: K3's per-thread buffers as dynamic LDS (the descriptor asked for 104 VGPRs per wave for 59 in use: the compiler pads the
# register request up to the occupancy it derives from STATIC LDS; now 72): parity, C3 and C5 A/B against the library before
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job45.txt
rm -f $out
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_nmea.py -m gpu -x -q 2>&1 | tail -2 ) >> $out
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for rep in 1 2; do
for lib in before new; do
  if [ $lib = new ]; then cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; else cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so; fi
  echo "lib $lib" >> $out
  ( REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
  ( timeout 300 python bench.py --config C5 --steps 20 --warmup 4 --no-cpu --no-others --no-e2e --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('C5 20-step', round(d['ms_per_step'], 3), 'steady', round(d.get('steady_state', {}).get('ms_per_step', 0), 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()})" ) >> $out 2>&1
done
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
cat $out
