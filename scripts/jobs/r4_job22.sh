# round 4, job 22: K1s's exact re-evaluation through a typed buffer descriptor (two VALU instructions per tap instead of twenty):
# the whole GPU suite, then C3 and C5 A/B against the tree before (scripts/ab/lib_vtaps.so = head with the 12-tap kernel's taps in
# VGPRs, measured level with head in job 21), then fuzz
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job22.txt
rm -f $out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) >> $out
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for rep in 1 2; do
for lib in head new; do
  if [ $lib = head ]; then cp scripts/ab/lib_vtaps.so gnuais_amd/libgnuais_hip.so; else cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; fi
  echo "C3 lib $lib" >> $out
  ( REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
  echo "C5 lib $lib" >> $out
  ( timeout 300 python bench.py --config C5 --steps 20 --warmup 4 --no-cpu --no-traffic --no-e2e --no-others 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('20-step', round(d['ms_per_step'], 3), 'steady', round(d.get('steady_state', {}).get('ms_per_step', 0), 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()}, 'isolated', {k: round(v, 3) for k, v in d['kernel_ms_isolated'].items()})
" ) >> $out 2>&1
done
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
( timeout 400 python scripts/fuzz_parity.py 300 530000 2>&1 | tail -1 ) >> $out
( TABLE=192k timeout 400 python scripts/fuzz_parity.py 300 540000 2>&1 | tail -1 ) >> $out
cat $out
