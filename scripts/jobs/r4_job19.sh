# round 4, job 19: open signs settled lane-parallel in both K1s kernels (FIR_SIGN_DEFER, PK_DEFER with an 8-entry list):
# the whole GPU suite, then C3 and C5 A/B against the same tree built with both switches off / the commit before, then fuzz
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/job19.txt
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) >> gpurun_out/r4/job19.txt
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for lib in nodefer new nodefer new; do
  if [ $lib = new ]; then cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; else cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so; fi
  echo "C3 lib $lib" >> gpurun_out/r4/job19.txt
  ( REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> gpurun_out/r4/job19.txt
done
for lib in head new head new; do
  if [ $lib = new ]; then cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; else cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so; fi
  timeout 600 python bench.py --config C5 --no-cpu --no-traffic --no-e2e --no-others --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C5 lib $lib: 20-step', round(d['ms_per_step'], 3), 'steady', round(d['steady_state']['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()}, 'isolated', {k: round(v, 3) for k, v in (d.get('kernel_ms_isolated') or {}).items()})
" >> gpurun_out/r4/job19.txt
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
( timeout 400 python scripts/fuzz_parity.py 400 490000 2>&1 | tail -1 ) >> gpurun_out/r4/job19.txt
( TABLE=192k timeout 300 python scripts/fuzz_parity.py 200 500000 2>&1 | tail -1 ) >> gpurun_out/r4/job19.txt
( GNUAIS_FIR_PK=1 timeout 300 python scripts/fuzz_parity.py 200 510000 2>&1 | tail -1 ) >> gpurun_out/r4/job19.txt
cat gpurun_out/r4/job19.txt
