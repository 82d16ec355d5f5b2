# round 4, job 18: the packed FIR settles its open signs lane-parallel at the end of a segment (PK_DEFER) -- parity first,
# then C5 A/B against the library of the commit before (scripts/ab/lib_head.so), then the 192k fuzz
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/job18.txt
( timeout 1200 python -m pytest tests -m gpu -x -q -k "c5 or 192 or threshold or pk or silen or fir or slicer" 2>&1 | tail -4 ) >> gpurun_out/r4/job18.txt
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for lib in head new head new; do
  if [ $lib = head ]; then cp scripts/ab/lib_head.so gnuais_amd/libgnuais_hip.so; else cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; fi
  timeout 600 python bench.py --config C5 --no-cpu --no-traffic --no-e2e --no-others --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C5 lib $lib: 20-step', round(d['ms_per_step'], 3), 'steady', round(d['steady_state']['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()}, 'isolated', {k: round(v, 3) for k, v in (d.get('kernel_ms_isolated') or {}).items()})
" >> gpurun_out/r4/job18.txt
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
( TABLE=192k timeout 300 python scripts/fuzz_parity.py 200 470000 2>&1 | tail -1 ) >> gpurun_out/r4/job18.txt
( GNUAIS_FIR_PK=1 timeout 300 python scripts/fuzz_parity.py 200 480000 2>&1 | tail -1 ) >> gpurun_out/r4/job18.txt
cat gpurun_out/r4/job18.txt
