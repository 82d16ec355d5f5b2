# round 4, job 11: C5 with the packed FIR's ring at v56 (three waves = 480 registers, 32 left per SIMD) against v44
# (432, 80 left: a K3 or a PLL wave fits beside three FIR waves), and longer segments; then C5's parity at full size
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/job11_c5.txt
for lib in b56 b44 b56 b44; do
  cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so
  for T in 1536 3072; do
    GNUAIS_FIR_T=$T timeout 600 python bench.py --config C5 --no-cpu --no-traffic --no-e2e --no-others --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C5 ring $lib fir_T $T: 20-step', round(d['ms_per_step'], 3), 'steady', round(d['steady_state']['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()})
" >> gpurun_out/r4/job11_c5.txt
  done
done
cp scripts/ab/lib_b44.so gnuais_amd/libgnuais_hip.so
( timeout 900 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "c5 or 192" 2>&1 | tail -3 ) >> gpurun_out/r4/job11_c5.txt
( TABLE=192k timeout 200 python scripts/fuzz_parity.py 120 460000 2>&1 | tail -1 ) >> gpurun_out/r4/job11_c5.txt
cat gpurun_out/r4/job11_c5.txt
