# round 4, job 30: (a) twelve central taps again now that the flags cost one instruction (band 0.125 instead of 0.5: a quarter of the
# open signs, each of which re-reads 32 rows), (b) the deframer held to 96 / 80 VGPRs (EV_WAVES_PER_EU 5 / 6: it then fits beside four FIR waves)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job30.txt
rm -f $out
for rep in 1 2; do
for nc in 0 12; do
  echo "fir_nc $nc" >> $out
  ( GNUAIS_FIR_NC=$nc REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
done
for nc in 0 12; do
  ( GNUAIS_FIR_NC=$nc timeout 300 python bench.py --no-cpu --no-others --no-e2e --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
t = d['roofline']['traffic_detail']['bytes_per_launch']
print('fir_nc $nc bench: 20-step', round(d['ms_per_step'], 4), 'steady', round(d['steady_state']['ms_per_step'], 4), 'iso', {k: round(v, 3) for k, v in d['kernel_ms_isolated'].items()}, 'fir traffic GB', round(t['fir_slice'] / 1e9, 3), 'fir VALU M', round(d['roofline']['valu_by_kernel']['fir_slice']['insts_per_launch'] / 1e6, 1))
" ) >> $out 2>&1
done
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for lib in evocc5 evocc6 new; do
  if [ $lib = new ]; then cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; else cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so; fi
  echo "lib $lib" >> $out
  ( REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
cat $out
