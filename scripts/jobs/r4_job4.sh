# round 4, job 4: cold start -- does the PLL stage of the first call get its place when the next FIR launch is held
# until the PLL launch reports its last workgroup running?  region timeline with and without, 20-step figures
mkdir -p gpurun_out/r4
( timeout 600 python scripts/time_sched.py 4,-1,1,1 4,0,1,1 4,30,1,1 4,-1,1,1 4,0,1,1 ) > gpurun_out/r4/job4_sched.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for hold in -1 0; do
  rm -rf /tmp/rt$hold
  GNUAIS_COLD_HOLD_US=$hold timeout 600 rocprofv3 --kernel-trace -d /tmp/rt$hold -o rt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-traffic --no-e2e --no-others --no-kernel-leg --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r4/job4_rt$hold.log 2>&1
  f=$(find /tmp/rt$hold -name '*kernel_trace.csv' | head -1)
  python $GRAFT_REPO_ROOT/scripts/region_timeline.py $f 20 > $GRAFT_REPO_ROOT/gpurun_out/r4/job4_region_hold$hold.txt 2>&1
done
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids gpurun_out/r4/job4_sched.txt; head -8 gpurun_out/r4/job4_region_hold-1.txt; head -8 gpurun_out/r4/job4_region_hold0.txt
# C5: the FIR-bound configuration with its launches on two streams (no gap, no tail between them)
for fs in 1 2 1 2; do
  GNUAIS_FIR_STREAMS=$fs timeout 600 python bench.py --config C5 --no-cpu --no-traffic --no-e2e --no-others --steps 20 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C5 fir_streams $fs: 20-step', round(d['ms_per_step'], 3), 'steady', round(d['steady_state']['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()})
" >> gpurun_out/r4/job4_c5.txt
done
cat gpurun_out/r4/job4_c5.txt
