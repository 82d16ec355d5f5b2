# round 4, job 5: (a) the timed region's real cold start in a kernel trace, with and without the hold; (b) the
# time-parallel PLL (pll_variant 7): parity suite with it forced, C2 timing
mkdir -p gpurun_out/r4
( GNUAIS_PLL_VARIANT=7 timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4/job5_pytest_tp.txt
for v in 0 7 6; do
  GNUAIS_PLL_VARIANT=$v timeout 600 python bench.py --config C2 --no-cpu --no-traffic --no-e2e --no-others --steps 60 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C2 pll_variant $v: ms/step', round(d['ms_per_step'], 4), 'steady', round(d.get('steady_state', {}).get('ms_per_step', 0), 4), {k: round(v, 4) for k, v in d['kernel_ms'].items()}, 'iso', {k: round(v, 4) for k, v in d['kernel_ms_isolated'].items()})
" >> gpurun_out/r4/job5_c2.txt
done
cd /tmp && export TMPDIR=/tmp
for hold in -1 0; do
  rm -rf /tmp/cr$hold
  GNUAIS_COLD_HOLD_US=$hold timeout 600 rocprofv3 --kernel-trace -d /tmp/cr$hold -o cr --output-format csv -- python $GRAFT_REPO_ROOT/scripts/cold_region.py 20 > $GRAFT_REPO_ROOT/gpurun_out/r4/job5_cold$hold.log 2>&1
  f=$(find /tmp/cr$hold -name '*kernel_trace.csv' | head -1)
  python $GRAFT_REPO_ROOT/scripts/region_timeline.py $f 20 > $GRAFT_REPO_ROOT/gpurun_out/r4/job5_region_hold$hold.txt 2>&1
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/r4/job5_pytest_tp.txt; cat gpurun_out/r4/job5_c2.txt; grep region gpurun_out/r4/job5_cold-1.log gpurun_out/r4/job5_cold0.log; head -9 gpurun_out/r4/job5_region_hold-1.txt; tail -4 gpurun_out/r4/job5_region_hold-1.txt; head -9 gpurun_out/r4/job5_region_hold0.txt
