cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ch in 0 1 0 1; do
  echo "== GNUAIS_COLD_HOLD $ch"
  for r in 1 2 3; do GNUAIS_COLD_HOLD=$ch python bench.py --steps 20 --warmup 5 --no-cpu --no-others --no-e2e --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20 steps: %.4f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'steady', round(d['steady_state']['ms_per_step'],4) if d.get('steady_state') else None)"; done
done
rm -rf gpurun_out/tl20; mkdir -p gpurun_out/tl20
GNUAIS_COLD_HOLD=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl20 -o t -- python bench.py --steps 20 --warmup 5 --no-cpu --no-others --no-e2e --no-traffic > gpurun_out/tl20/bench.json 2>/dev/null
F=$(find gpurun_out/tl20 -name "*kernel_trace.csv" | head -1)
python scripts/region_timeline.py $F 20 | head -8
find gpurun_out/tl20 -name "*.csv" -delete
