mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/job14_nbuf_bench.txt
for nb in 4 3 4 3; do
  GNUAIS_NBUF=$nb timeout 600 python bench.py --no-cpu --no-traffic --no-e2e --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('nbuf $nb: C3 20-step', round(d['ms_per_step'], 4), 'steady', round(d['steady_state']['ms_per_step'], 4), 'exact', round(d['exact_chain']['ms_per_step'], 3), {k: round(v['ms_per_step'], 4) for k, v in d['other_configs'].items()})
" >> gpurun_out/r4/job14_nbuf_bench.txt
done
cat gpurun_out/r4/job14_nbuf_bench.txt
